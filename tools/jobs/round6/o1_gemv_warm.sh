#!/bin/bash
# round 6 / O1: what a GEMV launch costs with its weights already on chip (1 copy replayed: L2 / MALL warm; 2 copies: MALL warm) against streamed from HBM
# (many copies cycled) — the price list for a weight prefetch beside the attention kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/gemv_warm.py > gpurun_out/r6/o1_gemv_warm.txt 2>&1; cat gpurun_out/r6/o1_gemv_warm.txt
