#!/bin/bash
# round 6 / I1: kernel table of the WIDE session (B = 64, 1.7B, 512-token prompts, 320 frames) — rocprofv3 --kernel-trace, frames through
# hipGraphLaunch (Q3_AQL=0: rocprofv3 cannot follow the own queue)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
Q3_AQL=0 Q3_PROF_NAME=r6/i1_rocprof_b64.txt bash tools/prof_bench_b8.sh --batch 64 --frames 320 | head -60
