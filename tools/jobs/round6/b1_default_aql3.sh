#!/bin/bash
# round 6 / B1: the fence-free own-queue path as the DEFAULT (Q3_AQL unset = 3). A/B at the bench's own run length, the whole
# -m gpu suite through it (parity constants tightened to MARGIN_EPS 2e-4 / LOGIT_NOISE 5e-5), the default bench line, the
# per-node timeline of the B = 8 frame.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 0,3,0,3 2>&1 | grep -v WARNING > gpurun_out/r6/b1_aql_ab_b8_640.txt
python tools/dev/aql_ab.py --model 0.6b --batch 1 --frames 300 --modes 0,3,0,3 2>&1 | grep -v WARNING > gpurun_out/r6/b1_aql_ab_06b_b1.txt
cat gpurun_out/r6/b1_aql_ab_b8_640.txt gpurun_out/r6/b1_aql_ab_06b_b1.txt
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6/b1_suite.txt 2>&1
tail -8 gpurun_out/r6/b1_suite.txt
timeout 900 python bench.py > gpurun_out/r6/b1_bench.json 2> gpurun_out/r6/b1_bench.err
head -c 400 gpurun_out/r6/b1_bench.json; echo
python tools/trace_frame.py 1.7b 8 64 512 > gpurun_out/r6/b1_trace_frame_b8.txt 2>&1
tail -30 gpurun_out/r6/b1_trace_frame_b8.txt
