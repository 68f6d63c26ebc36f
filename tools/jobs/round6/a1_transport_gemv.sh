#!/bin/bash
# round 6 / A1: the frame's kernels on write-through transport; fence-free boundaries (Q3_AQL=3) vs hipGraphLaunch vs own queue with HIP's fences;
# bisected by family (Q3_AQL_T_ONLY) and by fence half (Q3_AQL_T_ACQ / _REL)
set -x
mkdir -p gpurun_out/r6
python tools/dev/aql_ab.py --batch 8 --frames 300 --modes 0,1,3,3/Q3_AQL_T_ONLY=k_gemv,3/Q3_AQL_T_ONLY=k_attn,3/Q3_AQL_T_REL=0,3/Q3_AQL_T_ACQ=0,0,3 > gpurun_out/r6/a1_aql_ab_b8.txt 2>&1
python tools/dev/aql_ab.py --batch 1 --frames 300 --modes 0,3,0,3 > gpurun_out/r6/a1_aql_ab_b1.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -k "linear or teacher_forced or free_run or b8_b16 or sampler" > gpurun_out/r6/a1_tests.txt 2>&1
tail -5 gpurun_out/r6/a1_tests.txt
cat gpurun_out/r6/a1_aql_ab_b8.txt gpurun_out/r6/a1_aql_ab_b1.txt
