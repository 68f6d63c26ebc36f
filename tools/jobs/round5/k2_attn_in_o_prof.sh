#!/bin/bash
# B = 1 frame under rocprofv3 --kernel-trace: kernel table of the steady segment (is k_gemv4_attn running, what does it cost)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
bash tools/prof_frame.sh 1.7b 1 200 512 > /dev/null 2>&1; cp gpurun_out/frameprof/frame_1.7b_b1.txt gpurun_out/r5/k2_b1_kernels_fold.txt
Q3_CP_NO_ATTN_FOLD=1 bash tools/prof_frame.sh 1.7b 1 200 512 > /dev/null 2>&1; cp gpurun_out/frameprof/frame_1.7b_b1.txt gpurun_out/r5/k2_b1_kernels_nofold.txt
head -40 gpurun_out/r5/k2_b1_kernels_fold.txt; head -40 gpurun_out/r5/k2_b1_kernels_nofold.txt
