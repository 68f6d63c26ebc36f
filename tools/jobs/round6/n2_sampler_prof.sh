#!/bin/bash
# round 6 / N2: k_sample's own duration with the one-pass selection and with the radix select (rocprofv3 --kernel-trace, frames through hipGraphLaunch)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
Q3_AQL=0 Q3_PROF_NAME=r6/n2_prof_fast.txt bash tools/prof_bench_b8.sh --frames 160 > /dev/null 2>&1
Q3_SAMPLE_SLOW_TOPK=1 Q3_AQL=0 Q3_PROF_NAME=r6/n2_prof_slow.txt bash tools/prof_bench_b8.sh --frames 160 > /dev/null 2>&1
grep -E "k_sample|k_frame_embed|per frame" gpurun_out/r6/n2_prof_fast.txt gpurun_out/r6/n2_prof_slow.txt
