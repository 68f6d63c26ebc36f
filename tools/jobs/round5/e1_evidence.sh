#!/bin/bash
# round-5 evidence: PMC traffic of the GEMV shapes, the rocprofv3 kernel table of the headline command, vocoder table + MFMA
# counters, the whole GPU suite, the default bench line. Outputs under gpurun_out/final5/ (copied into profiles/ by hand).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/final5
bash tools/pmc_collect.sh 8 > gpurun_out/final5/pmc_collect.log 2>&1; cp gpurun_out/pmc/pmc_gemv_M8.json gpurun_out/final5/r5_pmc_gemv_M8.json
Q3_PROF_NAME=final5/r5_rocprof_kernel_stats_bench_b8.txt bash tools/prof_bench_b8.sh > /dev/null 2>&1
bash tools/prof_vocoder.sh 640 > /dev/null 2>&1; cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/final5/r5_vocoder_kernels_T640.txt
bash tools/pmc_vocoder.sh 640 > /dev/null 2>&1; cp gpurun_out/pmc/vocoder_mfma_T640.txt gpurun_out/final5/r5_pmc_vocoder_mfma_T640.txt
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final5/suite.txt
cp gpurun_out/final5/r5_pmc_gemv_M8.json gpurun_out/final5/r5_rocprof_kernel_stats_bench_b8.txt profiles/ 2>/dev/null      # bench.py reads this round's profiles
timeout 900 python bench.py > gpurun_out/final5/bench.json 2> gpurun_out/final5/bench.err
tail -3 gpurun_out/final5/suite.txt; head -c 600 gpurun_out/final5/bench.json
