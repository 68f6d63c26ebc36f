#!/bin/bash
# round-6 evidence OF THE TREE IT RUNS ON: PMC traffic of the GEMV shapes, the rocprofv3 kernel table of the headline command (frames
# submitted on the library's own AQL queue), the per-node timeline of the B = 8 / B = 1 frame (-DQ3_TRACE build), vocoder kernel table +
# MFMA counters, prefill kernel table, the whole GPU suite, smoke, the default bench line. Outputs under gpurun_out/final6/ (copied into
# profiles/ by hand; bench.py reads profiles/r6_* for its traffic / gemv_in_graph fields, so the PMC + rocprof tables are copied first).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final6; mkdir -p $O
git rev-parse HEAD > $O/head.txt 2>/dev/null
bash tools/pmc_collect.sh 8 > $O/pmc_collect.log 2>&1; cp gpurun_out/pmc/pmc_gemv_M8.json $O/r6_pmc_gemv_M8.json
Q3_PROF_NAME=final6/r6_rocprof_kernel_stats_bench_b8.txt bash tools/prof_bench_b8.sh > /dev/null 2>&1; head -12 $O/r6_rocprof_kernel_stats_bench_b8.txt
python tools/trace_frame.py 1.7b 8 300 512 > $O/r6_trace_frame_b8.txt 2>&1; sed -n 1,8p $O/r6_trace_frame_b8.txt | cut -c1-200
python tools/trace_frame.py 1.7b 1 300 512 > $O/r6_trace_frame_b1.txt 2>&1
bash tools/prof_vocoder.sh 640 > /dev/null 2>&1; cp gpurun_out/vocprof/vocoder_T640.txt $O/r6_vocoder_kernels_T640.txt; head -2 $O/r6_vocoder_kernels_T640.txt
bash tools/pmc_vocoder.sh 640 > /dev/null 2>&1; cp gpurun_out/pmc/vocoder_mfma_T640.txt $O/r6_pmc_vocoder_mfma_T640.txt
bash tools/prof_prefill.sh 1.7b 4096 > $O/r6_prefill_kernels_1.7b_4096.txt 2>&1; cat $O/r6_prefill_kernels_1.7b_4096.txt
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/suite.txt; tail -3 $O/suite.txt
cp $O/r6_pmc_gemv_M8.json $O/r6_rocprof_kernel_stats_bench_b8.txt profiles/ 2>/dev/null      # bench.py reads this round's profiles
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
head -c 500 $O/bench.json; echo
