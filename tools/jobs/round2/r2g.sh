#!/bin/bash
# round 2, GPU call G: evidence refresh — PMC HBM traffic per GEMV shape (M = 8), rocprofv3 kernel stats of the bench command
O=gpurun_out/r2g; mkdir -p $O
bash tools/pmc_collect.sh 8 > $O/pmc_collect.log 2>&1; cp gpurun_out/pmc/pmc_gemv_M8.json $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --also-batches "" > $GRAFT_REPO_ROOT/$O/rocprof_bench.log 2>&1
f=$(find /tmp/rp -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/$O/rocprof_kernel_stats_bench_b8.csv
cd $GRAFT_REPO_ROOT; head -25 $O/rocprof_kernel_stats_bench_b8.csv | cut -c1-160; cat $O/pmc_gemv_M8.json | head -40; tail -2 $O/rocprof_bench.log | cut -c1-600
