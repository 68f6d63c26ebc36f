#!/bin/bash
# r2l: prefill GEMM geometry 3 (LDS-DMA, one barrier per stage) + decode-step tail: parity, timing variants, kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2l.txt; : > $O
run() { echo "== $*" >> $O; env "$@" python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -2 >> $O; }
run Q3_GEMM_GEO=2 Q3_PREFILL_TAIL_PASSES=0
run Q3_GEMM_GEO=2
run Q3_GEMM_GEO=3
run Q3_GEMM_GEO=3 Q3_GEMM3_V=1
run Q3_X=1
run Q3_GEMM3_NSPLIT=1
run Q3_GEMM3_NSPLIT=2
echo "== 0.6b" >> $O
env Q3_GEMM_GEO=2 Q3_PREFILL_TAIL_PASSES=0 python tools/prof_prefill.py 0.6b 4096 1 2>&1 | tail -1 >> $O
python tools/prof_prefill.py 0.6b 4096 1 2>&1 | tail -1 >> $O
echo "== 2048 / 1000 positions" >> $O
env Q3_GEMM_GEO=2 python tools/prof_prefill.py 1.7b 2039 1 2>&1 | tail -1 >> $O
python tools/prof_prefill.py 1.7b 2039 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=3 python tools/prof_prefill.py 1.7b 2039 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=2 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=3 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gemm_prefill or prefill" 2>&1 | tail -5 >> $O
timeout 600 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -5 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2l_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r2l_prof -name "*kernel_stats.csv" | head -1); head -14 "$f" >> $O
cat $O
