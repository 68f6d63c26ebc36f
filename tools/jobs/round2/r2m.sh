#!/bin/bash
# r2m: bf16x3 flash attention for the prefill: parity, timing A/B, kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2m.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -2 >> $O; }
run Q3_PREFILL_ATTN_X3=0
run Q3_PREFILL_ATTN_X3=1
run Q3_PREFILL_ATTN_X3=1 Q3_PREFILL_ATTN_NOSPLIT=1
echo "== 0.6b, B=2 2048, 1000" >> $O
timeout 300 python tools/prof_prefill.py 0.6b 4096 1 2>&1 | tail -1 >> $O
env Q3_PREFILL_ATTN_X3=0 timeout 300 python tools/prof_prefill.py 1.7b 2039 2 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 1.7b 2039 2 2>&1 | tail -1 >> $O
env Q3_PREFILL_ATTN_X3=0 timeout 300 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
timeout 600 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -5 >> $O
cat gpurun_out/bench_prefill4k.json >> $O; echo >> $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gemm_prefill or prefill" 2>&1 | tail -5 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2m_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2m_prof 3 >> $O 2>&1
cat $O
