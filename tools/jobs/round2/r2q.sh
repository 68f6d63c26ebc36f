#!/bin/bash
# r2q: prefill after the 2-record merge; kernel trace; MFMA-busy PMC table for the prefill kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2q.txt; : > $O
timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -2 >> $O
timeout 300 python tools/prof_prefill.py 0.6b 4096 1 2>&1 | tail -1 >> $O
timeout 600 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -2 >> $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gemm_prefill or prefill" 2>&1 | tail -3 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2q_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2q_prof 3 >> $O 2>&1
cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2q_pmc -o p -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r2q_pmc/* | head >> $O; python tools/pmc_mfma_table.py gpurun_out/r2q_pmc 3 "prefill of 4096 instruct positions (1.7b), 3 prefills + 8 frames each" > gpurun_out/r2q_pmc_table.txt 2>> $O; cat gpurun_out/r2q_pmc_table.txt >> $O
find gpurun_out/r2q_pmc -name "*kernel_trace.csv" -delete; find gpurun_out/r2q_pmc -name "*.csv" -size +20M -delete
cat $O
