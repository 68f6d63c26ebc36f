#!/bin/bash
# r2t: PMC counters of the vocoder's conv kernels (what do the k = 7 convs wait on?)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2t.txt; : > $O
timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 >> $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2t_pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 1 > $GRAFT_REPO_ROOT/gpurun_out/r2t_pmc$i.log 2>&1
  cd $GRAFT_REPO_ROOT; find gpurun_out/r2t_pmc$i -name "*kernel_trace.csv" -delete
done
python tools/pmc_table.py conv_bf16x3 gpurun_out/r2t_pmc1 gpurun_out/r2t_pmc2 gpurun_out/r2t_pmc3 gpurun_out/r2t_pmc4 > gpurun_out/r2t_table.txt 2>&1
head -150 gpurun_out/r2t_table.txt >> $O
tail -5 gpurun_out/r2t_pmc4.log >> $O
cat $O
