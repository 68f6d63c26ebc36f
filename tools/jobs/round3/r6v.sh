#!/bin/bash
# round 3, job 6v: PMC counters of a 64-row frame (eager launches, 10 frames): matrix-core busy, VALU / VMEM instructions, L2 requests, HBM fetch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6v; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6v
cd /tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "TCC_REQ_sum TCC_HIT_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pm
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/prof_run.py 1.7b 64 10 eager > $O/pass$i.log 2>&1
  echo "== pass $i: $ctrs" >> $O/pmc_frame_b64.txt
  python $GRAFT_REPO_ROOT/tools/pmc_kernels.py /tmp/pm 2>&1 | grep -v "k_conv\|k_resunit\|k_norm_c\|k_attn_c_mfma\|k_rope_c\|k_lin_small\|k_silu\|k_dwconv\|k_rvq\|k_pack\|copyBuffer\|k_lm_gemm\|k_prefill\|k_kv_planes\|k_split" | head -22 >> $O/pmc_frame_b64.txt
done
cat $O/pmc_frame_b64.txt | cut -c1-190
