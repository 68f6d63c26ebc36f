#!/bin/bash
# round 6 / A3: minimal failing sets of release-free kernels (A2: k_gemv + k_attn_cp differs; k_attn + any single GEMV family does not)
mkdir -p gpurun_out/r6
G="k_attn_cpILi16ELb1+k_attn_cpILi8ELb1+k_attn_cpILi4ELb1"
N="k_attn_cpILi16ELb0+k_attn_cpILi8ELb0+k_attn_cpILi4ELb0"
M="0"
for set in "k_attn_cp+k_gemv_sk2+k_gemv_mfmaI" "k_attn_cp+k_gemv_sk2" "k_attn_cp+k_gemv_mfmaI" "k_gemv+$G" "k_gemv+$N" "k_gemv_sk2+k_gemv_mfmaI+$G" "k_gemv_sk2+k_gemv_mfmaI+$N"; do
  M="$M,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=$set"
done
python tools/dev/aql_ab.py --batch 8 --frames 120 --reps 1 --modes "$M" 2>&1 | grep -v WARNING > gpurun_out/r6/a3_bisect.txt
cat gpurun_out/r6/a3_bisect.txt
