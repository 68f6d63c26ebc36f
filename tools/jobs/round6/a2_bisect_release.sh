#!/bin/bash
# round 6 / A2: which release-free family pair gives different codes (A1: all release-free differs, each family alone does not)
mkdir -p gpurun_out/r6
python tools/dev/aql_ab.py --batch 8 --frames 120 --reps 1 --modes 0,3/Q3_AQL_T_ACQ=0,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_gemv+k_attn_cp,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_gemv+k_attn_fused,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_gemv+k_attn_merge,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_gemv+k_attn_first2,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_attn+k_gemv_sk2,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_attn+k_gemv_mfmaI,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_attn+k_gemv_gu24,3/Q3_AQL_T_ACQ=0/Q3_AQL_T_ONLY=k_attn+k_gemv_lds+k_gemv_mfma4 2>&1 | grep -v WARNING > gpurun_out/r6/a2_bisect.txt
cat gpurun_out/r6/a2_bisect.txt
