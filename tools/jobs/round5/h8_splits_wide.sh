#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
O=gpurun_out/r5/h8_splits_wide.txt
echo "B = 16" | tee -a $O; timeout 900 python tools/dev/env_ab.py "" "Q3_ATTN_SPLITS=8" "Q3_ATTN_SPLITS=6" --batch 16 --frames 400 --reps 2 --rounds 1 2>&1 | tee -a $O
echo "B = 32" | tee -a $O; timeout 900 python tools/dev/env_ab.py "" "Q3_ATTN_SPLITS=4" "Q3_ATTN_SPLITS=3" --batch 32 --frames 400 --reps 2 --rounds 1 2>&1 | tee -a $O
echo "B = 64" | tee -a $O; timeout 900 python tools/dev/env_ab.py "" "Q3_ATTN_SPLITS=2" "Q3_ATTN_SPLITS=3" --batch 64 --frames 400 --reps 2 --rounds 1 2>&1 | tee -a $O
echo "B = 8" | tee -a $O; timeout 900 python tools/dev/env_ab.py "" "Q3_ATTN_SPLITS=6" "Q3_ATTN_SPLITS=10" "Q3_ATTN_SPLITS=12" --batch 8 --frames 640 --reps 2 --rounds 1 2>&1 | tee -a $O
