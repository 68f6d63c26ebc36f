#!/bin/bash
# round 6 / L1: k_wide2_swiglu with 8 waves (K in eight slices: half as many dependent k-steps per wave) against 4 — isolated launch
# times of the two gate / up shapes at 64 rows, the frame of a 64-row session, then the wide-session parity tests with 8 waves
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for w in 4 8 4 8; do
  echo "Q3_WIDE2_WAVES=$w: $(Q3_WIDE2_WAVES=$w python tools/dev/wide2_ab.py --child --batch 64 --frames 200 2>&1 | grep -E 'us|ms/frame' | tr '\n' '|')"
done > gpurun_out/r6/l1_wide2_waves.txt 2>&1
for w in 4 8; do
  echo "Q3_WIDE2_WAVES=$w B=32: $(Q3_WIDE2_WAVES=$w python tools/dev/wide2_ab.py --child --batch 32 --frames 200 2>&1 | grep -E 'ms/frame' | tr '\n' '|')"
done >> gpurun_out/r6/l1_wide2_waves.txt 2>&1
cat gpurun_out/r6/l1_wide2_waves.txt
Q3_WIDE2_WAVES=8 timeout 900 python -m pytest tests -m gpu -q -k "wide or b64 or batch_64 or rows_64 or 64" > gpurun_out/r6/l1_tests.txt 2>&1; tail -4 gpurun_out/r6/l1_tests.txt
