#!/bin/bash
# round 6 / N5: shader-clock stamps inside k_sample, per row: cycles from kernel entry to (0) phase 1 done, (1) candidates known, (2) tail done;
# (3) the same span as (2) on the 100 MHz wall clock
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for sel in 0 1 2 3; do echo "stamp $sel: $(Q3_SAMPLE_SLOW_TOPK=$((64 + 256 * sel)) python tools/dev/sample_probe.py 2>&1 | tail -1)"; done > gpurun_out/r6/n5_stamps.txt 2>&1; cat gpurun_out/r6/n5_stamps.txt
