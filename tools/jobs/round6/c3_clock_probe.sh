#!/bin/bash
# round 6 / C3: the shader clock (and board power) the chip holds under the frame loop, the 4k-position prefill and the vocoder —
# what "matrix-core busy as a fraction of the NOMINAL clock" can be at best for the two MFMA-paced phases.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/clock_probe.py 6 > gpurun_out/r6/c3_clock_probe.txt 2>&1; cat gpurun_out/r6/c3_clock_probe.txt
rocm-smi --showclocks --showpower 2>&1 | head -30
