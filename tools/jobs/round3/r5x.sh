#!/bin/bash
# round 3, job x: continuous-batching fuzz + replace tests on the reserved-capacity session; vocoder geometry knob A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or replace or sampling_options or rows_end or streaming_several" 2>&1 | tail -15
bash tools/jobs/r5w.sh
