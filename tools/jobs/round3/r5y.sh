#!/bin/bash
# round 3, job y: continuous-batching fuzz with reserved capacity; vocoder with the 4-wave 32x128 geometry as default
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or replace or sampling_options or rows_end or streaming_several" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -m gpu -x -q -k "fused_residual or full_size_decoder or decoder_stages or vocoder or stream" 2>&1 | tail -3
python tools/prof_decode.py 640 5 | tail -1
