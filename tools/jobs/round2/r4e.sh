#!/bin/bash
# r4e: GPU suite in other file orders (the round's over-read surfaced in one order only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4e.txt; : > $O
flt() { grep -v "^  File\|Extension modules\|RCCL\|Librccl\|HIP version\|ROCm version\|Hostname\|^$" "$1" | tail -3 | cut -c1-200; }
timeout 1500 python -m pytest $(ls tests/test_*.py | sort -r) -q -m gpu > gpurun_out/r4e_rev.log 2>&1; echo "reversed rc=$?" >> $O; flt gpurun_out/r4e_rev.log >> $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_speech_encoder.py tests/test_speaker_encoder.py tests/test_examples.py tests/test_cli.py tests/test_c_host.py tests/test_bench_config_parity.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r4e_mix.log 2>&1; echo "mixed (gpu_parity twice) rc=$?" >> $O; flt gpurun_out/r4e_mix.log >> $O
cat $O
