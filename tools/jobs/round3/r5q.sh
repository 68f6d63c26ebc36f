#!/bin/bash
# round 3, job q: GPU suite in two file orders (flakiness of the atomic / zero-job paths), then the ICL replace test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py tests/test_speech_encoder.py tests/test_speaker_encoder.py tests/test_examples.py tests/test_cli.py tests/test_c_host.py tests/test_io_formats.py -m gpu -q -p no:randomly 2>&1 | grep "passed\|failed\|Error" | tail -3
timeout 1500 python -m pytest tests/test_bench_config_parity.py tests/test_gpu_parity.py -m gpu -q 2>&1 | grep "passed\|failed\|Error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
