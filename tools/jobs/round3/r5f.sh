#!/bin/bash
# round 3, job f: attn_cp on VALU cross-lane ops; kernarg preload A/B (library built without the flag)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -m gpu -x -q 2>&1 | tail -6
for rep in 1 2; do
  echo "== default"; python tools/prof_run.py 1.7b 8 300 | tail -1
  echo "== no preload"; Q3TTS_LIB=$GRAFT_REPO_ROOT/qwen3_tts_rs_amd/libq3tts_nopre.so python tools/prof_run.py 1.7b 8 300 | tail -1
done
timeout 600 python tools/trace_frame.py 1.7b 8 64 512 --full > gpurun_out/r5f_trace_b8.txt 2>&1
grep -A14 "mean per kernel" gpurun_out/r5f_trace_b8.txt | cut -c1-250
grep "attn_cp B8 splits1 pos1[0-5] " gpurun_out/r5f_trace_b8.txt | tail -8 | cut -c1-200
tail -1 gpurun_out/r5f_trace_b8.txt
