#!/bin/bash
# round 3, job t: fused residual unit (96 / 192 channels) — bit identity, oracle taps, per-launch table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_residual or full_size_decoder or decoder_stages" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "vocoder" 2>&1 | tail -3
bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/r5t_vocoder_T640.txt; grep -v "^ *[0-9]* k_conv_bf16x3<1, 1, 1\|k_norm\|k_rope\|k_silu\|k_dwconv\|k_rvq" gpurun_out/r5t_vocoder_T640.txt | cut -c1-150 | head -70
