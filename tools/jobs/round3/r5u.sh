#!/bin/bash
# round 3, job u: vocoder — fused 96-channel unit only, small-linear column tiles, K=2 stage width A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_residual or full_size_decoder or decoder_stages or stream or chunk or segment" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "vocoder or stream" 2>&1 | tail -3
for v in "" "Q3_CONV_NO_SMALL_TILES=1" "Q3_CONV_K2_CIS=64" "Q3_CODEC_UNIT_FUSE_192=1"; do
  echo "== $v"; env $v python tools/prof_decode.py 640 5 | tail -1
done
bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/r5u_vocoder_T640.txt; head -14 gpurun_out/r5u_vocoder_T640.txt | cut -c1-150
Q3_CONV_K2_CIS=64 bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
grep "k_conv_bf16x3<2" gpurun_out/vocprof/vocoder_T640.txt | head -8 | cut -c1-150
