#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
for v in "" "Q3_CODEC_UNIT_FUSE_192=1"; do
  env $v bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  echo "== ${v:-default}: $(head -1 gpurun_out/vocprof/vocoder_T640.txt)"; grep -E "k_resunit|ILi7|<7, 1, 4, 3" gpurun_out/vocprof/vocoder_T640.txt | head -8
done 2>&1 | tee gpurun_out/r5/f3_fuse192.txt
