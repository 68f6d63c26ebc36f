#!/bin/bash
# round 6 / G1: the batcher's decode worker (a finished row's vocoder beside the running frames): batcher / fuzz / soak tests with PCM,
# eos_mix with every finished row vocoded — worker vs synchronous decode (Q3_BAT_SYNC_DECODE=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -x -q -k "batcher or continuous or stages_a_4k or c_host or frame_submission or examples" > gpurun_out/r6/g1_tests.txt 2>&1; tail -4 gpurun_out/r6/g1_tests.txt
timeout 900 python tools/dev/soak_batcher.py 600 30 > gpurun_out/r6/g1_soak.txt 2>&1; tail -2 gpurun_out/r6/g1_soak.txt
EOS_MIX_PCM=1 python tools/dev/eos_mix_ab.py 2 > gpurun_out/r6/g1_eos_mix_pcm.txt 2>&1
EOS_MIX_PCM=1 Q3_BAT_SYNC_DECODE=1 python tools/dev/eos_mix_ab.py 2 >> gpurun_out/r6/g1_eos_mix_pcm.txt 2>&1
cat gpurun_out/r6/g1_eos_mix_pcm.txt
