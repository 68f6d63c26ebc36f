#!/bin/bash
# round 6 / B3: the batcher's prefill worker (staged swaps) — tests + the eos_mix leg with and without it; the split-K o-projection
# behind the code predictor's 2-token first pass, A/B on the B = 8 frame at 640 frames.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -m gpu -x -q -k "stream or chunk or batcher or variants or free_run or continuous or ragged or stages_a_4k or b8_b16" > gpurun_out/r6/b3_tests.txt 2>&1
tail -5 gpurun_out/r6/b3_tests.txt
python tools/dev/eos_mix_ab.py 2 > gpurun_out/r6/b3_eos_mix.txt 2>&1; Q3_BAT_NO_STAGE=1 python tools/dev/eos_mix_ab.py 2 >> gpurun_out/r6/b3_eos_mix.txt 2>&1; cat gpurun_out/r6/b3_eos_mix.txt
python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 3,3 2>&1 | grep -v WARNING > gpurun_out/r6/b3_first2_sk.txt
Q3_FIRST2_NO_SK=1 python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 3,3 2>&1 | grep -v WARNING | sed 's/^/Q3_FIRST2_NO_SK=1 /' >> gpurun_out/r6/b3_first2_sk.txt
cat gpurun_out/r6/b3_first2_sk.txt
