#!/bin/bash
# round 5: KV budget / batcher admission / ragged first batch — the tests that cover the engine changes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_paged_kv.py tests/test_ragged_batch.py -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/r5/b1_paged_ragged.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch or batcher or continuous or replace or rows_with or icl" 2>&1 | tail -30 | tee gpurun_out/r5/b1_parity_subset.txt
