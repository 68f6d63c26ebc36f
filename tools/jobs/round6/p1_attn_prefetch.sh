#!/bin/bash
# round 6 / P1: the talker's attention launch touches the o-projection's weight tiles into the L2 of the XCD that will stream them (eight extra
# planes of workgroups): headline step with it and without (Q3_ATTN_PREFETCH=0), alternating; then the talker parity tests with it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for v in "Q3_ATTN_PREFETCH=0" "Q3_ATTN_PREFETCH=1" "Q3_ATTN_PREFETCH=0" "Q3_ATTN_PREFETCH=1"; do
  env $v python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'], round(d['stage_ms']['generation_ms']/640,4))"
done > gpurun_out/r6/p1_attn_prefetch.txt 2>&1
cat gpurun_out/r6/p1_attn_prefetch.txt
timeout 1500 python -m pytest tests/test_bench_config_parity.py tests/test_frame_submission.py -m gpu -q -x > gpurun_out/r6/p1_tests.txt 2>&1; tail -3 gpurun_out/r6/p1_tests.txt
