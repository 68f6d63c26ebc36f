#!/bin/bash
# round 6 / N1: the sampler's one-pass top-k selection (five block-wide barriers instead of 21): sampler / free-run / per-row-options parity
# tests, then the headline step with it and with the four-pass radix select (Q3_SAMPLE_SLOW_TOPK=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests -m gpu -q -x -k "sampler or free_run or sampling_options or b8_b16_graph or kat" > gpurun_out/r6/n1_tests.txt 2>&1; tail -4 gpurun_out/r6/n1_tests.txt
for v in "Q3_SAMPLE_SLOW_TOPK=1" "Q3_X=0" "Q3_SAMPLE_SLOW_TOPK=1" "Q3_X=0"; do
  env $v python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'], round(d['stage_ms']['generation_ms']/640,4))"
done > gpurun_out/r6/n1_sampler_ab.txt 2>&1
cat gpurun_out/r6/n1_sampler_ab.txt
