#!/bin/bash
# round 6 / N4: k_sample with the wave-0 tail on SGPR loop bounds: isolated duration (variants 0 = one-pass, 1 = radix select), the sampler parity
# tests, the headline step both ways
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/sp; ( cd /tmp && Q3_SAMPLE_SLOW_TOPK=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -o t -- python "$GRAFT_REPO_ROOT/tools/dev/sample_probe.py" > /tmp/sp.log 2>&1 )
  f=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "k_sample" in r["Kernel_Name"]]
d=sorted(d[5:]); print(f"variant {sys.argv[2]}: {len(d)} launches, median {d[len(d)//2]/1e3:.2f} us, min {d[0]/1e3:.2f}")
PY
done > gpurun_out/r6/n4_sampler.txt 2>&1
cat gpurun_out/r6/n4_sampler.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "sampler or free_run or sampling_options or b8_b16_graph or kat" > gpurun_out/r6/n4_tests.txt 2>&1; tail -3 gpurun_out/r6/n4_tests.txt
Q3_SAMPLE_SLOW_TOPK=1 timeout 600 python -m pytest tests -m gpu -q -x -k "sampler or free_run_codes or sampling_options" > gpurun_out/r6/n4_tests_slow.txt 2>&1; tail -1 gpurun_out/r6/n4_tests_slow.txt
for v in "Q3_SAMPLE_SLOW_TOPK=1" "Q3_X=0" "Q3_SAMPLE_SLOW_TOPK=1" "Q3_X=0"; do
  env $v python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'], round(d['stage_ms']['generation_ms']/640,4))"
done >> gpurun_out/r6/n4_sampler.txt 2>&1
tail -4 gpurun_out/r6/n4_sampler.txt
