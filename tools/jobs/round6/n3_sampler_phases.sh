#!/bin/bash
# round 6 / N3: where k_sample's 20 us go — the kernel alone (tools/dev/sample_probe.py) under rocprofv3, cut short at successive points
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
for v in 8 2 16 32 0; do
  rm -rf /tmp/sp; ( cd /tmp && Q3_SAMPLE_SLOW_TOPK=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -o t -- python "$GRAFT_REPO_ROOT/tools/dev/sample_probe.py" > /tmp/sp.log 2>&1 )
  f=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "k_sample" in r["Kernel_Name"]]
d=sorted(d[5:]); print(f"variant {sys.argv[2]}: {len(d)} launches, median {d[len(d)//2]/1e3:.2f} us, min {d[0]/1e3:.2f}")
PY
done > gpurun_out/r6/n3_sampler_phases.txt 2>&1
cat gpurun_out/r6/n3_sampler_phases.txt
