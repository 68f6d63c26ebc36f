#!/bin/bash
# round 2, GPU call K: decode-attention KV splits at B = 8 (is the merge launch worth its splits?)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2k; mkdir -p $O
for ns in 1 2 4 8 16 32; do
  Q3_ATTN_SPLITS=$ns timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-batches "" --ttfa-reps 1 > $O/b_$ns.json 2> $O/b_$ns.err
  python - $ns <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r2k/b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"splits {sys.argv[1]:>2s}: ms/frame {d['stage_ms']['generation_ms']/640:7.4f}  b1 {d['latency'].get('b1_ms_per_frame',0):7.4f} ttfa {d['latency'].get('ttfa_ms_p50',0):6.2f}")
PY
done
