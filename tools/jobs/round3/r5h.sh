#!/bin/bash
# round 3, job h: evidence refresh: default bench line (roofline + cpu_baseline + other_configs), rocprof kernel stats of the bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5h; export TMPDIR=/tmp
O=gpurun_out/r5h
( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "default rc=$?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-batches "" --no-other-configs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py $O/prof 1 > $O/kernel_stats.txt 2>&1; rm -rf $O/prof
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5h/bench_default.json").read().strip().splitlines()[-1])
print("fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d["stage_ms"], {k: round(v, 2) for k, v in d["latency"].items() if isinstance(v, (int, float))})
r = d["roofline"]; print("roofline frac", round(r["frac"], 4), "avg us", round(r["avg_launch_us"], 2), "launches", r["launches_per_frame"], "in_situ", round(r["in_situ"]["frac"], 4), "rocprof", r.get("rocprof_in_situ"))
print("other_batches", {k: round(v.get("frames_per_s", 0), 1) for k, v in d["other_batches"].items()})
print("other_configs", {k: (round(v.get("frames_per_s", 0), 1), round(v.get("ms_per_frame", 0), 3)) if "error" not in v else v for k, v in d["other_configs"].items()})
print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, "rccl", d["rccl"])
for k, v in sorted(r["per_shape"].items(), key=lambda kv: -kv[1].get("us", 0) * kv[1]["launches_per_frame"]): print(f"  {k:60s} {v.get('us', 0):7.2f} us x {v['launches_per_frame']:3d}  {v.get('gbps', 0):7.0f} GB/s")
PY
head -32 $O/kernel_stats.txt
