#!/bin/bash
# round 3, job 6l: evidence refresh — default bench line (as the driver runs it), rocprof kernel stats of the bench command, smoke, GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6l; export TMPDIR=/tmp
O=gpurun_out/r6l
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "default rc=$?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-batches "" --no-other-configs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py $O/prof 1 > $O/kernel_stats.txt 2>&1; rm -rf $O/prof
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6l/bench_default.json").read().strip().splitlines()[-1])
print("fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d["stage_ms"], {k: round(v, 2) for k, v in d["latency"].items() if isinstance(v, (int, float))})
r = d["roofline"]; print("roofline frac", round(r["frac"], 4), "avg us", round(r["avg_launch_us"], 2), "launches", r["launches_per_frame"], "traffic", r["traffic"], r["traffic_source"], "rocprof", r.get("rocprof_in_situ"))
print("other_batches", {k: round(v.get("frames_per_s", 0), 1) for k, v in d["other_batches"].items()})
print("other_configs", {k: (round(v.get("frames_per_s", 0), 1), round(v.get("ms_per_frame", 0), 3)) if "error" not in v else v for k, v in d["other_configs"].items()})
print("eos_mix", d["eos_mix"]); print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
PY
head -14 $O/kernel_stats.txt
python -c "import __graft_entry__ as g; g.smoke()"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
