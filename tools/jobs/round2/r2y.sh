#!/bin/bash
# r2y: evidence refresh: default bench line (with cpu_baseline + roofline), other configs, rocprof kernel stats of the bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r2y; export TMPDIR=/tmp
O=gpurun_out/r2y
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
run() { tag=$1; shift; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches "" "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag rc=$?"; }
run cfg1_0.6b_b1 --model 0.6b --batch 1
run cfg3_greedy --sampling greedy
run cfg4_voicedesign4k_b1 --workload voicedesign4k --batch 1
run xvector_b8 --workload xvector
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-batches "" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py $O/prof 1 > $O/kernel_stats.txt 2>&1; rm -rf $O/prof
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2y/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d["stage_ms"], {k: round(v, 2) for k, v in d["latency"].items() if isinstance(v, (int, float))})
    except Exception as e:
        print(f, "ERR", e)
PY
head -30 $O/kernel_stats.txt
