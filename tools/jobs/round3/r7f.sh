#!/bin/bash
# round 3, job 7f: L2 prefetch of the next projection's weights (LinArgs::pf) — frame times with / without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" "Q3_NO_L2_PREFETCH=1" "" "Q3_NO_L2_PREFETCH=1"; do echo "== $v"; env $v python tools/prof_run.py 1.7b 8 300 2>&1 | tail -1; done
for v in "" "Q3_NO_L2_PREFETCH=1"; do echo "== B=1 $v"; env $v python tools/prof_run.py 1.7b 1 300 2>&1 | tail -1; done
for v in "" "Q3_NO_L2_PREFETCH=1"; do echo "== B=16 $v"; env $v python tools/prof_run.py 1.7b 16 200 2>&1 | tail -1; done
for v in "" "Q3_NO_L2_PREFETCH=1"; do echo "== 0.6b B=1 $v"; env $v python tools/prof_run.py 0.6b 1 300 2>&1 | tail -1; done
