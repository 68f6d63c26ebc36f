#!/bin/bash
# round 3, job l: waves per workgroup of the 16-row-tile GEMVs, re-measured under the bytes-in-flight reading of the timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for e in "Q3_X=1" "Q3_GEMV_WAVES=8" "Q3_GEMV_WAVES=16"; do echo "== $e"; env $e python tools/prof_run.py 1.7b 8 300 | tail -1; done
for e in "Q3_X=1" "Q3_GEMV_WAVES=8" "Q3_GEMV_WAVES=16"; do echo "== $e"; env $e Q3_BENCH_M=8 python tools/bench_kernels.py 2>&1 | cut -c1-100 | grep "talker\|cp qkv\|cp gate"; done
