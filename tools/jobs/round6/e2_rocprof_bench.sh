#!/bin/bash
# round 6 / E2: the rocprofv3 kernel table of the headline command. With the frames on the library's own AQL queue rocprofv3's queue
# interception faulted in aql_submit (E1: it reads a packet's kernel arguments from the host; ours live in device memory) — taken here
# with Q3_AQL_HOST_KERNARG=1 (pinned host kernargs, the placement HIP itself uses), and with Q3_AQL=0 (hipGraphLaunch) beside it;
# what the kernarg placement costs unprofiled.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final6; mkdir -p $O
Q3_AQL_HOST_KERNARG=1 Q3_PROF_NAME=final6/r6_rocprof_kernel_stats_bench_b8.txt bash tools/prof_bench_b8.sh > /dev/null 2>&1; head -14 $O/r6_rocprof_kernel_stats_bench_b8.txt | cut -c1-150
cp gpurun_out/prof_bench_run.log $O/prof_bench_run_hostkernarg.log
Q3_AQL=0 Q3_PROF_NAME=final6/r6_rocprof_kernel_stats_bench_b8_hipgraph.txt bash tools/prof_bench_b8.sh > /dev/null 2>&1; head -8 $O/r6_rocprof_kernel_stats_bench_b8_hipgraph.txt | cut -c1-150
python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 3,3/Q3_AQL_HOST_KERNARG=1,3,3/Q3_AQL_HOST_KERNARG=1 2>&1 | grep -v WARNING > $O/r6_kernarg_placement_ab.txt; cat $O/r6_kernarg_placement_ab.txt
for v in "Q3_DECODE_PAIRS=1" "Q3_DECODE_PAIRS=3" "Q3_DECODE_PAIRS=2"; do
  env $v python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'], d['config'].get('frame_packets_without_acquire_release_fence'))"
done > $O/decode_pairs_more.txt 2>&1; cat $O/decode_pairs_more.txt
