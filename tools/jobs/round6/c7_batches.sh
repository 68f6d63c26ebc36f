#!/bin/bash
# round 6 / C7: frames/s of one session at 1 / 16 / 32 / 64 utterances on the final tree (headline step only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for b in 1 16 32 64; do
  python bench.py --headline-only --batch $b --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$b', round(d['value'],1), 'frames/s', round(d['roofline']['frame_ms'],3), 'ms/frame', d['stage_ms'], d['config'].get('frame_packets'), d['config'].get('frame_packets_without_acquire_release_fence'))"
done > gpurun_out/r6/c7_batches.txt 2>&1; cat gpurun_out/r6/c7_batches.txt
