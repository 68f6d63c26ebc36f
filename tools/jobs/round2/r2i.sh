#!/bin/bash
# round 2, GPU call I: HIP runtime knobs that could change the per-graph-node cost of the 565-launch frame (kernarg placement,
# cache-flush scope at kernel boundaries, graph packet capture, dispatch path). Metric: generation ms per frame, B = 8, 200 frames.
O=gpurun_out/r2i; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 1 --warmup 1 --frames 200 --no-cpu-baseline --also-batches "" --ttfa-reps 1 > $O/b_$tag.json 2> $O/b_$tag.err
  python - $tag <<'PY'
import json, sys
try:
    d=json.loads(open(f"gpurun_out/r2i/b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:34s} ms/frame {d['stage_ms']['generation_ms']/200:7.4f}  b1 ms/frame {d['latency'].get('b1_ms_per_frame',0):7.4f}  ttfa {d['latency'].get('ttfa_ms_p50',0):6.2f}")
except Exception as e:
    print(sys.argv[1], "ERR", e, open(f"gpurun_out/r2i/b_{sys.argv[1]}.err").read()[-300:])
PY
}
run baseline A=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
run opt_flush_0 AMD_OPT_FLUSH=0
run opt_flush_1 AMD_OPT_FLUSH=1
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph_batch_1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graph_batch_4096 DEBUG_HIP_GRAPH_BATCH_SIZE=4096
run hdp_flush_wa_0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run hdp_flush_wa_1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run kernarg_copy_opt_0 DEBUG_HIP_KERNARG_COPY_OPT=0
run flush_on_exec_1 GPU_FLUSH_ON_EXECUTION=1
run direct_dispatch_0 AMD_DIRECT_DISPATCH=0
run hw_queues_1 GPU_MAX_HW_QUEUES=1
run baseline2 A=1
