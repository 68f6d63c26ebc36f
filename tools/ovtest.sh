mkdir -p gpurun_out/r1b
run() { python bench.py --no-cpu-baseline --steps 2 --ttfa-reps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step']), {k: round(v) for k, v in d['stage_ms'].items()})"; }
run prio
Q3_DECODE_CUS=64 run cus64
Q3_DECODE_CUS=96 run cus96
Q3_DECODE_CUS=128 run cus128
Q3_DECODE_CUS=32 run cus32
