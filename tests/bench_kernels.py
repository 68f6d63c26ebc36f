"""Kernel micro-benchmarks (development aid): GEMV shapes of the 1.7B talker / code predictor, both
kernel generations, M = 1 and 8. Prints µs per launch (incl. the graph-internal kernel boundary) and GB/s."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_rs_amd as q
lib = q._lib.lib
lib.q3_bench_linear.restype = ctypes.c_int
lib.q3_bench_linear.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_double)]

SHAPES = [  # name, N, K, epi, rms
    ("talker qkv", 4096, 2048, 0, 1), ("talker o", 2048, 2048, 1, 0), ("talker gate/up", 6144, 2048, 3, 1),
    ("talker down", 2048, 6144, 1, 0), ("codec head", 3072, 2048, 0, 0),
    ("cp qkv", 4096, 1024, 0, 1), ("cp o", 1024, 2048, 1, 0), ("cp gate/up", 3072, 1024, 3, 1), ("cp down", 1024, 3072, 1, 0),
    ("cp mtp proj", 1024, 2048, 0, 0), ("cp lm_head", 2048, 1024, 0, 1),
]
which = sys.argv[1:] or ["0", "1"]
Ms = [int(m) for m in os.environ.get("Q3_BENCH_M", "1,8").split(",")]
REPS = int(os.environ.get("Q3_BENCH_REPS", "3"))
for name, N, K, epi, rms in SHAPES:
    nbytes = N * K * 2 * (2 if epi == 3 else 1)
    copies = max(2, int(600e6 // nbytes))
    row = f"{name:16s} N={N:5d} K={K:5d} {nbytes / 1e6:6.1f} MB |"
    for tiled in which:
        for M in Ms:
            us = ctypes.c_double(); best = 1e30; st = 0
            for _ in range(REPS):      # min over repetitions: box-to-box / clock noise is ~0.5 us
                st = lib.q3_bench_linear(0, M, N, K, epi, rms, int(tiled), 200, copies, ctypes.byref(us))
                if st != 0:
                    break
                best = min(best, us.value)
            us.value = best
            if st != 0:
                row += f" t{tiled} M{M}: ERR {lib.q3_last_error().decode()[:40]} |"
            else:
                row += f" t{tiled} M{M}: {us.value:6.2f} us {nbytes / us.value / 1e3:6.0f} GB/s |"
    print(row, flush=True)
