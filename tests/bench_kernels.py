"""Kernel micro-benchmarks (development aid): GEMV shapes of the 1.7B talker / code predictor at M = 1 / 8 (Q3_BENCH_M).
Prints µs per launch (mean of 5 graph replays, incl. the graph-internal kernel boundary) and GB/s.
Columns: `rms1` = RMSNorm weight applied inside the consuming GEMV, `pre` = producer-side RMSNorm (pre-normed x + partial
sums), `+prod` = the launch also writes z_out / ssq_out. Environment: Q3_GEMV_NO_HALF=1 → two-instruction x loads."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_rs_amd as q
lib = q._lib.lib
lib.q3_bench_linear.restype = ctypes.c_int
lib.q3_bench_linear.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_double)]

SHAPES = [  # name, N, K, epi, rms consumer?, producer?
    ("talker qkv", 4096, 2048, 0, 1, 0), ("talker o", 2048, 2048, 1, 0, 1), ("talker gate/up", 6144, 2048, 3, 1, 0),
    ("talker down", 2048, 6144, 1, 0, 1), ("codec head", 3072, 2048, 0, 0, 0),
    ("cp qkv", 4096, 1024, 0, 1, 0), ("cp o", 1024, 2048, 1, 0, 1), ("cp gate/up", 3072, 1024, 3, 1, 0), ("cp down", 1024, 3072, 1, 0, 1),
    ("cp mtp proj", 1024, 2048, 0, 0, 0), ("cp lm_head", 2048, 1024, 0, 1, 0),
]
Ms = [int(m) for m in os.environ.get("Q3_BENCH_M", "1,8").split(",")]
REPS = int(os.environ.get("Q3_BENCH_REPS", "3"))


def run(M, N, K, epi, rms, tiled=-1):
    nbytes = N * K * 2 * (2 if (epi & 15) == 3 else 1)
    copies = max(2, int(600e6 // nbytes))
    us = ctypes.c_double(); tot = 0.0
    for _ in range(REPS):
        if lib.q3_bench_linear(0, M, N, K, epi, rms, tiled, 200, copies, ctypes.byref(us)) != 0:
            return None
        tot += us.value
    return tot / REPS


for name, N, K, epi, rmsc, prod in SHAPES:
    nbytes = N * K * 2 * (2 if epi == 3 else 1)
    row = f"{name:16s} N={N:5d} K={K:5d} {nbytes / 1e6:6.1f} MB |"
    for M in Ms:
        variants = [("rms1" if rmsc else "base", epi, 1 if rmsc else 0)]
        if rmsc:
            variants.append(("pre", epi, 2))
        if prod:
            variants.append(("+prod", epi | 16, 0))
        for tag, e, r in variants:
            us = run(M, N, K, e, r)
            row += f" M{M} {tag}: " + ("ERR" if us is None else f"{us:6.2f} us {nbytes / us / 1e3:5.0f} GB/s") + " |"
    print(row, flush=True)
