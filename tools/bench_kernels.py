"""Kernel micro-benchmarks (development aid): GEMV shapes of the 1.7B talker / code predictor at M = 1 / 8 (Q3_BENCH_M).
Prints µs per launch (mean of 5 graph replays, incl. the graph-internal kernel boundary) and GB/s.
Columns: t-1 = the engine's tiling choice, t1 = 16-row tiles, t2 = 4-row tiles. Environment (tuning aids of the launcher):
Q3_GEMV_NO_HALF=1 two-instruction x loads, Q3_GEMV_NO_LDS=1 no LDS-staged kernel, Q3_GEMV_BIG8=1 long-K shapes on 8 waves,
Q3TTS_LIB=build/libq3tts_nopipe.so the build without the software-pipelined groups."""
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
        for tiled in (-1, 1, 2):          # the engine's choice, 16-row tiles, 4-row tiles
            if tiled == 2 and (N >= 4096 or rmsc):
                continue
            us = run(M, N, K, epi, 1 if rmsc else 0, tiled)
            row += f" M{M} t{tiled}: " + ("ERR" if us is None else f"{us:6.2f} us {nbytes / us / 1e3:5.0f} GB/s") + " |"
    print(row, flush=True)
