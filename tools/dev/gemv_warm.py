"""Dev experiment: GEMV launch time with the weights L2-warm (1 copy replayed) vs cold (many copies cycled)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qwen3_tts_rs_amd as q
lib = q._lib.lib
SHAPES = [("talker qkv", 4096, 2048, 0, 1), ("talker o", 2048, 2048, 1, 0), ("talker gate/up", 6144, 2048, 3, 1), ("talker down", 2048, 6144, 1, 0),
          ("cp qkv", 4096, 1024, 0, 1), ("cp o", 1024, 2048, 1, 0), ("cp gate/up", 3072, 1024, 3, 1), ("cp down", 1024, 3072, 1, 0), ("cp lm_head", 2048, 1024, 0, 1)]
for M in (1, 8):
    for name, N, K, epi, rms in SHAPES:
        nbytes = N * K * 2 * (2 if epi == 3 else 1)
        row = f"M={M} {name:12s} {nbytes/1e6:5.1f} MB |"
        for copies in (max(2, int(600e6 // nbytes)), max(2, int(120e6 // nbytes)), 2, 1):
            best = 1e30
            for _ in range(3):
                us = ctypes.c_double()
                assert lib.q3_bench_linear(0, M, N, K, epi, rms, -1, 200, copies, ctypes.byref(us)) == 0
                best = min(best, us.value)
            row += f" copies {copies:3d}: {best:6.2f} us |"
        print(row, flush=True)
