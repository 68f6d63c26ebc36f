"""PMC driver: a fixed number of launches of ONE GEMV shape (run under rocprofv3 --pmc ...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd.api import bench_linear
name, N, K, epi, rms, tiled, M = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
us = bench_linear(M, N, K, epi, bool(rms), tiled=tiled, iters=50)       # tiled: -1 = the engine's unsplit choice, 3 = 16-row tiles with split-K in two
print(f"{name} M={M} N={N} K={K}: {us:.2f} us/launch, algorithmic {N*K*2*(2 if epi==3 else 1)} B")
