"""Vocoder-only profiling driver: decode T random frames a few times (run under rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qwen3_tts_rs_amd as q
T = int(sys.argv[1]) if len(sys.argv) > 1 else 640
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = q.tiny(); full = q.qwen3_tts_0_6b()
for f in ("dec_cb_dim", "dec_q_dim", "dec_latent", "dec_hidden", "dec_layers", "dec_heads", "dec_inter", "dec_dim"):
    setattr(cfg, f, getattr(full, f))          # tiny LM + full-size decoder: fast to build
m = q.Qwen3TTS.from_synthetic(cfg)
rng = np.random.default_rng(0)
codes = rng.integers(0, 2048, size=(T, 16)).astype(np.uint32)
m.decode_codes(codes)
t0 = time.time()
for _ in range(reps):
    m.decode_codes(codes)
print(f"decode T={T}: {(time.time() - t0) / reps * 1e3:.1f} ms per call (incl. alloc + copies)")
