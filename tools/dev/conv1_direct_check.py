#!/usr/bin/env python3
"""k_conv1_direct (the decoder blocks' 1x1 residual convs without LDS staging) against k_conv_bf16x3: the same codes decoded with
Q3_CONV1_DIRECT=0 and =1 (the switch is read once per process: one child each), PCM compared BIT FOR BIT, decode time per call.
   usage: conv1_direct_check.py [T=640] [reps=5]      (run ON the GPU box)"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 640
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
CHILD = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import numpy as np
import qwen3_tts_rs_amd as q
T, reps = int(sys.argv[1]), int(sys.argv[2])
cfg = q.tiny(); full = q.qwen3_tts_0_6b()
for f in ("dec_cb_dim", "dec_q_dim", "dec_latent", "dec_hidden", "dec_layers", "dec_heads", "dec_inter", "dec_dim"): setattr(cfg, f, getattr(full, f))
m = q.Qwen3TTS.from_synthetic(cfg)
codes = np.random.default_rng(0).integers(0, 2048, size=(T, 16)).astype(np.uint32)
pcm = m.decode_codes(codes).samples
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter(); m.decode_codes(codes); best = min(best, time.perf_counter() - t0)
print("RESULT", hashlib.sha1(np.ascontiguousarray(pcm).tobytes()).hexdigest(), f"{best * 1e3:.2f}", float(np.abs(pcm).max()), flush=True)
''' % ROOT
out = {}
for mode in ("0", "1", "0", "1"):
    env = dict(os.environ); env["Q3_CONV1_DIRECT"] = mode
    r = subprocess.run([sys.executable, "-c", CHILD, str(T), str(reps)], capture_output=True, text=True, env=env, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(f"Q3_CONV1_DIRECT={mode}: FAILED\n{r.stderr[-1500:]}"); sys.exit(1)
    _, digest, ms, peak = line[-1].split()
    print(f"Q3_CONV1_DIRECT={mode}: decode of {T} frames {ms} ms per call (incl. upload / copy-out), PCM sha1 {digest[:16]}, peak {peak}", flush=True)
    out.setdefault(mode, set()).add(digest)
same = out["0"] == out["1"] and len(out["0"]) == 1
print("PCM bit-identical" if same else "PCM DIFFERS")
sys.exit(0 if same else 2)
