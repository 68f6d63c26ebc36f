#!/usr/bin/env python3
"""A/B of the wide-session SwiGLU projection forms (Q3_WIDE2=0: split-K GEMM + slice-sum launch; 1: x split once + single-launch GEMM):
isolated launch times of the two gate/up shapes at M rows, frame time of a B-row session, and how many codes differ between the forms
(the summation order differs: near-ties may flip).   usage: wide2_ab.py [--batch 64] [--frames 160]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--frames", type=int, default=160); ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if a.child:
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import synth
    from qwen3_tts_rs_amd.api import bench_linear
    for (N, K) in ((3072, 1024), (6144, 2048)):
        print(f"gate/up M={a.batch} N={N} K={K}: {bench_linear(a.batch, N, K, 3, True, tiled=1, device=0):.2f} us", flush=True)
    cfg = q.qwen3_tts_1_7b()
    model = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=synth.DEFAULT_SEED)
    utts = [q.Utterance(synth.synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(a.batch)]
    opts = q.SynthesisOptions(max_length=a.frames, eos_token_id=None, seed=42)
    best = 1e9
    for r in range(2):
        s = model.session(utts, opts); s.prefill()
        t0 = time.perf_counter(); s.generate(a.frames, use_graph=True); dt = time.perf_counter() - t0
        codes = np.stack([s.codes(b) for b in range(a.batch)]); s.close()
        best = min(best, dt)
    np.save(f"/tmp/wide2_codes_{os.environ.get('Q3_WIDE2', '1')}{os.environ.get('Q3_WIDE_OPLANES', '')}.npy", codes)
    print(f"qkv M={a.batch} N=4096 K=1024: {bench_linear(a.batch, 4096, 1024, 0, True, tiled=1, device=0):.2f} us  K=2048: {bench_linear(a.batch, 4096, 2048, 0, True, tiled=1, device=0):.2f} us", flush=True)
    print(f"session B={a.batch}: {best * 1e3 / a.frames:.3f} ms/frame", flush=True)
    sys.exit(0)
import numpy as np
for rnd in range(2):
    for mode in os.environ.get("WIDE2_MODES", "0,1").split(","):
        env = dict(os.environ); env["Q3_WIDE2"] = mode[0]
        if len(mode) > 1: env["Q3_WIDE_OPLANES"] = mode[1]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(a.batch), "--frames", str(a.frames)], capture_output=True, text=True, env=env, timeout=900)
        print(f"Q3_WIDE2={mode}: " + " | ".join(l for l in r.stdout.splitlines() if "us" in l or "ms/frame" in l) + ("" if r.returncode == 0 else "  FAILED " + r.stderr[-400:]), flush=True)
ms = os.environ.get("WIDE2_MODES", "0,1").split(",")
c0, c1 = np.load(f"/tmp/wide2_codes_{ms[0]}.npy"), np.load(f"/tmp/wide2_codes_{ms[-1]}.npy")
rows_equal = int((c0 == c1).all(axis=(1, 2)).sum())
first = [int(np.argmax((c0[b] != c1[b]).any(axis=1))) if (c0[b] != c1[b]).any() else -1 for b in range(c0.shape[0])]
print(f"codes: {rows_equal} of {c0.shape[0]} rows identical over {c0.shape[1]} frames; first differing frame per row (-1 = none): {first}")
