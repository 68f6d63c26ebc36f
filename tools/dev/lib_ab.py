#!/usr/bin/env python3
"""A/B of two builds of libq3tts.so on the bench session (1.7B, B rows, 512-token prompts): each library runs in its own
process (Q3TTS_LIB), prints ms/frame and a digest of the codes.   usage: lib_ab.py libA.so libB.so [--batch 8] [--frames 300] [--reps 3]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+"); ap.add_argument("--model", default="1.7b"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--rounds", type=int, default=2); ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if a.child:
    sys.path.insert(0, ROOT)
    import time, hashlib
    import numpy as np
    import qwen3_tts_rs_amd as q
    from qwen3_tts_rs_amd import synth
    cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b}[a.model]()
    model = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=synth.DEFAULT_SEED)
    utts = [q.Utterance(synth.synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(a.batch)]
    opts = q.SynthesisOptions(max_length=a.frames, eos_token_id=None, seed=42)
    best = 1e9
    for r in range(a.reps):
        s = model.session(utts, opts); s.prefill()
        t0 = time.perf_counter(); s.generate(a.frames, use_graph=True); dt = time.perf_counter() - t0
        codes = np.stack([s.codes(b) for b in range(a.batch)]); path = s.submit_info()[0]; s.close()
        best = min(best, dt)
    print(f"{best * 1e3 / a.frames:.4f} ms/frame  path {path}  codes {hashlib.sha1(codes.tobytes()).hexdigest()[:12]}", flush=True)
    sys.exit(0)
for rnd in range(a.rounds):
    for lib in a.libs:
        env = dict(os.environ); env["Q3TTS_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), lib, "--child", "--model", a.model, "--batch", str(a.batch), "--frames", str(a.frames), "--reps", str(a.reps)],
                           capture_output=True, text=True, env=env, timeout=900)
        out = [l for l in r.stdout.splitlines() if "ms/frame" in l]
        print(f"{os.path.basename(lib):28s} {out[-1] if out else 'FAILED: ' + r.stderr[-300:]}", flush=True)
