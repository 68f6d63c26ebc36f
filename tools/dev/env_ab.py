#!/usr/bin/env python3
"""A/B of environment knobs on the bench session (1.7B, B rows, 512-token prompts), each setting in its own process, alternating:
   env_ab.py "" "Q3_GEMV_NO_FULL=1" [--batch 8] [--frames 300] [--reps 3] [--rounds 2]      ("" = the default build / settings)"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+"); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()
child = os.path.join(ROOT, "tools", "dev", "lib_ab.py")
for rnd in range(a.rounds):
    for st in a.settings:
        env = dict(os.environ)
        for kv in st.split():
            k, v = kv.split("=", 1); env[k] = v
        lib = env.get("Q3TTS_LIB", os.path.join(ROOT, "qwen3_tts_rs_amd", "libq3tts.so"))
        r = subprocess.run([sys.executable, child, lib, "--child", "--batch", str(a.batch), "--frames", str(a.frames), "--reps", str(a.reps)],
                           capture_output=True, text=True, env=env, timeout=900)
        out = [l for l in r.stdout.splitlines() if "ms/frame" in l]
        print(f"{st or '(default)':40s} {out[-1] if out else 'FAILED: ' + r.stderr[-300:]}", flush=True)
