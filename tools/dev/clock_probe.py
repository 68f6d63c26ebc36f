#!/usr/bin/env python3
"""What shader clock the chip holds under each phase of the hot path (development aid, run ON the GPU box): a child process loops one
phase — prefill of a 4105-position prompt / the B = 8 frame loop / the 640-frame vocoder — while this process samples
`rocm-smi --showclocks` (sclk) and `--showpower` every 0.25 s.   usage: clock_probe.py [seconds per phase = 6]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
from common import synthetic_prompt
phase, secs = sys.argv[1], float(sys.argv[2])
if phase == "vocoder":
    cfg = q.tiny(); full = q.qwen3_tts_0_6b()
    for f in ("dec_cb_dim", "dec_q_dim", "dec_latent", "dec_hidden", "dec_layers", "dec_heads", "dec_inter", "dec_dim"): setattr(cfg, f, getattr(full, f))
    m = q.Qwen3TTS.from_synthetic(cfg); codes = np.random.default_rng(0).integers(0, 2048, size=(640, 16)).astype(np.uint32)
    m.decode_codes(codes); print("READY", flush=True); t0 = time.time()
    while time.time() - t0 < secs: m.decode_codes(codes)
else:
    m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), seed=synth.DEFAULT_SEED)
    if phase == "prefill":
        u = [q.Utterance(synthetic_prompt(16, 0), language=q.Language.German, instruct_ids=synthetic_prompt(4096, 30), seed=5)]
        o = q.SynthesisOptions(max_length=8, seed=5, eos_token_id=None)
        s = m.session(u, o); s.prefill(); s.close(); print("READY", flush=True); t0 = time.time()
        while time.time() - t0 < secs:
            s = m.session(u, o); s.prefill(); s.close()
    else:
        u = [q.Utterance(synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(8)]
        o = q.SynthesisOptions(max_length=4000, seed=42, eos_token_id=None)
        s = m.session(u, o); s.prefill(); s.generate(50, use_graph=True); print("READY", flush=True); t0 = time.time()
        while time.time() - t0 < secs: s.generate(200, use_graph=True)
''' % (ROOT, ROOT)
def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out); pw = re.search(r"Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None, out)
idle = sample(); print(f"idle: sclk {idle[0]} MHz, power {idle[1]} W")
if idle[0] is None: print(idle[2][:1500])
for phase in ("frames", "prefill", "vocoder"):
    p = subprocess.Popen([sys.executable, "-c", CHILD, phase, str(SECS)], stdout=subprocess.PIPE, text=True)
    while True:
        line = p.stdout.readline()
        if not line or line.startswith("READY"): break
    time.sleep(0.5); s = []
    while p.poll() is None:
        s.append(sample()[:2]); time.sleep(0.25)
    ck = [a for a, b in s if a]; pw = [b for a, b in s if b]
    if ck: print(f"{phase:8s}: sclk median {sorted(ck)[len(ck)//2]} MHz (min {min(ck)}, max {max(ck)}, {len(ck)} samples), power median {sorted(pw)[len(pw)//2] if pw else None} W")
    else: print(f"{phase:8s}: no clock samples")
