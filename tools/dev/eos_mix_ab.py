#!/usr/bin/env python3
"""bench.py's `eos_mix` leg alone (32 requests with lengths uniform in 100 .. 640 frames through 8 rows of the native batcher,
8-frame steps): useful frames / wall, continuous vs until-the-queue-ran-dry. Run it once per setting of Q3_BAT_NO_STAGE (the
switch is read once per process):   python tools/dev/eos_mix_ab.py [reps=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if os.environ.get("EOS_MIX_TORCH") == "1":
    import torch      # bench.py loads torch first: its bundled HIP runtime then serves libq3tts.so too (tests/conftest.py)
    torch.cuda.set_device(0)
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
from qwen3_tts_rs_amd.synth import synthetic_prompt
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, FR = 8, 640
model = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), device=0, seed=synth.DEFAULT_SEED)
opts = q.SynthesisOptions(max_length=FR, eos_token_id=None, seed=42)
rng = np.random.default_rng(2026)
lens = [int(x) for x in rng.integers(100, FR + 1, size=4 * B)]
mix = []
for i, L in enumerate(lens):
    u = q.Utterance(synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i); u.max_length = L
    mix.append(u)
def run(reqs, poll, want_pcm=os.environ.get('EOS_MIX_PCM') == '1'):
    bt = q.Batcher(model, slots=B, frame_budget=FR, prompt_budget=0, options=opts)
    try:
        ta = time.perf_counter()
        tickets = [bt.submit(u, want_pcm=want_pcm) for u in reqs]
        steady = None
        while True:
            running, queued, _ = bt.step(poll, True)
            if queued == 0 and steady is None:
                steady = (sum(bt.poll(t)[1] for t in tickets), time.perf_counter() - ta)
            if running == 0 and queued == 0:
                break
        frames = sum(int(bt.fetch(t)[0].shape[0]) for t in tickets)
        wall = time.perf_counter() - ta
        return frames, wall, steady
    finally:
        bt.close()
run(mix[:B + 2], 8)
for r in range(reps):
    fr, wall, steady = run(mix, 8)
    print(f"Q3_BAT_NO_STAGE={os.environ.get('Q3_BAT_NO_STAGE', '-')} Q3_BAT_SYNC_DECODE={os.environ.get('Q3_BAT_SYNC_DECODE', '-')} pcm={os.environ.get('EOS_MIX_PCM', '0')}: continuous {fr / wall:.1f} frames/s ({fr} frames, {wall * 1e3:.1f} ms), until the queue ran dry {steady[0] / steady[1]:.1f}", flush=True)
