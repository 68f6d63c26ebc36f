"""Wall time of q3_session_replace at the bench configuration (1.7B, 8 rows, 512-token prompts): what a swap costs the other rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
from common import synthetic_prompt
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), seed=synth.DEFAULT_SEED)
def utt(i, n=512):
    return q.Utterance(synthetic_prompt(n, i), q.Speaker.Ryan, q.Language.English, seed=42 + i)
opts = q.SynthesisOptions(max_length=640, eos_token_id=None, seed=42)
s = m.session([utt(i) for i in range(8)], opts); s.prefill(); s.generate(16, use_graph=True)
for n in (512, 512, 512, 64, 64, 1000):
    t0 = time.perf_counter(); s.replace(3, utt(20, n)); t1 = time.perf_counter()
    s.generate(4, use_graph=True); t2 = time.perf_counter()
    print(f"replace with a {n}-token prompt: {1e3 * (t1 - t0):.2f} ms; next 4 frames {1e3 * (t2 - t1):.2f} ms")
s.close(); m.close()
