"""TTFA breakdown (development aid): session creation, prefill, first chunk's frames, first chunk's decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401
import numpy as np
import qwen3_tts_rs_amd as q
from common import synthetic_prompt
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b())
ids = synthetic_prompt(512, 0)
opts = q.SynthesisOptions(max_length=30, eos_token_id=None, seed=42, chunk_frames=10)
for rep in range(4):
    t0 = time.perf_counter()
    s = m.session([q.Utterance(ids, seed=42)], opts)
    t1 = time.perf_counter(); s.prefill()
    t2 = time.perf_counter(); s.generate(10)
    t3 = time.perf_counter(); pcm = s.decode(0, 0, 10)
    t4 = time.perf_counter(); s.close()
    print(f"create {1e3*(t1-t0):.2f} ms, prefill {1e3*(t2-t1):.2f}, 10 frames {1e3*(t3-t2):.2f}, decode(10) {1e3*(t4-t3):.2f}, total {1e3*(t4-t0):.2f}")
    t0 = time.perf_counter()
    ss = m.synthesize_streaming(ids, q.Speaker.Ryan, q.Language.English, opts); c = ss.next_chunk()
    print(f"   streaming TTFA {1e3*(time.perf_counter()-t0):.2f} ms"); ss._s.close()
