"""Dev aid: frame time of the graph-replayed decode behind a long prompt (BASELINE config[4]): long_ctx_decode.py <positions> [B] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: F401
import qwen3_tts_rs_amd as q
from common import synthetic_prompt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b())
utts = [q.Utterance(synthetic_prompt(16, i), language=q.Language.German, instruct_ids=synthetic_prompt(n, 30 + i), seed=5 + i) for i in range(B)]
s = m.session(utts, q.SynthesisOptions(max_length=frames + 8, seed=5, eos_token_id=None))
t0 = time.perf_counter(); s.prefill(); t1 = time.perf_counter()
s.generate(8); t2 = time.perf_counter()
s.generate(frames); t3 = time.perf_counter()
print(f"prompt {s.prefill_len(0)[0]} positions, B {B}: prefill {1e3*(t1-t0):.1f} ms, first 8 frames {1e3*(t2-t1):.1f} ms, then {1e3*(t3-t2)/frames:.3f} ms/frame")
