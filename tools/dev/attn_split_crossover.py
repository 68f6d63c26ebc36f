"""ms/frame per 32-frame chunk of a 1.7B session (context = 10 + frames) — run once per Q3_ATTN_SPLITS setting to find where the
split decode attention (+ merge launch) starts to pay. Usage: attn_split_crossover.py <B> [chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
from common import synthetic_prompt
B = int(sys.argv[1]); chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), seed=synth.DEFAULT_SEED)
utts = [q.Utterance(synthetic_prompt(512, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(B)]
opts = q.SynthesisOptions(max_length=32 * chunks, eos_token_id=None, seed=42)
for rep in range(2):
    s = m.session(utts, opts); s.prefill(); s.generate(1, use_graph=True); s.frames(0)
    out = []
    for c in range(chunks):
        t0 = time.perf_counter(); s.generate(32 if c else 31, use_graph=True); s.frames(0); out.append((time.perf_counter() - t0) / (32 if c else 31) * 1e3)
    s.close()
print(f"B={B} splits={os.environ.get('Q3_ATTN_SPLITS', 'default')}: " + " ".join(f"{x:.3f}" for x in out))
m.close()
