"""Long-prompt prefill timing (development aid): prof_prefill.py <model> <prompt positions> [B]. Q3_PREFILL_GEMM_MIN=0 selects
the chunked decode-step schedule (16 rows per weight pass) instead of the GEMM path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401
import qwen3_tts_rs_amd as q
from common import synthetic_prompt
model = sys.argv[1] if len(sys.argv) > 1 else "1.7b"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b, "tiny": q.tiny}[model]()
m = q.Qwen3TTS.from_synthetic(cfg)
utts = [q.Utterance(synthetic_prompt(16, i), language=q.Language.German, instruct_ids=synthetic_prompt(n, 30 + i), seed=5 + i) for i in range(B)]
opts = q.SynthesisOptions(max_length=8, seed=5, eos_token_id=None)
for rep in range(3):
    s = m.session(utts, opts)
    t0 = time.perf_counter(); s.prefill(); dt = time.perf_counter() - t0
    S, _ = s.prefill_len(0)
    s.generate(8); c = s.codes(0)
    s.close()
    print(f"model {model} B {B} prefill positions {S}: {dt*1e3:.1f} ms ({S*B/dt:.0f} positions/s), first codes {c[0][:4].tolist()}", flush=True)
