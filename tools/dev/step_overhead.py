"""Dev aid: where a bench step's wall time goes outside the three timed stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: F401
import qwen3_tts_rs_amd as q
from common import synthetic_prompt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2 and sys.argv[2] == "torchctx":       # mimic bench.py: torch's CUDA context is live in the process
    torch.cuda.set_device(0); torch.cuda.synchronize(0); _t = torch.zeros(1 << 20, device="cuda:0")
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b())
utts = [q.Utterance(synthetic_prompt(512, i), seed=42 + i) for i in range(B)]
opts = q.SynthesisOptions(max_length=640, eos_token_id=None, seed=42)
for rep in range(3):
    t0 = time.perf_counter(); s = m.session(utts, opts); t1 = time.perf_counter()
    tm = s.run_timing_only(use_graph=True); t2 = time.perf_counter()
    s.close(); t3 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.1f} ms, run {1e3*(t2-t1):.1f} ms (stages {tm.prefill_ms:.1f} + {tm.generation_ms:.1f} + {tm.decode_ms:.1f} = {tm.prefill_ms+tm.generation_ms+tm.decode_ms:.1f}), close {1e3*(t3-t2):.1f} ms")
