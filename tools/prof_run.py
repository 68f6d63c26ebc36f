"""Profiling driver (run under rocprofv3): prefill + N graph-replayed frames + vocoder for one batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401 — loads torch's bundled HIP runtime first: rocprofv3 + hipGraph segfaults with /opt/rocm's (ROCm 7.2)
import qwen3_tts_rs_amd as q
from common import synthetic_prompt

model = sys.argv[1] if len(sys.argv) > 1 else "1.7b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 32
graph = (sys.argv[4] != "eager") if len(sys.argv) > 4 else True
cfg = {"1.7b": q.qwen3_tts_1_7b, "0.6b": q.qwen3_tts_0_6b, "tiny": q.tiny}[model]()
m = q.Qwen3TTS.from_synthetic(cfg)
utts = [q.Utterance(synthetic_prompt(int(os.environ.get("Q3_PROMPT", "512")), i), seed=42 + i) for i in range(B)]
opts = q.SynthesisOptions(max_length=frames, eos_token_id=None, seed=42)
s = m.session(utts, opts)
t0 = time.time(); s.prefill(); t1 = time.time()
s.generate(frames, use_graph=graph); t2 = time.time()
for b in range(B):
    s.decode(b)
t3 = time.time()
print(f"model {model} B {B} frames {frames} graph {graph}: prefill {1e3*(t1-t0):.1f} ms, generate {1e3*(t2-t1):.1f} ms "
      f"({1e3*(t2-t1)/frames:.3f} ms/frame), decode {1e3*(t3-t2):.1f} ms")
