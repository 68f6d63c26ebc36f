"""Dev experiment: two half-batch sessions generating concurrently (two host threads, two streams) vs one full batch."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: F401
import qwen3_tts_rs_amd as q
from common import synthetic_prompt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b())
opts = q.SynthesisOptions(max_length=frames, eos_token_id=None, seed=42)
def mk(lo, hi):
    s = m.session([q.Utterance(synthetic_prompt(512, i), seed=42 + i) for i in range(lo, hi)], opts); s.prefill(); return s
full = mk(0, B)
t0 = time.time(); full.generate(frames); t1 = time.time()
print(f"one session  B={B}: {1e3 * (t1 - t0) / frames:.3f} ms/frame -> {B * frames / (t1 - t0):.0f} frames/s")
full.close()
per = B // parts
ss = [mk(i * per, (i + 1) * per) for i in range(parts)]
for s in ss: s.generate(2)          # capture graphs
th = [threading.Thread(target=lambda s=s: s.generate(frames - 2)) for s in ss]
t0 = time.time()
for t in th: t.start()
for t in th: t.join()
t1 = time.time()
print(f"{parts} sessions x B={per} concurrently: {1e3 * (t1 - t0) / (frames - 2):.3f} ms/frame-step -> {B * (frames - 2) / (t1 - t0):.0f} frames/s")
