"""GPU bring-up diagnostic (not a pytest): times each stage of a tiny session incl. hipGraph capture,
prints immediately (flush) so a hang is localised. Run under `timeout`."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np


def P(*a):
    print(f"[{time.time() - T0:8.2f}s]", *a, flush=True)


T0 = time.time()
import qwen3_tts_rs_amd as q
from common import model_pair, synthetic_prompt
import oracle as O
P("imports done; devices", q._lib.lib.q3_device_count(), "cpus", os.cpu_count())
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = {"tiny": q.tiny, "0.6b": q.qwen3_tts_0_6b, "1.7b": q.qwen3_tts_1_7b}[which]()
gm, om = model_pair(cfg, seed=1234)
P("models built")
utt = q.Utterance(synthetic_prompt(20), seed=42)
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = q.SynthesisOptions(max_length=nf, seed=42, eos_token_id=None)
s = gm.session([utt], opts); P("session created")
s.prefill(); P("prefill done")
s.generate(2, use_graph=False); P("2 frames eager")
s.generate(2, use_graph=True); P("2 frames graph (incl. capture+instantiate)")
t = time.time(); s.generate(nf - 4, use_graph=True); dt = time.time() - t
P(f"{nf - 4} frames graph replay: {dt * 1000 / max(nf - 4, 1):.3f} ms/frame")
codes = s.codes(0); P("codes", codes.shape)
osess = O.OracleSession(om, utt, opts); ocodes = osess.generate(); P("oracle done")
P("codes equal:", bool((codes == ocodes).all()))
if not (codes == ocodes).all():
    P("gpu\n", codes[:3], "\noracle\n", ocodes[:3])
t = time.time(); pcm = s.decode(0); P(f"decode {nf} frames: {(time.time() - t) * 1000:.1f} ms")
opcm = om.decode(ocodes); P("pcm rms err", float(np.sqrt(np.mean((pcm - opcm) ** 2))))
# small-vocab sampler KAT that failed in run 1
g = q.SynthesisOptions(temperature=0.001)
try:
    P("kat greedy", q.sample(np.array([[1.0, 2.0, 5.0, 1.0]], np.float32), np.zeros(1, np.float32), g))
    o = q.SynthesisOptions(temperature=1.0, top_k=0, top_p=1.0)
    lg = np.array([[-np.inf, 0.0, -np.inf, -np.inf]], np.float32)
    for u in (0.0, 0.3, 0.999, 1.0):
        P("kat onehot u", u, q.sample(lg, np.array([u], np.float32), o))
except Exception as e:
    P("kat error", e)
# eager timing
s2 = gm.session([utt], opts); s2.prefill()
t = time.time(); s2.generate(nf, use_graph=False); P(f"eager: {(time.time() - t) * 1000 / nf:.3f} ms/frame")
P("done")
