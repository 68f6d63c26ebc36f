"""Dev aid: dump the vocoder taps for a short decode so two runs (Q3_CONV_NO_SMALL set / unset) can be diffed."""
import sys, os, numpy as np
import torch  # noqa: F401  (first: one HIP runtime)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qwen3_tts_rs_amd as q
from common import model_pair
from qwen3_tts_rs_amd import synth
T = int(sys.argv[2]); full = sys.argv[3] == "full"
t = q.tiny()
cfg = q.Q3Config(text_dim=t.text_dim, hidden=t.hidden, inter=t.inter, n_layers=t.n_layers, n_heads=t.n_heads,
                 n_kv_heads=t.n_kv_heads, cp_hidden=t.cp_hidden, cp_inter=t.cp_inter, cp_layers=t.cp_layers,
                 cp_heads=t.cp_heads, cp_kv_heads=t.cp_kv_heads, name="tiny-lm-full-decoder") if full else t
gm, om = model_pair(cfg, seed=synth.DEFAULT_SEED)
codes = np.random.default_rng(T).integers(0, 2048, size=(T, 16)).astype(np.uint32)
_, otaps = om.decode(codes, taps=True)
taps = [np.zeros_like(x) for x in otaps]
pcm = gm.decode_codes(codes, taps=taps).samples
np.savez(sys.argv[1], pcm=pcm, **{f"t{i}": x for i, x in enumerate(taps)})
