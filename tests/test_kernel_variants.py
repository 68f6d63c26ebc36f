"""One function, several kernels: the engine picks among kernel variants by shape, and environment knobs switch the
alternatives back on. Each variant runs the bench session (1.7B, synthetic weights, default sampling, hipGraph) in its own process —
the knobs are read once per process — and must produce the default build's codes:
  * bit-identical BY CONSTRUCTION (same operands, same sums in the same order): contiguous instead of paged K/V, the frame replayed
    through the library's own AQL queue, the code predictor's gather as its own launch instead of folded into the attention, the
    two-instruction x loads instead of the half-slot split;
  * the same function with another summation order (the generic attention kernel instead of the code predictor's one-wave kernel,
    unsplit o / down projections instead of the two-half split-K): codes agree except where a near-tie flips, so at least 98 % of them.
Found in round 5: two template instances of ONE source line disagreed in the last place because hipcc contracted `__fmul_rn` +
`__fadd_rn` in one of them (tests/test_source_rules.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
B, F = int(sys.argv[1]), int(sys.argv[2])
m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), device=0, seed=synth.DEFAULT_SEED)
utts = [q.Utterance(synth.synthetic_prompt(64, i), q.Speaker.Ryan, q.Language.English, seed=42 + i) for i in range(B)]
s = m.session(utts, q.SynthesisOptions(max_length=F, eos_token_id=None, seed=42)); s.prefill(); s.generate(F, use_graph=True)
codes = np.stack([s.codes(b) for b in range(B)]).astype(np.uint32); s.close(); m.close()
np.save(sys.argv[3], codes)
""" % ROOT


def _run(tmp_path, name, env_extra, B, F):
    env = dict(os.environ); env.update(env_extra)
    out = tmp_path / (name + ".npy")
    r = subprocess.run([sys.executable, "-c", CHILD, str(B), str(F), str(out)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-2000:])
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 16])
def test_variants_give_the_default_codes(tmp_path, B):
    F = 48
    base = _run(tmp_path, "default", {}, B, F)
    assert base.shape == (B, F, 16)
    exact = {"contiguous_kv": {"Q3_KV_CONTIGUOUS": "1"}, "hip_graph_launch": {"Q3_AQL": "0"}, "own_aql_queue_hip_fences": {"Q3_AQL": "1"}, "gather_unfolded": {"Q3_CP_NO_FOLD": "1"},
             "two_instruction_x": {"Q3_GEMV_NO_HALF": "1"}}
    for name, env in exact.items():
        got = _run(tmp_path, name, env, B, F)
        assert (got == base).all(), (name, int((got != base).sum()))
    close = {"generic_cp_attention": {"Q3_NO_CP_ATTN": "1"}, "unsplit_o_down": {"Q3_NO_KSPLIT": "1"}}
    for name, env in close.items():
        got = _run(tmp_path, name, env, B, F)
        first = min((int(np.argmax((got[b] != base[b]).any(axis=1))) if (got[b] != base[b]).any() else F) for b in range(B))
        agree = float((got == base).mean())
        # sampling feeds back: after a flipped near-tie a row's later frames differ legitimately, so count rows up to their first flip
        rows_equal = sum(int((got[b] == base[b]).all()) for b in range(B))
        assert first >= 8 and rows_equal >= B - 2, (name, first, rows_equal, agree)
