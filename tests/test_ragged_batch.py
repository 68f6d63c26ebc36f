"""Ragged first batch (-m gpu; round 5): q3_session_create takes rows of DIFFERENT prompt kinds and prefill lengths — BASELINE
config[3] on a Base checkpoint mixes x-vector and ICL voice-clone prompts, and the reference's synthesize calls are per request
(lib.rs:718-784 CustomVoice, 802-870 VoiceDesign, 897-1046 voice clone). The session prefills the rows in groups of equal
prefill length and decodes them in ONE captured frame graph; every row must carry the bits of its own batch-1 run and the
oracle's codes for it."""
import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import _lib, api
import oracle as O
from common import model_pair, synthetic_prompt, pcm_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    cfg = q.tiny()
    gm, om = model_pair(cfg, seed=1234)
    yield cfg, gm, om
    gm.close(); om.close()


def _mixed(cfg):
    rng = np.random.default_rng(5)
    xv = rng.standard_normal(cfg.hidden).astype(np.float32)
    ref = rng.integers(0, 2048, size=(5, 16)).astype(np.uint32)
    greedy = q.SynthesisOptions(temperature=0.0, eos_token_id=None, max_length=20, seed=1)
    topk = q.SynthesisOptions(temperature=1.1, top_k=20, top_p=1.0, repetition_penalty=1.2, eos_token_id=None, max_length=20, seed=1)
    utts = [
        q.Utterance(synthetic_prompt(9, 0), q.Speaker.Ryan, q.Language.English, seed=42),                                    # CustomVoice: 10 positions
        q.Utterance(synthetic_prompt(6, 1), language=q.Language.German, instruct_ids=synthetic_prompt(7, 51), seed=43),      # VoiceDesign(7): 16
        q.Utterance(synthetic_prompt(8, 2), language=q.Language.German, instruct_ids=synthetic_prompt(600, 52), seed=44),    # VoiceDesign(600): 609 (GEMM prefill)
        q.Utterance(synthetic_prompt(7, 3), language=q.Language.French, xvector=xv, seed=45),                               # x-vector clone
        q.Utterance(synthetic_prompt(11, 4), language=q.Language.French, xvector=xv, ref_codes=ref, ref_text_ids=synthetic_prompt(3, 94), seed=46),   # ICL
        q.Utterance(synthetic_prompt(1, 5), q.Speaker.Ryan, q.Language.English, seed=47),                                    # CustomVoice again: same group as row 0
        q.Utterance(synthetic_prompt(12, 6), language=q.Language.German, instruct_ids=synthetic_prompt(7, 56), seed=48),     # VoiceDesign(7) again: same group as row 1
        q.Utterance(synthetic_prompt(0, 7), q.Speaker.Ryan, q.Language.English, seed=49),                                    # empty text: 9 positions
    ]
    limits = [14, 9, 12, 20, 16, 5, 11, 7]
    for i, (u, L) in enumerate(zip(utts, limits)):
        u.max_length = L
        if i in (2, 5): u.options = greedy
        if i in (3, 6): u.options = topk
    return utts


@pytest.mark.parametrize("use_graph", [True, False])
def test_ragged_first_batch(pair, use_graph):
    cfg, gm, om = pair
    utts = _mixed(cfg)
    host = q.SynthesisOptions(max_length=20, eos_token_id=None, seed=1)
    s = gm.session(utts, host)
    lens = [s.prefill_len(b)[0] for b in range(len(utts))]
    assert lens[0] == 10 and lens[1] == 16 and lens[2] == 609 and lens[7] == 9 and len(set(lens)) >= 5, lens      # really ragged
    s.prefill()
    assert [s.prefill_len(b)[0] for b in range(len(utts))] == lens
    s.generate(20, use_graph=use_graph)
    got = [s.codes(b) for b in range(len(utts))]
    pcm = [s.decode(b) for b in range(len(utts))]
    s.close()
    assert gm.kv_pool_info()["pages_in_use"] == 0
    for b, u in enumerate(utts):
        o = u.options or host
        assert got[b].shape == (u.max_length, 16), (b, got[b].shape)
        s1 = gm.session([u], o); s1.prefill(); s1.generate(20, use_graph=False)
        np.testing.assert_array_equal(got[b], s1.codes(0), err_msg=f"row {b} vs its batch-1 run")
        np.testing.assert_array_equal(pcm[b], s1.decode(0), err_msg=f"row {b} PCM vs its batch-1 run")
        s1.close()
        osess = O.OracleSession(om, u, o)
        np.testing.assert_array_equal(got[b], osess.generate(), err_msg=f"row {b} vs the oracle"); osess.close()


def test_ragged_session_run_and_swap(pair):
    """q3_session_run on a ragged batch (prefill + frames + vocoder in one call), then a continuous-batching swap into it."""
    cfg, gm, om = pair
    utts = _mixed(cfg)[:5]
    host = q.SynthesisOptions(max_length=20, eos_token_id=None, seed=1)
    s = api.Session(gm, utts, host, frame_budget=24, prompt_budget=700)
    audio, timing = s.run()
    assert timing.generation_frames > 0
    for b, u in enumerate(utts):
        s1 = gm.session([u], u.options or host); s1.prefill(); s1.generate(20)
        ref = s1.decode(0); s1.close()
        assert audio[b].samples.shape == ref.shape and pcm_rms(audio[b].samples, ref) == 0.0, b
    s.close()
    # a ragged session is an ordinary session afterwards: rows can be replaced
    s = api.Session(gm, utts[:3], host, frame_budget=24, prompt_budget=700)
    s.prefill(); s.generate(9)                       # row 1 (limit 9) has ended
    late = _mixed(cfg)[4]
    s.replace(1, late); s.generate(24)
    s1 = gm.session([late], host); s1.prefill(); s1.generate(24)
    np.testing.assert_array_equal(s.codes(1), s1.codes(0)); s1.close()
    s0 = gm.session([utts[0]], host); s0.prefill(); s0.generate(24)
    np.testing.assert_array_equal(s.codes(0), s0.codes(0)); s0.close()
    s.close()
    assert gm.kv_pool_info()["pages_in_use"] == 0


def test_ragged_bad_row_fails_at_prefill(pair):
    cfg, gm, om = pair
    ok = q.Utterance(synthetic_prompt(9, 0), q.Speaker.Ryan, q.Language.English, seed=42)
    bad = q.Utterance([cfg.text_vocab + 5], language=q.Language.German, instruct_ids=synthetic_prompt(7, 51), seed=43)      # id out of range, other length
    s = gm.session([ok, bad], q.SynthesisOptions(max_length=4, seed=1))
    with pytest.raises(_lib.Q3Error, match="out of range"):
        s.prefill()
    s.close()
    assert gm.kv_pool_info()["pages_in_use"] == 0


def test_ragged_icl_rows_above_their_cap(pair):
    """Two ICL rows of different reference length with the default-sized max_length (far above the ICL cap max(75, 6 * n_text),
    lib.rs:897-1046) next to an x-vector row: the session is sized from the RESOLVED limits, and so must the ragged prefill's check be
    (round-5 advisor finding: every such batch failed with `max_length 2048 outside 1..N`)."""
    cfg, gm, om = pair
    rng = np.random.default_rng(6)
    xv = rng.standard_normal(cfg.hidden).astype(np.float32)
    ref_a = rng.integers(0, 2048, size=(4, 16)).astype(np.uint32)
    ref_b = rng.integers(0, 2048, size=(9, 16)).astype(np.uint32)
    utts = [
        q.Utterance(synthetic_prompt(5, 10), language=q.Language.French, xvector=xv, ref_codes=ref_a, ref_text_ids=synthetic_prompt(3, 95), seed=50),
        q.Utterance(synthetic_prompt(14, 11), language=q.Language.French, xvector=xv, ref_codes=ref_b, ref_text_ids=synthetic_prompt(4, 96), seed=51),
        q.Utterance(synthetic_prompt(7, 12), language=q.Language.French, xvector=xv, seed=52),
    ]
    utts[0].max_length = 2048; utts[1].max_length = 2048; utts[2].max_length = 30
    caps = [75, 84, 30]                                   # max(75, 6 * n_text) for the ICL rows
    host = q.SynthesisOptions(max_length=2048, eos_token_id=None, seed=1)
    s = gm.session(utts, host)
    s.prefill()
    s.generate(12, use_graph=True)
    got = [s.codes(b) for b in range(3)]
    s.close()
    assert gm.kv_pool_info()["pages_in_use"] == 0
    for b, u in enumerate(utts):
        o = q.SynthesisOptions(max_length=u.max_length, eos_token_id=None, seed=1)
        s1 = gm.session([u], o); s1.prefill(); s1.generate(12, use_graph=False)
        assert got[b].shape == (12, 16) and caps[b] >= 12
        np.testing.assert_array_equal(got[b], s1.codes(0), err_msg=f"row {b} vs its batch-1 run"); s1.close()
