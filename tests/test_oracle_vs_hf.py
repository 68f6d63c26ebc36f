"""The C oracle against Hugging Face's own PyTorch implementations of the same published blocks (third-party code, run in
the build container by tests/make_golden_hf.py; this test only reads the fixture it wrote — no torch / transformers here,
and nothing of this runs on the GPU box's product path). It does not pin the oracle to the reference binary (only a
reference run could), but every decoder stage D1-D9 and the talker's decoder layers A4 / A2 are held to an implementation
the builder did not write. Known, documented differences between the Qwen3-Omni blocks and the reference's Rust are
handled in make_golden_hf.py (docstring items 1-4)."""
import os

import numpy as np

import qwen3_tts_rs_amd as q
import oracle as O
from common import oracle_model, synthetic_prompt

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_crosscheck.npz")
SEED = 4321


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(b).max() + 1e-30))


def test_decoder_stages_match_hf_code2wav_and_mimi_rvq():
    fx = np.load(FX)
    cfg = q.tiny()
    om = oracle_model(cfg, seed=SEED, which=2)
    pcm, taps = om.decode(fx["codes"], taps=True)
    names = ["quant", "pre_conv", "pre_transformer", "up0", "up1", "init", "blk0", "blk1", "blk2", "blk3"]
    errs = {n: _rel(t, fx[n]) for n, t in zip(names, taps)}
    # f32 everywhere on both sides; torch sums in different orders (and uses a two-pass LayerNorm variance). The error
    # grows along the chain (each SnakeBeta's sin² of a large argument amplifies it): ~1e-6 up to the first block, 5e-5 at blk3
    print(errs)
    assert all(errs[n] <= 1e-5 for n in names[:7]), errs
    assert all(errs[n] <= 2e-4 for n in names[7:]), errs
    assert float(np.sqrt(np.mean((pcm - fx["pcm"]) ** 2))) <= 5e-4            # north-star tolerance is 1e-3
    # the pre-clamp waveform too (the clamp hides most samples with these synthetic weights)
    un = np.abs(fx["pcm_preclamp"]) < 1.0
    assert np.abs(pcm[un] - fx["pcm_preclamp"][un]).max() <= 5e-4 * max(1.0, float(np.abs(fx["pcm_preclamp"]).max()))
    om.close()


def test_trans_conv_trim_rule_vs_hf_module():
    """Documented difference 1: HF's CausalTransConvNet trims k - s samples at BOTH ends, the reference at the right end
    only (causal_trans_conv.rs:76-99). Oracle == reference rule; HF module output == the same signal without its first r samples."""
    fx = np.load(FX)
    cfg = q.tiny()
    x = np.ascontiguousarray(fx["transconv_blk0_input"])                       # [cin][L] after SnakeBeta
    cin, L = x.shape; r = cfg.dec_up_rates[0]; cout = cin // 2
    from common import manifest_handle
    from qwen3_tts_rs_amd import synth
    h = manifest_handle(cfg)
    W = {n: a for n, a, dt in synth.synthetic_checkpoint(cfg, h, SEED) if n.startswith("decoder.decoder.1.block.1.conv")}
    q._lib.lib.q3_model_free(h)
    w = np.ascontiguousarray(W["decoder.decoder.1.block.1.conv.weight"], np.float32); b = np.ascontiguousarray(W["decoder.decoder.1.block.1.conv.bias"], np.float32)
    y = np.zeros((cout, L * r), np.float32)
    O.olib.q3o_causal_trans_conv1d(O.ptr(x), O.ptr(w), O.ptr(b), O.ptr(y), cin, cout, L, 2 * r, r)
    assert _rel(y, fx["ref_rule_transconv_blk0"]) <= 1e-5
    assert fx["hf_transconv_blk0"].shape == (cout, (L - 1) * r)
    assert _rel(y[:, r:], fx["hf_transconv_blk0"]) <= 1e-5


def test_talker_layers_match_hf_qwen3_decoder_layer():
    fx = np.load(FX)
    cfg = q.tiny()
    om = oracle_model(cfg, seed=SEED, which=1)
    utt = q.Utterance(synthetic_prompt(9, 3), q.Speaker.Ryan, q.Language.English, seed=1)
    s = O.OracleSession(om, utt, q.SynthesisOptions(max_length=4, seed=1))
    np.testing.assert_array_equal(s.prefill_embeds(), fx["talker_prefill_embeds"])
    hid, lg = s.prefill_out()
    assert _rel(hid, fx["talker_last_hidden"]) <= 2e-5, _rel(hid, fx["talker_last_hidden"])
    assert np.abs(lg - fx["talker_logits"]).max() <= 2e-5 * max(1.0, float(np.abs(fx["talker_logits"]).max()))
    s.close(); om.close()


def test_code_predictor_loop_matches_hf_code_predictor_module():
    """A3 (code_predictor.rs:320-416): the 15-pass loop of the oracle — q3o_session_cp_generate: the two-token first pass, which
    embedding table and which lm_head every later pass uses, the cache position of each pass — against Hugging Face's own
    Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration driven greedily on the same seeded weights (VERDICT r4 item 3:
    the loop was only ever compared with tests/np_reference.py, written by the same author). Two configurations (tiny; five
    layers with 2-way GQA), three rows each: all 15 ids equal wherever upstream's own top-2 margin is not a near-tie, logits
    of every pass within 5e-5. The two documented differences (small_to_mtp_projection applied outside the module; upstream
    samples where the reference takes the argmax) are handled in make_golden_hf.py::cp_loop_fixture."""
    import dataclasses
    fx = np.load(os.path.join(os.path.dirname(FX), "hf_cp_loop.npz"))
    for tag in ("tiny", "mid"):
        c = [int(x) for x in fx[f"{tag}_cfg"]]
        cfg = dataclasses.replace(q.tiny(), hidden=c[0], inter=c[1], n_heads=c[2], n_kv_heads=c[3], cp_hidden=c[4], cp_inter=c[5], cp_layers=c[6], cp_heads=c[7], cp_kv_heads=c[8])
        om = oracle_model(cfg, seed=SEED, which=1)
        s = O.OracleSession(om, q.Utterance(synthetic_prompt(5, 1), q.Speaker.Ryan, q.Language.English, seed=1), q.SynthesisOptions(max_length=2, seed=1))
        for b in range(fx[f"{tag}_ids"].shape[0]):
            codes, logits = s.cp_generate(fx[f"{tag}_last_hidden"][b], fx[f"{tag}_sem_embed"][b])
            want = fx[f"{tag}_ids"][b]; marg = fx[f"{tag}_top2_margin"][b]
            same = np.asarray(codes) == want
            first_bad = int(np.argmin(same)) if not same.all() else 15
            if first_bad < 15:          # a different greedy code changes every later pass: only an upstream near-tie may cause it
                assert marg[first_bad] < 1e-4, (tag, b, first_bad, float(marg[first_bad]))
            for g in range(min(first_bad + 1, 15)):
                e = float(np.abs(logits[g] - fx[f"{tag}_logits"][b, g]).max())
                assert e <= 5e-5 * max(1.0, float(np.abs(fx[f"{tag}_logits"][b, g]).max())), (tag, b, g, e)
            assert first_bad >= 8, (tag, b, first_bad)             # the fixtures hold no early near-tie: most of the loop is always compared
        s.close(); om.close()


def _frame_loop_case(fx, tag):
    import dataclasses
    c = [int(x) for x in fx[f"{tag}_cfg"]]
    cfg = dataclasses.replace(q.tiny_same_width(), hidden=c[0], inter=c[1], n_layers=c[2], n_heads=c[3], n_kv_heads=c[4], cp_hidden=c[5], cp_inter=c[6],
                              cp_layers=c[7], cp_heads=c[8], cp_kv_heads=c[9])
    utt = q.Utterance(synthetic_prompt(9, 3), q.Speaker.Ryan, q.Language.English, seed=1)
    runs = {"plain": dict(repetition_penalty=1.0, min_new_tokens=0, eos_token_id=None),
            "penalties": dict(repetition_penalty=1.05, min_new_tokens=2, eos_token_id=q.api.CODEC_EOS_TOKEN_ID)}
    return cfg, utt, runs


def check_codes_against_upstream_loop(fx, tag, rn, codes):
    """`codes` [n][16] of one run against upstream's: code 0 of every frame, all sixteen codes of every frame but the last (upstream
    computes a frame's residual codes while preparing the NEXT step). Only an upstream near-tie of the talker may excuse a divergence."""
    want0, want = fx[f"{tag}_{rn}_code0"], fx[f"{tag}_{rn}_codes"]
    marg = fx[f"{tag}_{rn}_top2_margin"]
    assert len(codes) == len(want0), (tag, rn, len(codes), len(want0))
    same0 = np.asarray(codes)[:, 0] == want0
    if not same0.all():
        fb = int(np.argmin(same0))
        assert marg[fb] < 1e-4, (tag, rn, "code 0 differs at frame", fb, "upstream margin", float(marg[fb]))
        raise AssertionError(f"{tag}/{rn}: upstream near-tie at frame {fb} — regenerate the fixture with another prompt seed")
    np.testing.assert_array_equal(np.asarray(codes)[:len(want)], want)


def test_frame_loop_matches_hf_talker_generate():
    """A1 + A10 (lib.rs:530-656): the oracle's frame loop — q3o_session_generate: code 0 from the talker's logits, the code
    predictor on [last hidden, embed(code 0)], the next talker input = the sum of the sixteen code embeddings + the next
    trailing-text row or the tts_pad row, one KV position per frame — against Hugging Face's own loop for this model family:
    Qwen3OmniMoeTalkerForConditionalGeneration.generate() with its prepare_inputs_for_generation() and the code predictor's
    generate() inside, on the module's own caches (the loop had only ever met tests/np_reference.py, same
    author). 24 frames, two configurations, greedy, with and without the default penalties riding along: every code of every
    compared frame equal, talker logits of every step within 5e-5. The documented differences (dense MLP swapped into upstream's
    MoE layer, 0.6B topology, greedy) are in make_golden_hf.py::frame_loop_fixture."""
    fx = np.load(os.path.join(os.path.dirname(FX), "hf_frame_loop.npz"))
    for tag in ("tiny", "mid"):
        cfg, utt, runs = _frame_loop_case(fx, tag)
        om = oracle_model(cfg, seed=SEED, which=1)
        for rn, ro in runs.items():
            s = O.OracleSession(om, utt, q.SynthesisOptions(max_length=len(fx[f"{tag}_{rn}_code0"]), seed=1, temperature=0.0, **ro))
            # the loop's inputs are the ones upstream was given (the prompt assembly is not what this test holds)
            np.testing.assert_array_equal(s.prefill_embeds(), fx[f"{tag}_{rn}_prefill_embeds"])
            tr, pad = s.trailing()
            np.testing.assert_array_equal(tr, fx[f"{tag}_{rn}_trailing"]); np.testing.assert_array_equal(pad, fx[f"{tag}_{rn}_pad"])
            assert 0 < tr.shape[0] < len(fx[f"{tag}_{rn}_code0"]) - 2          # text rows first, then the pad row: both branches
            codes, tl, _ = s.generate(capture=True)
            check_codes_against_upstream_loop(fx, tag, rn, codes)
            if rn == "plain":
                ref = fx[f"{tag}_{rn}_logits"]
                for i in range(len(ref)):
                    e = float(np.abs(tl[i] - ref[i]).max())
                    assert e <= 5e-5 * max(1.0, float(np.abs(ref[i]).max())), (tag, i, e)
            s.close()
        om.close()


def _speech_oracle(scfg):
    from qwen3_tts_rs_amd.speech_encoder import SpeechEncoder, synthetic_speech_checkpoint
    enc = SpeechEncoder(scfg, device=-1)              # manifest-only handle: names / sizes (no GPU)
    o = O.OracleSpeechEncoder(scfg)
    for name, arr in synthetic_speech_checkpoint(enc, SEED):
        o.set_tensor(name, arr)
    enc.close()
    return o


def _clip(n, seed):          # the generator's test signal (tests/make_golden_hf.py::test_clip)
    t = np.arange(n) / 24000.0
    rng = np.random.default_rng(seed)
    x = 0.3 * np.sin(2 * np.pi * 180.0 * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 2.5 * t)) + 0.15 * np.sin(2 * np.pi * 1250.0 * t + 0.7) + \
        0.05 * rng.standard_normal(n)
    return x.astype(np.float32)


def test_speech_encoder_matches_hf_mimi():
    """ICL reference-audio encoder (encoder_12hz.rs:34-144 over candle's Mimi): oracle/q3_oracle_mimi.c against Hugging
    Face's MimiModel on the same seeded weights — SEANet output, transformer output, down-sampled latents, and the 16
    codebook indices per frame (a differing index is tolerated only where the oracle's own decision margin is tiny)."""
    import qwen3_tts_rs_amd as q
    fx = np.load(os.path.join(os.path.dirname(FX), "hf_mimi.npz"))
    for tag, scfg in (("tiny", q.tiny_speech_config()), ("full", q.SpeechEncoderConfig())):
        n = int(fx[f"mimi_{tag}_n"][0])
        o = _speech_oracle(scfg)
        codes, taps = o.encode(_clip(n, 7), taps=True)
        assert _rel(taps[0], fx[f"mimi_{tag}_seanet"]) <= 2e-5, (tag, _rel(taps[0], fx[f"mimi_{tag}_seanet"]))
        assert _rel(taps[1], fx[f"mimi_{tag}_transformer"]) <= 5e-5, (tag, _rel(taps[1], fx[f"mimi_{tag}_transformer"]))
        assert _rel(taps[2], fx[f"mimi_{tag}_downsample"]) <= 5e-5, (tag, _rel(taps[2], fx[f"mimi_{tag}_downsample"]))
        ref = fx[f"mimi_{tag}_codes"]
        assert codes.shape == ref.shape
        bad = 0
        for t in range(codes.shape[0]):
            if not (codes[t] == ref[t]).all():
                l = int(np.argmin(codes[t] == ref[t]))            # first differing layer; later layers follow from it
                assert taps[3][t, l] <= 1e-4 * max(1.0, float(np.abs(taps[2]).max()) ** 2), (tag, t, l, float(taps[3][t, l]))
                bad += 1
        assert bad <= max(1, codes.shape[0] // 20), (tag, bad)
        o.close()


def test_sampler_filters_match_hf_logits_processors():
    """A9 sampler filters against transformers.generation.logits_process (fixture: make_golden_hf.py sampling): repetition
    penalty values, the top-k kept set and the top-p kept set on 16 seeded logit rows (three logit scales, k in {50, 30, 5,
    off}, p in {0.9, 0.8, 0.95, off}). HF's top-p sums the ascending tail, the reference the descending head: the kept
    sets must agree except for a token whose inclusion hangs on the last ulps of those two different f32 sums."""
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_sampling.npz"))
    n = int(fx["n"]); V = fx["c0_logits"].shape[0]
    boundary = 0
    for i in range(n):
        g = lambda k: fx[f"c{i}_{k}"]
        logits = g("logits").astype(np.float32); pen = float(g("pen")); k = int(g("k")); p = float(g("p")); temp = float(g("temp"))
        seen = np.zeros(V, np.uint8); seen[g("seen")] = 1
        l = np.ascontiguousarray(logits.copy())
        O.olib.q3o_apply_penalties(l.ctypes.data, V, seen.ctypes.data, pen, 100, 0, -1)
        lo = V - 1024                                                    # the oracle also suppresses [V-1024, V): not an HF rule
        # same rule (x > 0 ? x / pen : x * pen); the reference multiplies by the f32 reciprocal where HF divides: <= 1 ulp
        np.testing.assert_allclose(l[:lo], g("after_pen")[:lo], rtol=2.5e-7, atol=0)
        t_hf = g("after_t").astype(np.float32)
        t_or = (g("after_pen").astype(np.float32) * np.float32(1.0 / temp) + np.float32(0.0)) if temp != 1.0 else g("after_pen")
        assert np.abs(t_or[:lo] - t_hf[:lo]).max() <= 2e-6 * max(1.0, np.abs(t_hf[:lo]).max())     # x * (1/t) vs x / t
        x = np.ascontiguousarray(t_hf.copy())                          # filters on HF's own input row: only the filter rules differ
        if k > 0:
            O.olib.q3o_top_k_filter(x.ctypes.data, V, k)
            np.testing.assert_array_equal(np.isfinite(x), g("keep_k"))
        if p < 1.0:
            O.olib.q3o_top_p_filter(x.ctypes.data, V, p)
            diff = np.flatnonzero(np.isfinite(x) != g("keep_p"))
            if diff.size:
                boundary += 1
                assert diff.size == 1, (i, diff)                        # at most the one boundary token
    assert boundary <= 1, boundary
