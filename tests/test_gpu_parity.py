"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs. Tolerances are stated per test; ids must be bit-exact except at oracle
near-ties (top-2 logit margin below MARGIN_EPS), which are reported and bounded."""
import json
import os

import numpy as np
import pytest

import ctypes

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth, api, _lib
import oracle as O
from common import model_pair, synthetic_prompt, top2_margin, rel_err, pcm_rms

pytestmark = pytest.mark.gpu

MARGIN_EPS = 2e-3      # a GPU/CPU id mismatch is tolerated only if the oracle's top-2 margin is below this
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _dump(name, obj):
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, name), "w") as f:
        json.dump(obj, f, indent=1)


# ---------------------------------------------------------------- ops
@pytest.mark.parametrize("M,N,K", [(1, 64, 64), (1, 1000, 1024), (2, 4096, 2048), (4, 2048, 6144), (8, 3072, 2048),
                                   (8, 1024, 3072), (3, 520, 136), (5, 12288, 2048), (11, 256, 512),
                                   # wide batches (k_gemv_wide: 2, 3 and 4 column tiles, ragged row counts, K tails)
                                   (17, 4096, 2048), (32, 1024, 3072), (33, 2048, 1024), (48, 3072, 2048), (64, 2048, 6144),
                                   (64, 520, 136), (40, 64, 64)])
def test_linear_matches_oracle(M, N, K):
    rng = np.random.default_rng(M * 1000 + N + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = synth.f32_to_bf16((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(N).astype(np.float32)
    y = q.linear(x, w, b)
    wf = synth.bf16_to_f32(w).reshape(N, K)
    yo = np.zeros((M, N), dtype=np.float32)
    O.olib.q3o_linear(O.ptr(x), O.ptr(np.ascontiguousarray(wf)), O.ptr(b), O.ptr(yo), M, N, K)
    ref64 = x.astype(np.float64) @ wf.astype(np.float64).T + b
    # f32 accumulation in a different order: both sides within 2e-5 (relative to max |y|) of the f64 result
    assert rel_err(y, ref64) < 2e-5 and rel_err(yo, ref64) < 2e-5
    assert rel_err(y, yo) < 2e-5


@pytest.mark.parametrize("rows,cols", [(1, 1024), (1, 2048), (4, 2048), (3, 96), (2, 4100)])
def test_fused_residual_rmsnorm_f32(rows, cols):
    rng = np.random.default_rng(rows * 7 + cols)
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    r = rng.standard_normal((rows, cols)).astype(np.float32)
    w = (1 + 0.02 * rng.standard_normal(cols)).astype(np.float32)
    n, s = q.fused_residual_rmsnorm(x, r, w, 1e-6)
    no = np.zeros_like(x); so = np.zeros_like(x)
    O.olib.q3o_fused_residual_rmsnorm(O.ptr(x), O.ptr(r), O.ptr(w), rows, cols, 1e-6, O.ptr(no), O.ptr(so))
    assert (s == so).all()                       # the sum is a single f32 add: bit-exact
    assert np.abs(n - no).max() <= 4e-6          # normalisation differs only by the Σx² order


def test_fused_residual_rmsnorm_bf16():
    rng = np.random.default_rng(5)
    rows, cols = 2, 2048
    x = synth.f32_to_bf16(rng.standard_normal((rows, cols)).astype(np.float32)).reshape(rows, cols)
    r = synth.f32_to_bf16(rng.standard_normal((rows, cols)).astype(np.float32)).reshape(rows, cols)
    w = synth.f32_to_bf16((1 + 0.02 * rng.standard_normal(cols)).astype(np.float32))
    n, s = q.fused_residual_rmsnorm(x, r, w, 1e-6)
    xf, rf, wf = synth.bf16_to_f32(x), synth.bf16_to_f32(r), synth.bf16_to_f32(w)
    sf = (xf + rf)
    s_ref = synth.f32_to_bf16(sf).reshape(rows, cols)
    assert (s == s_ref).all()                    # kernels/fused_residual_rmsnorm.cu: sum stored rounded to T
    sr = synth.bf16_to_f32(s_ref).reshape(rows, cols)
    den = np.sqrt((sf.reshape(rows, cols).astype(np.float64) ** 2).mean(axis=1, keepdims=True) + 1e-6)
    n_ref = sr / den * wf
    assert np.abs(synth.bf16_to_f32(n).reshape(rows, cols) - n_ref).max() <= 2e-2   # one bf16 ulp at |x|<=4


# ---------------------------------------------------------------- sampler
def _oracle_sample(logits, seen, opts, token_count, u_seed):
    """penalties + sample with the oracle; returns (token, u) where u is the draw it consumed."""
    import ctypes
    lg = np.array(logits, dtype=np.float32)
    V = lg.shape[0]
    if token_count >= 0:
        O.olib.q3o_apply_penalties(O.ptr(lg), V, O.ptr(seen), float(opts.repetition_penalty), token_count,
                                   opts.min_new_tokens, -1 if opts.eos_token_id is None else opts.eos_token_id)
    st = ctypes.c_uint64(); O.olib.q3o_rng_seed(u_seed, ctypes.byref(st))
    st2 = ctypes.c_uint64(st.value)
    u = O.olib.q3o_rng_next(ctypes.byref(st2))
    tok = O.olib.q3o_sample(O.ptr(lg), V, float(opts.temperature), opts.top_k, float(opts.top_p), ctypes.byref(st))
    return tok, u


@pytest.mark.parametrize("temp,top_k,top_p,rep", [(0.9, 50, 0.9, 1.05), (0.0, 50, 0.9, 1.05), (1.0, 0, 1.0, 1.0),
                                                  (0.7, 10, 0.5, 1.5), (1.3, 0, 0.8, 1.05), (0.9, 3072, 0.99, 2.0),
                                                  (0.5, 1, 0.9, 1.0)])
def test_sampler_ids_bit_exact(temp, top_k, top_p, rep):
    V, rows = 3072, 64
    rng = np.random.default_rng(int(temp * 100) + top_k)
    opts = q.SynthesisOptions(temperature=temp, top_k=top_k, top_p=top_p, repetition_penalty=rep, seed=1)
    logits = (3.5 * rng.standard_normal((rows, V))).astype(np.float32)
    logits[0] = (5.0 * np.sin(0.1 * np.arange(V))).astype(np.float32)       # benches/sampling.rs:12-17 pattern
    logits[1, :100] = 2.0                                                    # exact ties straddling top-k
    seen = (rng.random((rows, V)) < 0.05).astype(np.uint8)
    us = np.zeros(rows, dtype=np.float32); ref = np.zeros(rows, dtype=np.uint32)
    for r in range(rows):
        ref[r], us[r] = _oracle_sample(logits[r], seen[r], opts, 1, 100 + r)
    got = q.sample(logits, us, opts, seen=seen, token_count=1)
    bad = np.nonzero(got != ref)[0]
    assert len(bad) == 0, f"rows {bad[:8]} differ: gpu {got[bad[:8]]} oracle {ref[bad[:8]]}"


def test_sampler_reference_kats_small_vocab():
    """The reference's own sampler unit tests (sampling.rs:473-624), on the device sampler."""
    g = q.SynthesisOptions(temperature=0.001)
    assert q.sample(np.array([[1.0, 2.0, 5.0, 1.0]], np.float32), np.zeros(1, np.float32), g)[0] == 2
    assert list(q.sample(np.array([[1.0, 5.0, 2.0], [3.0, 1.0, 2.0], [1.0, 2.0, 10.0]], np.float32), np.zeros(3, np.float32), g)) == [1, 0, 2]
    assert q.sample(np.array([[1.0, 10.0, 2.0, 1.0]], np.float32), np.zeros(1, np.float32), g)[0] == 1
    # multinomial with a one-hot distribution always picks that index (sampling.rs:601-609)
    o = q.SynthesisOptions(temperature=1.0, top_k=0, top_p=1.0)
    lg = np.array([[-np.inf, 0.0, -np.inf, -np.inf]], np.float32)
    for u in (0.3, 0.999, 1.0, 1e-30):
        assert q.sample(lg, np.array([u], np.float32), o)[0] == 1
    # u == 0.0 exactly: `first i: cdf[i] >= u` is index 0 even at zero probability (sampling.rs:300-318)
    assert q.sample(lg, np.array([0.0], np.float32), o)[0] == 0


# ---------------------------------------------------------------- model-level
@pytest.fixture(scope="module", params=["tiny", "tiny_same_width"])
def pair(request):
    cfg = q.tiny() if request.param == "tiny" else q.tiny_same_width()
    gm, om = model_pair(cfg, seed=1234)
    yield cfg, gm, om
    gm.close(); om.close()


def _utts(mode, n_text=9, index=0, hidden=64):
    text = synthetic_prompt(n_text, index)
    if mode == "custom":
        return q.Utterance(text, q.Speaker.Ryan, q.Language.English, seed=42 + index)
    if mode == "design":
        return q.Utterance(text, language=q.Language.German, instruct_ids=synthetic_prompt(7, 50 + index), seed=42 + index)
    rng = np.random.default_rng(index)
    return q.Utterance(text, language=q.Language.French, xvector=rng.standard_normal(hidden).astype(np.float32), seed=42 + index)


@pytest.mark.parametrize("mode", ["custom", "design", "clone"])
@pytest.mark.parametrize("n_text", [0, 1, 9])
def test_prefill_stages(pair, mode, n_text):
    cfg, gm, om = pair
    utt = _utts(mode, n_text, hidden=cfg.hidden)
    opts = q.SynthesisOptions(max_length=4, seed=42)
    s = gm.session([utt], opts); s.prefill()
    osess = O.OracleSession(om, utt, opts)
    S, Ttr = s.prefill_len(0)
    assert S == osess.prefill_len()
    emb = s.get(0, (S, cfg.hidden)); oemb = osess.prefill_embeds()
    assert np.abs(emb - oemb).max() <= 2e-5 * max(1.0, np.abs(oemb).max())
    otr, opad = osess.trailing()
    assert Ttr == otr.shape[0]
    tr = s.get(3, (Ttr, cfg.hidden)); pad = s.get(4, (cfg.hidden,))
    assert np.abs(tr - otr).max() <= 2e-5 * max(1.0, np.abs(otr).max())
    assert np.abs(pad - opad).max() <= 2e-5 * max(1.0, np.abs(opad).max())
    hid = s.get(1, (cfg.hidden,)); lg = s.get(2, (cfg.codec_vocab,))
    ohid, olg = osess.prefill_out()
    assert np.abs(hid - ohid).max() <= 1e-4, np.abs(hid - ohid).max()
    assert np.abs(lg - olg).max() <= 1e-3, np.abs(lg - olg).max()
    s.close(); osess.close()


def test_teacher_forced_steps(pair):
    cfg, gm, om = pair
    utt = _utts("custom", 9, hidden=cfg.hidden)
    opts = q.SynthesisOptions(max_length=8, seed=42)
    s = gm.session([utt], opts); s.prefill()
    osess = O.OracleSession(om, utt, opts)
    rng = np.random.default_rng(3)
    ohid, _ = osess.prefill_out()
    for step in range(4):
        sem = rng.standard_normal(cfg.hidden).astype(np.float32)
        ocodes, ocl = osess.cp_generate(ohid, sem)
        codes, cl = s.cp_generate(ohid[None], sem[None])
        assert np.abs(cl[0] - ocl).max() <= 2e-3, (step, np.abs(cl[0] - ocl).max())
        for g in range(15):
            if codes[0, g] != ocodes[g]:
                assert top2_margin(ocl[g]) < MARGIN_EPS, (step, g, codes[0, g], ocodes[g], top2_margin(ocl[g]))
                break
        emb = rng.standard_normal(cfg.hidden).astype(np.float32)
        ohid, olg = osess.talker_step(emb)
        hid, lg = s.talker_step(emb[None])
        assert np.abs(hid[0] - ohid).max() <= 2e-4, (step, np.abs(hid[0] - ohid).max())
        assert np.abs(lg[0] - olg).max() <= 2e-3, (step, np.abs(lg[0] - olg).max())
    s.close(); osess.close()


def _free_run_compare(cfg, gm, om, utt, opts, use_graph, tag):
    s = gm.session([utt], opts, debug=not use_graph); s.prefill()
    s.generate(opts.max_length, use_graph=use_graph)
    codes = s.codes(0)
    osess = O.OracleSession(om, utt, opts)
    ocodes, otl, ocl = osess.generate(capture=True)
    n = min(len(codes), len(ocodes))
    first_div = None
    for f in range(n):
        if not (codes[f] == ocodes[f]).all():
            first_div = f; break
    report = {"tag": tag, "frames_gpu": int(len(codes)), "frames_oracle": int(len(ocodes)), "first_divergence": first_div}
    if first_div is None and len(codes) != len(ocodes):
        first_div = n
    if first_div is not None:
        # a divergence is acceptable only at an oracle near-tie: find the deciding margin
        f = first_div
        margins = []
        if f < len(ocodes):
            g = int(np.nonzero(codes[f] != ocodes[f])[0][0]) if f < len(codes) else 0
            margins.append(top2_margin(otl[f]) if g == 0 else top2_margin(ocl[f][g - 1]))
        report["margin_at_divergence"] = margins
        _dump(f"divergence_{tag}.json", report)
        assert margins and margins[0] < MARGIN_EPS, report
    s.close(); osess.close()
    return report, codes, ocodes


@pytest.mark.parametrize("sampling", ["greedy", "default"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_free_run_codes_bit_exact_tiny(pair, sampling, use_graph):
    cfg, gm, om = pair
    utt = _utts("custom", 20, hidden=cfg.hidden)
    opts = q.SynthesisOptions(max_length=24, seed=42, eos_token_id=None) if sampling == "default" else \
        q.SynthesisOptions(max_length=24, temperature=0.0, seed=42, eos_token_id=None)
    rep, codes, ocodes = _free_run_compare(cfg, gm, om, utt, opts, use_graph, f"{cfg.name}_{sampling}_{int(use_graph)}")
    assert rep["first_divergence"] is None, rep
    assert (codes == ocodes).all()


def test_eos_and_min_new_tokens(pair):
    """EOS handling: frame whose semantic token is EOS is never emitted (lib.rs:581-585); frame counts
    match the oracle with the default eos id."""
    cfg, gm, om = pair
    for idx in range(3):
        utt = _utts("custom", 5, index=idx, hidden=cfg.hidden)
        opts = q.SynthesisOptions(max_length=40, temperature=1.3, top_k=0, top_p=1.0, seed=7 + idx)
        s = gm.session([utt], opts); s.prefill(); s.generate(40, use_graph=True)
        codes = s.codes(0)
        osess = O.OracleSession(om, utt, opts); ocodes = osess.generate()
        assert len(codes) == len(ocodes) and (codes == ocodes).all()
        assert not (codes[:, 0] == q.CODEC_EOS_TOKEN_ID).any()
        s.close(); osess.close()


@pytest.mark.parametrize("B", [5, 17, 33, 64])
def test_batch_equals_single(pair, B):
    """Every sequence of a batch behaves exactly like its own batch-1 run (bit-exact codes) — up to the 64 sequences of one
    session (B > 16: the wide-batch GEMV with 2 / 3 / 4 column tiles, one prefill position per step)."""
    cfg, gm, om = pair
    utts = [_utts("custom", 11, index=i, hidden=cfg.hidden) for i in range(B)]
    for i, u in enumerate(utts):
        u.seed = 100 + i
    opts = q.SynthesisOptions(max_length=12, seed=1, eos_token_id=None)
    sb = gm.session(utts, opts); sb.prefill(); sb.generate(12, use_graph=True)
    for i in sorted(set([0, 1, B // 2, B - 2, B - 1])):
        s1 = gm.session([utts[i]], opts); s1.prefill(); s1.generate(12, use_graph=False)
        assert (sb.codes(i) == s1.codes(0)).all(), i
        s1.close()
    audio, _ = gm.session(utts, opts).run()
    assert len(audio) == B and all(len(a) == 12 * 1920 for a in audio)
    sb.close()
    with pytest.raises(_lib.Q3Error, match="unsupported"):
        gm.session([utts[0]] * 65, opts)


@pytest.mark.parametrize("T", [1, 2, 10])
def test_decoder_stages_and_pcm(pair, T):
    cfg, gm, om = pair
    rng = np.random.default_rng(T)
    codes = rng.integers(0, 2048, size=(T, 16)).astype(np.uint32)
    codes[:, 0] = rng.integers(0, 3072, size=T)          # semantic ids use the 3072 vocab (mod 2048 in the decoder)
    opcm, otaps = om.decode(codes, taps=True)
    taps = [np.zeros_like(t) for t in otaps]
    pcm = gm.decode_codes(codes, taps=taps).samples
    names = ["quant", "pre_conv", "pre_transformer", "up0", "up1", "init", "blk0", "blk1", "blk2", "blk3"]
    for nme, a, b in zip(names, taps, otaps):
        e = np.abs(a - b).max() / (np.abs(b).max() + 1e-9)
        assert e <= 2e-4, (nme, e)
    rms = pcm_rms(pcm, opcm)
    assert rms <= 1e-3, rms                                # north-star tolerance: PCM within 1e-3 RMS, on unsaturated samples
    assert pcm.shape[0] == T * 1920


def test_streaming_chunks(pair):
    cfg, gm, om = pair
    utt = _utts("custom", 9, hidden=cfg.hidden)
    opts = q.SynthesisOptions(max_length=25, seed=42, eos_token_id=None, chunk_frames=10)
    ss = gm.synthesize_streaming(utt.text_ids, utt.speaker, utt.language, opts)
    chunks = list(ss)
    assert [len(c) for c in chunks] == [19200, 19200, 9600]
    # each chunk is decoded context-free (lib.rs:1755-1758): equals decoding those frames alone
    s = gm.session([q.Utterance(utt.text_ids, utt.speaker, utt.language)], opts); s.prefill(); s.generate(25)
    codes = s.codes(0)
    osess = O.OracleSession(om, q.Utterance(utt.text_ids, utt.speaker, utt.language), opts); ocodes = osess.generate()
    assert (codes == ocodes).all()
    ref = om.decode(ocodes[10:20])
    assert pcm_rms(chunks[1].samples, ref) <= 1e-3
    s.close(); osess.close()


def test_frame_embed_exact(pair):
    cfg, gm, om = pair
    rng = np.random.default_rng(9)
    for _ in range(4):
        sem = int(rng.integers(0, 3072)); codes = rng.integers(0, 2048, 15).astype(np.uint32)
        t = rng.standard_normal(cfg.hidden).astype(np.float32)
        assert (gm.frame_embed(sem, codes, t) == om.frame_embed(sem, codes, t)).all()   # pure f32 adds: bit-exact


# ---------------------------------------------------------------- full-size shapes
@pytest.mark.parametrize("size", ["0.6b", "1.7b"])
def test_full_size_frames(size):
    cfg = q.qwen3_tts_0_6b() if size == "0.6b" else q.qwen3_tts_1_7b()
    gm, om = model_pair(cfg, seed=synth.DEFAULT_SEED)
    utt = q.Utterance(synthetic_prompt(32, 0), seed=42)
    opts = q.SynthesisOptions(max_length=6, seed=42, eos_token_id=None)
    rep, codes, ocodes = _free_run_compare(cfg, gm, om, utt, opts, False, f"full_{size}")
    _dump(f"full_{size}.json", rep)
    if rep["first_divergence"] is None:
        pcm = gm.decode_codes(codes).samples
        opcm = om.decode(ocodes)
        rms = pcm_rms(pcm, opcm)
        assert rms <= 1e-3, rms
    gm.close(); om.close()


@pytest.mark.parametrize("n_text,n_ref_text,n_ref", [(6, 4, 3), (2, 1, 8)])
def test_icl_voice_clone(pair, n_text, n_ref_text, n_ref):
    """ICL voice clone (lib.rs:897-1046): prompt, adjusted options, codes, and the ref-prepended / cut decode."""
    cfg, gm, om = pair
    rng = np.random.default_rng(n_text * 100 + n_ref)
    ref_codes = rng.integers(0, 2048, size=(n_ref, 16)).astype(np.uint32); ref_codes[:, 0] = rng.integers(0, 3072, n_ref)
    utt = q.Utterance(synthetic_prompt(n_text, 3), language=q.Language.Korean, xvector=rng.standard_normal(cfg.hidden).astype(np.float32),
                      ref_codes=ref_codes, ref_text_ids=synthetic_prompt(n_ref_text, 4), seed=5)
    opts = q.SynthesisOptions(max_length=200, seed=5, eos_token_id=None)
    s = gm.session([utt], opts); s.prefill()
    osess = O.OracleSession(om, utt, opts)
    S, Ttr = s.prefill_len(0)
    assert S == osess.prefill_len() == 9 + n_ref + 1
    emb = s.get(0, (S, cfg.hidden)); oemb = osess.prefill_embeds()
    assert np.abs(emb - oemb).max() <= 2e-5 * max(1.0, np.abs(oemb).max())
    hid = s.get(1, (cfg.hidden,)); ohid, olg = osess.prefill_out()
    assert np.abs(hid - ohid).max() <= 1e-4
    s.generate(200)
    codes = s.codes(0); ocodes = osess.generate()
    assert len(ocodes) == max(75, 6 * n_text) == len(codes)
    assert (codes == ocodes).all()
    pcm = s.decode(0)
    full = om.decode(np.concatenate([ref_codes, ocodes], 0))
    cut = n_ref * len(full) // (n_ref + len(ocodes))
    ref = full[cut:]
    assert pcm.shape == ref.shape and pcm_rms(pcm, ref) <= 1e-3
    s.close(); osess.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["tiny", "tiny_same_width"])
def test_from_pretrained_matches_from_tensors(tmp_path, variant):
    """On-disk formats (lib.rs:180-262): a checkpoint directory written by the `safetensors` package (bf16 talker,
    f32 decoder, two tensors stored F16, unrelated speaker_encoder.* / encoder.* tensors present) loaded by the C++
    loader gives the same codes and PCM as the same tensors pushed through q3_model_set_tensor; then the WAV / dump
    writers round-trip the result."""
    from common import write_checkpoint_dir
    base = getattr(q, variant)()
    f16 = ("talker.model.norm.weight", "decoder.pre_conv.conv.bias")
    raw = write_checkpoint_dir(base, str(tmp_path), f16_names=f16)
    # config.json carries no decoder shapes (Decoder12HzConfig::default in the reference, lib.rs:345) and the tiny
    # decoder is smaller than that, so load file by file into a handle built for the tiny config; the directory-level
    # entry point is exercised by test_from_pretrained_directory below
    m = q.Qwen3TTS(base, 0)
    n = ctypes.c_int()
    _lib.check(_lib.lib.q3_model_load_safetensors(m._h, str(tmp_path / "model.safetensors").encode(), ctypes.byref(n)))
    n_main = n.value
    _lib.check(_lib.lib.q3_model_load_safetensors(m._h, str(tmp_path / "speech_tokenizer" / "model.safetensors").encode(), ctypes.byref(n)))
    assert n_main + n.value == len(raw)
    m.finalize()
    ref = q.Qwen3TTS.from_tensors(base, raw, 0)
    opts = q.SynthesisOptions(max_length=12, seed=9, eos_token_id=None)
    utt = q.Utterance(synthetic_prompt(7, 1), seed=9)
    outs = []
    for mm in (m, ref):
        s = mm.session([utt], opts); s.prefill(); s.generate(12)
        outs.append((s.codes(0).copy(), s.decode(0).copy())); s.close()
    assert (outs[0][0] == outs[1][0]).all() and outs[0][0].shape == (12, 16)
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    api.save_codes_binary(str(tmp_path / "codes.bin"), outs[0][0])
    np.testing.assert_array_equal(api.load_codes_binary(str(tmp_path / "codes.bin")), outs[0][0])
    q.AudioBuffer(outs[0][1], 24000).save(str(tmp_path / "o.wav"))
    back = api.load_wav(str(tmp_path / "o.wav"))
    assert len(back) == 12 * 1920
    np.testing.assert_array_equal(back.samples, api.pcm16(outs[0][1]).astype(np.float32) / np.float32(32768.0))
    m.close(); ref.close()


@pytest.mark.gpu
def test_from_pretrained_directory(tmp_path):
    """q3_model_load end to end: tiny talker / code predictor shapes from config.json + the full-size 12 Hz decoder
    (its shapes are compiled-in defaults, as in the reference)."""
    from common import write_checkpoint_dir
    t = q.tiny()
    cfg = q.Q3Config(text_dim=t.text_dim, hidden=t.hidden, inter=t.inter, n_layers=t.n_layers, n_heads=t.n_heads,
                     n_kv_heads=t.n_kv_heads, cp_hidden=t.cp_hidden, cp_inter=t.cp_inter, cp_layers=t.cp_layers,
                     cp_heads=t.cp_heads, cp_kv_heads=t.cp_kv_heads, name="tiny-lm-full-decoder")
    raw = write_checkpoint_dir(cfg, str(tmp_path), model_type="voice_design")
    m = q.Qwen3TTS.from_pretrained(str(tmp_path), 0)
    assert m.model_type == api.ModelType.VoiceDesign and not m.supports_preset_speakers() and m.supports_voice_design()
    assert bytes(m.config.to_c()) == bytes(cfg.to_c())
    ref = q.Qwen3TTS.from_tensors(cfg, raw, 0)
    opts = q.SynthesisOptions(max_length=4, seed=2, eos_token_id=None)
    utt = q.Utterance(synthetic_prompt(5, 2), seed=2)
    outs = []
    for mm in (m, ref):
        s = mm.session([utt], opts); s.prefill(); s.generate(4)
        outs.append((s.codes(0).copy(), s.decode(0).copy())); s.close()
    assert (outs[0][0] == outs[1][0]).all()
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    m.close(); ref.close()


@pytest.mark.gpu
def test_model_load_reports_missing_weight(tmp_path):
    """finalize's "Missing weight: <name>" (decoder_12hz.rs:176-181) surfaces through q3_model_load."""
    import torch
    from safetensors.torch import save_file
    (tmp_path / "speech_tokenizer").mkdir()
    save_file({"talker.model.norm.weight": torch.ones(1024)}, str(tmp_path / "model.safetensors"))
    save_file({"decoder.x": torch.ones(1)}, str(tmp_path / "speech_tokenizer" / "model.safetensors"))
    with pytest.raises(_lib.Q3Error, match="Missing weight"):
        q.Qwen3TTS.from_pretrained(str(tmp_path), 0)
    # wrong element count → shape error naming the tensor
    save_file({"talker.model.norm.weight": torch.ones(1000)}, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text("{}")
    with pytest.raises(_lib.Q3Error, match="talker.model.norm.weight has 1000 elements, expected 1024"):
        q.Qwen3TTS.from_pretrained(str(tmp_path), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("eos", [False, True])
def test_run_overlapped_segment_decode_is_exact(pair, eos, monkeypatch):
    """With Q3_DECODE_OVERLAP=1 q3_session_run decodes 128-frame segments on a second stream while the frame loop
    continues (12 frames of left context re-run per segment). The PCM must equal the whole-utterance decode of the
    same codes bit for bit — with EOS off (all sequences 300 frames) and with sequences that end at different frames."""
    cfg, gm, om = pair
    monkeypatch.setenv("Q3_DECODE_OVERLAP", "1")          # opt-in path (off by default: slower on MI355X); read at every q3_session_run
    utts = [_utts("custom", 4 + i, index=i, hidden=cfg.hidden) for i in range(3)]
    if eos:
        opts = q.SynthesisOptions(max_length=300, temperature=1.3, top_k=0, top_p=1.0, seed=11, min_new_tokens=2)
    else:
        opts = q.SynthesisOptions(max_length=300, seed=11, eos_token_id=None)
    s = gm.session(utts, opts)
    audio, timing = s.run()
    lens = [len(s.codes(b)) for b in range(3)]
    assert timing.generation_frames == sum(lens)
    if not eos:
        assert lens == [300, 300, 300]
    for b in range(3):
        full = s.decode(b)                       # whole-utterance path (q3_session_decode)
        assert audio[b].samples.shape == full.shape == (lens[b] * 1920,)
        np.testing.assert_array_equal(audio[b].samples, full)
    s.close()
    # the oracle agrees on the first sequence (whole-utterance CPU decode)
    s2 = gm.session([utts[0]], opts); s2.prefill(); s2.generate(300); c = s2.codes(0); s2.close()
    if len(c):
        ref = om.decode(c)
        assert pcm_rms(audio[0].samples, ref) <= 1e-3


@pytest.mark.gpu
def test_run_side_by_side_decode_is_exact(pair, monkeypatch):
    """The DEFAULT q3_session_run decode (utterances vocoded four at a time, each on its own stream and workspace) at a
    length beyond one segment: PCM of every sequence equals its whole-utterance decode bit for bit."""
    cfg, gm, om = pair
    monkeypatch.delenv("Q3_DECODE_OVERLAP", raising=False)
    utts = [_utts("custom", 4 + i, index=i, hidden=cfg.hidden) for i in range(6)]
    opts = q.SynthesisOptions(max_length=200, seed=11, eos_token_id=None)
    s = gm.session(utts, opts)
    audio, timing = s.run()
    assert timing.generation_frames == 6 * 200
    for b in range(6):
        np.testing.assert_array_equal(audio[b].samples, s.decode(b))
    ref = om.decode(s.codes(5))
    assert pcm_rms(audio[5].samples, ref) <= 1e-3
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [3, 11])
def test_full_size_decoder_stages(T):
    """The production vocoder shapes (1024/1536/768/384/192/96 channels: the bf16x3 matrix-core conv kernel, all four
    workgroup geometries, the polyphase transposed convs) against the oracle, stage by stage."""
    t = q.tiny()
    cfg = q.Q3Config(text_dim=t.text_dim, hidden=t.hidden, inter=t.inter, n_layers=t.n_layers, n_heads=t.n_heads,
                     n_kv_heads=t.n_kv_heads, cp_hidden=t.cp_hidden, cp_inter=t.cp_inter, cp_layers=t.cp_layers,
                     cp_heads=t.cp_heads, cp_kv_heads=t.cp_kv_heads, name="tiny-lm-full-decoder")
    gm, om = model_pair(cfg, seed=synth.DEFAULT_SEED)
    rng = np.random.default_rng(T)
    codes = rng.integers(0, 2048, size=(T, 16)).astype(np.uint32)
    opcm, otaps = om.decode(codes, taps=True)
    taps = [np.zeros_like(x) for x in otaps]
    pcm = gm.decode_codes(codes, taps=taps).samples
    names = ["quant", "pre_conv", "pre_transformer", "up0", "up1", "init", "blk0", "blk1", "blk2", "blk3"]
    for nme, a, b in zip(names, taps, otaps):
        e = np.abs(a - b).max() / (np.abs(b).max() + 1e-9)
        assert e <= 2e-4, (nme, e)
    rms = pcm_rms(pcm, opcm)
    assert rms <= 1e-3, rms
    gm.close(); om.close()


@pytest.mark.gpu
def test_fused_residual_unit_is_bit_identical(tmp_path):
    """96 / 192-channel residual units run as ONE launch (conv7 + SnakeBeta + 1x1 + residual, the intermediate kept in
    LDS; decoder_block.rs:81-92). Per output element it performs the two-launch arithmetic in the same order, so the PCM
    must be the same bits as with Q3_CODEC_NO_UNIT_FUSE=1 (another process: the switch is read once). 26 frames: both
    widths take the fused path, the last time tile of each layer is partial, all three dilations see real history."""
    import subprocess, sys
    code = (
        "import sys, numpy as np, qwen3_tts_rs_amd as q\n"
        "from qwen3_tts_rs_amd import synth\n"
        "t = q.tiny()\n"
        "cfg = q.Q3Config(text_dim=t.text_dim, hidden=t.hidden, inter=t.inter, n_layers=t.n_layers, n_heads=t.n_heads,"
        " n_kv_heads=t.n_kv_heads, cp_hidden=t.cp_hidden, cp_inter=t.cp_inter, cp_layers=t.cp_layers, cp_heads=t.cp_heads,"
        " cp_kv_heads=t.cp_kv_heads, name='tiny-lm-full-decoder')\n"
        "m = q.Qwen3TTS.from_synthetic(cfg, seed=synth.DEFAULT_SEED)\n"
        "codes = np.random.default_rng(26).integers(0, 2048, size=(26, 16)).astype(np.uint32)\n"
        "np.save(sys.argv[1], m.decode_codes(codes).samples)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, extra in (("fused.npy", {}), ("split.npy", {"Q3_CODEC_NO_UNIT_FUSE": "1"})):
        out = str(tmp_path / name)
        r = subprocess.run([sys.executable, "-c", code, out], cwd=root, env={**os.environ, **extra}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert outs[0].shape == (26 * 1920,) and np.abs(outs[0]).max() > 0.01
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.gpu
def test_weight_arena_broadcast_plumbing(pair):
    """The one data-parallel collective: the weight arena, wrapped without a copy as a torch tensor over
    the library's device pointer, goes through a real RCCL broadcast (world size 1 here — the 8-GPU run is the driver's;
    the 2-rank protocol itself is covered on gloo in test_dp_gloo.py) and the model still produces the same codes."""
    import torch
    import torch.distributed as dist
    from qwen3_tts_rs_amd import dp
    cfg, gm, om = pair
    ptr, nbytes = gm.arena()
    t = torch.as_tensor(dp._DevMem(ptr, nbytes), device="cuda:0")
    assert t.data_ptr() == ptr and t.numel() == nbytes and t.dtype == torch.uint8
    before = int(t[: 1 << 20].to(torch.int64).sum().item())
    opts = q.SynthesisOptions(max_length=5, seed=3, eos_token_id=None)
    utt = _utts("custom", 5)
    s = gm.session([utt], opts); s.prefill(); s.generate(5); c0 = s.codes(0).copy(); s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        for off in range(0, nbytes, 1 << 26):
            dist.broadcast(t[off:off + (1 << 26)], src=0)
        x = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize(0)
        assert float(x.item()) == 1.5
    finally:
        if created:
            dist.destroy_process_group()
    assert int(t[: 1 << 20].to(torch.int64).sum().item()) == before
    s = gm.session([utt], opts); s.prefill(); s.generate(5); c1 = s.codes(0).copy(); s.close()
    assert (c0 == c1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["design", "icl"])
def test_long_prompt_4k(pair, mode):
    """BASELINE config[4]: long-form prompts (≈4k prefill positions — VoiceDesign instruct text, or an ICL reference of
    2000 transcript tokens + 2000 codec frames), hipGraph-replayed decode. Prefill logits and codes against the oracle."""
    cfg, gm, om = pair
    rng = np.random.default_rng(17)
    if mode == "design":
        utt = q.Utterance(synthetic_prompt(64, 1), language=q.Language.German, instruct_ids=synthetic_prompt(4000, 7), seed=4)
        opts = q.SynthesisOptions(max_length=24, seed=4, eos_token_id=None)
    else:
        ref = rng.integers(0, 2048, size=(2000, 16)).astype(np.uint32); ref[:, 0] = rng.integers(0, 3072, 2000)
        utt = q.Utterance(synthetic_prompt(8, 2), language=q.Language.Chinese, xvector=rng.standard_normal(cfg.hidden).astype(np.float32),
                          ref_codes=ref, ref_text_ids=synthetic_prompt(2000, 9), seed=4)
        opts = q.SynthesisOptions(max_length=24, seed=4, eos_token_id=None)
    s = gm.session([utt], opts); s.prefill()
    osess = O.OracleSession(om, utt, opts)
    S, _ = s.prefill_len(0)
    assert S == osess.prefill_len() and S >= 2000
    hid = s.get(1, (cfg.hidden,)); ohid, olg = osess.prefill_out()
    assert np.abs(hid - ohid).max() <= 2e-4
    s.generate(24, use_graph=True)
    codes = s.codes(0); ocodes = osess.generate()
    n = min(len(codes), len(ocodes))
    assert n == len(ocodes) == len(codes) and (codes[:n] == ocodes[:n]).all()
    s.close(); osess.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [10, 7, 33])
def test_streaming_continuous_mode_is_seamless(pair, chunk):
    """SURVEY.md §8(f) rank 1 "improved overlap mode": with q3_session_set_stream_mode(1) the streamed chunks concatenate
    to exactly the non-streaming decode (the reference's context-free chunks do not: every chunk restarts from zero
    padding). Chunk sizes below, equal to and above the 12-frame context."""
    cfg, gm, om = pair
    utt = _utts("custom", 9, hidden=cfg.hidden)
    opts = q.SynthesisOptions(max_length=70, seed=42, eos_token_id=None, chunk_frames=chunk)
    ss = gm.synthesize_streaming(utt.text_ids, utt.speaker, utt.language, opts, continuous=True)
    chunks = list(ss)
    assert sum(len(c) for c in chunks) == 70 * 1920 and all(len(c) == chunk * 1920 for c in chunks[:-1])
    got = np.concatenate([c.samples for c in chunks])
    s = gm.session([q.Utterance(utt.text_ids, utt.speaker, utt.language)], opts); s.prefill(); s.generate(70)
    full = s.decode(0); s.close()
    np.testing.assert_array_equal(got, full)
    # and the default mode really differs (seams), so the test above is not vacuous
    ss0 = gm.synthesize_streaming(utt.text_ids, utt.speaker, utt.language, opts)
    got0 = np.concatenate([c.samples for c in ss0])
    assert got0.shape == full.shape and not np.array_equal(got0, full)


@pytest.mark.gpu
@pytest.mark.parametrize("size", ["0.6b", "1.7b"])
def test_full_size_gemm_prefill(size):
    """Long-prompt prefill (S >= 48: the GEMM + query-blocked attention path of q3_kernels_prefill.hip) at production
    widths: VoiceDesign with 300 instruct tokens (>= 256 positions: the bf16x3 flash attention), two sequences of different text in one batch; last hidden state,
    first-token logits and 4 frames of codes against the oracle."""
    cfg = q.qwen3_tts_0_6b() if size == "0.6b" else q.qwen3_tts_1_7b()
    gm, om = model_pair(cfg, seed=synth.DEFAULT_SEED)
    utts = [q.Utterance(synthetic_prompt(12, i), language=q.Language.German, instruct_ids=synthetic_prompt(300, 30 + i), seed=5 + i) for i in range(2)]
    opts = q.SynthesisOptions(max_length=4, seed=5, eos_token_id=None)
    s = gm.session(utts, opts, debug=False); s.prefill()
    S, _ = s.prefill_len(0)
    assert S >= 300
    for b in range(2):
        osess = O.OracleSession(om, utts[b], opts)
        hid = s.get(1, (cfg.hidden,), b=b); ohid, olg = osess.prefill_out()
        assert np.abs(hid - ohid).max() <= 2e-4 * max(1.0, np.abs(ohid).max()), (b, np.abs(hid - ohid).max())
        lg = s.get(2, (cfg.codec_vocab,), b=b)
        assert np.abs(lg - olg).max() <= 2e-4 * max(1.0, np.abs(olg).max())
        osess.close()
    s.generate(4)
    for b in range(2):
        osess = O.OracleSession(om, utts[b], opts); ocodes = osess.generate(); codes = s.codes(b)
        assert codes.shape == ocodes.shape
        if not (codes == ocodes).all():      # tolerated only at an oracle near-tie
            f, g = np.argwhere(codes != ocodes)[0]
            pytest.fail(f"sequence {b}: first divergence at frame {f} group {g}")
        osess.close()
    ref_codes = [s.codes(b).copy() for b in range(2)]
    s.close()
    # sessions come and go on one model: their buffers are recycled device blocks (DevCache) that must be zeroed and
    # ordered against the non-blocking session stream — a long 4k-position prompt three times in a row, then the first
    # batch again, bit-identical each time
    long_utt = q.Utterance(synthetic_prompt(8, 3), language=q.Language.German, instruct_ids=synthetic_prompt(4000, 77), seed=11)
    got = []
    for rep in range(3):
        s2 = gm.session([long_utt], q.SynthesisOptions(max_length=3, seed=11, eos_token_id=None)); s2.prefill(); s2.generate(3)
        got.append((s2.get(2, (cfg.codec_vocab,), b=0).copy(), s2.codes(0).copy())); s2.close()
    assert np.isfinite(got[0][0]).all() and got[0][1].max() > 0
    for lg, cd in got[1:]:
        assert np.array_equal(lg, got[0][0]) and np.array_equal(cd, got[0][1])
    s3 = gm.session(utts, opts); s3.prefill(); s3.generate(4)
    for b in range(2):
        assert np.array_equal(s3.codes(b), ref_codes[b])
    s3.close(); gm.close(); om.close()


@pytest.mark.gpu
def test_long_prompt_batch_equals_single():
    """Two 2111-position prompts in one session (0.6B width): 4224-row budget => two GEMM passes per layer (2048 + 63
    positions per sequence, the second one attending over the first one's K/V planes), then every sequence must match
    its own batch-1 run (first logits bit for bit, codes equal) — rows of the prefill GEMMs, the bf16x3 flash attention and the 128-tile / tail
    split are all per-sequence independent. Also a 4105-position pair: GEMM passes over 4096 positions + a 9-position
    decode-step tail at 8 rows per sequence."""
    gm = q.Qwen3TTS.from_synthetic(q.qwen3_tts_0_6b(), seed=synth.DEFAULT_SEED)
    opts = q.SynthesisOptions(max_length=6, seed=3, eos_token_id=None)
    for n_ins in (2102, 4096):
        utts = [q.Utterance(synthetic_prompt(12, i), language=q.Language.German, instruct_ids=synthetic_prompt(n_ins, 50 + i), seed=20 + i) for i in range(2)]
        sb = gm.session(utts, opts); sb.prefill()
        assert sb.prefill_len(0)[0] == n_ins + 9
        lb = [sb.get(2, (gm.config.codec_vocab,), b=b).copy() for b in range(2)]
        sb.generate(6, use_graph=True)
        for b in range(2):
            s1 = gm.session([utts[b]], opts); s1.prefill()
            l1 = s1.get(2, (gm.config.codec_vocab,), b=0)
            # GEMM-only prefill: the same bits. With a decode-step tail the KV-split count of the decode attention is a
            # function of the session's batch size (as in every decode step), so only the decisions must agree.
            if n_ins == 2102: assert np.array_equal(l1, lb[b]), (n_ins, b)
            else: assert np.abs(l1 - lb[b]).max() <= 1e-4, (n_ins, b, np.abs(l1 - lb[b]).max())
            s1.generate(6, use_graph=False)
            assert np.array_equal(s1.codes(0), sb.codes(b)), (n_ins, b)
            s1.close()
        sb.close()
    gm.close()


@pytest.mark.gpu
def test_native_rccl_comm_world1(pair, tmp_path):
    """q3_dp_* (the C-ABI data-parallel boundary): RCCL resolved at run time, communicator from a
    file-shipped unique id, the arena broadcast and the timing all-gather run through real RCCL calls (world size 1 on
    this box; the N-rank protocol is the same calls), and the model still produces the same codes afterwards."""
    from qwen3_tts_rs_amd import dp
    cfg, gm, om = pair
    opts = q.SynthesisOptions(max_length=5, seed=3, eos_token_id=None)
    utt = _utts("custom", 5)
    s = gm.session([utt], opts); s.prefill(); s.generate(5); c0 = s.codes(0).copy(); s.close()
    comm = dp.NativeComm.from_file(str(tmp_path / "rccl_id"), 0, 1, 0)
    comm.broadcast_weights(gm, 0)
    got = comm.allgather([1.5, 2.5, 3.5])
    assert got.shape == (1, 3) and got[0].tolist() == [1.5, 2.5, 3.5]
    with pytest.raises(_lib.Q3Error):
        comm.broadcast_weights(gm, 3)          # root outside the communicator
    comm.close()
    s = gm.session([utt], opts); s.prefill(); s.generate(5); c1 = s.codes(0).copy(); s.close()
    assert (c0 == c1).all()


@pytest.mark.gpu
def test_error_contract(pair):
    """Error behaviour of the boundary (SURVEY.md §8b "Errors"): every misuse returns a status with a message, nothing
    aborts; the session stays usable after a refused call."""
    cfg, gm, om = pair
    E = _lib.Q3Error
    opts = q.SynthesisOptions(max_length=4, seed=1, eos_token_id=None)
    with pytest.raises(E, match="out of range"):                      # token id beyond the text vocabulary
        gm.session([q.Utterance([cfg.text_vocab + 5])], opts)
    with pytest.raises(E, match="max_length"):
        gm.session([_utts("custom", 4)], q.SynthesisOptions(max_length=0))
    with pytest.raises(E, match="RoPE table|exceeds"):                # kv_cache.rs:293-300's bail! → Q3_KV_OVERFLOW
        gm.session([_utts("custom", 4)], q.SynthesisOptions(max_length=10_000_000))
    bad = np.zeros((3, 16), np.uint32); bad[1, 5] = cfg.cp_vocab + 1
    with pytest.raises(E, match="out of range for codebook"):         # decode_codes validates every code
        gm.decode_codes(bad)
    s = gm.session([_utts("custom", 4)], opts)
    with pytest.raises(E, match="not prefilled"):
        s.generate(1)
    s.prefill()
    with pytest.raises(E, match="already prefilled"):
        s.prefill()
    with pytest.raises(E, match="bad sequence index"):
        s.codes(3)
    s.generate(4)                                                     # still healthy
    assert s.codes(0).shape == (4, 16) and s.frames(0) == (4, True)
    s.generate(4)                                                     # past max_length: a no-op, not an error
    assert s.frames(0) == (4, True)
    s.close()
    with pytest.raises(E, match="unknown tensor name"):
        gm.set_tensor("talker.model.layers.99.nope", np.zeros(4, np.float32), 0)
    assert _lib.lib.q3_last_error().decode().startswith("unknown tensor name")


@pytest.mark.gpu
def test_soak_mixed_sessions_are_deterministic(pair):
    """Sessions of changing shape come and go on one model (recycled device blocks, graph capture per session, streaming
    read-ahead): the same request gives the same codes and PCM whenever it runs, in whatever company."""
    cfg, gm, om = pair
    rng = np.random.default_rng(2024)

    def request(i):      # key i = (kind, text length); a batch holds one kind (equal prefill lengths, as the engine requires)
        kind = ["custom", "design", "clone"][i // 9]
        return _utts(kind, 5 + (i % 9), hidden=cfg.hidden)

    def run_batch(idx, frames):
        opts = q.SynthesisOptions(max_length=frames, seed=100, eos_token_id=None)
        us = [request(i) for i in idx]
        for k, u in enumerate(us):
            u.seed = 100 + idx[k]
        s = gm.session(us, opts); s.prefill(); s.generate(frames)
        out = [(s.codes(b).copy(), s.decode(b).copy()) for b in range(len(us))]
        s.close()
        return out

    ref = {}
    for it in range(40):
        B = int(rng.integers(1, 6)); frames = int(rng.integers(3, 14))
        kind = int(rng.integers(0, 3))
        idx = [9 * kind + int(x) for x in rng.integers(0, 9, size=B)]
        if it % 5 == 4:                                     # a streaming session in between (read-ahead, second stream)
            i0 = idx[0]; u = request(i0); u.seed = 100 + i0
            ss = api.StreamingSession(gm, u, q.SynthesisOptions(max_length=frames, seed=100, eos_token_id=None, chunk_frames=4), continuous=(it % 10 == 9))
            chunks = [c.samples for c in ss]
            codes = ss._s.codes(0).copy(); ss._s.close()
            assert sum(len(c) for c in chunks) == frames * 1920
            key = (i0, frames)
            if key in ref:
                assert np.array_equal(codes, ref[key][0])
            continue
        out = run_batch(idx, frames)
        for k, i in enumerate(idx):
            key = (i, frames)
            if key in ref:
                assert np.array_equal(out[k][0], ref[key][0]), (it, key)
                assert np.array_equal(out[k][1], ref[key][1]), (it, key)
            else:
                ref[key] = out[k]
    assert len(ref) > 20


@pytest.mark.gpu
def test_run_decodes_utterances_side_by_side(pair):
    """q3_session_run vocodes up to four utterances at a time on separate streams / workspaces: PCM and sample counts of a
    5-utterance batch with different end frames (EOS on) equal the one-at-a-time q3_session_decode results bit for bit."""
    cfg, gm, om = pair
    utts = [_utts("custom", 4 + i, index=i, hidden=cfg.hidden) for i in range(5)]
    opts = q.SynthesisOptions(max_length=40, seed=7)             # EOS enabled: ragged lengths
    s = gm.session(utts, opts)
    audio, timing = s.run()
    lens = [s.frames(b)[0] for b in range(5)]
    assert timing.generation_frames == sum(lens) and all(len(audio[b]) == lens[b] * 1920 for b in range(5))
    for b in range(5):
        np.testing.assert_array_equal(audio[b].samples, s.decode(b))
    s.close()


@pytest.mark.gpu
def test_synthesize_batch_mixed_prompt_kinds(pair):
    """A mixed batch (CustomVoice, voice design with two instruct lengths, x-vector clone: 20 requests, four prefill shapes)
    is served by ONE session — rows of a shape prefilled together, all rows decoded in one frame graph (round 5: ragged first
    batch) — and every utterance equals its own batch-1 run."""
    cfg, gm, om = pair
    kinds = ["custom", "design", "clone", "custom", "design"]
    utts = []
    for i in range(20):
        u = _utts(kinds[i % 5], 3 + i % 4, index=i, hidden=cfg.hidden)
        if kinds[i % 5] == "design" and i % 2:
            u.instruct_ids = synthetic_prompt(11, 60 + i)         # a second instruct length → a third prefill shape
        u.seed = 300 + i
        utts.append(u)
    opts = q.SynthesisOptions(max_length=6, seed=1, eos_token_id=None)
    audio, timing = gm.synthesize_batch(utts, opts)
    assert len(audio) == 20 and timing.generation_frames == 20 * 6
    for i in (0, 1, 6, 12, 19):
        s = gm.session([utts[i]], opts); a1, _ = s.run(); s.close()
        np.testing.assert_array_equal(audio[i].samples, a1[0].samples)
    s = gm.session([utts[0], utts[1]], opts)                          # two prefill lengths in one session: accepted since round 5
    assert s.prefill_len(0)[0] != s.prefill_len(1)[0]
    a2, _ = s.run(); s.close()
    np.testing.assert_array_equal(a2[1].samples, audio[1].samples)


@pytest.mark.gpu
def test_ref_codes_without_transcript_are_prepended_at_decode(pair):
    """A voice-clone prompt that carries reference frames but no transcript keeps the x-vector-only prefill, yet the decode
    still runs over [reference frames ; generated] and cuts the reference's share (lib.rs:1022-1041)."""
    cfg, gm, om = pair
    rng = np.random.default_rng(8)
    ref = rng.integers(0, 2048, size=(5, 16)).astype(np.uint32); ref[:, 0] = rng.integers(0, 3072, 5)
    xv = rng.standard_normal(cfg.hidden).astype(np.float32)
    opts = q.SynthesisOptions(max_length=9, seed=3, eos_token_id=None)
    plain = q.Utterance(synthetic_prompt(6, 1), language=q.Language.French, xvector=xv, seed=3)
    withref = q.Utterance(synthetic_prompt(6, 1), language=q.Language.French, xvector=xv, ref_codes=ref, seed=3)
    s0 = gm.session([plain], opts); s0.prefill(); s0.generate(9); c0 = s0.codes(0); s0.close()
    s1 = gm.session([withref], opts); s1.prefill(); s1.generate(9); c1 = s1.codes(0); pcm = s1.decode(0); s1.close()
    np.testing.assert_array_equal(c0, c1)                        # same prefill, same codes (no ICL block)
    full = om.decode(np.concatenate([ref, c1]))
    cut = 5 * len(full) // 14
    assert pcm.shape == (len(full) - cut,)
    assert pcm_rms(pcm, full[cut:]) <= 1e-3


@pytest.mark.gpu
def test_synthesize_batch_icl_requests_with_different_text_lengths(pair):
    """Two ICL voice-clone requests with the same reference but different text lengths resolve to different max_length
    caps (max(75, 6 * n_text), lib.rs:913-929): one session, every row ending at its own cap."""
    cfg, gm, om = pair
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 2048, size=(4, 16)).astype(np.uint32); ref[:, 0] = rng.integers(0, 3072, 4)
    xv = rng.standard_normal(cfg.hidden).astype(np.float32)
    utts = [q.Utterance(synthetic_prompt(n, 3 + n), language=q.Language.Korean, xvector=xv, ref_codes=ref,
                        ref_text_ids=synthetic_prompt(3, 4), seed=9) for n in (13, 20, 13)]
    opts = q.SynthesisOptions(seed=9, eos_token_id=None)           # default max_length 2048 → caps 78 / 120 / 78
    audio, timing = gm.synthesize_batch(utts, opts)
    # decode = [reference frames ; generated] with the reference's share cut from the front (lib.rs:1022-1041)
    assert [len(a) for a in audio] == [78 * 1920, 120 * 1920, 78 * 1920]
    s = gm.session([utts[1]], opts); a1, _ = s.run(); s.close()
    np.testing.assert_array_equal(audio[1].samples, a1[0].samples)


@pytest.mark.gpu
def test_dp_two_ranks_on_one_gpu():
    """The data-parallel start-up of bench.py with real GPU memory on both sides: 2 ranks (gloo, both on cuda:0), arena
    broadcast from rank 0, non-root finalize, identical codes and PCM on both ranks (tests/dp_same_gpu_check.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(root, "tests", "dp_same_gpu_check.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 0: arena" in r.stdout and "rank 1: arena" in r.stdout and "DIFFER" not in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_same_gpu():
    """The WHOLE bench.py flow at N = 2: self-spawned ranks, arena broadcast, non-root finalize, barriers,
    max-over-ranks timing, the `rccl` record, rank 0's extras while rank 1 waits at the final barrier, destroy_process_group
    on every rank, exit code 0 — both ranks on cuda:0 over gloo (RCCL refuses two ranks per device), tiny model."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["Q3_DP_TEST_SAME_GPU"] = "1"; env["MASTER_ADDR"] = "127.0.0.1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--model", "tiny", "--steps", "1", "--warmup", "1", "--frames", "12",
                        "--batch", "2", "--also-batches", "", "--no-cpu-baseline", "--no-other-configs", "--ttfa-reps", "1", "--prompt-tokens", "16"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["rccl"]["world"] == 2 and sorted(x[0] for x in out["rccl"]["ranks_seen"]) == [0, 1] and out["rccl"]["same_gpu_test_mode"] is True
    assert out["config"]["utterances_per_gpu"] == 2 and out["weight_broadcast"]["bytes"] > 0


@pytest.mark.gpu
def test_bench_eight_ranks_same_gpu():
    """8-rank rehearsal of the driver's SCALE run: `bench.py --gpus 8` spawns eight ranks itself, all on
    cuda:0 over gloo (Q3_DP_TEST_SAME_GPU=1; RCCL refuses several ranks per device), tiny model. Arena broadcast to seven
    non-root ranks, their finalize, barriers, max-over-ranks timing, sum of frames over ranks, eight entries in
    rccl.ranks_seen, every rank leaves with exit code 0."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["Q3_DP_TEST_SAME_GPU"] = "1"; env["MASTER_ADDR"] = "127.0.0.1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--model", "tiny", "--steps", "2", "--warmup", "1", "--frames", "12",
                        "--batch", "2", "--also-batches", "", "--no-cpu-baseline", "--no-other-configs", "--ttfa-reps", "1", "--prompt-tokens", "16"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["value"] > 0
    assert out["rccl"]["world"] == 8 and sorted(x[0] for x in out["rccl"]["ranks_seen"]) == list(range(8)) and out["rccl"]["same_gpu_test_mode"] is True
    assert out["config"]["utterances_per_gpu"] == 2 and out["weight_broadcast"]["bytes"] > 0
    # weak scaling bookkeeping: 8 ranks x 2 utterances x 12 frames x 2 steps over the max-over-ranks time
    assert abs(out["value"] * out["ms_per_step"] / 1000.0 - 8 * 2 * 12) < 1e-6 * 8 * 2 * 12
    _dump("bench_eight_ranks_same_gpu.json", {k: out[k] for k in ("n_gpus", "value", "ms_per_step", "rccl", "weight_broadcast")})


@pytest.mark.gpu
def test_native_rccl_world2_same_gpu(tmp_path):
    """q3_dp_* at world 2 on a one-GPU box: two processes, both on device 0, id shipped through a file. RCCL
    is expected to refuse the duplicate device; the outcome is RECORDED (gpurun_out/native_rccl_world2_same_gpu.json), and the
    requirement is that it is clean either way — both ranks report within the timeout, as `ok` with matching codes or as a
    status + message, never a hang or a crash."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "rccl_w2")
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "dp_native_world2_check.py"), str(r), path], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate(); o = (o or "") + "\n[killed after 240 s]"
        outs.append((p.returncode, o))
    lines = [[l for l in o.splitlines() if l.startswith("rank ")] for _, o in outs]
    _dump("native_rccl_world2_same_gpu.json", {"returncodes": [rc for rc, _ in outs], "lines": lines, "tails": [o[-600:] for _, o in outs]})
    for rc, o in outs:
        assert rc == 0, o[-2000:]
    assert all(len(l) == 1 for l in lines), outs
    kinds = [l[0].split()[2] for l in lines]
    assert all(k in ("ok", "refused", "error") for k in kinds)
    if all(k == "ok" for k in kinds):
        assert all("match" in l[0] for l in lines)


# ---------------------------------------------------------------- continuous batching (q3_session_replace, per-row limits, per-row streaming)
@pytest.mark.gpu
def test_rows_end_at_their_own_limit(pair):
    """Each row of a session stops at its own max_length (SampleArgs::limit): the session runs until the longest row is done, a
    frozen row records nothing further, and every row equals its own batch-1 run bit for bit."""
    cfg, gm, om = pair
    lims = [5, 12, 9, 1]
    utts = [_utts("custom", 7, index=i, hidden=cfg.hidden) for i in range(4)]
    for i, u in enumerate(utts):
        u.seed = 50 + i; u.max_length = lims[i]
    opts = q.SynthesisOptions(max_length=12, seed=1, eos_token_id=None)
    s = gm.session(utts, opts); s.prefill(); s.generate(100, use_graph=True)
    for i, u in enumerate(utts):
        n, done = s.frames(i)
        assert (n, done) == (lims[i], True)
        s1 = gm.session([u], opts); s1.prefill(); s1.generate(100, use_graph=False)
        assert s1.frames(0) == (lims[i], True)
        np.testing.assert_array_equal(s.codes(i), s1.codes(0))
        np.testing.assert_array_equal(s.decode(i), s1.decode(0))
        s1.close()
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False])
def test_replace_staggered_arrivals(pair, graph):
    """Continuous batching: nine requests with different lengths, texts and seeds go through a THREE-row session;
    a row that ends is swapped for the next waiting request at the 4-frame poll (q3_session_replace) while the other rows keep
    running. Every request's codes and PCM are bit-equal to its own batch-1 run — the swap is invisible to the row that is
    swapped in (it starts from its own prefill) and to the rows that stay."""
    cfg, gm, om = pair
    lens = [6, 14, 9, 4, 11, 7, 13, 5, 8]
    utts = [_utts("custom", 5 + (i % 4), index=i, hidden=cfg.hidden) for i in range(9)]
    for i, u in enumerate(utts):
        u.seed = 300 + i; u.max_length = lens[i]
    opts = q.SynthesisOptions(max_length=14, seed=1, eos_token_id=None)
    codes, pcm, frames, wall = gm.synthesize_continuous(utts, opts, slots=3, poll_frames=4, use_graph=graph)
    assert frames == sum(lens)
    for i, u in enumerate(utts):
        s1 = gm.session([u], opts); s1.prefill(); s1.generate(100, use_graph=False)
        np.testing.assert_array_equal(codes[i], s1.codes(0), err_msg=f"request {i}")
        np.testing.assert_array_equal(pcm[i], s1.decode(0), err_msg=f"request {i}")
        s1.close()


@pytest.mark.gpu
def test_replace_with_eos_and_other_modes(pair):
    """Rows that end by EOS are detected at the poll and replaced; the replacement may be another prompt flavour (x-vector /
    voice design: a different prefill length — positions are per row) as long as it fits the row; misuse is refused."""
    cfg, gm, om = pair
    opts = q.SynthesisOptions(max_length=40, seed=3)                      # EOS on: synthetic weights sample it now and then
    base = [_utts("custom", 6, index=i, hidden=cfg.hidden) for i in range(2)]
    s = gm.session(base, opts); s.prefill(); s.generate(8, use_graph=True)
    n0, d0 = s.frames(0)
    repl = _utts("clone", 4, index=7, hidden=cfg.hidden); repl.seed = 77; repl.max_length = 10
    s.replace(0, repl)                                                    # row 0 abandoned mid-utterance: allowed, its frames restart at 0
    s.generate(200, use_graph=True)
    s1 = gm.session([repl], opts); s1.prefill(); s1.generate(200, use_graph=False)
    np.testing.assert_array_equal(s.codes(0), s1.codes(0))
    s1.close()
    s2 = gm.session([base[1]], opts); s2.prefill(); s2.generate(200, use_graph=False)      # the row that stayed is untouched
    np.testing.assert_array_equal(s.codes(1), s2.codes(0))
    s2.close()
    design = _utts("design", 3, index=2, hidden=cfg.hidden); design.max_length = 6
    s.replace(1, design)
    s.generate(200, use_graph=True)
    s3 = gm.session([design], opts); s3.prefill(); s3.generate(200, use_graph=False)
    np.testing.assert_array_equal(s.codes(1), s3.codes(0))
    s3.close()
    too_long = _utts("custom", 1100, index=1, hidden=cfg.hidden)
    with pytest.raises(_lib.Q3Error, match="exceed the session's slot"):
        s.replace(0, too_long)
    over = _utts("custom", 3, index=1, hidden=cfg.hidden); over.max_length = 41
    with pytest.raises(_lib.Q3Error, match="frame budget"):
        s.replace(0, over)
    s.close()


@pytest.mark.gpu
def test_streaming_several_rows(pair):
    """One StreamingSession per row of a three-row session (q3_session_next_chunk_row): chunks of every row equal the chunks
    of its own batch-1 streaming session, in both chunk-decode modes."""
    cfg, gm, om = pair
    utts = [_utts("custom", 6, index=i, hidden=cfg.hidden) for i in range(3)]
    for i, u in enumerate(utts):
        u.seed = 20 + i; u.max_length = [23, 10, 17][i]
    opts = q.SynthesisOptions(max_length=23, seed=1, eos_token_id=None, chunk_frames=7)
    for continuous in (False, True):
        s = gm.session(utts, opts)
        if continuous:
            _lib.check(_lib.lib.q3_session_set_stream_mode(s._h, 1))
        got = [[] for _ in utts]; done = [False] * 3
        while not all(done):
            for b in range(3):
                if not done[b]:
                    chunk, d = s.next_chunk_row(b)
                    if chunk is not None:
                        got[b].append(chunk.samples)
                    done[b] = d or chunk is None
        s.close()
        for b, u in enumerate(utts):
            ref = [c.samples for c in api.StreamingSession(gm, u, opts, continuous=continuous)]
            assert len(ref) == len(got[b]) and all(np.array_equal(x, y) for x, y in zip(ref, got[b])), (continuous, b)


@pytest.mark.gpu
def test_replace_icl_rows(pair):
    """ICL requests (reference codes + reference text: repetition penalty floored at 1.5, length capped, reference frames
    prepended and cut at decode) can be swapped into a running session, and so can a plain request afterwards: every row
    carries its own sampler settings (SampleRow), read by the captured sampler from device memory."""
    cfg, gm, om = pair
    rng = np.random.default_rng(5)
    def icl(i, n_ref, n_text, n_ref_text):
        xv = rng.standard_normal(cfg.hidden).astype(np.float32)
        ref = rng.integers(0, 2048, size=(n_ref, 16)).astype(np.uint32)
        return q.Utterance(synthetic_prompt(n_text, i), language=q.Language.French, xvector=xv, ref_codes=ref,
                           ref_text_ids=synthetic_prompt(n_ref_text, 40 + i), seed=60 + i)
    a, b, c = icl(0, 4, 5, 3), icl(1, 4, 7, 3), icl(2, 6, 4, 2)
    a.max_length = 9; b.max_length = 14; c.max_length = 11
    opts = q.SynthesisOptions(max_length=20, seed=1, eos_token_id=None)
    s = gm.session([a, b], opts); s.prefill(); s.generate(9, use_graph=True)
    assert s.frames(0) == (9, True)
    pcm_a = s.decode(0)
    s.replace(0, c)                                     # a different prefill length (6 reference frames instead of 4)
    s.generate(100, use_graph=True)
    for row, u, pcm_row in ((0, c, None), (1, b, None)):
        s1 = gm.session([u], opts); s1.prefill(); s1.generate(100, use_graph=False)
        np.testing.assert_array_equal(s.codes(row), s1.codes(0))
        np.testing.assert_array_equal(s.decode(row), s1.decode(0))
        s1.close()
    s1 = gm.session([a], opts); s1.prefill(); s1.generate(100, use_graph=False)
    np.testing.assert_array_equal(pcm_a, s1.decode(0)); s1.close()
    plain = _utts("custom", 5, index=3, hidden=cfg.hidden); plain.max_length = 5     # no ICL penalty floor: its own sampler row
    s.replace(1, plain); s.generate(100, use_graph=True)
    s1 = gm.session([plain], opts); s1.prefill(); s1.generate(100, use_graph=False)
    np.testing.assert_array_equal(s.codes(1), s1.codes(0))
    np.testing.assert_array_equal(s.decode(1), s1.decode(0)); s1.close()
    s.close()


@pytest.mark.gpu
def test_rows_with_their_own_sampling_options(pair):
    """SynthesisOptions is per call in the reference (lib.rs:1786-1836), so requests that share a session may differ in
    every sampler setting: greedy, temperature / top-k / top-p, repetition penalty, min_new_tokens and the EOS id.  Each
    row of the mixed batch equals the same request run alone, and equals the oracle with that request's options."""
    cfg, gm, om = pair
    base = dict(max_length=12, seed=1)
    variants = [q.SynthesisOptions(temperature=0.0, eos_token_id=None, **base),                                 # greedy
                q.SynthesisOptions(temperature=1.3, top_k=5, top_p=1.0, repetition_penalty=1.0, eos_token_id=None, **base),
                q.SynthesisOptions(temperature=0.7, top_k=0, top_p=0.8, repetition_penalty=1.4, eos_token_id=None, **base),
                q.SynthesisOptions(min_new_tokens=4, **base),                                                   # default sampler, EOS live
                q.SynthesisOptions(temperature=0.9, top_k=50, top_p=0.9, repetition_penalty=1.05, eos_token_id=None, **base)]
    utts = []
    for i, o in enumerate(variants):
        u = _utts("custom", 5 + i, index=i, hidden=cfg.hidden); u.seed = 70 + i; u.options = o
        utts.append(u)
    host = q.SynthesisOptions(**base)
    for use_graph in (False, True):
        s = gm.session(utts, host); s.prefill(); s.generate(100, use_graph=use_graph)
        for i, u in enumerate(utts):
            s1 = gm.session([u], u.options); s1.prefill(); s1.generate(100, use_graph=False)
            np.testing.assert_array_equal(s.codes(i), s1.codes(0))
            np.testing.assert_array_equal(s.decode(i), s1.decode(0)); s1.close()
            osess = O.OracleSession(om, u, u.options)
            np.testing.assert_array_equal(s.codes(i), osess.generate()); osess.close()
        s.close()
    # and a request with other options swapped into a finished slot
    s = gm.session(utts[:2], host); s.prefill(); s.generate(100, use_graph=True)
    s.replace(0, utts[2]); s.generate(100, use_graph=True)
    s1 = gm.session([utts[2]], utts[2].options); s1.prefill(); s1.generate(100, use_graph=False)
    np.testing.assert_array_equal(s.codes(0), s1.codes(0)); s1.close(); s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_continuous_batching_fuzz(pair, seed):
    """Random traffic through one three-row session: twelve requests of all four prompt kinds (CustomVoice, VoiceDesign,
    x-vector clone, ICL clone), random text lengths, frame limits, sampler settings and EOS handling, swapped into whichever
    row ends first at a random poll interval. Every request must come out with the codes and the PCM of its own batch-1
    run (the reference keeps all state per call: lib.rs:743-756) — whatever its neighbours were doing."""
    cfg, gm, om = pair
    rng = np.random.default_rng(1000 + seed)
    def options():
        kind = rng.integers(0, 4)
        eos = None if rng.integers(0, 2) else q.SynthesisOptions().eos_token_id
        if kind == 0: return q.SynthesisOptions(temperature=0.0, eos_token_id=eos, max_length=16, seed=1)
        if kind == 1: return q.SynthesisOptions(temperature=1.2, top_k=int(rng.integers(2, 40)), top_p=1.0, eos_token_id=eos, max_length=16, seed=1)
        if kind == 2: return q.SynthesisOptions(temperature=0.8, top_k=0, top_p=0.85, repetition_penalty=1.3, eos_token_id=eos, max_length=16, seed=1)
        return q.SynthesisOptions(eos_token_id=eos, min_new_tokens=int(rng.integers(0, 5)), max_length=16, seed=1)
    def request(i, kind):
        n_text = int(rng.integers(1, 13))
        if kind == 3:
            u = q.Utterance(synthetic_prompt(n_text, i), language=q.Language.French, xvector=rng.standard_normal(cfg.hidden).astype(np.float32),
                            ref_codes=rng.integers(0, 2048, size=(int(rng.integers(2, 7)), 16)).astype(np.uint32),
                            ref_text_ids=synthetic_prompt(int(rng.integers(1, 5)), 90 + i))
        else:
            u = _utts(("custom", "design", "clone")[kind], n_text, index=i, hidden=cfg.hidden)
        u.seed = 500 + 17 * i + seed; u.max_length = int(rng.integers(3, 15)); u.options = options()
        return u
    # seed 0: the first batch shares one prefill shape (one batched prefill); the others: any kinds from the first row on
    utts = [request(i, 0 if seed == 0 else int(rng.integers(0, 4))) for i in range(3)] + [request(i, int(rng.integers(0, 4))) for i in range(3, 12)]
    host = q.SynthesisOptions(max_length=16, seed=1)
    codes, pcm, frames, _ = gm.synthesize_continuous(utts, host, slots=3, poll_frames=int(rng.choice([1, 3, 8])), use_graph=bool(seed != 1))
    assert frames == sum(c.shape[0] for c in codes)
    for i, u in enumerate(utts):
        s1 = gm.session([u], u.options); s1.prefill(); s1.generate(100, use_graph=False)
        np.testing.assert_array_equal(codes[i], s1.codes(0), err_msg=f"request {i}")
        np.testing.assert_array_equal(pcm[i], s1.decode(0), err_msg=f"request {i}")
        s1.close()
    if seed == 0:                                                      # and against the oracle, request by request
        for i, u in enumerate(utts):
            osess = O.OracleSession(om, u, u.options)
            np.testing.assert_array_equal(codes[i], osess.generate(), err_msg=f"request {i} vs oracle"); osess.close()


@pytest.mark.gpu
def test_native_batcher(pair):
    """q3_batcher: the native serving loop. Requests of all four prompt kinds with their own options are queued — some before
    the first step, some while others are running — and every one comes back with the codes and PCM of its batch-1 run. A
    request that cannot be placed fails alone, with the engine's message, and the queue moves on."""
    cfg, gm, om = pair
    rng = np.random.default_rng(77)
    def request(i, kind, L):
        n_text = int(rng.integers(1, 11))
        if kind == 3:
            u = q.Utterance(synthetic_prompt(n_text, i), language=q.Language.French, xvector=rng.standard_normal(cfg.hidden).astype(np.float32),
                            ref_codes=rng.integers(0, 2048, size=(4, 16)).astype(np.uint32), ref_text_ids=synthetic_prompt(3, 90 + i))
        else:
            u = _utts(("custom", "design", "clone")[kind], n_text, index=i, hidden=cfg.hidden)
        u.seed = 900 + i; u.max_length = L
        u.options = q.SynthesisOptions(temperature=0.0 if i % 3 == 0 else 0.9, eos_token_id=None, max_length=16, seed=1)
        return u
    utts = [request(i, k, L) for i, (k, L) in enumerate([(1, 5), (0, 12), (3, 7), (2, 9), (0, 3), (1, 14), (3, 6), (2, 11), (0, 8)])]
    b = q.Batcher(gm, slots=3, frame_budget=16, prompt_budget=40, options=q.SynthesisOptions(max_length=16, seed=1))
    tickets = [b.submit(u) for u in utts[:5]]
    assert b.poll(tickets[0])[0] == q.Batcher.QUEUED
    finished = 0
    running, queued, f = b.step(4); finished += f
    assert running == 3 and queued == 2 and f == 0                       # three rows busy (limits 5 / 12 / 7), two requests waiting
    tickets += [b.submit(u) for u in utts[5:]]                           # late arrivals
    too_long = q.Utterance(synthetic_prompt(1100, 5), q.Speaker.Ryan, q.Language.English); too_long.max_length = 4
    bad = b.submit(too_long)
    with pytest.raises(_lib.Q3Error, match="frame budget"):
        over = _utts("custom", 4, index=1, hidden=cfg.hidden); over.max_length = 17
        b.submit(over)
    for _ in range(200):
        running, queued, f = b.step(3); finished += f
        if running == 0 and queued == 0:
            break
    assert finished == len(utts) + 1
    assert b.poll(bad)[0] == q.Batcher.FAILED
    with pytest.raises(_lib.Q3Error, match="exceed the session's slot"):
        b.fetch(bad)
    for i, (u, t) in enumerate(zip(utts, tickets)):
        assert b.poll(t)[0] == q.Batcher.DONE
        codes, pcm = b.fetch(t)
        s1 = gm.session([u], u.options); s1.prefill(); s1.generate(100, use_graph=False)
        np.testing.assert_array_equal(codes, s1.codes(0), err_msg=f"request {i}")
        np.testing.assert_array_equal(pcm, s1.decode(0), err_msg=f"request {i}"); s1.close()
    with pytest.raises(_lib.Q3Error, match="unknown ticket"):
        b.poll(tickets[0])                                              # released by fetch
    # the batcher stays usable after draining: a new request goes into an idle row of the same session
    t = b.submit(utts[2]); b.step(50)
    codes, _ = b.fetch(t)
    s1 = gm.session([utts[2]], utts[2].options); s1.prefill(); s1.generate(100, use_graph=False)
    np.testing.assert_array_equal(codes, s1.codes(0)); s1.close()
    b.close()


@pytest.mark.gpu
def test_batcher_bad_first_request_fails_alone(pair):
    """the batcher opens its session on idle rows built from fixed valid values, never from the first queued
    request — a malformed FIRST request (repetition_penalty <= 0, NaN temperature) fails alone at its own swap and the queue
    moves on; budgets beyond the RoPE table are refused at q3_batcher_create."""
    cfg, gm, om = pair
    opts = q.SynthesisOptions(max_length=6, seed=1, eos_token_id=None)
    with pytest.raises(_lib.Q3Error, match="RoPE table"):
        q.Batcher(gm, slots=2, frame_budget=8000, prompt_budget=4000, options=opts)
    b = q.Batcher(gm, slots=2, frame_budget=6, prompt_budget=0, options=opts)
    try:
        bad = _utts("custom", 5, index=1, hidden=cfg.hidden)
        bad.options = q.SynthesisOptions(max_length=6, seed=1, eos_token_id=None, repetition_penalty=0.0)
        bad2 = _utts("custom", 5, index=2, hidden=cfg.hidden)
        bad2.options = q.SynthesisOptions(max_length=6, seed=1, eos_token_id=None, temperature=float("nan"))
        good = [_utts("custom", 4 + i, index=3 + i, hidden=cfg.hidden) for i in range(3)]
        t_bad = b.submit(bad, want_pcm=False); t_bad2 = b.submit(bad2, want_pcm=False)
        t_good = [b.submit(u, want_pcm=False) for u in good]
        for _ in range(30):
            running, queued, _ = b.step(4)
            if running == 0 and queued == 0:
                break
        assert b.poll(t_bad)[0] == q.Batcher.FAILED and b.poll(t_bad2)[0] == q.Batcher.FAILED
        for u, t in zip(good, t_good):
            assert b.poll(t)[0] == q.Batcher.DONE
            codes, _ = b.fetch(t)
            s1 = gm.session([u], opts); s1.prefill(); s1.generate(6, use_graph=False)
            np.testing.assert_array_equal(codes, s1.codes(0)); s1.close()
    finally:
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_engine_frame_loop_matches_upstream_talker_generate(tag):
    """The PRODUCT (HIP engine through the C-ABI) against third-party code directly: the codes of 24 greedy frames equal those of
    Hugging Face's own talker loop (Qwen3OmniMoeTalkerForConditionalGeneration.generate() with its own code predictor and caches,
    tests/golden/hf_frame_loop.npz; what differs from the reference is listed in make_golden_hf.py::frame_loop_fixture) — not via the
    same-author oracle. Both runs of the fixture: plain logits, and the default penalties riding along."""
    from test_oracle_vs_hf import _frame_loop_case, check_codes_against_upstream_loop, SEED as HF_SEED
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_frame_loop.npz"))
    cfg, utt, runs = _frame_loop_case(fx, tag)
    gm, om = model_pair(cfg, seed=HF_SEED)
    for rn, ro in runs.items():
        n = len(fx[f"{tag}_{rn}_code0"])
        s = gm.session([utt], q.SynthesisOptions(max_length=n, seed=1, temperature=0.0, **ro))
        s.prefill(); s.generate(n, use_graph=True)
        codes = np.asarray(s.codes(0))
        s.close()
        check_codes_against_upstream_loop(fx, tag, rn, codes)
    gm.close(); om.close()
