"""Parity at the BENCHMARK's own configuration (-m gpu): Qwen3-TTS-1.7B, 8 / 16 utterances per GPU, 512-token prompts,
hipGraph frame loop, full-size vocoder at 128 / 640 frames, 1.7B streaming chunks, 4k-position prefill, 0.6B single
utterance — against fixtures the CPU oracle produced in the build container (tests/make_golden_bench.py). A codec-id
mismatch is tolerated only where the oracle itself is at a near-tie; then the oracle is re-run LIVE for that one sequence
to adjudicate, and the case is written to gpurun_out/."""
import ctypes
import json
import os

import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth
import oracle as O
from common import oracle_model, synthetic_prompt, top2_margin, pcm_rms
from make_golden_bench import bench_utt, tap_indices, pcm_decimate_idx, N_FRAMES

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
# Round 6: the allowances follow what is MEASURED — every logit of every configuration is within 3.6e-5 of the
# oracle (gpurun_out/bench_teacher_forced_m8.json, bench_long_context_b8.json) — instead of 2e-3 / 2e-4, which a real 1e-3 kernel
# regression would have passed as a "near-tie".
MARGIN_EPS = 2e-4          # greedy decisions: tolerated only below this oracle top-2 logit margin
LOGIT_NOISE = 5e-5         # sampled decisions: tolerated only if logit noise of this size reproduces the GPU's token


def _dump(name, obj):
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, name), "w") as f:
        json.dump(obj, f, indent=1)


@pytest.fixture(scope="module")
def gm17():
    m = q.Qwen3TTS.from_synthetic(q.qwen3_tts_1_7b(), seed=synth.DEFAULT_SEED)
    yield m
    m.close()


_oracles = {}


def _oracle(cfg_name):
    """Live oracle, built only when a mismatch has to be adjudicated."""
    if cfg_name not in _oracles:
        cfg = q.qwen3_tts_1_7b() if cfg_name == "1.7b" else q.qwen3_tts_0_6b()
        _oracles[cfg_name] = oracle_model(cfg, seed=synth.DEFAULT_SEED, which=1)
    return _oracles[cfg_name]


def _adjudicate(cfg_name, utt, opts, gpu_codes, tag):
    """Re-run the oracle for one sequence; the first differing decision must sit at an oracle near-tie."""
    om = _oracle(cfg_name)
    s = O.OracleSession(om, utt, opts)
    ocodes, tl, cl = s.generate(capture=True)
    s.close()
    n = min(len(ocodes), len(gpu_codes))
    f = next(i for i in range(n) if not (ocodes[i] == gpu_codes[i]).all())
    g = int(np.nonzero(ocodes[f] != gpu_codes[f])[0][0])
    rep = {"tag": tag, "frame": f, "group": g, "oracle": int(ocodes[f][g]), "gpu": int(gpu_codes[f][g])}
    if g > 0:                       # code predictor: always greedy (code_predictor.rs:404-410)
        rep["margin"] = top2_margin(cl[f][g - 1])
        ok = rep["margin"] < MARGIN_EPS
    else:
        V = tl.shape[1]
        seen = np.zeros(V, np.uint8)
        for t in ocodes[:f, 0]:
            if t < V:
                seen[t] = 1
        eos = -1 if opts.eos_token_id is None else opts.eos_token_id
        seed = utt.seed if utt.seed is not None else opts.seed
        rng = np.random.default_rng(f)
        hits = 0
        for trial in range(128):
            lg = (tl[f] + (LOGIT_NOISE * rng.standard_normal(V) if trial else 0.0)).astype(np.float32)
            O.olib.q3o_apply_penalties(O.ptr(lg), V, O.ptr(seen), float(opts.repetition_penalty), f, opts.min_new_tokens, eos)
            st = ctypes.c_uint64(); O.olib.q3o_rng_seed(int(seed), ctypes.byref(st))
            for _ in range(f):
                O.olib.q3o_rng_next(ctypes.byref(st))
            tok = O.olib.q3o_sample(O.ptr(lg), V, float(opts.temperature), opts.top_k, float(opts.top_p), ctypes.byref(st))
            hits += int(tok == gpu_codes[f][0])
        rep["noise_trials_reproducing_gpu_token"] = hits
        ok = hits > 0
    _dump(f"bench_divergence_{tag}.json", rep)
    return ok, rep


def _check_free_run(gm, cfg_name, utts, opts, ref_codes, use_graph, tag, margins=None):
    s = gm.session(utts, opts); s.prefill()
    s.generate(opts.max_length, use_graph=use_graph)
    report = []
    for b, u in enumerate(utts):
        codes = s.codes(b)
        assert codes.shape == ref_codes[b].shape, (tag, b, codes.shape)
        if not (codes == ref_codes[b]).all():
            ok, rep = _adjudicate(cfg_name, u, opts, codes, f"{tag}_seq{b}")
            report.append(rep)
            assert ok, rep
    s.close()
    _dump(f"bench_parity_{tag}.json", {"tag": tag, "sequences": len(utts), "frames": int(opts.max_length), "near_tie_divergences": report})
    return report


@pytest.mark.parametrize("B", [8, 16])
@pytest.mark.parametrize("sampling", ["default", "greedy"])
def test_b8_b16_graph_codes(gm17, B, sampling):
    """BASELINE config[3] per GPU: B utterances, 512-token prompts, seeds 42+i, hipGraph — codec ids of every sequence
    bit-exact against the oracle fixture (SURVEY §8d cfg B = default sampling, cfg A = greedy)."""
    ref = np.load(os.path.join(G, "bench_1_7b_codes.npz"))[f"{sampling}_codes"]
    if B > ref.shape[0]:
        pytest.skip("greedy fixture holds 8 sequences")
    kw = dict(temperature=0.0) if sampling == "greedy" else {}
    opts = q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42, **kw)
    utts = [bench_utt(i) for i in range(B)]
    rep = _check_free_run(gm17, "1.7b", utts, opts, ref[:B], True, f"1_7b_b{B}_{sampling}_graph")
    assert len(rep) <= max(1, B // 8), rep          # near-ties are rare: at most one per eight sequences


def test_continuous_batching_1_7b(gm17):
    """Continuous batching at the benchmark's own size: the 16 fixture sequences (512-token prompts, default sampling) with
    frame limits between 6 and 32 go through ONE eight-row hipGraph session — whenever a row reaches its limit the next
    waiting request is prefilled on the side and swapped into it (q3_session_replace) while the other seven keep running.
    Every request's codes must be the oracle's for that sequence, cut at its limit (a row's i-th random draw and its
    arithmetic do not depend on its neighbours or on when it started)."""
    ref = np.load(os.path.join(G, "bench_1_7b_codes.npz"))["default_codes"]
    limits = [6, 32, 11, 24, 9, 32, 17, 28, 32, 7, 20, 13, 32, 10, 26, 15]
    utts = []
    for i, L in enumerate(limits):
        u = bench_utt(i); u.max_length = L
        utts.append(u)
    opts = q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42)
    codes, _, frames, _ = gm17.synthesize_continuous(utts, opts, slots=8, poll_frames=4, decode=False, use_graph=True)
    assert frames == sum(limits)
    bad = []
    for i, L in enumerate(limits):
        assert codes[i].shape == (L, 16), (i, codes[i].shape)
        if not (codes[i] == ref[i][:L]).all():
            o1 = q.SynthesisOptions(max_length=L, eos_token_id=None, seed=42)
            ok, rep = _adjudicate("1.7b", utts[i], o1, codes[i], f"1_7b_continuous_seq{i}")
            assert ok, rep
            bad.append(rep)
    assert len(bad) <= 2, bad
    _dump("bench_parity_1_7b_continuous.json", {"requests": len(limits), "slots": 8, "frames": int(frames), "limits": limits, "near_tie_divergences": bad})


@pytest.mark.parametrize("B", [32, 64])
def test_b32_b64_session_1_7b(gm17, B):
    """One session carrying 32 / 64 utterances at 1.7B (B = 32: wide-batch GEMV; B = 64: split-K GEMM + slice sums for the
    wide outputs, the two-addend split-K kernel over blocks of 16 rows for o / down; two attention splits): the 16
    sequences the oracle fixture holds are compared with it, the others with their own 8-utterance sessions (HIP vs HIP)."""
    ref = np.load(os.path.join(G, "bench_1_7b_codes.npz"))["default_codes"]
    opts = q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42)
    utts = [bench_utt(i) for i in range(B)]
    s = gm17.session(utts, opts); s.prefill(); s.generate(N_FRAMES, use_graph=True)
    codes = [s.codes(b) for b in range(B)]
    s.close()
    bad = []
    for b in range(16):
        if not (codes[b] == ref[b]).all():
            ok, rep = _adjudicate("1.7b", utts[b], opts, codes[b], f"1_7b_b{B}_seq{b}")
            assert ok, rep
            bad.append(rep)
    assert len(bad) <= 2, bad
    # rows 16 .. B-1: the 64-utterance fixture holds the oracle's first 8 frames of every utterance of the job (round 5)
    w64 = np.load(os.path.join(G, "bench_1_7b_wide64.npz"))
    o8 = q.SynthesisOptions(max_length=8, eos_token_id=None, seed=42)
    bad64 = []
    for b in range(16, B):
        if not (codes[b][:8] == w64["codes"][b]).all():
            ok, rep = _adjudicate("1.7b", utts[b], o8, codes[b][:8], f"1_7b_b{B}_seq{b}_first8")
            assert ok, rep
            bad64.append(rep)
    assert len(bad64) <= max(1, B // 32), bad64
    flips = 0
    for g0 in range(16, B, 8):
        s8 = gm17.session(utts[g0:g0 + 8], opts); s8.prefill(); s8.generate(N_FRAMES, use_graph=True)
        same = sum(int((s8.codes(i) == codes[g0 + i]).all()) for i in range(8))
        s8.close()
        assert same >= 7, (g0, same)        # different GEMV kernels (M = 8 vs M = 32 / 64): a near-tie may flip one sequence
        flips += 8 - same
    assert flips <= max(2, B // 16), flips
    _dump(f"bench_parity_1_7b_b{B}.json", {"oracle_near_ties": bad, "oracle_near_ties_rows_16_up_first8": bad64, "hip_vs_hip_flips": flips})


def test_config3_every_rank_shard_vs_oracle(gm17):
    """BASELINE config[3] = 64 utterances data-parallel over 8 GPUs, utterance i on rank i mod 8. The fixture holds
    the oracle's first 8 frames (default sampling, seed 42 + i) of ALL 64 utterances: every rank's shard — eight 8-row hipGraph
    sessions, run here one after the other on the one GPU — is compared with it, so no utterance of the job is only ever
    checked HIP against HIP."""
    w64 = np.load(os.path.join(G, "bench_1_7b_wide64.npz"))
    opts = q.SynthesisOptions(max_length=8, eos_token_id=None, seed=42)
    rep_all = []
    for rank in range(8):
        idx = list(range(rank, 64, 8))
        utts = [bench_utt(i) for i in idx]
        rep = _check_free_run(gm17, "1.7b", utts, opts, w64["codes"][idx], True, f"1_7b_config3_rank{rank}")
        rep_all += rep
    assert len(rep_all) <= 2, rep_all
    _dump("bench_parity_1_7b_config3_shards.json", {"ranks": 8, "utterances": 64, "frames": 8, "near_tie_divergences": rep_all})


def test_native_batcher_1_7b(gm17):
    """The native serving loop at full width with prompt kinds mixed: six CustomVoice fixture sequences with their own frame
    limits (default sampling) and the x-vector / ICL voice-clone fixtures (greedy, their own options) go through four rows of
    one q3_batcher session — every request is prefilled on the side and swapped in — and each must come back with the
    oracle's codes for it."""
    from make_golden_bench import clone_utts
    ref = np.load(os.path.join(G, "bench_1_7b_codes.npz"))["default_codes"]
    fx = np.load(os.path.join(G, "bench_1_7b_clone.npz"))
    limits = [8, 12, 6, 10, 7, 9]
    utts = []
    for i, L in enumerate(limits):
        u = bench_utt(i); u.max_length = L
        utts.append(u)
    clones = clone_utts(gm17.config)
    for name in ("xvector", "icl"):
        u = clones[name]; u.options = q.SynthesisOptions(max_length=4, temperature=0.0, eos_token_id=None, seed=42)
        utts.insert(3 if name == "xvector" else 5, u)                  # in the middle of the queue, between CustomVoice requests
    b = q.Batcher(gm17, slots=4, frame_budget=32, prompt_budget=int(fx["icl_prefill_len"][0]) + 8,
                  options=q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42))
    res = b.run_all(utts, want_pcm=False, poll_frames=5)
    b.close()
    bad = 0
    for u, (codes, _) in zip(utts, res):
        if u.xvector is not None:
            name = "icl" if u.ref_codes is not None else "xvector"
            want = fx[f"{name}_codes"]
            assert codes.shape == want.shape, (name, codes.shape)
            if not (codes == want).all():
                f = next(i for i in range(len(want)) if not (codes[i] == want[i]).all())
                g = int(np.nonzero(codes[f] != want[f])[0][0])
                m = float(fx[f"{name}_talker_top2_margin"][f]) if g == 0 else float(fx[f"{name}_cp_top2_margin"][f][g - 1])
                assert m < MARGIN_EPS, (name, f, g, m)
                bad += 1
        else:
            i = u.seed - 42; L = u.max_length
            assert codes.shape == (L, 16), (i, codes.shape)
            if not (codes == ref[i][:L]).all():
                ok, rep = _adjudicate("1.7b", u, q.SynthesisOptions(max_length=L, eos_token_id=None, seed=42), codes, f"1_7b_batcher_seq{i}")
                assert ok, rep
                bad += 1
    assert bad <= 1, bad


def test_batcher_stages_a_4k_prompt_beside_running_rows_1_7b(gm17):
    """a swap no longer stalls the session. Eight CustomVoice rows run; the head of the queue — the 4105-position
    VoiceDesign prompt of config[4], ~45 ms of prefill — is prefilled AHEAD on the batcher's worker thread and stream while the
    captured frame keeps replaying (q3_batcher::Stage), enters the row that ends first at a frame boundary, and a CustomVoice request
    follows it the same way. Every request must come back with the oracle's codes: the 4k request its fixture's first 24 default-
    sampling frames, the CustomVoice rows theirs — and the same with staging off (Q3_BAT_NO_STAGE is read once per process, so the
    unstaged run is the batch-1 comparison of test_frames_behind_4k_prompt_1_7b)."""
    from make_golden_bench import prefill4k_utt
    ref = np.load(os.path.join(G, "bench_1_7b_codes.npz"))["default_codes"]
    fx4k = np.load(os.path.join(G, "bench_1_7b_prefill4k_frames.npz"))["free_codes"]
    limits = [32, 30, 8, 32, 31, 28, 32, 26]                # row 2 ends first: the staged 4k request enters there
    utts = []
    for i, L in enumerate(limits):
        u = bench_utt(i); u.max_length = L
        utts.append(u)
    big = prefill4k_utt(); big.max_length = 24
    utts.append(big)
    tail = bench_utt(8); tail.max_length = 12                # staged behind the 4k request while that one runs
    utts.append(tail)
    b = q.Batcher(gm17, slots=8, frame_budget=32, prompt_budget=4105 + 8,
                  options=q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42))
    res = b.run_all(utts, want_pcm=False, poll_frames=4)
    b.close()
    bad = 0
    for u, (codes, _) in zip(utts, res):
        if u.instruct_ids is not None:
            want = fx4k[:24]
            assert codes.shape == want.shape, codes.shape
            if not (codes == want).all():
                ok, rep = _adjudicate("1.7b", u, q.SynthesisOptions(max_length=24, eos_token_id=None, seed=42), codes, "1_7b_batcher_staged_4k")
                assert ok, rep
                bad += 1
        else:
            i = u.seed - 42; L = u.max_length
            assert codes.shape == (L, 16), (i, codes.shape)
            if not (codes == ref[i][:L]).all():
                ok, rep = _adjudicate("1.7b", u, q.SynthesisOptions(max_length=L, eos_token_id=None, seed=42), codes, f"1_7b_batcher_staged_seq{i}")
                assert ok, rep
                bad += 1
    assert bad <= 1, bad


def test_wide_session_is_deterministic(gm17):
    """The wide-session kernels reduce through f32 atomics (two addends per element onto zeros: order-independent) and through
    slice sums added in slice order: two runs of one 64-row session must give the same codes bit for bit, graph or eager."""
    opts = q.SynthesisOptions(max_length=12, eos_token_id=None, seed=42)
    utts = [bench_utt(i) for i in range(64)]
    runs = []
    for use_graph in (True, True, False):
        s = gm17.session(utts, opts); s.prefill(); s.generate(12, use_graph=use_graph)
        runs.append(np.stack([s.codes(b) for b in range(64)])); s.close()
    np.testing.assert_array_equal(runs[0], runs[1])
    np.testing.assert_array_equal(runs[0], runs[2])


def test_teacher_forced_m8(gm17):
    """talker step + code predictor at M = 8 rows, full width: RMS-fused / SwiGLU / residual epilogues, the split attention
    and its merge, compared logit by logit with the oracle's values for 8 DIFFERENT sequences."""
    fx = np.load(os.path.join(G, "bench_1_7b_steps.npz"))
    cfg = gm17.config
    B = 8
    utts = [bench_utt(b, 64) for b in range(B)]
    s = gm17.session(utts, q.SynthesisOptions(max_length=8, seed=42)); s.prefill()
    stats = {}
    for b in range(B):
        hid = s.get(1, (cfg.hidden,), b=b); lg = s.get(2, (cfg.codec_vocab,), b=b)
        stats.setdefault("prefill_hidden", []).append(float(np.abs(hid - fx["prefill_hidden"][b]).max()))
        stats.setdefault("prefill_logits", []).append(float(np.abs(lg - fx["prefill_logits"][b]).max()))
    assert max(stats["prefill_hidden"]) <= 1e-4 and max(stats["prefill_logits"]) <= 1e-3, stats
    h = fx["prefill_hidden"]
    for st in range(fx["sem"].shape[0]):
        codes, cl = s.cp_generate(h, fx["sem"][st])
        for b in range(B):
            same = codes[b] == fx["cp_codes"][st, b]
            first_bad = int(np.argmin(same)) if not same.all() else 15
            if first_bad < 15:          # a wrong greedy code poisons the later groups: only an oracle near-tie may cause it
                assert fx["cp_top2_margin"][st, b, first_bad] < MARGIN_EPS, (st, b, first_bad, float(fx["cp_top2_margin"][st, b, first_bad]))
            for k, g in enumerate((0, 7, 14)):
                if g <= first_bad:
                    e = float(np.abs(cl[b, g] - fx["cp_logits_g0_7_14"][st, b, k]).max())
                    stats.setdefault(f"cp_logits_g{g}", []).append(e)
                    assert e <= 2e-3, (st, b, g, e)
        hid, lg = s.talker_step(fx["emb"][st])
        eh = np.abs(hid - fx["hidden"][st]).max(axis=1); el = np.abs(lg - fx["talker_logits"][st]).max(axis=1)
        stats.setdefault("step_hidden", []).extend(float(x) for x in eh); stats.setdefault("step_logits", []).extend(float(x) for x in el)
        assert eh.max() <= 2e-4 and el.max() <= 2e-3, (st, eh.max(), el.max())
        h = fx["hidden"][st]
    s.close()
    _dump("bench_teacher_forced_m8.json", {k: {"max": max(v), "mean": float(np.mean(v))} for k, v in stats.items()})


@pytest.mark.parametrize("T", [128, 640])
def test_full_size_vocoder_long(gm17, T):
    """The production vocoder kernels at the bench's own length (T = 640: 20 query tiles per attention head, hundreds of
    time tiles per conv, the XCD-grouped tile orders) and at 128 frames: every stage tap at 4096 seeded positions
    (relative to the tap's max magnitude) and the PCM within the north-star 1e-3 RMS."""
    fx = np.load(os.path.join(G, f"bench_vocoder_T{T}.npz"))
    codes = fx["codes"]
    shapes = O.decoder_tap_shapes(gm17.config, T)
    taps = [np.zeros(sh, dtype=np.float32) for sh in shapes]
    pcm = gm17.decode_codes(codes, taps=taps).samples
    names = ["quant", "pre_conv", "pre_transformer", "up0", "up1", "init", "blk0", "blk1", "blk2", "blk3"]
    errs = {}
    for i, (nme, t) in enumerate(zip(names, taps)):
        flat = t.reshape(-1)
        got = flat[tap_indices(flat.size, i, T)]
        errs[nme] = float(np.abs(got - fx[f"tap{i}"]).max() / (float(fx[f"tap{i}_absmax"][0]) + 1e-9))
    assert pcm.shape[0] == T * 1920
    if T <= 128:
        ref = fx["pcm"]; got = pcm
    else:
        ref = fx["pcm_decimated"]; got = pcm[pcm_decimate_idx(pcm.size)]
    rms = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
    unsat = np.abs(ref) < 0.999           # the clamp must not be what makes the PCM agree (synth.py scales the final conv for that)
    rms_unsat = float(np.sqrt(np.mean((got[unsat].astype(np.float64) - ref[unsat]) ** 2))) if unsat.any() else 0.0
    _dump(f"bench_vocoder_T{T}.json", {"tap_rel_err": errs, "pcm_rms_err": rms, "pcm_rms_err_unsaturated": rms_unsat,
                                       "unsaturated_fraction": float(unsat.mean())})
    for nme, e in errs.items():
        assert e <= 2e-4, (nme, e)
    # north-star tolerance: PCM within 1e-3 RMS, over ALL samples, of a waveform whose RMS is speech-like (0.19) and that
    # practically never touches the clamp — so the tolerance is exercised on real values, not on +-1 against +-1
    assert float(unsat.mean()) >= 0.9 and 0.05 <= float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) <= 0.5
    assert rms <= 1e-3 and rms_unsat <= 1e-3, (rms, rms_unsat)


@pytest.mark.parametrize("T", [128, 640])
def test_vocoder_two_plane_mode_1_7b(gm17, T):
    """q3_model_set_codec_planes(2): the vocoder's matrix-core convs use the hi + mid bf16 planes of each f32 operand
    (three products per multiply-accumulate instead of six). Opt-in; the stated bounds: PCM within 3e-4 RMS of the
    reference CPU path (measured 1.0e-4; the path's tolerance is 1e-3), every stage tap within 1e-3 of its max magnitude
    (measured <= 2.9e-4). The mode must really be engaged (the PCM differs from the three-plane bits) and must leave the
    default untouched afterwards (bit-identical decode before and after)."""
    fx = np.load(os.path.join(G, f"bench_vocoder_T{T}.npz"))
    codes = fx["codes"]
    exact = gm17.decode_codes(codes).samples
    shapes = O.decoder_tap_shapes(gm17.config, T)
    taps = [np.zeros(sh, dtype=np.float32) for sh in shapes]
    gm17.set_codec_planes(2)
    try:
        pcm = gm17.decode_codes(codes, taps=taps).samples
    finally:
        gm17.set_codec_planes(3)
    errs = {}
    for i, t in enumerate(taps):
        flat = t.reshape(-1)
        errs[i] = float(np.abs(flat[tap_indices(flat.size, i, T)] - fx[f"tap{i}"]).max() / (float(fx[f"tap{i}_absmax"][0]) + 1e-9))
        assert errs[i] <= 1e-3, (i, errs[i])
    if T <= 128:
        ref = fx["pcm"]; got = pcm
    else:
        ref = fx["pcm_decimated"]; got = pcm[pcm_decimate_idx(pcm.size)]
    rms = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
    _dump(f"bench_vocoder_2planes_T{T}.json", {"tap_rel_err": errs, "pcm_rms_err": rms,
                                               "pcm_rms_vs_three_planes": float(np.sqrt(np.mean((pcm.astype(np.float64) - exact) ** 2)))})
    assert rms <= 3e-4, rms
    assert not np.array_equal(pcm, exact)
    assert np.array_equal(gm17.decode_codes(codes).samples, exact)
    from qwen3_tts_rs_amd import _lib
    with pytest.raises(_lib.Q3Error):
        gm17.set_codec_planes(4)


@pytest.mark.parametrize("planes", [3, 2])
def test_continuous_streaming_full_size_vocoder_is_seamless(gm17, planes):
    """Continuous streaming at production widths, in both vocoder modes: the 10-frame chunks (the short-sequence linear
    kernel in the pre-transformer / ConvNeXt stages) concatenate to the bits of the whole-utterance decode (the tiled
    kernel) — both kernels follow one summation order and one plane count."""
    u = bench_utt(0)
    opts = q.SynthesisOptions(max_length=30, eos_token_id=None, seed=42, chunk_frames=10)
    gm17.set_codec_planes(planes)
    try:
        ss = gm17.synthesize_streaming(u.text_ids, u.speaker, u.language, opts, continuous=True)
        got = np.concatenate([c.samples for c in ss])
        s = gm17.session([q.Utterance(u.text_ids, u.speaker, u.language)], opts); s.prefill(); s.generate(30)
        full = s.decode(0); s.close()
    finally:
        gm17.set_codec_planes(3)
    np.testing.assert_array_equal(got, full)


def test_streaming_chunks_1_7b(gm17):
    """config[2]: 1.7B CustomVoice streaming; the first two 10-frame chunks against the oracle's context-free decodes."""
    fx = np.load(os.path.join(G, "bench_1_7b_stream.npz"))
    ref_codes = np.load(os.path.join(G, "bench_1_7b_codes.npz"))["default_codes"][0]
    u = bench_utt(0)
    opts = q.SynthesisOptions(max_length=20, eos_token_id=None, seed=42, chunk_frames=10)
    ss = gm17.synthesize_streaming(u.text_ids, u.speaker, u.language, opts)
    chunks = list(ss)
    assert [len(c) for c in chunks] == [19200, 19200]
    codes = ss._s.codes(0)
    # the vocoder on the oracle's frames, chunk by chunk (independent of any near-tie in the frame loop)
    for k, name in enumerate(("chunk0", "chunk1")):
        got = gm17.decode_codes(ref_codes[10 * k:10 * k + 10]).samples
        assert float((np.abs(fx[name]) < 0.999).mean()) >= 0.9             # tolerance checked on unsaturated samples
        assert pcm_rms(got, fx[name]) <= 1e-3
    if (codes == ref_codes[:20]).all():
        for k, name in enumerate(("chunk0", "chunk1")):
            assert pcm_rms(chunks[k].samples, fx[name]) <= 1e-3
    else:
        ok, rep = _adjudicate("1.7b", u, q.SynthesisOptions(max_length=20, eos_token_id=None, seed=42), codes, "1_7b_stream")
        assert ok, rep
    ss._s.close()


def test_prefill_4k_1_7b(gm17):
    """config[4]: VoiceDesign prompt with 4096 instruct tokens (4105 prefill positions) through the GEMM + flash-attention
    prefill at 1.7B width; last hidden state, first logits and four hipGraph-decoded greedy frames against the oracle."""
    fx = np.load(os.path.join(G, "bench_1_7b_prefill4k.npz"))
    utt = q.Utterance(synthetic_prompt(32, 0), language=q.Language.English, instruct_ids=synthetic_prompt(4096, 77), seed=42)
    opts = q.SynthesisOptions(max_length=4, temperature=0.0, eos_token_id=None, seed=42)
    s = gm17.session([utt], opts); s.prefill()
    assert s.prefill_len(0)[0] == 4105
    hid = s.get(1, (gm17.config.hidden,)); lg = s.get(2, (gm17.config.codec_vocab,))
    eh = float(np.abs(hid - fx["hidden"]).max()); el = float(np.abs(lg - fx["logits"]).max())
    _dump("bench_prefill4k.json", {"hidden_max_abs_err": eh, "logits_max_abs_err": el, "hidden_absmax": float(np.abs(fx["hidden"]).max()),
                                   "logits_absmax": float(np.abs(fx["logits"]).max())})
    assert eh <= 5e-4 and el <= 5e-3, (eh, el)
    s.generate(4, use_graph=True)
    codes = s.codes(0)
    if not (codes == fx["greedy_codes"]).all():
        f = next(i for i in range(4) if not (codes[i] == fx["greedy_codes"][i]).all())
        g = int(np.nonzero(codes[f] != fx["greedy_codes"][f])[0][0])
        m = float(fx["talker_top2_margin"][f]) if g == 0 else float(fx["cp_top2_margin"][f][g - 1])
        assert m < MARGIN_EPS, (f, g, m)
    s.close()


def test_frames_behind_4k_prompt_1_7b(gm17):
    """config[4] where it is long: behind the 4105-position VoiceDesign prompt (33 KV pages, the decode
    attention's 64-way key split + merge) 32 DEFAULT-SAMPLING hipGraph frames must be the oracle's — four greedy frames were all
    that was compared before — and two teacher-forced talker steps + code-predictor runs at positions 4105 / 4106 are held to
    the oracle logit by logit."""
    from make_golden_bench import prefill4k_utt
    fx = np.load(os.path.join(G, "bench_1_7b_prefill4k_frames.npz"))
    cfg = gm17.config
    s = gm17.session([prefill4k_utt()], q.SynthesisOptions(max_length=8, seed=42)); s.prefill()
    assert s.prefill_len(0)[0] == 4105
    stats = {}
    hid = s.get(1, (cfg.hidden,)); lg = s.get(2, (cfg.codec_vocab,))
    stats["prefill_hidden"] = [float(np.abs(hid - fx["prefill_hidden"]).max())]; stats["prefill_logits"] = [float(np.abs(lg - fx["prefill_logits"]).max())]
    assert stats["prefill_hidden"][0] <= 5e-4 and stats["prefill_logits"][0] <= 5e-3, stats
    fx1 = {k: (fx[k][None] if k.startswith("prefill_") else fx[k][:, None]) for k in
           ("sem", "emb", "hidden", "talker_logits", "cp_logits_g0_7_14", "cp_codes", "cp_top2_margin", "prefill_hidden")}
    _tf_steps(s, fx1, cfg, 1, stats, "frames4k")
    s.close()
    _dump("bench_frames_behind_4k_prompt.json", {k: {"max": max(v), "mean": float(np.mean(v))} for k, v in stats.items()})
    FR = fx["free_codes"].shape[0]
    assert FR >= 32
    opts = q.SynthesisOptions(max_length=FR, eos_token_id=None, seed=42)
    rep = _check_free_run(gm17, "1.7b", [prefill4k_utt()], opts, fx["free_codes"][None], True, "1_7b_frames4k_graph")
    assert len(rep) <= 1, rep


def test_ragged_first_batch_1_7b(gm17):
    """VERDICT r4 item 6 at full width: ONE eight-row session whose rows are CustomVoice (512-token prompts), VoiceDesign with a
    600-position prompt, an x-vector voice clone and an ICL voice clone — three prefill lengths, three prompt builders
    (lib.rs:718-784, 802-870, 897-1046) — decoded in one hipGraph; every row against the oracle fixture it already has."""
    from make_golden_bench import clone_utts, longctx_utt
    ref = np.load(os.path.join(G, "bench_1_7b_codes.npz"))["default_codes"]
    lc = np.load(os.path.join(G, "bench_1_7b_longctx.npz"))
    fc = np.load(os.path.join(G, "bench_1_7b_clone.npz"))
    greedy4 = q.SynthesisOptions(max_length=4, temperature=0.0, eos_token_id=None, seed=42)
    clones = clone_utts(gm17.config)
    rows = []          # (utterance, expected codes, adjudication options)
    for i, L in ((0, 32), (1, 20)):
        u = bench_utt(i); u.max_length = L
        rows.append((u, ref[i][:L], q.SynthesisOptions(max_length=L, eos_token_id=None, seed=42)))
    for b in (0, 1):
        u = longctx_utt(b); u.max_length = 10
        rows.append((u, lc["free_codes"][b], q.SynthesisOptions(max_length=10, eos_token_id=None, seed=42)))
    for name in ("xvector", "icl"):
        u = clones[name]; u.options = greedy4; u.max_length = 4
        rows.append((u, fc[f"{name}_codes"], greedy4))
    for i, L in ((2, 12), (3, 32)):
        u = bench_utt(i); u.max_length = L
        rows.append((u, ref[i][:L], q.SynthesisOptions(max_length=L, eos_token_id=None, seed=42)))
    order = [0, 2, 4, 1, 5, 3, 6, 7]                       # prompt kinds interleaved
    rows = [rows[k] for k in order]
    host = q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42)
    s = gm17.session([r[0] for r in rows], host)
    assert len({s.prefill_len(b)[0] for b in range(8)}) >= 3           # CustomVoice and the x-vector clone share 10 positions; 600; the ICL block
    s.prefill(); s.generate(N_FRAMES, use_graph=True)
    bad = []
    for b, (u, want, o1) in enumerate(rows):
        codes = s.codes(b)
        assert codes.shape == want.shape, (b, codes.shape, want.shape)
        if not (codes == want).all():
            ok, rep = _adjudicate("1.7b", u, o1, codes, f"1_7b_ragged_row{b}")
            assert ok, rep
            bad.append(rep)
    s.close()
    assert len(bad) <= 1, bad
    _dump("bench_parity_1_7b_ragged.json", {"rows": 8, "prefill_lengths": 3, "near_tie_divergences": bad})


@pytest.mark.parametrize("sampling", ["default", "greedy"])
def test_0_6b_single_utterance(sampling):
    """config[1]: Qwen3-TTS-0.6B, one utterance, non-streaming, hipGraph: 32 frames bit-exact against the oracle fixture."""
    ref = np.load(os.path.join(G, "bench_0_6b_codes.npz"))[f"{sampling}_codes"]
    gm = q.Qwen3TTS.from_synthetic(q.qwen3_tts_0_6b(), seed=synth.DEFAULT_SEED)
    kw = dict(temperature=0.0) if sampling == "greedy" else {}
    opts = q.SynthesisOptions(max_length=N_FRAMES, eos_token_id=None, seed=42, **kw)
    _check_free_run(gm, "0.6b", [bench_utt(0)], opts, ref[:1], True, f"0_6b_b1_{sampling}_graph")
    gm.close()


def _tf_steps(s, fx, cfg, B, stats, tag):
    """teacher-forced code-predictor runs + talker steps of a prefilled B-row session against a steps fixture"""
    h = fx["prefill_hidden"]
    for st in range(fx["sem"].shape[0]):
        codes, cl = s.cp_generate(h, fx["sem"][st])
        for b in range(B):
            same = codes[b] == fx["cp_codes"][st, b]
            first_bad = int(np.argmin(same)) if not same.all() else 15
            if first_bad < 15:          # a wrong greedy code poisons the later groups: only an oracle near-tie may cause it
                assert fx["cp_top2_margin"][st, b, first_bad] < MARGIN_EPS, (tag, st, b, first_bad, float(fx["cp_top2_margin"][st, b, first_bad]))
            for k, g in enumerate((0, 7, 14)):
                if g <= first_bad:
                    e = float(np.abs(cl[b, g] - fx["cp_logits_g0_7_14"][st, b, k]).max())
                    stats.setdefault(f"cp_logits_g{g}", []).append(e)
                    assert e <= 2e-3, (tag, st, b, g, e)
        hid, lg = s.talker_step(fx["emb"][st])
        eh = np.abs(hid - fx["hidden"][st]).max(axis=1); el = np.abs(lg - fx["talker_logits"][st]).max(axis=1)
        stats.setdefault("step_hidden", []).extend(float(x) for x in eh); stats.setdefault("step_logits", []).extend(float(x) for x in el)
        assert eh.max() <= 5e-4 and el.max() <= 5e-3, (tag, st, eh.max(), el.max())
        h = fx["hidden"][st]


def test_long_context_b8(gm17):
    """The bench batch BEHIND LONG CONTEXT: 8 sequences with equal 600-position VoiceDesign prompts — GEMM +
    flash-attention prefill of 8 x 600 rows, then the decode step at positions 600 / 601: `k_attn_fused` with its 8-way key split
    over ~75 keys per split + `k_attn_merge`, the split-K projections and the code predictor at M = 8, teacher-forced against
    the oracle logit by logit; then a 10-frame default-sampling hipGraph free run of the same batch, codes bit-exact."""
    from make_golden_bench import longctx_utt
    fx = np.load(os.path.join(G, "bench_1_7b_longctx.npz"))
    cfg = gm17.config
    B = 8
    utts = [longctx_utt(b) for b in range(B)]
    s = gm17.session(utts, q.SynthesisOptions(max_length=10, seed=42)); s.prefill()
    assert s.prefill_len(0)[0] == 600
    stats = {}
    for b in range(B):
        hid = s.get(1, (cfg.hidden,), b=b); lg = s.get(2, (cfg.codec_vocab,), b=b)
        stats.setdefault("prefill_hidden", []).append(float(np.abs(hid - fx["prefill_hidden"][b]).max()))
        stats.setdefault("prefill_logits", []).append(float(np.abs(lg - fx["prefill_logits"][b]).max()))
    assert max(stats["prefill_hidden"]) <= 5e-4 and max(stats["prefill_logits"]) <= 5e-3, stats
    _tf_steps(s, fx, cfg, B, stats, "longctx")
    s.close()
    _dump("bench_long_context_b8.json", {k: {"max": max(v), "mean": float(np.mean(v))} for k, v in stats.items()})
    opts = q.SynthesisOptions(max_length=10, eos_token_id=None, seed=42)
    rep = _check_free_run(gm17, "1.7b", utts, opts, fx["free_codes"], True, "1_7b_longctx_b8_graph")
    assert len(rep) <= 1, rep


def test_640_frames_b8(gm17):
    """The benchmark's own run length: the 8-utterance bench session carried to all 640 frames under the
    hipGraph, sequences 0 and 5 compared with the oracle's 640-frame runs — the decode attention's key splits at 100 .. 650
    positions, every per-frame counter, the penalty mask after hundreds of tokens. Each 640-frame run holds ~20 code-predictor
    decisions with an oracle top-2 margin below 2e-3 (two of them below 3e-5), so a sequence may leave the fixture — but only
    AT such a decision (margins are in the fixture; a sampled talker token is adjudicated live), never before the first
    decision whose margin is below 1e-4, and every frame before that point must be bit-exact (150+ frames for one sequence).
    The decode attention at 600 keys is compared separately, logit by logit (test_long_context_b8)."""
    x8 = os.path.join(G, "bench_1_7b_long640x8.npz")           # round 5: all eight sequences of the headline batch
    fx = np.load(x8 if os.path.exists(x8) else os.path.join(G, "bench_1_7b_long640.npz"))
    FR = 640
    opts = q.SynthesisOptions(max_length=FR, eos_token_id=None, seed=42)
    utts = [bench_utt(i) for i in range(8)]
    s = gm17.session(utts, opts); s.prefill(); s.generate(FR, use_graph=True)
    report = []; exact = []
    for i in [int(x) for x in fx["seqs"]]:
        codes = s.codes(i); ref = fx[f"codes_{i}"]
        assert codes.shape == ref.shape == (FR, 16)
        if (codes == ref).all():
            exact.append(FR); continue
        f = next(k for k in range(FR) if not (codes[k] == ref[k]).all())
        g = int(np.nonzero(codes[f] != ref[f])[0][0])
        exact.append(f)
        risky = np.argwhere(fx[f"cp_top2_margin_{i}"] < 1e-4)
        first_risky = int(risky[0][0]) if len(risky) else FR
        assert f >= first_risky or g == 0, (i, f, g, first_risky)          # kernels agree to ~1e-5: nothing may flip before a < 1e-4 margin
        if g > 0:
            m = float(fx[f"cp_top2_margin_{i}"][f][g - 1])
            report.append({"seq": i, "frame": f, "group": g, "oracle_margin": m})
            assert m < MARGIN_EPS, report[-1]
        else:
            ok, rep = _adjudicate("1.7b", utts[i], opts, codes, f"1_7b_long640_seq{i}")
            report.append(rep)
            assert ok, rep
    s.close()
    _dump("bench_640_frames_b8.json", {"exact_frames": exact, "near_tie_divergences": report,
                                       "min_cp_margin": {str(int(i)): float(fx[f"cp_top2_margin_{int(i)}"].min()) for i in fx["seqs"]}})
    assert max(exact) >= 150, (exact, report)


@pytest.mark.parametrize("flavour", ["xvector", "icl"])
def test_clone_prefill_1_7b(gm17, flavour):
    """BASELINE config[3] ("1.7B-Base") / config[4] ("ICL voice-clone") prompt flavours at full width: the
    x-vector prefill (speaker embedding row instead of a preset speaker token) and the ICL prefill (reference codes summed
    over 16 codebooks + reference text, talker.rs:646-710; repetition penalty floored at 1.5, lib.rs:913-929): last hidden
    state, first logits and four greedy hipGraph frames against the oracle."""
    from make_golden_bench import clone_utts
    fx = np.load(os.path.join(G, "bench_1_7b_clone.npz"))
    cfg = gm17.config
    utt = clone_utts(cfg)[flavour]
    opts = q.SynthesisOptions(max_length=4, temperature=0.0, eos_token_id=None, seed=42)
    s = gm17.session([utt], opts); s.prefill()
    assert s.prefill_len(0)[0] == int(fx[f"{flavour}_prefill_len"][0])
    hid = s.get(1, (cfg.hidden,)); lg = s.get(2, (cfg.codec_vocab,))
    eh = float(np.abs(hid - fx[f"{flavour}_hidden"]).max()); el = float(np.abs(lg - fx[f"{flavour}_logits"]).max())
    _dump(f"bench_clone_{flavour}.json", {"hidden_max_abs_err": eh, "logits_max_abs_err": el, "prefill_len": int(fx[f"{flavour}_prefill_len"][0])})
    assert eh <= 2e-4 and el <= 2e-3, (eh, el)
    s.generate(4, use_graph=True)
    codes = s.codes(0); ref = fx[f"{flavour}_codes"]
    if not (codes == ref).all():
        f = next(i for i in range(4) if not (codes[i] == ref[i]).all())
        g = int(np.nonzero(codes[f] != ref[f])[0][0])
        m = float(fx[f"{flavour}_talker_top2_margin"][f]) if g == 0 else float(fx[f"{flavour}_cp_top2_margin"][f][g - 1])
        assert m < MARGIN_EPS, (f, g, m)
    s.close()


@pytest.mark.gpu
def test_icl_2k_reference_1_7b(gm17):
    """config[4] thickened: ONE 1.7B ICL voice-clone request whose reference block makes the prefill 2110
    positions (2100 reference frames, 600 reference-text tokens, talker.rs:646-710): the GEMM prefill over an ICL prompt
    (paged KV: 17 pages), last hidden state and first logits, four greedy hipGraph frames, and the prepend-and-cut decode of
    lib.rs:1022-1041 over 2104 frames (PCM of the generated share, RMS <= 1e-3) against the oracle fixture."""
    from make_golden_bench import icl2k_utt
    fx = np.load(os.path.join(G, "bench_1_7b_icl2k.npz"))
    cfg = gm17.config
    utt = icl2k_utt(cfg)
    opts = q.SynthesisOptions(max_length=4, temperature=0.0, eos_token_id=None, seed=42)
    s = gm17.session([utt], opts); s.prefill()
    assert s.prefill_len(0)[0] == int(fx["prefill_len"][0]) == 2110
    hid = s.get(1, (cfg.hidden,)); lg = s.get(2, (cfg.codec_vocab,))
    eh = float(np.abs(hid - fx["hidden"]).max()); el = float(np.abs(lg - fx["logits"]).max())
    assert eh <= 5e-4 and el <= 5e-3, (eh, el)              # the 600-position thresholds of test_long_context_b8
    s.generate(4, use_graph=True)
    codes = s.codes(0); ref = fx["codes"]
    exact = bool((codes == ref).all())
    if not exact:
        f = next(i for i in range(4) if not (codes[i] == ref[i]).all())
        g = int(np.nonzero(codes[f] != ref[f])[0][0])
        m = float(fx["talker_top2_margin"][f]) if g == 0 else float(fx["cp_top2_margin"][f][g - 1])
        assert m < MARGIN_EPS, (f, g, m)
    pcm = s.decode(0)
    s.close()
    assert pcm.shape == fx["pcm"].shape == (4 * 1920,)
    rms = float(np.sqrt(np.mean((pcm.astype(np.float64) - fx["pcm"].astype(np.float64)) ** 2)))
    _dump("bench_icl2k.json", {"hidden_max_abs_err": eh, "logits_max_abs_err": el, "codes_exact": exact, "pcm_rms_err": rms,
                               "pcm_ref_rms": float(np.sqrt(np.mean(fx["pcm"].astype(np.float64) ** 2)))})
    if exact:
        assert rms <= 1e-3, rms


@pytest.mark.gpu
def test_bf16_kv_mode_1_7b(gm17):
    """The opt-in bf16 K/V mode at the benchmark's size (8 rows, 1.7B): three teacher-forced talker steps after the prefill stay within
    bf16 distance of the f32 session (hidden states: 2e-2 of their scale; logits: 5e-2 of theirs), and a 40-frame graph run is
    deterministic. (The mode is never the headline: bench.py reports it as separate other_configs lines.)"""
    from make_golden_bench import bench_utt
    cfg = gm17.config
    utts = [bench_utt(i) for i in range(8)]
    opts = q.SynthesisOptions(max_length=40, eos_token_id=None, seed=42)
    rng = np.random.default_rng(5)
    emb = (0.5 * rng.standard_normal((3, 8, cfg.hidden))).astype(np.float32)
    out = {}
    for mode in (False, True):
        s = gm17.session(utts, opts, kv_bf16=mode); s.prefill()
        out[mode] = [s.talker_step(emb[i]) for i in range(3)]
        s.close()
    eh = max(float(np.abs(a[0] - b[0]).max()) for a, b in zip(out[True], out[False])); sh = max(float(np.abs(b[0]).max()) for b in out[False])
    el = max(float(np.abs(a[1] - b[1]).max()) for a, b in zip(out[True], out[False])); sl = max(float(np.abs(b[1]).max()) for b in out[False])
    _dump("bench_bf16_kv_1_7b.json", {"hidden_max_abs_diff": eh, "hidden_scale": sh, "logits_max_abs_diff": el, "logits_scale": sl})
    assert 0 < eh <= 2e-2 * sh and el <= 5e-2 * sl, (eh, sh, el, sl)
    codes = []
    for _ in range(2):
        s = gm17.session(utts, opts, kv_bf16=True); s.prefill(); s.generate(40, use_graph=True)
        codes.append(np.stack([s.codes(b) for b in range(8)])); s.close()
    np.testing.assert_array_equal(codes[0], codes[1])
