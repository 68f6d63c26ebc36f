"""Paged talker KV (-m gpu): pages of 128 positions from one pool per model (north_star "in-place paged KV"; replaces the
reference's per-call preallocated cache and its overflow bail, kv_cache.rs:234-310 / :293-300).
  * bit-identity with one contiguous extent per row (Q3_KV_CONTIGUOUS=1) across page boundaries, for the decode kernel, the
    chunked prefill and the GEMM prefill;
  * a 4105-position row and seven short rows share a pool far smaller than 8 x the worst case;
  * pool exhaustion is the reference's overflow bail: Q3_KV_OVERFLOW before anything runs, the ticket fails alone;
  * a continuous-batching swap relinks pages (none copied, none leaked)."""
import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import _lib, api
from common import model_pair, synthetic_prompt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    cfg = q.tiny()
    m = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=1234)
    yield m
    m.close()


def _run(model, utts, frames, graph, monkeypatch, contiguous):
    if contiguous:
        monkeypatch.setenv("Q3_KV_CONTIGUOUS", "1")
    else:
        monkeypatch.delenv("Q3_KV_CONTIGUOUS", raising=False)
    s = model.session(utts, q.SynthesisOptions(max_length=frames, eos_token_id=None, seed=7))
    s.prefill(); s.generate(frames, use_graph=graph)
    out = [s.codes(b) for b in range(len(utts))]
    pcm = s.decode(0, 0, min(frames, 8))
    s.close()
    monkeypatch.delenv("Q3_KV_CONTIGUOUS", raising=False)
    return out, pcm


@pytest.mark.parametrize("kind,B,frames", [("custom", 3, 300), ("custom", 1, 140), ("design200", 2, 150), ("design600", 1, 40), ("design1100", 1, 30)])
def test_paged_equals_contiguous_bit_for_bit(gm, kind, B, frames, monkeypatch):
    """Same sessions with paged and with contiguous KV: identical codes. 300 frames cross two page boundaries inside the
    captured frame; a 200-token instruct prompt fills page 0 and 1 in the chunked prefill, 600 tokens go through the GEMM
    prefill (bf16x3 planes built from pages) and end in page 4; 1100 tokens make the row longer than 8 pages (the vector-loaded
    form of the page-table row; shorter sessions use the scalar-loaded one)."""
    def utt(i):
        if kind == "custom":
            return q.Utterance(synthetic_prompt(12, i), q.Speaker.Ryan, q.Language.English, seed=40 + i)
        n = int(kind[6:])
        return q.Utterance(synthetic_prompt(9, i), language=q.Language.German, instruct_ids=synthetic_prompt(n, 50 + i), seed=40 + i)
    utts = [utt(i) for i in range(B)]
    info0 = gm.kv_pool_info()
    assert info0["page_positions"] == 128 and info0["pages_in_use"] == 0
    for graph in (True, False):
        a, pa = _run(gm, utts, frames, graph, monkeypatch, contiguous=False)
        b, pb = _run(gm, utts, frames, graph, monkeypatch, contiguous=True)
        for x, y in zip(a, b):
            assert x.shape == (frames, 16)
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(pa, pb)
    info = gm.kv_pool_info()
    assert info["pages_in_use"] == 0 and info["pages_peak"] >= B          # every page came back


def test_long_and_short_rows_share_a_small_pool():
    """a 4105-position row and seven ten-position rows in one session, out of a pool smaller than one eighth
    of 8 x the worst case. Contiguous extents would reserve 8 x (4200 + 64 + 1) positions = 8 x 34 pages."""
    cfg = q.tiny()
    m = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=1234)
    try:
        F = 64
        worst_pages = 8 * -(-(4200 + F + 1) // 128)
        limit = 48                                             # 7 short rows x 1 + 33 for the long one + the swap's transient + spare
        assert limit * 5 < worst_pages
        m.kv_pool_limit(limit)
        opts = q.SynthesisOptions(max_length=F, eos_token_id=None, seed=3)
        long_u = q.Utterance(synthetic_prompt(9, 0), language=q.Language.German, instruct_ids=synthetic_prompt(4096, 50), seed=11)
        shorts = [q.Utterance(synthetic_prompt(5 + i, i + 1), q.Speaker.Ryan, q.Language.English, seed=20 + i) for i in range(7)]
        b = q.Batcher(m, slots=8, frame_budget=F, prompt_budget=4200, options=opts)
        try:
            tickets = [b.submit(u, want_pcm=False) for u in [long_u] + shorts]
            peak_seen = 0
            for _ in range(100):
                running, queued, _ = b.step(16)
                peak_seen = max(peak_seen, m.kv_pool_info()["pages_in_use"])
                if running == 0 and queued == 0:
                    break
            got = [b.fetch(t)[0] for t in tickets]
        finally:
            b.close()
        assert peak_seen <= limit and m.kv_pool_info()["pages_peak"] <= limit
        assert m.kv_pool_info()["pages_in_use"] == 0
        # every row equals its own batch-1 session (rows of a session are independent), the long one included
        for u, codes in zip([long_u] + shorts, got):
            s1 = m.session([u], opts); s1.prefill(); s1.generate(F)
            assert s1.prefill_len(0)[0] == (4105 if u is long_u else 10)
            np.testing.assert_array_equal(codes, s1.codes(0)); s1.close()
    finally:
        m.close()


def test_pool_exhaustion_is_the_overflow_bail():
    """A session that needs a page the pool may not hold fails with Q3_KV_OVERFLOW before it runs (kv_cache.rs:293-300's
    bail), the session and the pool stay usable, and in the batcher the request that does not fit fails alone."""
    cfg = q.tiny()
    m = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=1234)
    try:
        m.kv_pool_limit(3)
        opts = q.SynthesisOptions(max_length=300, eos_token_id=None, seed=3)
        u = q.Utterance(synthetic_prompt(9, 0), q.Speaker.Ryan, q.Language.English, seed=5)
        s = m.session([u, u], opts); s.prefill()               # 2 pages
        s.generate(100)                                        # position 110 < 128: no new page
        with pytest.raises(_lib.Q3Error, match="KV page pool exhausted") as ei:
            s.generate(200)                                    # both rows cross into page 1 and 2: 4 more pages, 1 free
        assert ei.value.status == 4                            # Q3_KV_OVERFLOW
        assert s.frames(0)[0] == 100                           # nothing ran
        s.close()
        assert m.kv_pool_info()["pages_in_use"] == 0
        s = m.session([u], opts); s.prefill(); s.generate(300)  # one row: 3 pages — fits
        assert s.frames(0) == (300, True); s.close()
        # batcher: the 600-position prompt (5 pages) cannot be placed; the short requests run
        m.kv_pool_limit(4)
        b = q.Batcher(m, slots=2, frame_budget=20, prompt_budget=700, options=q.SynthesisOptions(max_length=20, eos_token_id=None, seed=3))
        big = q.Utterance(synthetic_prompt(9, 0), language=q.Language.German, instruct_ids=synthetic_prompt(600, 50), seed=11)
        t_big = b.submit(big, want_pcm=False); t_ok = b.submit(u, want_pcm=False); t_ok2 = b.submit(u, want_pcm=False)
        for _ in range(20):
            running, queued, _ = b.step(8)
            if running == 0 and queued == 0:
                break
        assert b.poll(t_big)[0] == q.Batcher.FAILED and b.poll(t_ok)[0] == q.Batcher.DONE and b.poll(t_ok2)[0] == q.Batcher.DONE
        with pytest.raises(_lib.Q3Error, match="KV page pool exhausted"):
            b.fetch(t_big)
        assert b.fetch(t_ok)[0].shape == (20, 16)
        b.close()
        assert m.kv_pool_info()["pages_in_use"] == 0
    finally:
        m.close()


def test_replace_relinks_pages(monkeypatch):
    """q3_session_replace hands the side session's prefilled pages to the row: the pool's occupancy after the swap is the
    other rows' pages plus the new prompt's, the swapped row's codes equal its batch-1 run, its neighbour is untouched."""
    cfg = q.tiny()
    m = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=1234)
    try:
        opts = q.SynthesisOptions(max_length=40, eos_token_id=None, seed=3)
        a = q.Utterance(synthetic_prompt(9, 0), q.Speaker.Ryan, q.Language.English, seed=5)
        c = q.Utterance(synthetic_prompt(9, 2), language=q.Language.German, instruct_ids=synthetic_prompt(300, 52), seed=6)      # 309 positions: 3 pages
        s = api.Session(m, [a, a], opts, frame_budget=40, prompt_budget=400)
        s.prefill(); s.generate(10)
        assert m.kv_pool_info()["pages_in_use"] == 2
        s.replace(1, c)
        assert m.kv_pool_info()["pages_in_use"] == 1 + 3
        s.generate(40)
        s1 = m.session([c], opts); s1.prefill(); s1.generate(40)
        np.testing.assert_array_equal(s.codes(1), s1.codes(0)); s1.close()
        s0 = m.session([a], opts); s0.prefill(); s0.generate(40)
        np.testing.assert_array_equal(s.codes(0), s0.codes(0)); s0.close()
        s.close()
        assert m.kv_pool_info()["pages_in_use"] == 0
    finally:
        m.close()


def test_bf16_kv_session_mode(gm):
    """Opt-in bf16 K/V (q3_session_set_kv_dtype; the reference GPU path's cache dtype, kv_cache.rs:234-310): the prompt is prefilled
    in f32 pages and converted once into pages of the bf16 pool, decode steps append / read bf16. Not bit-comparable with the F32
    oracle by construction, so the checks are: teacher-forced hidden states within bf16 distance of the f32 session's (crossing a
    page boundary), deterministic free runs, a swap into a bf16 session equal to the request's own bf16 batch-1 run, and every
    page of both pools returned."""
    cfg = gm.config
    opts = q.SynthesisOptions(max_length=150, eos_token_id=None, seed=7)
    utts = [q.Utterance(synthetic_prompt(12, i), language=q.Language.German, instruct_ids=synthetic_prompt(110, 50 + i), seed=40 + i) for i in range(2)]   # 119 positions: page 0 nearly full
    rng = np.random.default_rng(3)
    emb = (0.5 * rng.standard_normal((20, 2, cfg.hidden))).astype(np.float32)
    hs = {}
    for mode in (False, True):
        s = gm.session(utts, opts, kv_bf16=mode); s.prefill()
        hs[mode] = np.stack([s.talker_step(emb[i])[0] for i in range(20)])      # positions 119 .. 138: across the page boundary
        s.close()
    scale = float(np.abs(hs[False]).max())
    err = float(np.abs(hs[True] - hs[False]).max())
    assert 0 < err <= 4e-2 * scale, (err, scale)             # bf16 K/V: 8 mantissa bits on every cached key and value; not zero (the mode is really on)
    runs = []
    for _ in range(2):
        s = gm.session(utts, opts, kv_bf16=True); s.prefill(); s.generate(150)
        runs.append([s.codes(b) for b in range(2)]); pcm = s.decode(0, 0, 8); s.close()
    for b in range(2):
        assert runs[0][b].shape == (150, 16)
        np.testing.assert_array_equal(runs[0][b], runs[1][b])
    assert np.isfinite(pcm).all()
    # continuous batching in a bf16 session: the side session converts too, its bf16 pages are relinked
    a = q.Utterance(synthetic_prompt(9, 0), q.Speaker.Ryan, q.Language.English, seed=5)
    c = q.Utterance(synthetic_prompt(9, 2), language=q.Language.German, instruct_ids=synthetic_prompt(200, 52), seed=6)
    o40 = q.SynthesisOptions(max_length=40, eos_token_id=None, seed=3)
    s = api.Session(gm, [a, a], o40, frame_budget=40, prompt_budget=300, kv_bf16=True)
    s.prefill(); s.generate(10); s.replace(1, c); s.generate(40)
    s1 = gm.session([c], o40, kv_bf16=True); s1.prefill(); s1.generate(40)
    np.testing.assert_array_equal(s.codes(1), s1.codes(0)); s1.close(); s.close()
    with pytest.raises(_lib.Q3Error, match="before prefill"):
        s = gm.session([a], o40); s.prefill(); _lib.check(_lib.lib.q3_session_set_kv_dtype(s._h, 1))
    s.close()
    assert gm.kv_pool_info()["pages_in_use"] == 0


def test_batcher_under_a_tight_pool_limit_waits_instead_of_wedging():
    """with a page limit, admission used to be optimistic (prompt + 1 positions). Three 300-frame requests through three
    rows of a 4-page pool were all admitted at one page each, met the limit together at position 128, and q3_session_generate
    then returned Q3_KV_OVERFLOW for the whole session on every step — no ticket failed, nothing moved. Now a request enters a
    row only when its worst case (3 pages) fits beside the running rows' (so they run one after the other here), finished rows
    are idled at once and give their pages back, and every ticket completes with the codes of its batch-1 run."""
    cfg = q.tiny()
    m = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=1234)
    try:
        F = 300
        opts = q.SynthesisOptions(max_length=F, eos_token_id=None, seed=3)
        utts = [q.Utterance(synthetic_prompt(5 + i, i + 1), q.Speaker.Ryan, q.Language.English, seed=20 + i) for i in range(4)]
        utts[3].max_length = 40                                  # 1 page: fits beside a running 3-page request
        want = []
        for u in utts:
            s1 = m.session([u], opts); s1.prefill(); s1.generate(F); want.append(s1.codes(0)); s1.close()
        m.kv_pool_limit(3 + 3)                                   # three idle rows (1 page each) + one 3-page request
        b = q.Batcher(m, slots=3, frame_budget=F, prompt_budget=32, options=opts)
        try:
            tickets = [b.submit(u, want_pcm=False) for u in utts]
            peak, most_running = 0, 0
            for _ in range(400):
                running, queued, _ = b.step(32)
                peak = max(peak, m.kv_pool_info()["pages_in_use"]); most_running = max(most_running, running)
                if running == 0 and queued == 0:
                    break
            assert [b.poll(t)[0] for t in tickets] == [q.Batcher.DONE] * 4
            got = [b.fetch(t)[0] for t in tickets]
        finally:
            b.close()
        assert peak <= 6 and m.kv_pool_info()["pages_peak"] <= 6
        assert most_running <= 2                                 # never two 3-page requests at once
        for w, g in zip(want, got):
            np.testing.assert_array_equal(w, g)
        assert m.kv_pool_info()["pages_in_use"] == 0
        # a finished row stops taking pages while the queue is empty: one short request, then many idle steps
        m.kv_pool_limit(0)
        b = q.Batcher(m, slots=2, frame_budget=F, prompt_budget=32, options=opts)
        try:
            u = q.Utterance(synthetic_prompt(5, 1), q.Speaker.Ryan, q.Language.English, seed=20); u.max_length = 200
            u.options = q.SynthesisOptions(max_length=200, seed=3)            # EOS live: the row ends early or at 200
            t = b.submit(u, want_pcm=False)
            for _ in range(20):
                b.step(32)
            assert b.poll(t)[0] == q.Batcher.DONE
            assert m.kv_pool_info()["pages_in_use"] == 2         # two idle rows, one page each — not the finished row's three
        finally:
            b.close()
        assert m.kv_pool_info()["pages_in_use"] == 0
    finally:
        m.close()


def test_pool_limit_covers_bf16_sessions():
    """the page limit and the occupancy figures acted on the f32 pool only; a bf16-KV session took its decode pages from
    a second, unlimited pool. One budget now covers both, in f32-equivalent pages (a bf16 page is half of one)."""
    cfg = q.tiny()
    m = q.Qwen3TTS.from_synthetic(cfg, device=0, seed=1234)
    try:
        m.kv_pool_limit(2)                                       # = 4 bf16 pages
        opts = q.SynthesisOptions(max_length=600, eos_token_id=None, seed=3)
        u = q.Utterance(synthetic_prompt(9, 0), q.Speaker.Ryan, q.Language.English, seed=5)
        s = m.session([u], opts, kv_bf16=True); s.prefill()      # 1 f32 page -> 1 bf16 page
        info = m.kv_pool_info()
        assert info["pages_in_use"] == 1 and info["pages_peak"] == 2, info      # half a page, rounded up; f32 + bf16 page side by side at the conversion
        s.generate(400)                                          # positions .. 410: 4 bf16 pages = the limit
        assert m.kv_pool_info()["pages_in_use"] == 2
        with pytest.raises(_lib.Q3Error, match="KV page pool exhausted") as ei:
            s.generate(200)                                      # a fifth bf16 page
        assert ei.value.status == 4 and s.frames(0)[0] == 400
        s.close()
        assert m.kv_pool_info()["pages_in_use"] == 0
        # two f32 pages are the limit for an f32 session too, and a bf16 prompt that needs 2 f32 pages + their 2 bf16 copies does not fit
        long_u = q.Utterance(synthetic_prompt(9, 0), language=q.Language.German, instruct_ids=synthetic_prompt(200, 50), seed=11)   # 209 positions: 2 pages
        s = m.session([long_u], opts, kv_bf16=True)
        with pytest.raises(_lib.Q3Error, match="exhausted"):
            s.prefill()
        s.close()
        assert m.kv_pool_info()["pages_in_use"] == 0
        freed = m.kv_pool_trim()
        assert freed > 0 and m.kv_pool_info()["pages_total"] == 0           # nothing held: every slab goes back
        s = m.session([u], q.SynthesisOptions(max_length=8, eos_token_id=None, seed=3)); s.prefill(); s.generate(8); s.close()      # and the pool grows again on demand
        assert m.kv_pool_info()["pages_total"] > 0
    finally:
        m.close()
