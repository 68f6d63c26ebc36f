"""How the captured frame is submitted (-m gpu). Since round 6 the default is the library's own AQL queue with the
boundaries between the frame's write-through kernels free of HIP's agent-scope fences; hipGraphLaunch (Q3_AQL=0) and the own queue
with HIP's fences on every packet (Q3_AQL=1) replay the very same kernels. Held here: which path a session takes, how many packets
go out without fences (and that a frame's first packet still acquires, its last still releases), that every path gives the same
codes as eager launches, that the frame captured under the prefill (q3_session_run / streaming) is the frame captured late, and
that a model is destroyed by whoever drops its last reference."""
import os

import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from common import synthetic_prompt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_gm():
    gm = q.Qwen3TTS.from_synthetic(q.tiny(), seed=11)
    yield gm
    gm.close()


def _codes(gm, utts, frames, use_graph, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        s = gm.session(utts, q.SynthesisOptions(max_length=frames, seed=42, eos_token_id=None))
        s.prefill(); s.generate(frames, use_graph=use_graph)
        out = np.stack([s.codes(b) for b in range(len(utts))]), s.submit_info(), s.submit_fences()
        s.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("B", [1, 3, 8])
def test_paths_and_fence_policy(tiny_gm, B):
    utts = [q.Utterance(synthetic_prompt(12, i), seed=42 + i) for i in range(B)]
    eager, info_e, fences_e = _codes(tiny_gm, utts, 24, False)
    assert info_e == (0, 0) and fences_e == (0, 0)
    default, info, fences = _codes(tiny_gm, utts, 24, True)
    assert info[0] == 4 and info[1] > 50, info                       # own queue, fence-free boundaries: the product path
    acq_free, rel_free = fences
    # the GEMV / attention families are nearly all of the frame; the first packet keeps its acquire, the last its release, and the
    # glue kernels (sampler, frame embed, final norm, gather) keep both
    assert 0.85 * info[1] <= acq_free <= info[1] - 1, (info, fences)
    assert 0.85 * info[1] <= rel_free <= info[1] - 1, (info, fences)
    graph, info_g, fences_g = _codes(tiny_gm, utts, 24, True, {"Q3_AQL": "0"})
    assert info_g == (1, 0) and fences_g == (0, 0)
    fenced, info_f, fences_f = _codes(tiny_gm, utts, 24, True, {"Q3_AQL": "1"})
    assert info_f[0] == 2 and info_f[1] == info[1] and fences_f == (0, 0)
    np.testing.assert_array_equal(default, eager)
    np.testing.assert_array_equal(graph, eager)
    np.testing.assert_array_equal(fenced, eager)


def test_unsafe_probe_needs_its_opt_in(tiny_gm):
    """Q3_AQL=2 (no fence at all: wrong results) must not be reachable through a stray environment variable."""
    utts = [q.Utterance(synthetic_prompt(12, 0), seed=42)]
    os.environ.pop("Q3_AQL_UNSAFE", None)
    _, info, _ = _codes(tiny_gm, utts, 4, True, {"Q3_AQL": "2"})
    assert info == (1, 0), info                                       # stayed on hipGraphLaunch


def test_frame_captured_under_the_prefill_is_the_same_frame(tiny_gm):
    """q3_session_run captures and converts the frame while the prompt's kernels run (time to first audio); the codes and the PCM
    must be those of a session that prefills, then captures at its first generate call."""
    utts = [q.Utterance(synthetic_prompt(12, i), seed=42 + i) for i in range(4)]
    opts = q.SynthesisOptions(max_length=16, seed=42, eos_token_id=None)
    late, info_late, _ = _codes(tiny_gm, utts, 16, True)
    s = tiny_gm.session(utts, opts)
    s.run_timing_only(use_graph=True)
    early = np.stack([s.codes(b) for b in range(4)]); info_early = s.submit_info()
    s.close()
    assert info_early == info_late and info_early[0] == 4
    np.testing.assert_array_equal(early, late)


def test_streaming_read_ahead_on_the_own_queue(tiny_gm):
    """Chunks of a streaming session (frames of the next chunk submitted on the own queue while the current one is vocoded) carry
    the codes of the non-streaming run."""
    u = q.Utterance(synthetic_prompt(12, 0), seed=42)
    opts = q.SynthesisOptions(max_length=25, seed=42, eos_token_id=None, chunk_frames=10)
    ref, _, _ = _codes(tiny_gm, [u], 25, True)
    ss = tiny_gm.synthesize_streaming(u.text_ids, q.Speaker.Ryan, q.Language.English, opts)
    n = 0
    while True:
        c = ss.next_chunk()
        if c is None:
            break
        n += 1
    assert n == 3 and ss._s.submit_info()[0] == 4
    np.testing.assert_array_equal(ss._s.codes(0), ref[0])
    ss._s.close()


def test_model_outlives_its_handle_while_sessions_run():
    """q3_model_free is ONE reference among the sessions' (ADVICE r5): freeing the handle first must leave a running session intact,
    and the last session's close destroys the model."""
    gm = q.Qwen3TTS.from_synthetic(q.tiny(), seed=11)
    u = [q.Utterance(synthetic_prompt(12, 0), seed=42)]
    opts = q.SynthesisOptions(max_length=6, seed=42, eos_token_id=None)
    s0 = gm.session(u, opts); s0.prefill(); s0.generate(6, use_graph=True); want = s0.codes(0).copy(); s0.close()
    s = gm.session(u, opts); s.prefill()
    gm.close()                                   # the handle's reference goes first
    s.generate(6, use_graph=True)
    got = s.codes(0).copy()
    s.close()                                    # ... the session's last: destroys the model
    np.testing.assert_array_equal(got, want)


def test_two_sessions_on_two_threads_share_the_queue(tiny_gm):
    """One user-mode queue per process and device carries every session's frames; bursts of sessions driven from different host
    threads interleave at frame boundaries (the lock is per frame). Each session must still produce its own codes."""
    import threading
    utts = [[q.Utterance(synthetic_prompt(12, 10 * k + i), seed=100 * k + i) for i in range(3)] for k in range(2)]
    want = [_codes(tiny_gm, u, 40, False)[0] for u in utts]
    got = [None, None]; err = []

    def run(k):
        try:
            s = tiny_gm.session(utts[k], q.SynthesisOptions(max_length=40, seed=42, eos_token_id=None))
            s.prefill()
            for _ in range(8):
                s.generate(5, use_graph=True)
            got[k] = np.stack([s.codes(b) for b in range(3)]); assert s.submit_info()[0] == 4
            s.close()
        except Exception as e:      # pragma: no cover
            err.append(f"thread {k}: {e}")
    ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not err, "\n".join(err)
    for k in range(2):
        np.testing.assert_array_equal(got[k], want[k])


def test_concurrent_synthesize_calls(tiny_gm):
    """What a server built on the reference API does (one synthesize call per request, lib.rs:718-784, from several threads): four host
    threads each open, prefill, generate (every session captures its own frame), vocode and close sessions in a loop on ONE model.
    Captures, the shared packet queue, the device-block cache, the KV page pool and the vocoder's per-thread state all meet here;
    every call must return the codes and the samples of its serial run."""
    import threading
    reqs = [[q.Utterance(synthetic_prompt(8 + (3 * t + j) % 7, 100 * t + j), seed=1000 * t + j) for j in range(5)] for t in range(4)]
    opts = q.SynthesisOptions(max_length=14, seed=42, eos_token_id=None)

    def one(u):
        s = tiny_gm.session([u], opts)
        s.run_timing_only(use_graph=True)
        out = (s.codes(0).copy(), s.decode(0).copy())
        s.close()
        return out
    want = [[one(u) for u in row] for row in reqs]
    got = [[None] * 5 for _ in range(4)]; err = []

    def run(t):
        try:
            for j, u in enumerate(reqs[t]):
                got[t][j] = one(u)
        except Exception as e:      # pragma: no cover
            err.append(f"thread {t}: {e}")
    ts = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in ts: th.start()
    for th in ts: th.join()
    assert not err, "\n".join(err)
    for t in range(4):
        for j in range(5):
            np.testing.assert_array_equal(got[t][j][0], want[t][j][0])
            np.testing.assert_array_equal(got[t][j][1], want[t][j][1])
    info = tiny_gm.kv_pool_info()
    assert info["pages_in_use"] == 0, info
