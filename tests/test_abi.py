"""C-ABI surface (no GPU needed): the library loads, exports every symbol include/q3tts.h declares,
host-only helpers work, and compute entry points fail loudly (never silently fall back) without a GPU."""
import ctypes
import os
import re

import numpy as np

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "q3tts.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(q3_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(_lib.lib, n), f"libq3tts.so does not export {n}"
        assert n in _lib.SYMBOLS, f"python binding missing for {n}"
    assert _lib.lib.q3_abi_version() == 1


def test_struct_layouts_match_header_sizes():
    assert ctypes.sizeof(q.config.CConfig) == 36 * 4
    assert ctypes.sizeof(_lib.COptions) == 3 * 8 + 8 + 6 * 4
    assert ctypes.sizeof(_lib.CTiming) == 32


def test_manifest_only_handle_lists_reference_tensor_names():
    cfg = q.qwen3_tts_1_7b()
    h = ctypes.c_void_p(); c = cfg.to_c()
    _lib.check(_lib.lib.q3_model_create(ctypes.byref(c), -1, ctypes.byref(h)))
    items = {n: (cnt, dt) for n, cnt, dt in synth.manifest(h)}
    assert items["talker.model.layers.27.self_attn.q_proj.weight"] == (2048 * 2048, 1)
    assert items["talker.code_predictor.small_to_mtp_projection.weight"] == (1024 * 2048, 1)
    assert items["talker.code_predictor.lm_head.14.weight"] == (2048 * 1024, 1)
    assert items["decoder.decoder.1.block.1.conv.weight"] == (1536 * 768 * 16, 0)
    assert items["decoder.decoder.6.conv.weight"] == (96 * 7, 0)
    lm = sum(cnt for n, (cnt, dt) in items.items() if n.startswith("talker."))
    assert abs(lm / 1e9 - 1.92) < 0.03          # ≈1.92 B parameters (README.md:115-121 of the reference)
    # a manifest-only handle holds no weights and says so
    a = np.zeros(2048, dtype=np.float32)
    st = _lib.lib.q3_model_set_tensor(h, b"talker.model.norm.weight", 0, a.ctypes.data_as(ctypes.c_void_p), 2048)
    assert st == 7 and b"manifest-only" in _lib.lib.q3_last_error()
    _lib.lib.q3_model_free(h)
    cfg6 = q.qwen3_tts_0_6b()
    h = ctypes.c_void_p(); c = cfg6.to_c()
    _lib.check(_lib.lib.q3_model_create(ctypes.byref(c), -1, ctypes.byref(h)))
    names = [n for n, _, _ in synth.manifest(h)]
    assert "talker.code_predictor.small_to_mtp_projection.weight" not in names     # 0.6B has no projection
    _lib.lib.q3_model_free(h)


def test_unsupported_configs_are_rejected():
    cfg = q.tiny(); cfg.head_dim = 64
    h = ctypes.c_void_p(); c = cfg.to_c()
    assert _lib.lib.q3_model_create(ctypes.byref(c), -1, ctypes.byref(h)) == 7
    assert b"head_dim" in _lib.lib.q3_last_error()


def test_synth_fill_is_deterministic_and_normalish():
    a = synth._fill(5, "some.tensor", 0, 1.0, 0.0, 200000)
    b = synth._fill(5, "some.tensor", 0, 1.0, 0.0, 200000)
    c = synth._fill(5, "other.tensor", 0, 1.0, 0.0, 200000)
    assert (a == b).all() and not (a == c).all()
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1.0) < 0.01
    h = synth._fill(5, "some.tensor", 1, 1.0, 0.0, 1000)
    assert (h == synth.f32_to_bf16(a[:1000])).all()       # bf16 output = RNE of the f32 stream


def test_rng_matches_reference_formula():
    st = ctypes.c_uint64(); _lib.lib.q3_rng_seed(42, ctypes.byref(st))
    assert st.value == (42 * 2685821657736338717 + 1442695040888963407) & ((1 << 64) - 1)
    v = [_lib.lib.q3_rng_next(ctypes.byref(st)) for _ in range(4)]
    assert all(0.0 <= x <= 1.0 for x in v) and len(set(v)) == 4


def test_codes_to_tensor_layout():
    frames = np.arange(32, dtype=np.uint32).reshape(2, 16)
    t = q.codes_to_tensor(frames)
    assert t.shape == (1, 16, 2) and t.dtype == np.int64
    assert (t[0, :, 0] == np.arange(16)).all() and (t[0, :, 1] == np.arange(16, 32)).all()


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, not route to a CPU path."""
    if _lib.lib.q3_device_count() > 0:
        return
    cfg = q.tiny()
    try:
        q.Qwen3TTS(cfg, device=0)
    except _lib.Q3Error as e:
        assert e.status == 5 or e.status == 1
    else:
        raise AssertionError("model creation succeeded without a GPU")
    try:
        q.linear(np.zeros((1, 8), np.float32), np.zeros((8, 8), np.uint16))
    except _lib.Q3Error as e:
        assert e.status == 5
    else:
        raise AssertionError("q3_linear succeeded without a GPU")


def test_speaker_language_tables():
    assert q.Speaker.from_str("ryan").token_id() == 3061 and q.Speaker.from_str("uncle_fu") is q.Speaker.UncleFu
    assert q.Language.from_str("en").token_id() == 2050 and q.Language.from_str("Chinese").token_id() == 2055
    assert q.Speaker.Sohee.native_language() is q.Language.Korean
    o = q.SynthesisOptions()
    assert (o.max_length, o.temperature, o.top_k, o.top_p, o.repetition_penalty, o.eos_token_id, o.chunk_frames, o.min_new_tokens) == \
        (2048, 0.9, 50, 0.9, 1.05, 2150, 10, 2)
    assert q.CODEC_EOS_TOKEN_ID == 2150 and q.SAMPLES_PER_FRAME == 1920


def test_integration_doc_covers_every_symbol():
    """INTEGRATION.md maps every exported symbol to the reference interface it replaces (or marks it new)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "q3tts.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(q3_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 70
    assert [n for n in names if n not in doc] == []


def test_null_handles_return_status_not_crash():
    """"Never abort / unwind across the ABI" (Errors): every entry point checks its handles before it
    touches the GPU, so misuse from a host without a device still comes back as a status + message."""
    import ctypes
    from qwen3_tts_rs_amd import _lib
    L = _lib.lib
    null = ctypes.c_void_p(None)
    i = ctypes.c_int(); sz = ctypes.c_size_t(); d = ctypes.c_double(); t64 = ctypes.c_int64()
    calls = [
        lambda: L.q3_model_create(None, 0, ctypes.byref(null)),
        lambda: L.q3_model_set_tensor(None, b"x", 0, None, 0),
        lambda: L.q3_model_finalize(None),
        lambda: L.q3_model_arena(None, None, None),
        lambda: L.q3_session_create(None, None, 1, ctypes.byref(null)),
        lambda: L.q3_session_prefill(None),
        lambda: L.q3_session_generate(None, 1, 1),
        lambda: L.q3_session_frames(None, 0, ctypes.byref(i), ctypes.byref(i)),
        lambda: L.q3_session_codes(None, 0, None, 0, ctypes.byref(i)),
        lambda: L.q3_session_decode(None, 0, 0, 0, None, 0, ctypes.byref(sz)),
        lambda: L.q3_session_run(None, 1, None, None, None, None),
        lambda: L.q3_session_next_chunk(None, None, 0, ctypes.byref(sz), ctypes.byref(i)),
        lambda: L.q3_session_set_stream_mode(None, 1),
        lambda: L.q3_session_set_kv_dtype(None, 1),
        lambda: L.q3_model_set_codec_planes(None, 2),
        lambda: L.q3_session_create_reserved(None, None, 1, 8, 16, ctypes.byref(null)),
        lambda: L.q3_session_replace(None, 0, None),
        lambda: L.q3_session_next_chunk_row(None, 0, None, 0, ctypes.byref(sz), ctypes.byref(i)),
        lambda: L.q3_batcher_create(None, 8, 64, 0, ctypes.byref(null)),
        lambda: L.q3_batcher_submit(None, None, 0, ctypes.byref(t64)),
        lambda: L.q3_batcher_step(None, 8, 1, ctypes.byref(i), ctypes.byref(i), ctypes.byref(i)),
        lambda: L.q3_batcher_poll(None, 1, ctypes.byref(i), ctypes.byref(i), ctypes.byref(sz)),
        lambda: L.q3_batcher_fetch(None, 1, None, 0, None, 0),
        lambda: L.q3_decode_codes(None, None, 0, None, None),
        lambda: L.q3_spk_create(None, 0, ctypes.byref(null)),
        lambda: L.q3_spk_finalize(None),
        lambda: L.q3_spk_encode(None, None, 0, 24000, None),
        lambda: L.q3_spk_load_safetensors(None, b"/nonexistent"),
        lambda: L.q3_dp_init(0, 1, None, 0, ctypes.byref(null)),
        lambda: L.q3_dp_broadcast_weights(None, None, 0),
        lambda: L.q3_config_from_json(b"/nonexistent/config.json", None, None),
        lambda: L.q3_model_load(b"/nonexistent", -1, ctypes.byref(null), ctypes.byref(i)),
        lambda: L.q3_wav_read(b"/nonexistent.wav", None, 0, None, None),
        lambda: L.q3_resample(None, 1, 16000, 24000, None, 0, None),
    ]
    for k, f in enumerate(calls):
        st = f()
        assert st != 0, k
        assert len(L.q3_last_error()) > 0, k
    L.q3_model_free(None); L.q3_session_free(None); L.q3_spk_free(None); L.q3_dp_free(None); L.q3_batcher_free(None)     # free(NULL) is a no-op
