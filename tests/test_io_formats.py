"""On-disk formats either side of the hot path: config.json, safetensors, WAV, code/audio
dumps. CPU-only: the C++ loader's parsing is checked against Python's json / safetensors / wave / numpy; the GPU
end-to-end load is in test_gpu_parity.py."""
import ctypes
import json
import os
import struct
import wave

import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import _lib, api, synth
from qwen3_tts_rs_amd.config import CConfig, Q3Config
from common import write_checkpoint_dir


def same(a: Q3Config, b: Q3Config) -> bool:
    """equal as the C struct sees them (f32 eps/theta fields)"""
    return bytes(a.to_c()) == bytes(b.to_c())


def test_config_defaults_match_python_presets():
    for variant, preset in ((0, q.qwen3_tts_0_6b()), (1, q.qwen3_tts_1_7b())):
        c = CConfig()
        _lib.check(_lib.lib.q3_config_default(variant, ctypes.byref(c)))
        got = Q3Config.from_c(c, name=preset.name)
        assert same(got, preset)
    with pytest.raises(_lib.Q3Error):
        _lib.check(_lib.lib.q3_config_default(2, ctypes.byref(CConfig())))


def test_config_from_json_1_7b(tmp_path):
    write = {
        "tts_model_type": "voice_design", "tts_model_size": "1b7",
        "talker_config": {"hidden_size": 2048, "intermediate_size": 6144, "rope_theta": 1000000,
                          "rms_norm_eps": 1e-06, "spk_id": {'a"b\\é\n': [1, 2.5e3, None, True, "\U0001F600"]},
                          "code_predictor_config": {"hidden_size": 1024, "num_hidden_layers": 5}},
    }
    p = tmp_path / "config.json"
    p.write_text(json.dumps(write))
    cfg, mt = Q3Config.from_json(p)
    assert mt == 2
    assert same(cfg, q.qwen3_tts_1_7b())


def test_config_from_json_defaults_and_type_fallbacks(tmp_path):
    # every key missing → the reference's unwrap_or defaults (config.rs:256-297) = the 0.6B shapes, type "base"
    p = tmp_path / "config.json"
    p.write_text("{}")
    cfg, mt = Q3Config.from_json(p)
    assert mt == 0 and same(cfg, q.qwen3_tts_0_6b())
    # wrong-typed values behave like serde's as_u64() = None → default; unknown type string → Base
    p.write_text(json.dumps({"tts_model_type": "something", "talker_config": {"hidden_size": "2048", "num_hidden_layers": -3,
                                                                             "intermediate_size": 12.5, "vocab_size": 3000}}))
    cfg, mt = Q3Config.from_json(p)
    assert mt == 0 and cfg.hidden == 1024 and cfg.n_layers == 28 and cfg.inter == 3072 and cfg.codec_vocab == 3000
    p.write_text(json.dumps({"tts_model_type": "custom_voice"}))
    assert Q3Config.from_json(p)[1] == 1


def test_config_from_json_errors(tmp_path):
    with pytest.raises(_lib.Q3Error, match="Failed to read config"):
        Q3Config.from_json(tmp_path / "nope.json")
    p = tmp_path / "bad.json"
    p.write_text('{"talker_config": {"hidden_size": 1024,}')
    with pytest.raises(_lib.Q3Error, match="Failed to parse config"):
        Q3Config.from_json(p)
    p.write_text(json.dumps({"talker_config": {"rope_theta": 1e6, "code_predictor_config": {"rope_theta": 1e4}}}))
    with pytest.raises(_lib.Q3Error, match="differ"):
        Q3Config.from_json(p)


def test_pcm16_conversion_matches_reference_formula():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1.5, 1.5, 4096).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 0.99999, -0.99999, 3.05e-5, -3.05e-5, 2.0, -2.0, np.nan, np.inf, -np.inf],
                                 dtype=np.float32)])
    got = api.pcm16(x)
    with np.errstate(invalid="ignore"):
        want = np.trunc(np.nan_to_num(np.clip(x, -1.0, 1.0), nan=0.0).astype(np.float32) * np.float32(32767.0)).astype(np.int16)
    np.testing.assert_array_equal(got, want)
    assert got[-3] == 0 and got[-2] == 32767 and got[-1] == -32767      # NaN → 0 (Rust `as`), ±inf clamp


def test_wav_write_is_readable_by_python_wave(tmp_path):
    rng = np.random.default_rng(5)
    x = rng.uniform(-1.2, 1.2, 24000 + 17).astype(np.float32)
    p = str(tmp_path / "out.wav")
    api.AudioBuffer(x, 24000).save(p)
    with wave.open(p, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 24000, x.size)
        pcm = np.frombuffer(w.readframes(x.size), dtype="<i2")
    np.testing.assert_array_equal(pcm, api.pcm16(x))
    assert os.path.getsize(p) == 44 + 2 * x.size
    # round trip through our reader: i16 / 32768
    back = api.AudioBuffer.load(p)
    assert back.sample_rate == 24000
    np.testing.assert_array_equal(back.samples, pcm.astype(np.float32) / np.float32(32768.0))
    api.save_wav(str(tmp_path / "empty.wav"), np.zeros(0, np.float32))
    assert len(api.load_wav(str(tmp_path / "empty.wav"))) == 0


def _write_wav_raw(path, fmt_tag, channels, rate, bits, payload, extra_chunk=True):
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:
        body += b"LIST" + struct.pack("<I", 3) + b"abc" + b"\x00"       # odd-sized chunk + pad byte
    body += b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_read_formats(tmp_path):
    rng = np.random.default_rng(7)
    # stereo PCM16 → averaged mono
    st = rng.integers(-32768, 32767, size=(1000, 2)).astype("<i2")
    p = str(tmp_path / "st16.wav")
    _write_wav_raw(p, 1, 2, 16000, 16, st.tobytes())
    a = api.load_wav(p)
    assert a.sample_rate == 16000
    want = (st.astype(np.float32) / np.float32(32768.0))
    np.testing.assert_array_equal(a.samples, (want[:, 0] + want[:, 1]) / np.float32(2))
    # float32 mono
    fl = rng.uniform(-1, 1, 777).astype("<f4")
    p = str(tmp_path / "f32.wav")
    _write_wav_raw(p, 3, 1, 24000, 32, fl.tobytes(), extra_chunk=False)
    np.testing.assert_array_equal(api.load_wav(p).samples, fl)
    # 24-bit PCM
    v = rng.integers(-(1 << 23), (1 << 23) - 1, size=500)
    b = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    p = str(tmp_path / "p24.wav")
    _write_wav_raw(p, 1, 1, 48000, 24, b)
    np.testing.assert_array_equal(api.load_wav(p).samples, v.astype(np.float32) / np.float32(1 << 23))
    # 8-bit unsigned
    u = rng.integers(0, 255, size=300).astype(np.uint8)
    p = str(tmp_path / "p8.wav")
    _write_wav_raw(p, 1, 1, 8000, 8, u.tobytes())
    np.testing.assert_array_equal(api.load_wav(p).samples, (u.astype(np.float32) - 128) / np.float32(128))
    # errors
    (tmp_path / "junk.wav").write_bytes(b"not a wave file at all")
    with pytest.raises(_lib.Q3Error, match="RIFF"):
        api.load_wav(str(tmp_path / "junk.wav"))
    with pytest.raises(_lib.Q3Error, match="Failed to open WAV"):
        api.load_wav(str(tmp_path / "missing.wav"))
    _write_wav_raw(str(tmp_path / "alaw.wav"), 6, 1, 8000, 8, b"\x00" * 16)
    with pytest.raises(_lib.Q3Error, match="unsupported WAV"):
        api.load_wav(str(tmp_path / "alaw.wav"))


def test_code_and_audio_dumps(tmp_path):
    rng = np.random.default_rng(11)
    codes = rng.integers(0, 3072, size=(37, 16)).astype(np.uint32)
    p = str(tmp_path / "codes_seed42_frames37.bin")
    api.save_codes_binary(p, codes)
    raw = np.fromfile(p, dtype="<i8")       # the Python side of the reference's comparison reads int64
    np.testing.assert_array_equal(raw.reshape(37, 16), codes.astype(np.int64))
    np.testing.assert_array_equal(api.load_codes_binary(p), codes)
    api.save_codes_binary(str(tmp_path / "e.bin"), np.zeros((0, 16), np.uint32))
    assert api.load_codes_binary(str(tmp_path / "e.bin")).shape == (0, 16)
    (tmp_path / "ragged.bin").write_bytes(b"\x00" * 100)
    with pytest.raises(_lib.Q3Error, match="whole number"):
        api.load_codes_binary(str(tmp_path / "ragged.bin"))
    (tmp_path / "big.bin").write_bytes(np.full(16, -1, dtype="<i8").tobytes())
    with pytest.raises(_lib.Q3Error, match="does not fit"):
        api.load_codes_binary(str(tmp_path / "big.bin"))
    x = rng.standard_normal(1921).astype(np.float32)
    p = str(tmp_path / "audio.bin")
    api.save_audio_binary(p, x)
    np.testing.assert_array_equal(np.fromfile(p, dtype="<f4"), x)


def _info(path, name):
    dt = ctypes.c_int(); nd = ctypes.c_int(); shape = (ctypes.c_int64 * 8)()
    _lib.check(_lib.lib.q3_safetensors_info(str(path).encode(), name.encode(), ctypes.byref(dt), shape, 8, ctypes.byref(nd)))
    return dt.value, list(shape[:nd.value])


def test_safetensors_header_and_manifest_only_load(tmp_path):
    cfg = q.tiny()
    root = tmp_path / "ckpt"
    write_checkpoint_dir(cfg, str(root))
    assert _info(root / "model.safetensors", "talker.model.norm.weight") == (synth.F32, [cfg.hidden]) or \
        _info(root / "model.safetensors", "talker.model.norm.weight")[1] == [cfg.hidden]
    dt, shape = _info(root / "model.safetensors", "talker.codec_head.weight")
    assert dt == synth.BF16 and shape == [cfg.codec_vocab, cfg.hidden]
    assert _info(root / "speech_tokenizer" / "model.safetensors", "encoder.downsample.conv.weight") == (-1, [2, 2])
    with pytest.raises(_lib.Q3Error, match="Missing weight: nope"):
        _info(root / "model.safetensors", "nope")
    # device -1: config.json is parsed and the manifest built, no GPU touched
    h = ctypes.c_void_p(); mt = ctypes.c_int(-7)
    _lib.check(_lib.lib.q3_model_load(str(root).encode(), -1, ctypes.byref(h), ctypes.byref(mt)))
    c = CConfig(); _lib.check(_lib.lib.q3_model_config(h, ctypes.byref(c)))
    got = Q3Config.from_c(c, name=cfg.name)
    # decoder shapes are not in config.json (Decoder12HzConfig::default in the reference too)
    for f in ("hidden", "inter", "n_layers", "n_heads", "n_kv_heads", "cp_hidden", "cp_inter", "cp_layers", "text_dim"):
        assert getattr(got, f) == getattr(cfg, f), f
    assert got.dec_latent == 1024 and mt.value == 1
    _lib.lib.q3_model_free(h)


def test_model_load_errors_and_weight_inspection_fallback(tmp_path):
    import torch
    from safetensors.torch import save_file
    root = tmp_path / "m"
    root.mkdir()
    h = ctypes.c_void_p()
    with pytest.raises(_lib.Q3Error, match="Model weights not found at .*model.safetensors. Please download the model first."):
        _lib.check(_lib.lib.q3_model_load(str(root).encode(), -1, ctypes.byref(h), None))
    save_file({"talker.model.norm.weight": torch.ones(2048)}, str(root / "model.safetensors"))
    with pytest.raises(_lib.Q3Error, match="Speech tokenizer weights not found"):
        _lib.check(_lib.lib.q3_model_load(str(root).encode(), -1, ctypes.byref(h), None))
    # speech tokenizer in the PARENT directory (lib.rs:239-247); no config.json → hidden 2048 → 1.7B shapes
    (tmp_path / "speech_tokenizer").mkdir()
    save_file({"decoder.x": torch.ones(1)}, str(tmp_path / "speech_tokenizer" / "model.safetensors"))
    mt = ctypes.c_int(5)
    _lib.check(_lib.lib.q3_model_load((str(root) + "/").encode(), -1, ctypes.byref(h), ctypes.byref(mt)))
    c = CConfig(); _lib.check(_lib.lib.q3_model_config(h, ctypes.byref(c)))
    assert same(Q3Config.from_c(c), q.qwen3_tts_1_7b()) and mt.value == -1
    _lib.lib.q3_model_free(h)
    # an unparsable config.json falls back to weight inspection as well (lib.rs:206-214)
    (root / "config.json").write_text("{ not json")
    save_file({"talker.model.norm.weight": torch.ones(1024)}, str(root / "model.safetensors"))
    _lib.check(_lib.lib.q3_model_load(str(root).encode(), -1, ctypes.byref(h), ctypes.byref(mt)))
    _lib.check(_lib.lib.q3_model_config(h, ctypes.byref(c)))
    assert same(Q3Config.from_c(c), q.qwen3_tts_0_6b()) and mt.value == -1
    _lib.lib.q3_model_free(h)
    os.remove(root / "config.json")
    save_file({"something.else": torch.ones(3)}, str(root / "model.safetensors"))
    with pytest.raises(_lib.Q3Error, match="Missing talker.model.norm.weight"):
        _lib.check(_lib.lib.q3_model_load(str(root).encode(), -1, ctypes.byref(h), None))
    # corrupt safetensors
    (root / "model.safetensors").write_bytes(struct.pack("<Q", 1 << 40) + b"{}")
    with pytest.raises(_lib.Q3Error):
        _lib.check(_lib.lib.q3_model_load(str(root).encode(), -1, ctypes.byref(h), None))


# ---------------- resampler (audio/resample.rs tests :186-283, on q3_resample) ----------------
def test_resample_reference_properties():
    from qwen3_tts_rs_amd import api
    a = api.AudioBuffer(np.zeros(1000, np.float32), 24000)
    r = api.resample(a, 24000)
    assert r.sample_rate == 24000 and len(r) == len(a)                       # test_no_resample_needed
    r = api.resample(api.AudioBuffer(np.zeros(4800, np.float32), 48000), 24000)
    assert r.sample_rate == 24000 and 2000 < len(r) < 3000                   # test_downsample
    r = api.resample_to_24k(api.AudioBuffer(np.zeros(1600, np.float32), 16000))
    assert r.sample_rate == 24000 and 2000 < len(r) < 4000                   # test_upsample / test_resample_to_24k
    sine = np.sin(2 * np.pi * 100.0 * np.arange(4800) / 48000.0).astype(np.float32)
    r = api.resample(api.AudioBuffer(sine, 48000), 24000)
    assert np.abs(r.samples).max() > 0.5                                     # test_resample_preserves_sine_wave


@pytest.mark.parametrize("sr_in,sr_out", [(16000, 24000), (48000, 24000), (44100, 24000), (22050, 24000)])
def test_resample_accuracy(sr_in, sr_out):
    """Band-limited signal resampled = the same analytic signal sampled at the new rate (interior; the ends see the
    truncated filter), and components above the new Nyquist are removed when downsampling."""
    from qwen3_tts_rs_amd import api
    n = sr_in // 2
    t = np.arange(n) / sr_in
    x = (0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 3000.0 * t + 0.7)).astype(np.float32)
    hi = 0.4 * np.sin(2 * np.pi * 15000.0 * t) if sr_in >= 44100 else 0.0     # above 12 kHz: must vanish at 24 kHz
    r = api.resample(api.AudioBuffer((x + hi).astype(np.float32), sr_in), sr_out)
    assert len(r) == round(n * sr_out / sr_in)
    to = np.arange(len(r)) / sr_out
    ref = 0.5 * np.sin(2 * np.pi * 440.0 * to) + 0.3 * np.sin(2 * np.pi * 3000.0 * to + 0.7)
    m = slice(200, len(r) - 200)
    assert np.abs(r.samples[m] - ref[m]).max() < 2e-3
