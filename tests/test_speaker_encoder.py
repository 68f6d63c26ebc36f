"""Speaker-embedding path: mel front end + ECAPA-TDNN.

CPU part: the oracle (oracle/q3_oracle_spk.c) against every fact the reference's own unit tests hold for this path
(speaker.rs:402-470, mel.rs tests) and against an independent numpy restatement. GPU part: the HIP path against the
oracle — tolerance 2e-4 relative on activations / embedding (f32 arithmetic, different summation order), 2e-3 absolute
on log-mel (f32 log of sums that differ in the last bits)."""
import numpy as np
import pytest

import qwen3_tts_rs_amd as q
import oracle as O


def _audio(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 24000.0
    x = 0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.2 * np.sin(2 * np.pi * 1330.0 * t + 1.0) + 0.05 * rng.standard_normal(n)
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 3.0 * t)
    return (x * env).astype(np.float32)


# ---------------- reference unit-test facts, on the oracle ----------------
def test_reflect_pad_kats():      # speaker.rs:412-452
    x = np.arange(5, dtype=np.float32).reshape(1, 5)
    assert O.reflect_pad_1d(x, 0, 0).tolist() == [[0, 1, 2, 3, 4]]
    assert O.reflect_pad_1d(x, 2, 0).tolist() == [[2, 1, 0, 1, 2, 3, 4]]
    assert O.reflect_pad_1d(x, 0, 2).tolist() == [[0, 1, 2, 3, 4, 3, 2]]
    assert O.reflect_pad_1d(x, 2, 2).tolist() == [[2, 1, 0, 1, 2, 3, 4, 3, 2]]


def test_hann_window_kat():       # mel.rs test_hann_window: len 4 → w[0] = 0, w[2] = 1
    w = O.hann_window(4)
    assert abs(w[0]) < 1e-6 and abs(w[2] - 1.0) < 1e-6
    w = O.hann_window(1024)
    ref = 0.5 * (1 - np.cos(2 * np.pi * np.arange(1024) / 1024))
    assert np.abs(w - ref).max() < 1e-6


def test_mel_filterbank_properties():   # mel.rs:271-318: shape, non-negative, every band non-empty, Slaney area norm
    fb = O.mel_filterbank(24000, 1024, 128, 0.0, 12000.0)
    assert fb.shape == (128, 513) and (fb >= 0).all() and (fb.sum(1) > 0).all()
    # independent numpy restatement (librosa's slaney formulas, f64)
    def hz2mel(f):
        f = np.asarray(f, np.float64); return np.where(f < 1000, f / (200 / 3), 15 + np.log(np.maximum(f, 1e-9) / 1000) / (np.log(6.4) / 27))
    def mel2hz(m):
        m = np.asarray(m, np.float64); return np.where(m < 15, m * (200 / 3), 1000 * np.exp((m - 15) * (np.log(6.4) / 27)))
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(12000.0), 130))
    freqs = np.arange(513) * 24000 / 1024
    ref = np.zeros((128, 513))
    for i in range(128):
        lo, ce, up = pts[i], pts[i + 1], pts[i + 2]
        ref[i] = np.maximum(0, np.minimum((freqs - lo) / (ce - lo), (up - freqs) / (up - ce))) * 2 / (up - lo)
    assert np.abs(fb - ref).max() <= 2e-3 * ref.max()       # f32 band edges move a triangle's slope slightly


def test_mel_frame_count_and_values():
    for n in (24000, 24000 * 3 + 17, 1025, 300):
        T = O.olib.q3o_mel_frames(n, 1024, 256)
        assert T == (n + 768 - 1024) // 256 + 1          # mel.rs:197
    x = _audio(12000, 1)
    mel = O.mel_speaker(x)
    # numpy restatement: reflect pad 384, frames of 1024 hop 256, rfft magnitude, filterbank, log clamp
    xp = np.pad(x.astype(np.float64), 384, mode="reflect")
    T = (len(xp) - 1024) // 256 + 1
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(1024) / 1024))
    fr = np.stack([xp[i * 256:i * 256 + 1024] * win for i in range(T)])
    mag = np.sqrt(np.abs(np.fft.rfft(fr, axis=1)) ** 2 + 1e-9)
    fb = O.mel_filterbank(24000, 1024, 128, 0.0, 12000.0).astype(np.float64)
    ref = np.log(np.maximum(mag @ fb.T, 1e-5)).T
    assert mel.shape == ref.shape == (128, T)
    assert np.abs(mel - ref).max() < 2e-3


def _np_forward(cfg, W, mel):
    """Independent numpy (f64) restatement of speaker.rs:443-469 for the oracle cross-check."""
    def conv(x, name, k, dil, act=True):
        w = W[name + ".weight"].astype(np.float64); b = W[name + ".bias"].astype(np.float64)
        cout = b.size; w = w.reshape(cout, -1, k); T = x.shape[1]
        tot = dil * (k - 1); pl = tot // 2
        xp = np.pad(x, ((0, 0), (pl, tot - pl)), mode="reflect") if tot else x
        y = sum(w[:, :, kk] @ xp[:, kk * dil:kk * dil + T] for kk in range(k)) + b[:, None]
        return np.maximum(y, 0) if act else y
    c = cfg
    h = conv(mel.astype(np.float64), "speaker_encoder.blocks.0.conv", c.enc_kernel_sizes[0], c.enc_dilations[0])
    outs = []
    for bi in (1, 2, 3):
        p = f"speaker_encoder.blocks.{bi}"
        C = c.enc_channels[bi]; ch = C // c.enc_res2net_scale
        o = conv(h, p + ".tdnn1.conv", 1, 1)
        parts = [o[:ch]]
        for i in range(c.enc_res2net_scale - 1):
            chunk = o[(i + 1) * ch:(i + 2) * ch]
            parts.append(conv(chunk if i == 0 else chunk + parts[-1], f"{p}.res2net_block.blocks.{i}.conv", c.enc_kernel_sizes[bi], c.enc_dilations[bi]))
        o = conv(np.concatenate(parts), p + ".tdnn2.conv", 1, 1)
        s = o.mean(1, keepdims=True)
        s = conv(s, p + ".se_block.conv1", 1, 1)
        s = 1 / (1 + np.exp(-conv(s, p + ".se_block.conv2", 1, 1, act=False)))
        h = o * s + h
        outs.append(h)
    m = conv(np.concatenate(outs), "speaker_encoder.mfa.conv", c.enc_kernel_sizes[4], c.enc_dilations[4])
    T = m.shape[1]
    mean = m.mean(1, keepdims=True); std = np.sqrt(((m - mean) ** 2).mean(1, keepdims=True) + 1e-5)
    a = np.tanh(conv(np.concatenate([m, np.repeat(mean, T, 1), np.repeat(std, T, 1)]), "speaker_encoder.asp.tdnn.conv", 1, 1))
    a = conv(a, "speaker_encoder.asp.conv", 1, 1, act=False)
    a = np.exp(a - a.max(1, keepdims=True)); a /= a.sum(1, keepdims=True)
    wm = (m * a).sum(1, keepdims=True); ws = np.sqrt((((m - wm) ** 2) * a).sum(1, keepdims=True) + 1e-5)
    pooled = np.concatenate([wm, ws])[:, 0]
    return W["speaker_encoder.fc.weight"].astype(np.float64).reshape(c.enc_dim, -1) @ pooled + W["speaker_encoder.fc.bias"].astype(np.float64)


def _synth_weights(cfg, seed=7):
    """Seeded weights without touching the GPU library's device side (device = -1 handle gives the manifest)."""
    enc = q.SpeakerEncoder(cfg, device=-1)
    from qwen3_tts_rs_amd.speaker import synthetic_speaker_checkpoint
    W = dict(synthetic_speaker_checkpoint(enc, seed))
    enc.close()
    return W


def test_oracle_forward_matches_numpy_and_shape():    # speaker.rs:462-470: mel [128,100] → [enc_dim]
    cfg = q.tiny_speaker_config(enc_dim=64)
    W = _synth_weights(cfg)
    om = O.OracleSpeakerEncoder(cfg)
    for k, v in W.items():
        om.set_tensor(k, v)
    mel = np.random.default_rng(3).standard_normal((128, 100)).astype(np.float32)
    out = om.forward(mel)
    assert out.shape == (64,) and np.isfinite(out).all()
    ref = _np_forward(cfg, W, mel)
    assert np.abs(out - ref).max() <= 1e-4 * np.abs(ref).max()
    om.close()


def test_manifest_names_follow_the_reference():      # weight keys documented in speaker.rs:113-345
    enc = q.SpeakerEncoder(q.SpeakerEncoderConfig(), device=-1)
    names = dict(enc.manifest())
    assert names["speaker_encoder.blocks.0.conv.weight"] == 512 * 128 * 5
    assert names["speaker_encoder.blocks.2.res2net_block.blocks.6.conv.weight"] == 64 * 64 * 3
    assert names["speaker_encoder.blocks.3.se_block.conv1.weight"] == 128 * 512 and names["speaker_encoder.blocks.3.se_block.conv2.bias"] == 512
    assert names["speaker_encoder.mfa.conv.weight"] == 1536 * 1536 and names["speaker_encoder.asp.tdnn.conv.weight"] == 128 * 4608
    assert names["speaker_encoder.asp.conv.weight"] == 1536 * 128 and names["speaker_encoder.fc.weight"] == 1024 * 3072
    assert len(names) == 2 * (1 + 3 * (4 + 7) + 4)
    enc.close()


def test_config_from_json(tmp_path):
    import json
    p = tmp_path / "config.json"
    p.write_text(json.dumps({"tts_model_type": "base", "speaker_encoder_config": {"enc_dim": 2048, "enc_channels": [256, 256, 256, 256, 768]}}))
    cfg, present = q.SpeakerEncoderConfig.from_json(str(p))
    assert present and cfg.enc_dim == 2048 and cfg.enc_channels == [256, 256, 256, 256, 768] and cfg.enc_kernel_sizes == [5, 3, 3, 3, 1]
    p.write_text(json.dumps({"tts_model_type": "custom_voice"}))
    cfg, present = q.SpeakerEncoderConfig.from_json(str(p))
    assert not present and cfg == q.SpeakerEncoderConfig()


# ---------------- HIP path vs oracle ----------------
def _pair(cfg, seed=7):
    om = O.OracleSpeakerEncoder(cfg)
    gm = q.SpeakerEncoder.from_synthetic(cfg, seed=seed, sink=lambda n, a: om.set_tensor(n, a))
    return gm, om


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [24000 * 2 + 123, 5000, 24000 * 6])
def test_gpu_mel_matches_oracle(n):
    gm = q.SpeakerEncoder.from_synthetic(q.tiny_speaker_config())
    x = _audio(n, n)
    mel = gm.mel(x); ref = O.mel_speaker(x)
    assert mel.shape == ref.shape
    assert np.abs(mel - ref).max() <= 2e-3
    gm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [9, 100, 257])
def test_gpu_tiny_forward_stages(T):
    cfg = q.tiny_speaker_config()
    gm, om = _pair(cfg)
    mel = np.random.default_rng(T).standard_normal((128, T)).astype(np.float32) * 2 - 3
    ref, otaps = om.forward(mel, taps=True)
    taps = [np.zeros_like(t) for t in otaps]
    out = gm.forward(mel, taps=taps)
    for i, (a, b) in enumerate(zip(taps, otaps)):
        assert _rel(a, b) <= 2e-4, (i, _rel(a, b))
    assert _rel(out, ref) <= 2e-4
    gm.close(); om.close()


@pytest.mark.gpu
@pytest.mark.parametrize("enc_dim,secs", [(1024, 1.3), (2048, 4.0)])
def test_gpu_full_size_encode(enc_dim, secs):
    """Production shapes (512/1536 channels: the bf16x3 matrix-core convs, k = 5 / 3 reflect-padded convs) end to end
    from audio, stage by stage and on the embedding."""
    cfg = q.SpeakerEncoderConfig(enc_dim=enc_dim)
    gm, om = _pair(cfg, seed=11)
    x = _audio(int(24000 * secs), 5)
    mel = gm.mel(x)
    ref, otaps = om.forward(O.mel_speaker(x), taps=True)
    taps = [np.zeros_like(t) for t in otaps]
    out = gm.forward(mel, taps=taps)
    for i, (a, b) in enumerate(zip(taps, otaps)):
        assert _rel(a, b) <= 5e-4, (i, _rel(a, b))
    assert _rel(out, ref) <= 5e-4
    emb = gm.encode(x)
    assert _rel(emb, om.encode(x)) <= 5e-4 and np.array_equal(emb, out)
    gm.close(); om.close()


@pytest.mark.gpu
def test_gpu_errors():
    gm = q.SpeakerEncoder.from_synthetic(q.tiny_speaker_config())
    with pytest.raises(q.api._lib.Q3Error, match="resample"):
        gm.encode(_audio(24000), sample_rate=16000)
    with pytest.raises(q.api._lib.Q3Error, match="too short"):
        gm.encode(_audio(300))
    e2 = q.SpeakerEncoder(q.tiny_speaker_config())
    with pytest.raises(q.api._lib.Q3Error, match="Missing weight|not finalized"):
        e2.finalize()
    gm.close(); e2.close()


@pytest.mark.gpu
def test_gpu_voice_clone_from_checkpoint_dir(tmp_path):
    """create_voice_clone_prompt → synthesize_voice_clone (lib.rs:1132-1262) on a Base checkpoint directory: the loader
    attaches the speaker encoder from `speaker_encoder.*` + `speaker_encoder_config`; its embedding matches the oracle's,
    and the codes generated from it equal the oracle's run on the same embedding bit for bit."""
    from common import write_checkpoint_dir, synthetic_prompt
    t = q.tiny()
    cfg = q.Q3Config(text_dim=t.text_dim, hidden=t.hidden, inter=t.inter, n_layers=t.n_layers, n_heads=t.n_heads,
                     n_kv_heads=t.n_kv_heads, cp_hidden=t.cp_hidden, cp_inter=t.cp_inter, cp_layers=t.cp_layers,
                     cp_heads=t.cp_heads, cp_kv_heads=t.cp_kv_heads, name="tiny-lm-full-decoder")
    scfg = q.tiny_speaker_config(enc_dim=cfg.hidden)
    raw, spk_raw = write_checkpoint_dir(cfg, str(tmp_path), model_type="base", speaker_cfg=scfg)
    m = q.Qwen3TTS.from_pretrained(str(tmp_path), 0)
    assert m.model_type == q.api.ModelType.Base and m.supports_voice_cloning() and m.has_speaker_encoder() and not m.has_speech_encoder()
    assert m.speaker_encoder.config == scfg
    audio = q.AudioBuffer(_audio(24000 * 2, 3), 24000)
    prompt = m.create_voice_clone_prompt(audio)
    os_ = O.OracleSpeakerEncoder(scfg)
    for k, v in spk_raw.items():
        os_.set_tensor(k, v)
    assert _rel(prompt.speaker_embedding, os_.encode(audio.samples)) <= 2e-4
    with pytest.raises(q.api._lib.Q3Error, match="speech encoder"):
        m.create_voice_clone_prompt(audio, ref_text_ids=[1, 2, 3])
    # non-24 kHz reference audio is resampled first (lib.rs:1156-1166)
    a16 = q.api.resample(audio, 16000)
    p16 = m.create_voice_clone_prompt(a16)
    cos = float(np.dot(p16.speaker_embedding, prompt.speaker_embedding) / (np.linalg.norm(p16.speaker_embedding) * np.linalg.norm(prompt.speaker_embedding)))
    assert cos > 0.9, cos           # band-limited copy of the same clip (8-12 kHz noise gone) → a close embedding
    np.testing.assert_array_equal(p16.speaker_embedding, m.speaker_encoder.encode(q.api.resample_to_24k(a16).samples))
    # codes from the embedding: GPU vs oracle on identical inputs
    om = O.OracleModel(cfg)
    for k, (arr, dt) in raw.items():
        om.set_tensor(k, arr, dt)
    om.finalize(3)
    opts = q.SynthesisOptions(max_length=6, seed=4, eos_token_id=None)
    text = synthetic_prompt(9, 4)
    utt = q.Utterance(text, language=q.Language.English, xvector=prompt.speaker_embedding, seed=4)
    s = m.session([utt], opts); s.prefill(); s.generate(6); codes = s.codes(0).copy(); s.close()
    osess = O.OracleSession(om, utt, opts); ocodes = osess.generate(); osess.close()
    assert codes.shape == (6, 16) and (codes == ocodes).all()
    # a CustomVoice checkpoint has no encoder: the reference's hint (lib.rs:1137-1153)
    m.speaker_encoder = None; m.model_type = q.api.ModelType.CustomVoice
    with pytest.raises(q.api._lib.Q3Error, match="preset speakers"):
        m.create_voice_clone_prompt(audio)
    m.close(); om.close(); os_.close()
