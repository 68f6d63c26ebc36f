"""Speech-tokenizer encoder (ICL reference codes from raw audio; encoder_12hz.rs:34-144): the gfx950 implementation through
the C ABI (q3_mimi_*) against the CPU oracle (oracle/q3_oracle_mimi.c, itself held to HF's MimiModel by
tests/test_oracle_vs_hf.py). Stage taps to a relative tolerance, codec indices bit-exact except where the oracle's own
nearest-neighbour decision is a near-tie."""
import os

import numpy as np
import pytest

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import _lib
import oracle as O
from common import synthetic_prompt

SEED = 99


def _clip(n, seed=3):
    t = np.arange(n) / 24000.0
    rng = np.random.default_rng(seed)
    x = 0.3 * np.sin(2 * np.pi * 210.0 * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + 0.12 * np.sin(2 * np.pi * 1730.0 * t + 0.3) + \
        0.05 * rng.standard_normal(n)
    return x.astype(np.float32)


def test_manifest_and_frame_count_cpu():
    """Tensor manifest = the HF-format `encoder.*` keys of speech_tokenizer/model.safetensors (encoder_12hz.rs:9-17); frame
    count = ceil over the strides (24 kHz -> 12.5 Hz: 1920 samples per frame)."""
    e = q.SpeechEncoder(device=-1)
    names = dict(e.manifest())
    assert names["encoder.encoder.layers.0.conv.weight"] == 64 * 7 and names["encoder.encoder.layers.12.conv.weight"] == 1024 * 512 * 16
    assert names["encoder.downsample.conv.weight"] == 512 * 512 * 4
    assert names["encoder.quantizer.acoustic_residual_vector_quantizer.layers.14.codebook.embed_sum"] == 2048 * 256
    assert "encoder.quantizer.acoustic_residual_vector_quantizer.layers.15.codebook.embed_sum" not in names      # 1 semantic + 15 acoustic
    assert len([n for n in names if "encoder_transformer.layers." in n]) == 8 * 12
    assert [e.frames(n) for n in (1, 1919, 1920, 1921, 3840, 24000 * 5)] == [1, 1, 1, 2, 2, 63]
    e.close()
    with pytest.raises(_lib.Q3Error):
        q.SpeechEncoder(q.SpeechEncoderConfig(head_dim=32), device=-1)


def _pair(scfg):
    o = O.OracleSpeechEncoder(scfg)
    g = q.SpeechEncoder.from_synthetic(scfg, seed=SEED, sink=o.set_tensor)
    return g, o


def _compare(g, o, x, tag):
    ocodes, otaps = o.encode(x, taps=True)
    taps = [np.zeros(s, np.float32) for s in g.tap_shapes(x.size)]
    codes = g.encode(x, taps=taps)
    errs = [float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) for a, b in zip(taps, otaps[:3])]
    assert errs[0] <= 2e-4 and errs[1] <= 2e-4 and errs[2] <= 2e-4, (tag, errs)
    assert codes.shape == ocodes.shape == (o.frames(x.size), g.config.n_q)
    bad = 0
    for t in range(codes.shape[0]):
        if not (codes[t] == ocodes[t]).all():
            l = int(np.argmin(codes[t] == ocodes[t]))
            assert otaps[3][t, l] <= 2e-3 * max(1.0, float(np.abs(otaps[2]).max()) ** 2), (tag, t, l, float(otaps[3][t, l]))
            bad += 1
    assert bad <= max(1, codes.shape[0] // 10), (tag, bad, codes.shape)
    return errs, bad


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 37, 480, 2000, 5003])
def test_tiny_config_against_oracle(n):
    """Small widths (f32 fallback kernels), a 7-frame attention window shorter than the clip, ragged lengths down to one sample."""
    g, o = _pair(q.tiny_speech_config())
    _compare(g, o, _clip(n), f"tiny_{n}")
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seconds", [0.31, 2.0])
def test_full_config_against_oracle(seconds):
    """The production shapes (64..1024-channel SEANet on the matrix-core conv kernel, 8 x 512-wide layers, 16 x 2048-entry codebooks)."""
    g, o = _pair(q.SpeechEncoderConfig())
    errs, bad = _compare(g, o, _clip(int(24000 * seconds) + 7), f"full_{seconds}")
    g.close(); o.close()


@pytest.mark.gpu
def test_icl_prompt_from_raw_audio(tmp_path):
    """create_voice_clone_prompt with a transcript (lib.rs:1132-1190, ICL branch 1172-1178): a Base-style checkpoint directory
    whose speech_tokenizer/model.safetensors carries `encoder.*` -> from_pretrained attaches the speech encoder -> the prompt's
    ref_codes are the encoder's frames -> ICL synthesis runs from raw audio alone."""
    import torch
    from safetensors.torch import load_file, save_file
    from common import write_checkpoint_dir
    from qwen3_tts_rs_amd.speech_encoder import synthetic_speech_checkpoint
    t = q.tiny()          # tiny talker / code predictor, full-size decoder (config.json carries the talker's shapes only)
    cfg = q.Q3Config(text_dim=t.text_dim, hidden=t.hidden, inter=t.inter, n_layers=t.n_layers, n_heads=t.n_heads, n_kv_heads=t.n_kv_heads,
                     cp_hidden=t.cp_hidden, cp_inter=t.cp_inter, cp_layers=t.cp_layers, cp_heads=t.cp_heads, cp_kv_heads=t.cp_kv_heads,
                     name="tiny-lm-full-decoder")
    scfg = q.tiny_speaker_config(cfg.hidden)
    write_checkpoint_dir(cfg, str(tmp_path), model_type="base", speaker_cfg=scfg, extra=False)
    # add the (full-size) speech encoder to the speech tokenizer file, as the released checkpoints have it
    tokp = os.path.join(str(tmp_path), "speech_tokenizer", "model.safetensors")
    dec = load_file(tokp)
    man = q.SpeechEncoder(device=-1)
    o = O.OracleSpeechEncoder(man.config)
    for name, arr in synthetic_speech_checkpoint(man, SEED):
        dec[name] = torch.from_numpy(arr.copy()); o.set_tensor(name, arr)
    man.close()
    save_file(dec, tokp)
    m = q.Qwen3TTS.from_pretrained(str(tmp_path))
    assert m.has_speaker_encoder() and m.has_speech_encoder() and m.supports_voice_cloning()
    ref = q.AudioBuffer(_clip(24000 + 11))
    prompt = m.create_voice_clone_prompt(ref, ref_text_ids=synthetic_prompt(5, 2))
    ocodes = o.encode(ref.samples)
    assert prompt.ref_codes.shape == ocodes.shape == (13, 16)
    assert (prompt.ref_codes == ocodes).mean() >= 0.9                       # near-ties aside (checked layer by layer above)
    audio, codes = m.synthesize_voice_clone_prompt(synthetic_prompt(7, 1), prompt, q.Language.English, q.SynthesisOptions(max_length=20, seed=4, eos_token_id=None))
    assert codes.shape == (20, 16) and len(audio) == 20 * 1920
    # a speech tokenizer WITHOUT encoder keys: no speech encoder, ICL raises the reference's message, x-vector-only still works
    save_file({k: v for k, v in dec.items() if not k.startswith("encoder.")}, tokp)
    m2 = q.Qwen3TTS.from_pretrained(str(tmp_path))
    assert not m2.has_speech_encoder()
    with pytest.raises(_lib.Q3Error, match="requires a speech encoder"):
        m2.create_voice_clone_prompt(ref, ref_text_ids=synthetic_prompt(5, 2))
    assert m2.create_voice_clone_prompt(ref).ref_codes is None
    m.close(); m2.close(); o.close()
