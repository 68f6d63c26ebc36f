"""Shared helpers of the parity tests: build the product model and the CPU oracle from the SAME
seeded synthetic checkpoint, synthetic prompts, comparison utilities."""
import ctypes

import numpy as np

import qwen3_tts_rs_amd as q
from qwen3_tts_rs_amd import synth, _lib
import oracle as O


def manifest_handle(cfg):
    h = ctypes.c_void_p(); c = cfg.to_c()
    _lib.check(_lib.lib.q3_model_create(ctypes.byref(c), -1, ctypes.byref(h)))
    return h


def oracle_model(cfg, seed=synth.DEFAULT_SEED, which=3):
    """CPU oracle loaded with the synthetic checkpoint (no GPU needed)."""
    h = manifest_handle(cfg)
    om = O.OracleModel(cfg)
    for name, arr, dt in synth.synthetic_checkpoint(cfg, h, seed):
        if which == 1 and name.startswith("decoder."):
            continue
        if which == 2 and name.startswith("talker."):
            continue
        om.set_tensor(name, arr, dt)
    _lib.lib.q3_model_free(h)
    om.finalize(which)
    return om


def model_pair(cfg, seed=synth.DEFAULT_SEED, device=0):
    """(GPU model, CPU oracle) holding identical weights."""
    om = O.OracleModel(cfg)
    gm = q.Qwen3TTS.from_synthetic(cfg, device=device, seed=seed, sink=om.set_tensor)
    om.finalize(3)
    return gm, om


from qwen3_tts_rs_amd.synth import synthetic_prompt      # noqa: E402,F401 — the benchmark's prompt generator lives in the package (bench.py uses it)


def top2_margin(logits):
    s = np.sort(np.asarray(logits, dtype=np.float64))
    return float(s[-1] - s[-2])


def pcm_rms(got, ref, min_unsat=0.9):
    """RMS error of a PCM buffer against the oracle's, for the north-star tolerance (1e-3). Asserts first that the reference
    waveform is real signal — at least `min_unsat` of its samples strictly inside the clamp(-1, 1) of decoder_12hz.rs:496-504 and
    a non-trivial level — so that agreement cannot come from both sides sitting at +-1."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    unsat = float((np.abs(ref) < 0.999).mean())
    level = float(np.sqrt(np.mean(ref ** 2)))
    assert unsat >= min_unsat and level >= 0.02, f"reference PCM is saturated or silent (unsaturated {unsat:.3f}, rms {level:.3f})"
    return float(np.sqrt(np.mean((got - ref) ** 2)))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def write_checkpoint_dir(cfg, root, seed=synth.DEFAULT_SEED, with_config=True, model_type="custom_voice",
                         f16_names=(), extra=True, speaker_cfg=None):
    """Write a synthetic checkpoint in the reference's on-disk layout (lib.rs:180-262) with the independent
    `safetensors` package: <root>/config.json, <root>/model.safetensors (bf16 talker + code predictor),
    <root>/speech_tokenizer/model.safetensors (f32 decoder). Tensors named in `f16_names` are stored as F16
    (values rounded to f16 first, so they survive exactly); `extra` adds tensors the hot path must skip."""
    import json, os
    import torch
    from safetensors.torch import save_file
    os.makedirs(os.path.join(root, "speech_tokenizer"), exist_ok=True)
    h = manifest_handle(cfg)
    main, dec, raw = {}, {}, {}
    for name, arr, dt in synth.synthetic_checkpoint(cfg, h, seed):
        if dt == synth.BF16:
            t = torch.from_numpy(arr.view(np.int16).copy()).view(torch.bfloat16)
            fan = synth._fan_in(cfg, name, arr.size)
            if fan > 1 and arr.size % fan == 0:
                t = t.reshape(arr.size // fan, fan)
        else:
            t = torch.from_numpy(arr.copy())
        if name in f16_names:
            t = t.float().half()
            arr = t.float().numpy().reshape(-1)
            dt = synth.F32
        raw[name] = (arr, dt)
        (dec if name.startswith("decoder.") else main)[name] = t
    _lib.lib.q3_model_free(h)
    spk_raw = {}
    if speaker_cfg is not None:
        # Base checkpoints hold the ECAPA-TDNN under speaker_encoder.* in the main file, bf16 like everything else there
        from qwen3_tts_rs_amd.speaker import SpeakerEncoder, synthetic_speaker_checkpoint
        enc = SpeakerEncoder(speaker_cfg, device=-1)
        for name, arr in synthetic_speaker_checkpoint(enc, seed):
            b = synth.f32_to_bf16(arr)
            spk_raw[name] = synth.bf16_to_f32(b)
            main[name] = torch.from_numpy(b.view(np.int16).copy()).view(torch.bfloat16)
        enc.close()
    elif extra:
        main["speaker_encoder.blocks.0.conv.weight"] = torch.zeros(4, 3, 5)
        dec["encoder.downsample.conv.weight"] = torch.ones(2, 2, dtype=torch.float64)
    save_file(main, os.path.join(root, "model.safetensors"), metadata={"format": "pt"})
    save_file(dec, os.path.join(root, "speech_tokenizer", "model.safetensors"))
    if with_config:
        conf = {
            "architectures": ["Qwen3TTSForConditionalGeneration"], "model_type": "qwen3_tts",
            "tts_model_type": model_type, "tts_model_size": "1b7" if cfg.hidden == 2048 else "0b6",
            "talker_config": {
                "hidden_size": cfg.hidden, "intermediate_size": cfg.inter, "num_hidden_layers": cfg.n_layers,
                "num_attention_heads": cfg.n_heads, "num_key_value_heads": cfg.n_kv_heads, "head_dim": cfg.head_dim,
                "vocab_size": cfg.codec_vocab, "text_vocab_size": cfg.text_vocab, "text_hidden_size": cfg.text_dim,
                "rms_norm_eps": 1e-06, "rope_theta": 1000000, "max_position_embeddings": 32768,
                "rope_scaling": {"mrope_section": [24, 20, 20], "interleaved": True, "type": "default"},
                "spk_id": {"ryan": 3061, "中文": 1}, "hidden_act": "silu", "use_cache": True,
                "code_predictor_config": {
                    "hidden_size": cfg.cp_hidden, "intermediate_size": cfg.cp_inter, "num_hidden_layers": cfg.cp_layers,
                    "num_attention_heads": cfg.cp_heads, "num_key_value_heads": cfg.cp_kv_heads, "head_dim": cfg.head_dim,
                    "vocab_size": cfg.cp_vocab, "num_code_groups": cfg.n_groups, "rms_norm_eps": 1e-06,
                    "rope_theta": 1000000.0, "tie_word_embeddings": False, "sliding_window": None,
                },
            },
        }
        if speaker_cfg is not None:
            conf["speaker_encoder_config"] = {
                "mel_dim": speaker_cfg.mel_dim, "enc_dim": speaker_cfg.enc_dim, "enc_channels": speaker_cfg.enc_channels,
                "enc_kernel_sizes": speaker_cfg.enc_kernel_sizes, "enc_dilations": speaker_cfg.enc_dilations,
                "enc_attention_channels": speaker_cfg.enc_attention_channels, "enc_res2net_scale": speaker_cfg.enc_res2net_scale,
                "enc_se_channels": speaker_cfg.enc_se_channels, "sample_rate": speaker_cfg.sample_rate}
        with open(os.path.join(root, "config.json"), "w") as f:
            json.dump(conf, f, indent=2)
    if speaker_cfg is not None:
        return raw, spk_raw
    return raw
