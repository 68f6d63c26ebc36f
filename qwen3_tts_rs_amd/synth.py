"""Deterministic synthetic checkpoints (SURVEY.md Appendix B rules) — test/bench DATA, not product
logic: there is no network for real weights, so parity and benchmarks run on seeded random tensors
of the exact Qwen3-TTS shapes. Values come from the library's host-side counter-hash generator
(q3_synth_fill), so the same (seed, name) gives the same tensor everywhere.

Talker / code-predictor tensors are produced as bf16 bit patterns (uint16) — a real checkpoint's
native dtype (reference lib.rs:1394-1396); decoder tensors are f32.
"""
import ctypes
import re
from typing import Dict, Iterator, Tuple

import numpy as np

from . import _lib
from .config import Q3Config

DEFAULT_SEED = 0x51337755

F32, BF16 = 0, 1


def _fill(seed: int, name: str, dtype: int, scale: float, offset: float, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint16 if dtype == BF16 else np.float32)
    _lib.check(_lib.lib.q3_synth_fill(seed, name.encode(), dtype, scale, offset, n, out.ctypes.data_as(ctypes.c_void_p)))
    return out


def bf16_to_f32(a: np.ndarray) -> np.ndarray:
    return (a.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u >> 16).astype(np.uint16)


def manifest(model_handle) -> Iterator[Tuple[str, int, int]]:
    """(name, n_elements, stored_dtype) for every tensor the model expects."""
    n = _lib.lib.q3_model_n_tensors(model_handle)
    for i in range(n):
        name = ctypes.c_char_p(); cnt = ctypes.c_int64(); dt = ctypes.c_int()
        _lib.check(_lib.lib.q3_model_tensor_info(model_handle, i, ctypes.byref(name), ctypes.byref(cnt), ctypes.byref(dt)))
        yield name.value.decode(), cnt.value, dt.value


def _fan_in(cfg: Q3Config, name: str, n: int) -> int:
    H, TD, CH = cfg.hidden, cfg.text_dim, cfg.cp_hidden
    if "code_predictor.model.layers" in name:
        return cfg.cp_inter if "down_proj" in name else (cfg.cp_heads * cfg.head_dim if "o_proj" in name else CH)
    if "talker.model.layers" in name:
        return cfg.inter if "down_proj" in name else (cfg.n_heads * cfg.head_dim if "o_proj" in name else H)
    if "linear_fc" in name:
        return TD
    if "small_to_mtp" in name or "codec_head" in name:
        return H
    if "lm_head" in name:
        return CH
    return 1


def synth_tensor(cfg: Q3Config, seed: int, name: str, n: int, stored: int) -> Tuple[np.ndarray, int]:
    """Returns (array, source_dtype) for one tensor name."""
    if name.startswith("talker."):
        if name.endswith("bias"):
            return bf16_to_f32(_fill(seed, name, BF16, 0.02, 0.0, n)), F32
        if re.search(r"(layernorm|norm)\.weight$", name):
            return bf16_to_f32(_fill(seed, name, BF16, 0.02, 1.0, n)), F32
        if "text_embedding" in name:
            return _fill(seed, name, BF16, 1.0, 0.0, n), BF16
        if "codec_embedding" in name:
            return _fill(seed, name, BF16, 0.3, 0.0, n), BF16
        fan = _fan_in(cfg, name, n)
        boost = 4.0 if ("codec_head" in name or "lm_head" in name) else 1.0   # realistic logit margins
        return _fill(seed, name, BF16, boost / np.sqrt(fan), 0.0, n), BF16
    # decoder (f32)
    if name.endswith("cluster_usage"):
        return 1.0 + np.abs(_fill(seed, name, F32, 1.0, 0.0, n)), F32
    if name.endswith("embedding_sum"):
        cb = _fill(seed, name, F32, 1.0, 0.0, n)
        usage = 1.0 + np.abs(_fill(seed, name.replace("embedding_sum", "cluster_usage"), F32, 1.0, 0.0, cfg.dec_cb_size))
        return (cb.reshape(cfg.dec_cb_size, -1) * usage[:, None]).astype(np.float32).reshape(-1), F32
    if name.endswith(".alpha") or name.endswith(".beta"):
        return _fill(seed, name, F32, 0.1, 0.0, n), F32
    if name.endswith("layer_scale.scale"):
        return _fill(seed, name, F32, 0.001, 0.01, n), F32
    if name.endswith(".gamma"):
        return _fill(seed, name, F32, 0.01, 0.1, n), F32
    if name.endswith("bias"):
        return _fill(seed, name, F32, 0.02, 0.0, n), F32
    if re.search(r"(layernorm|norm)\.weight$", name):
        return _fill(seed, name, F32, 0.02, 1.0, n), F32
    # conv / linear weights: fan_in from the shape
    fan = _decoder_fan_in(cfg, name, n)
    # The stack above the final conv has no normalisation, so its input arrives with RMS ~ 80 on seeded weights; the final
    # 96 -> 1 conv is scaled (2^-9 / sqrt(fan)) so that the PRE-CLAMP waveform has a speech-like RMS of 0.15-0.2 and the
    # clamp(-1, 1) almost never engages — otherwise the PCM tolerance of the parity tests would be checked on +-1 samples only.
    boost = 2.0 ** -9 if name == "decoder.decoder.6.conv.weight" else 1.0
    return _fill(seed, name, F32, boost / np.sqrt(fan), 0.0, n), F32


def _decoder_fan_in(cfg: Q3Config, name: str, n: int) -> int:
    LAT, DH, Q, CD = cfg.dec_latent, cfg.dec_hidden, cfg.dec_q_dim, cfg.dec_cb_dim
    QD, DI = cfg.dec_heads * cfg.dec_head_dim, cfg.dec_inter
    if "output_proj.weight" in name and "quantizer" in name:
        return CD
    if name == "decoder.pre_conv.conv.weight":
        return Q * 3
    if "pre_transformer" in name:
        if "input_proj" in name:
            return LAT
        if "o_proj" in name:
            return QD
        if "down_proj" in name:
            return DI
        return DH
    if "dwconv" in name:
        return 7
    if "pwconv1" in name:
        return LAT
    if "pwconv2" in name:
        return 4 * LAT
    m = re.match(r"decoder\.upsample\.(\d)\.0\.conv\.weight", name)
    if m:
        return LAT
    if name == "decoder.decoder.0.conv.weight":
        return LAT * 7
    m = re.match(r"decoder\.decoder\.(\d)\.block\.(\d)\.(.*)", name)
    if m:
        b, u, rest = int(m.group(1)), int(m.group(2)), m.group(3)
        cin = cfg.dec_dim // (2 ** (b - 1))
        cout = cin // 2
        if u == 1:
            return cin * 2
        return cout * 7 if rest.startswith("conv1") else cout
    if name == "decoder.decoder.6.conv.weight":
        return n
    return 1


def synthetic_checkpoint(cfg: Q3Config, model_handle, seed: int = DEFAULT_SEED) -> Iterator[Tuple[str, np.ndarray, int]]:
    """Yields (name, array, source_dtype) for the whole manifest, one tensor at a time."""
    for name, n, stored in manifest(model_handle):
        arr, dt = synth_tensor(cfg, seed, name, n, stored)
        yield name, arr, dt


def synthetic_prompt(n: int, index: int = 0, vocab: int = 151643) -> np.ndarray:
    """n text ids uniform in [0, vocab) from the PCG stream of seed 1000 + index (the benchmark's prompts;
    host-side only — q3_rng_* are plain C)."""
    import ctypes
    from . import _lib
    st = ctypes.c_uint64()
    _lib.lib.q3_rng_seed(1000 + index, ctypes.byref(st))
    out = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        u = _lib.lib.q3_rng_next(ctypes.byref(st))
        out[i] = min(int(u * vocab), vocab - 1)
    return out
