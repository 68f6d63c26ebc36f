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


def synthetic_prompt(n, index=0, vocab=151643):
    """n text ids uniform in [0, vocab) from PCG seed 1000+index."""
    st = ctypes.c_uint64()
    _lib.lib.q3_rng_seed(1000 + index, ctypes.byref(st))
    out = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        u = _lib.lib.q3_rng_next(ctypes.byref(st))
        out[i] = min(int(u * vocab), vocab - 1)
    return out


def top2_margin(logits):
    s = np.sort(np.asarray(logits, dtype=np.float64))
    return float(s[-1] - s[-2])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
