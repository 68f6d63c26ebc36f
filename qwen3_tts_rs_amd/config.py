"""Shape constants of the Qwen3-TTS variants (reference: src/models/talker.rs:176-290,
src/models/code_predictor.rs:48-113, src/models/codec/decoder_12hz.rs:14-67).

`Q3Config` mirrors the C struct `q3_config` (include/q3tts.h) field for field.
"""
import ctypes
from dataclasses import dataclass, field, asdict
from typing import List


class CConfig(ctypes.Structure):
    _fields_ = [
        ("text_vocab", ctypes.c_int32), ("text_dim", ctypes.c_int32), ("hidden", ctypes.c_int32),
        ("inter", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("n_heads", ctypes.c_int32),
        ("n_kv_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("codec_vocab", ctypes.c_int32),
        ("cp_hidden", ctypes.c_int32), ("cp_inter", ctypes.c_int32), ("cp_layers", ctypes.c_int32),
        ("cp_heads", ctypes.c_int32), ("cp_kv_heads", ctypes.c_int32), ("cp_vocab", ctypes.c_int32),
        ("n_groups", ctypes.c_int32), ("rms_eps", ctypes.c_float), ("rope_theta", ctypes.c_float),
        ("dec_cb_dim", ctypes.c_int32), ("dec_q_dim", ctypes.c_int32), ("dec_latent", ctypes.c_int32),
        ("dec_hidden", ctypes.c_int32), ("dec_layers", ctypes.c_int32), ("dec_heads", ctypes.c_int32),
        ("dec_head_dim", ctypes.c_int32), ("dec_inter", ctypes.c_int32), ("dec_cb_size", ctypes.c_int32),
        ("dec_dim", ctypes.c_int32), ("dec_up_ratios", ctypes.c_int32 * 2), ("dec_up_rates", ctypes.c_int32 * 4),
        ("dec_eps", ctypes.c_float), ("dec_theta", ctypes.c_float),
    ]


@dataclass
class Q3Config:
    text_vocab: int = 151936
    text_dim: int = 2048
    hidden: int = 1024
    inter: int = 3072
    n_layers: int = 28
    n_heads: int = 16
    n_kv_heads: int = 8
    head_dim: int = 128
    codec_vocab: int = 3072
    cp_hidden: int = 1024
    cp_inter: int = 3072
    cp_layers: int = 5
    cp_heads: int = 16
    cp_kv_heads: int = 8
    cp_vocab: int = 2048
    n_groups: int = 16
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    dec_cb_dim: int = 256
    dec_q_dim: int = 512
    dec_latent: int = 1024
    dec_hidden: int = 512
    dec_layers: int = 8
    dec_heads: int = 16
    dec_head_dim: int = 64
    dec_inter: int = 1024
    dec_cb_size: int = 2048
    dec_dim: int = 1536
    dec_up_ratios: List[int] = field(default_factory=lambda: [2, 2])
    dec_up_rates: List[int] = field(default_factory=lambda: [8, 5, 4, 3])
    dec_eps: float = 1e-5
    dec_theta: float = 1e4
    name: str = "qwen3-tts-0.6b"

    def to_c(self) -> CConfig:
        c = CConfig()
        for f, _ in CConfig._fields_:
            v = getattr(self, f)
            if f == "dec_up_ratios":
                c.dec_up_ratios = (ctypes.c_int32 * 2)(*v)
            elif f == "dec_up_rates":
                c.dec_up_rates = (ctypes.c_int32 * 4)(*v)
            else:
                setattr(c, f, v)
        return c

    @classmethod
    def from_c(cls, c: CConfig, name: str = "") -> "Q3Config":
        kw = {}
        for f, _ in CConfig._fields_:
            v = getattr(c, f)
            kw[f] = list(v) if f in ("dec_up_ratios", "dec_up_rates") else v
        if not name:
            name = "qwen3-tts-1.7b" if kw["hidden"] == 2048 else ("qwen3-tts-0.6b" if kw["hidden"] == 1024 else "custom")
        return cls(name=name, **kw)

    @classmethod
    def from_json(cls, path: str):
        """ParsedModelConfig::from_file (config.rs:238-336) → (Q3Config, model_type); parsed by the C++ loader."""
        from . import _lib
        c = CConfig(); mt = ctypes.c_int(-1)
        _lib.check(_lib.lib.q3_config_from_json(str(path).encode(), ctypes.byref(c), ctypes.byref(mt)))
        return cls.from_c(c), mt.value

    @property
    def samples_per_frame(self) -> int:
        n = 1
        for r in list(self.dec_up_ratios) + list(self.dec_up_rates):
            n *= r
        return n


def qwen3_tts_0_6b() -> Q3Config:
    """TalkerConfig::default (talker.rs:208-230): hidden 1024, intermediate 3072."""
    return Q3Config(name="qwen3-tts-0.6b")


def qwen3_tts_1_7b() -> Q3Config:
    """TalkerConfig::custom_voice (talker.rs:257-274): hidden 2048, intermediate 6144;
    code predictor stays 1024 with a 2048->1024 small_to_mtp_projection (code_predictor.rs:99-113)."""
    return Q3Config(hidden=2048, inter=6144, name="qwen3-tts-1.7b")


def tiny(decoder: bool = True) -> Q3Config:
    """Shrunk configuration for CPU-sized parity runs: same op graph, same vocabularies (token ids
    are hard-wired in the reference), small widths / depths. The vocoder keeps its 1920x upsampling."""
    return Q3Config(
        text_dim=32, hidden=64, inter=128, n_layers=2, n_heads=2, n_kv_heads=1,
        cp_hidden=32, cp_inter=64, cp_layers=2, cp_heads=2, cp_kv_heads=1,
        dec_cb_dim=16, dec_q_dim=32, dec_latent=64, dec_hidden=32, dec_layers=2, dec_heads=2,
        dec_inter=64, dec_dim=96, name="tiny")


def tiny_same_width() -> Q3Config:
    """Tiny config whose code predictor has the talker width (no small_to_mtp_projection; the
    0.6B topology)."""
    c = tiny()
    c.cp_hidden = 64
    c.cp_inter = 128
    c.name = "tiny-0.6b-like"
    return c
