"""Speaker encoder (ECAPA-TDNN) for x-vector voice cloning — host-side mirror of the reference's
`SpeakerEncoder` (models/speaker.rs:345-469) and `SpeakerEncoderConfig` (models/config.rs:100-174) over the C ABI
(q3_spk_* in include/q3tts.h). All arithmetic runs in the gfx950 library; there is no CPU fallback."""
import ctypes
import dataclasses
from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import CSpkConfig, check, lib


@dataclass
class SpeakerEncoderConfig:        # config.rs:100-174 (defaults = serde defaults)
    mel_dim: int = 128
    enc_dim: int = 1024
    enc_channels: List[int] = field(default_factory=lambda: [512, 512, 512, 512, 1536])
    enc_kernel_sizes: List[int] = field(default_factory=lambda: [5, 3, 3, 3, 1])
    enc_dilations: List[int] = field(default_factory=lambda: [1, 2, 3, 4, 1])
    enc_attention_channels: int = 128
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    sample_rate: int = 24000

    def to_c(self) -> CSpkConfig:
        c = CSpkConfig()
        c.mel_dim, c.enc_dim = self.mel_dim, self.enc_dim
        for i in range(5):
            c.channels[i] = self.enc_channels[i]; c.kernel_sizes[i] = self.enc_kernel_sizes[i]; c.dilations[i] = self.enc_dilations[i]
        c.attention_channels, c.res2net_scale = self.enc_attention_channels, self.enc_res2net_scale
        c.se_channels, c.sample_rate = self.enc_se_channels, self.sample_rate
        return c

    @classmethod
    def from_c(cls, c: CSpkConfig) -> "SpeakerEncoderConfig":
        return cls(c.mel_dim, c.enc_dim, list(c.channels), list(c.kernel_sizes), list(c.dilations), c.attention_channels,
                   c.res2net_scale, c.se_channels, c.sample_rate)

    @classmethod
    def from_json(cls, path: str) -> Tuple["SpeakerEncoderConfig", bool]:
        """(config, present): `speaker_encoder_config` of a config.json; defaults and present=False when absent."""
        c = CSpkConfig(); present = ctypes.c_int(0)
        check(lib.q3_spk_config_from_json(str(path).encode(), ctypes.byref(c), ctypes.byref(present)))
        return cls.from_c(c), bool(present.value)


def tiny_speaker_config(enc_dim: int = 64) -> SpeakerEncoderConfig:
    """Small shapes for fast tests (channel counts off the matrix-core tile sizes on purpose: the f32 fallbacks run)."""
    return SpeakerEncoderConfig(enc_dim=enc_dim, enc_channels=[48, 48, 48, 48, 80], enc_attention_channels=24,
                                enc_res2net_scale=4, enc_se_channels=20)


class SpeakerEncoder:
    def __init__(self, config: Optional[SpeakerEncoderConfig] = None, device: int = 0):
        self.config = config or SpeakerEncoderConfig()
        self.device_index = device
        h = ctypes.c_void_p(); c = self.config.to_c()
        check(lib.q3_spk_create(ctypes.byref(c), device, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.q3_spk_free(self._h); self._h = None

    __del__ = close

    def manifest(self) -> Iterator[Tuple[str, int]]:
        for i in range(lib.q3_spk_n_tensors(self._h)):
            name = ctypes.c_char_p(); n = ctypes.c_int64()
            check(lib.q3_spk_tensor_info(self._h, i, ctypes.byref(name), ctypes.byref(n)))
            yield name.value.decode(), n.value

    def set_tensor(self, name: str, arr: np.ndarray, dtype: int = 0):
        a = np.ascontiguousarray(arr)
        check(lib.q3_spk_set_tensor(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), dtype, a.size))

    def finalize(self):
        check(lib.q3_spk_finalize(self._h))

    @classmethod
    def from_safetensors(cls, path: str, config: Optional[SpeakerEncoderConfig] = None, device: int = 0) -> "SpeakerEncoder":
        e = cls(config, device)
        check(lib.q3_spk_load_safetensors(e._h, str(path).encode()))
        return e

    @classmethod
    def from_synthetic(cls, config: Optional[SpeakerEncoderConfig] = None, device: int = 0, seed: int = 0x51337755, sink=None) -> "SpeakerEncoder":
        """Seeded random weights of the configured shapes (test/bench data; `sink(name, f32 array)` feeds a checker)."""
        e = cls(config, device)
        for name, arr in synthetic_speaker_checkpoint(e, seed):
            e.set_tensor(name, arr, 0)
            if sink is not None:
                sink(name, arr)
        e.finalize()
        return e

    @staticmethod
    def mel_frames(n_samples: int) -> int:
        return lib.q3_spk_mel_frames(n_samples)

    def mel(self, samples: np.ndarray) -> np.ndarray:
        """MelSpectrogram::compute_for_speaker_encoder (mel.rs:135-166) → [mel_dim, T] log-mel."""
        x = np.ascontiguousarray(samples, dtype=np.float32)
        T = lib.q3_spk_mel_frames(x.size)
        out = np.empty((self.config.mel_dim, T), np.float32); nf = ctypes.c_int()
        check(lib.q3_spk_mel(self._h, x.ctypes.data_as(ctypes.c_void_p), x.size, out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(nf)))
        return out

    def forward(self, mel: np.ndarray, taps: Optional[List[Optional[np.ndarray]]] = None) -> np.ndarray:
        """SpeakerEncoder::forward (speaker.rs:443-469) on one mel [mel_dim, T] → [enc_dim]."""
        m = np.ascontiguousarray(mel, dtype=np.float32)
        assert m.ndim == 2 and m.shape[0] == self.config.mel_dim
        out = np.empty(self.config.enc_dim, np.float32)
        tp = None
        if taps is not None:
            tp = (ctypes.c_void_p * 6)(*[t.ctypes.data_as(ctypes.c_void_p) if t is not None else None for t in taps])
        check(lib.q3_spk_forward(self._h, m.ctypes.data_as(ctypes.c_void_p), m.shape[1], out.ctypes.data_as(ctypes.c_void_p), tp))
        return out

    def encode(self, samples: np.ndarray, sample_rate: int = 24000) -> np.ndarray:
        """SpeakerEncoder::encode (speaker.rs:431-438): raw (unnormalised) embedding [enc_dim]."""
        x = np.ascontiguousarray(samples, dtype=np.float32)
        out = np.empty(self.config.enc_dim, np.float32)
        check(lib.q3_spk_encode(self._h, x.ctypes.data_as(ctypes.c_void_p), x.size, sample_rate, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def tap_shapes(self, T: int) -> List[Tuple[int, ...]]:
        c = self.config
        return [(c.enc_channels[0], T), (c.enc_channels[1], T), (c.enc_channels[2], T), (c.enc_channels[3], T), (c.enc_channels[4], T), (2 * c.enc_channels[4],)]


def synthetic_speaker_checkpoint(enc: SpeakerEncoder, seed: int) -> Iterator[Tuple[str, np.ndarray]]:
    """Conv weights ~ U·sqrt(3/fan_in) style scaling via the library's counter-hash generator; biases small."""
    c = enc.config
    fan = {}
    fan["speaker_encoder.blocks.0.conv.weight"] = c.mel_dim * c.enc_kernel_sizes[0]
    for bi in range(1, 4):
        C = c.enc_channels[bi]; ch = C // c.enc_res2net_scale
        fan[f"speaker_encoder.blocks.{bi}.tdnn1.conv.weight"] = C
        fan[f"speaker_encoder.blocks.{bi}.tdnn2.conv.weight"] = C
        for i in range(c.enc_res2net_scale - 1):
            fan[f"speaker_encoder.blocks.{bi}.res2net_block.blocks.{i}.conv.weight"] = ch * c.enc_kernel_sizes[bi]
        fan[f"speaker_encoder.blocks.{bi}.se_block.conv1.weight"] = C
        fan[f"speaker_encoder.blocks.{bi}.se_block.conv2.weight"] = c.enc_se_channels
    fan["speaker_encoder.mfa.conv.weight"] = sum(c.enc_channels[1:4]) * c.enc_kernel_sizes[4]
    fan["speaker_encoder.asp.tdnn.conv.weight"] = 3 * c.enc_channels[4]
    fan["speaker_encoder.asp.conv.weight"] = c.enc_attention_channels
    fan["speaker_encoder.fc.weight"] = 2 * c.enc_channels[4]
    for name, n in enc.manifest():
        out = np.empty(n, np.float32)
        if name.endswith("bias"):
            scale = 0.05
        else:
            scale = 1.5 / np.sqrt(fan[name])          # keeps activations O(1) through the ReLU stack
        check(lib.q3_synth_fill(seed, name.encode(), 0, float(scale), 0.0, n, out.ctypes.data_as(ctypes.c_void_p)))
        yield name, out


@dataclass
class VoiceClonePrompt:            # lib.rs:123-134
    speaker_embedding: np.ndarray                 # [enc_dim] f32
    ref_codes: Optional[np.ndarray] = None        # [T, 16] u32 (ICL mode)
    ref_text_ids: Optional[np.ndarray] = None     # tokenized reference text (ICL mode)
