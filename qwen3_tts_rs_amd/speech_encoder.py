"""Speech-tokenizer encoder (Mimi) for ICL voice cloning — host-side mirror of the reference's `Encoder12Hz`
(models/codec/encoder_12hz.rs:34-144) over the C ABI (q3_mimi_* in include/q3tts.h): 24 kHz reference audio -> [T, 16]
codec frames, the `ref_codes` of a VoiceClonePrompt. All arithmetic runs in the gfx950 library; there is no CPU fallback."""
import ctypes
from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Tuple

import numpy as np

from ._lib import CMimiConfig, check, lib


@dataclass
class SpeechEncoderConfig:         # mimi::Config::v0_1(Some(16)) (encoder_12hz.rs:73) = HF MimiConfig(num_quantizers=16)
    n_filters: int = 64
    hidden: int = 512
    ratios: List[int] = field(default_factory=lambda: [4, 5, 6, 8])      # encoder order (reversed upsampling_ratios)
    kernel: int = 7
    res_kernel: int = 3
    last_kernel: int = 3
    compress: int = 2
    n_layers: int = 8
    n_heads: int = 8
    head_dim: int = 64
    inter: int = 2048
    window: int = 250
    cb_size: int = 2048
    cb_dim: int = 256
    n_q: int = 16
    n_sem: int = 1
    norm_eps: float = 1e-5
    rope_theta: float = 1e4

    def to_c(self) -> CMimiConfig:
        c = CMimiConfig()
        for f, _ in CMimiConfig._fields_:
            v = getattr(self, f)
            if f == "ratios":
                c.ratios = (ctypes.c_int32 * 4)(*v)
            else:
                setattr(c, f, v)
        return c


def tiny_speech_config() -> SpeechEncoderConfig:
    """Small shapes for fast tests; widths off the matrix-core tile sizes in places, so the f32 fallback kernels run too."""
    return SpeechEncoderConfig(n_filters=8, hidden=128, ratios=[2, 3, 2, 4], n_layers=2, n_heads=2, inter=96, window=7, cb_size=64, cb_dim=24,
                               n_q=5, n_sem=1)


class SpeechEncoder:
    def __init__(self, config: Optional[SpeechEncoderConfig] = None, device: int = 0):
        self.config = config or SpeechEncoderConfig()
        self.device_index = device
        self._c = self.config.to_c()
        h = ctypes.c_void_p()
        check(lib.q3_mimi_create(ctypes.byref(self._c), device, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.q3_mimi_free(self._h); self._h = None

    __del__ = close

    def manifest(self) -> Iterator[Tuple[str, int]]:
        for i in range(lib.q3_mimi_n_tensors(self._h)):
            name = ctypes.c_char_p(); n = ctypes.c_int64()
            check(lib.q3_mimi_tensor_info(self._h, i, ctypes.byref(name), ctypes.byref(n)))
            yield name.value.decode(), n.value

    def set_tensor(self, name: str, arr: np.ndarray, dtype: int = 0):
        a = np.ascontiguousarray(arr)
        check(lib.q3_mimi_set_tensor(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), dtype, a.size))

    def finalize(self):
        check(lib.q3_mimi_finalize(self._h))

    @classmethod
    def from_safetensors(cls, path: str, config: Optional[SpeechEncoderConfig] = None, device: int = 0) -> "SpeechEncoder":
        """Encoder12Hz::from_safetensors (encoder_12hz.rs:45-48): the `encoder.*` keys of speech_tokenizer/model.safetensors."""
        e = cls(config, device)
        check(lib.q3_mimi_load_safetensors(e._h, str(path).encode()))
        return e

    @classmethod
    def from_synthetic(cls, config: Optional[SpeechEncoderConfig] = None, device: int = 0, seed: int = 0x51337755, sink=None) -> "SpeechEncoder":
        """Seeded random weights of the configured shapes (test/bench data; `sink(name, f32 array)` feeds a checker)."""
        e = cls(config, device)
        for name, arr in synthetic_speech_checkpoint(e, seed):
            e.set_tensor(name, arr, 0)
            if sink is not None:
                sink(name, arr)
        e.finalize()
        return e

    def frames(self, n_samples: int) -> int:
        return lib.q3_mimi_frames(ctypes.byref(self._c), n_samples)

    def tap_shapes(self, n_samples: int) -> List[Tuple[int, ...]]:
        T25 = n_samples
        for r in self.config.ratios:
            T25 = -(-T25 // r)
        return [(self.config.hidden, T25), (self.config.hidden, T25), (self.config.hidden, self.frames(n_samples))]

    def encode(self, samples: np.ndarray, sample_rate: int = 24000, taps: Optional[List[Optional[np.ndarray]]] = None) -> np.ndarray:
        """Encoder12Hz::encode (encoder_12hz.rs:119-144): [T, n_q] u32 codes at 12.5 Hz."""
        x = np.ascontiguousarray(samples, dtype=np.float32)
        nf = ctypes.c_int()
        check(lib.q3_mimi_encode(self._h, x.ctypes.data_as(ctypes.c_void_p), x.size, sample_rate, None, 0, ctypes.byref(nf), None))
        codes = np.zeros((nf.value, self.config.n_q), np.uint32)
        tp = None
        if taps is not None:
            tp = (ctypes.c_void_p * 3)(*[t.ctypes.data_as(ctypes.c_void_p) if t is not None else None for t in taps])
        check(lib.q3_mimi_encode(self._h, x.ctypes.data_as(ctypes.c_void_p), x.size, sample_rate, codes.ctypes.data_as(ctypes.c_void_p),
                                 codes.shape[0], ctypes.byref(nf), tp))
        return codes


def synthetic_speech_checkpoint(enc: SpeechEncoder, seed: int) -> Iterator[Tuple[str, np.ndarray]]:
    """Conv / linear weights ~ 1.4 / sqrt(fan_in) (activations stay O(1) through ELU / LayerNorm), LayerNorm weights near 1,
    layer scales 0.3, codebook sums O(1) with usage counts in [0.5, 1.5] — drawn with the library's counter-hash generator."""
    c = enc.config
    for name, n in enc.manifest():
        out = np.empty(n, np.float32)
        scale, offset = 1.0, 0.0
        if name.endswith("cluster_usage"):
            scale, offset = 0.25, 1.0
        elif name.endswith("embed_sum"):
            scale = 1.0
        elif "layernorm.weight" in name:
            scale, offset = 0.05, 1.0
        elif name.endswith("layer_scale.scale"):
            scale, offset = 0.02, 0.3
        elif name.endswith(".bias"):
            scale = 0.05
        elif name.endswith(".weight"):
            if "encoder.encoder.layers" in name or "downsample" in name:
                # conv [cout][cin][k]: fan_in = cin * k = n / cout; cout is the bias length of the same layer (or hidden for the bias-free downsample)
                cout = dict(enc.manifest()).get(name[:-6] + "bias", c.hidden)
                fan = n // cout
            elif "input_proj" in name or "self_attn" in name or "fc1" in name:
                fan = c.hidden
            else:
                fan = c.inter                        # fc2
            scale = 1.4 / np.sqrt(fan)
        check(lib.q3_synth_fill(seed, name.encode(), 0, float(scale), float(offset), n, out.ctypes.data_as(ctypes.c_void_p)))
        if name.endswith("cluster_usage"):
            out = np.maximum(out, 0.25)
        yield name, out
