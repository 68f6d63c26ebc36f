"""ctypes wrapper of the CPU oracle (oracle/libq3oracle.so) — TEST INFRASTRUCTURE. Imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libq3oracle.so")


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("q3_oracle.c", "q3_oracle_spk.c", "q3_oracle_mimi.c", "q3_oracle.h")]
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libq3oracle.so"], stdout=subprocess.DEVNULL)


build_oracle()
olib = ctypes.CDLL(LIB)


class OConfig(ctypes.Structure):
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


class OOptions(ctypes.Structure):
    _fields_ = [
        ("temperature", ctypes.c_double), ("top_p", ctypes.c_double), ("repetition_penalty", ctypes.c_double),
        ("seed", ctypes.c_uint64), ("max_length", ctypes.c_int32), ("top_k", ctypes.c_int32),
        ("eos_token_id", ctypes.c_int32), ("chunk_frames", ctypes.c_int32), ("min_new_tokens", ctypes.c_int32),
        ("has_seed", ctypes.c_int32),
    ]


class ORequest(ctypes.Structure):
    _fields_ = [
        ("mode", ctypes.c_int32),
        ("text_ids", ctypes.POINTER(ctypes.c_uint32)), ("n_text", ctypes.c_int32),
        ("instruct_ids", ctypes.POINTER(ctypes.c_uint32)), ("n_instruct", ctypes.c_int32),
        ("speaker_id", ctypes.c_uint32), ("language_id", ctypes.c_uint32),
        ("xvector", ctypes.POINTER(ctypes.c_float)),
        ("opts", OOptions),
        ("ref_codes", ctypes.POINTER(ctypes.c_uint32)), ("n_ref", ctypes.c_int32),
        ("ref_text_ids", ctypes.POINTER(ctypes.c_uint32)), ("n_ref_text", ctypes.c_int32),
    ]


vp, ci = ctypes.c_void_p, ctypes.c_int
olib.q3o_last_error.restype = ctypes.c_char_p
olib.q3o_model_new.restype = vp; olib.q3o_model_new.argtypes = [ctypes.POINTER(OConfig)]
olib.q3o_model_free.argtypes = [vp]
olib.q3o_model_set_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64]
olib.q3o_model_finalize.argtypes = [vp, ci]
olib.q3o_set_threads.argtypes = [ci]
olib.q3o_linear.argtypes = [vp, vp, vp, vp, ci, ci, ci]
olib.q3o_rms_norm.argtypes = [vp, vp, vp, ci, ci, ctypes.c_float]
olib.q3o_fused_residual_rmsnorm.argtypes = [vp, vp, vp, ci, ci, ctypes.c_float, vp, vp]
olib.q3o_rope_table.argtypes = [ctypes.c_float, ci, ci, ci, vp, vp]
olib.q3o_rng_seed.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
olib.q3o_rng_next.restype = ctypes.c_float; olib.q3o_rng_next.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
olib.q3o_build_suppression_mask.argtypes = [ci, ci, vp]
olib.q3o_apply_penalties.argtypes = [vp, ci, vp, ctypes.c_double, ci, ci, ci]
olib.q3o_top_k_filter.argtypes = [vp, ci, ci]
olib.q3o_top_p_filter.argtypes = [vp, ci, ctypes.c_double]
olib.q3o_sample.restype = ctypes.c_uint32
olib.q3o_sample.argtypes = [vp, ci, ctypes.c_double, ci, ctypes.c_double, ctypes.POINTER(ctypes.c_uint64)]
olib.q3o_codes_to_tensor.argtypes = [vp, ci, vp]
olib.q3o_session_new.restype = vp; olib.q3o_session_new.argtypes = [vp, ctypes.POINTER(ORequest)]
olib.q3o_session_free.argtypes = [vp]
olib.q3o_session_prefill_len.argtypes = [vp]
olib.q3o_session_effective.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
olib.q3o_session_prefill_out.argtypes = [vp, vp, vp]
olib.q3o_session_prefill_embeds.argtypes = [vp, vp]
olib.q3o_session_trailing_len.argtypes = [vp]
olib.q3o_session_trailing.argtypes = [vp, vp, vp]
olib.q3o_session_generate.argtypes = [vp, vp, vp, vp]
olib.q3o_session_talker_step.argtypes = [vp, vp, vp, vp]
olib.q3o_session_cp_generate.argtypes = [vp, vp, vp, vp, vp]
olib.q3o_frame_embed.argtypes = [vp, ctypes.c_uint32, vp, vp, vp]
olib.q3o_decode.argtypes = [vp, vp, ci, vp]
olib.q3o_decode_taps.argtypes = [vp, vp, ci, vp, ctypes.POINTER(vp)]
olib.q3o_causal_conv1d.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci]
olib.q3o_causal_trans_conv1d.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci]
olib.q3o_snake_beta.argtypes = [vp, vp, vp, vp, ci, ci]


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def err():
    return olib.q3o_last_error().decode()


def to_oconfig(cfg) -> OConfig:
    c = OConfig()
    for f, _ in OConfig._fields_:
        v = getattr(cfg, f)
        if f == "dec_up_ratios":
            c.dec_up_ratios = (ctypes.c_int32 * 2)(*v)
        elif f == "dec_up_rates":
            c.dec_up_rates = (ctypes.c_int32 * 4)(*v)
        else:
            setattr(c, f, v)
    return c


def to_ooptions(o) -> OOptions:
    r = OOptions()
    r.temperature = float(o.temperature); r.top_p = float(o.top_p); r.repetition_penalty = float(o.repetition_penalty)
    r.seed = 0 if o.seed is None else int(o.seed)
    r.max_length = int(o.max_length); r.top_k = int(o.top_k)
    r.eos_token_id = -1 if o.eos_token_id is None else int(o.eos_token_id)
    r.chunk_frames = int(o.chunk_frames); r.min_new_tokens = int(o.min_new_tokens)
    r.has_seed = 0 if o.seed is None else 1
    return r


class OracleModel:
    def __init__(self, cfg):
        self.cfg = cfg
        oc = to_oconfig(cfg)
        self.h = olib.q3o_model_new(ctypes.byref(oc))

    def set_tensor(self, name, arr, dtype):
        """arr: f32 array, or uint16 bf16 bits (dtype 1) — upconverted exactly to f32."""
        if dtype == 1:
            a = (np.ascontiguousarray(arr, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
        else:
            a = np.ascontiguousarray(arr, dtype=np.float32)
        olib.q3o_model_set_tensor(self.h, name.encode(), ptr(a), a.size)

    def finalize(self, which=3):
        if olib.q3o_model_finalize(self.h, which) != 0:
            raise RuntimeError(err())

    def close(self):
        if self.h:
            olib.q3o_model_free(self.h); self.h = None

    __del__ = close

    def decode(self, codes_frames, taps=False):
        """codes_frames [n][16] u32 → pcm (and optional stage taps dict)."""
        c = np.ascontiguousarray(codes_frames, dtype=np.uint32).reshape(-1, 16)
        T = c.shape[0]
        t64 = np.zeros((16, T), dtype=np.int64)
        olib.q3o_codes_to_tensor(ptr(c), T, ptr(t64))
        pcm = np.zeros(T * self.cfg.samples_per_frame, dtype=np.float32)
        if not taps:
            n = olib.q3o_decode(self.h, ptr(t64), T, ptr(pcm))
            if n < 0:
                raise RuntimeError(err())
            return pcm
        shapes = decoder_tap_shapes(self.cfg, T)
        bufs = [np.zeros(s, dtype=np.float32) for s in shapes]
        arr = (ctypes.c_void_p * 10)(*[ptr(b) for b in bufs])
        n = olib.q3o_decode_taps(self.h, ptr(t64), T, ptr(pcm), arr)
        if n < 0:
            raise RuntimeError(err())
        return pcm, bufs

    def frame_embed(self, sem, codes15, text_add):
        c = np.ascontiguousarray(codes15, dtype=np.uint32); t = np.ascontiguousarray(text_add, dtype=np.float32)
        out = np.zeros(self.cfg.hidden, dtype=np.float32)
        olib.q3o_frame_embed(self.h, int(sem), ptr(c), ptr(t), ptr(out))
        return out


def decoder_tap_shapes(cfg, T):
    LAT, Q = cfg.dec_latent, cfg.dec_q_dim
    shapes = [(Q, T), (LAT, T), (LAT, T)]
    L = T
    for r in cfg.dec_up_ratios:
        L *= r
        shapes.append((LAT, L))
    C = cfg.dec_dim
    shapes.append((C, L))
    for r in cfg.dec_up_rates:
        L *= r; C //= 2
        shapes.append((C, L))
    return shapes


class OracleSession:
    def __init__(self, model: OracleModel, utt, options):
        self.model = model; self.cfg = model.cfg; self.options = options
        r = ORequest()
        r.mode = utt.mode()
        self._t = np.ascontiguousarray(utt.text_ids, dtype=np.uint32)
        r.text_ids = self._t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_text = len(self._t)
        if utt.instruct_ids is not None:
            self._i = np.ascontiguousarray(utt.instruct_ids, dtype=np.uint32)
            r.instruct_ids = self._i.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_instruct = len(self._i)
        r.speaker_id = utt.speaker.token_id(); r.language_id = utt.language.token_id()
        if utt.xvector is not None:
            self._x = np.ascontiguousarray(utt.xvector, dtype=np.float32)
            r.xvector = self._x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        if getattr(utt, "ref_codes", None) is not None:
            self._rc = np.ascontiguousarray(utt.ref_codes, dtype=np.uint32).reshape(-1, 16)
            self._rt = np.ascontiguousarray(utt.ref_text_ids, dtype=np.uint32)
            r.ref_codes = self._rc.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_ref = self._rc.shape[0]
            r.ref_text_ids = self._rt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_ref_text = len(self._rt)
        o = to_ooptions(options)
        if utt.seed is not None:
            o.seed = int(utt.seed); o.has_seed = 1
        if getattr(utt, "max_length", None) is not None:       # per-request frame limit (Utterance.max_length), as Session._fill
            o.max_length = int(utt.max_length)
        r.opts = o
        self.h = olib.q3o_session_new(model.h, ctypes.byref(r))
        if not self.h:
            raise RuntimeError(err())

    def close(self):
        if self.h:
            olib.q3o_session_free(self.h); self.h = None

    __del__ = close

    def prefill_len(self):
        return olib.q3o_session_prefill_len(self.h)

    def effective(self):
        rp = ctypes.c_double(); ml = ctypes.c_int()
        olib.q3o_session_effective(self.h, ctypes.byref(rp), ctypes.byref(ml))
        return rp.value, ml.value

    def prefill_out(self):
        hid = np.zeros(self.cfg.hidden, dtype=np.float32); lg = np.zeros(self.cfg.codec_vocab, dtype=np.float32)
        olib.q3o_session_prefill_out(self.h, ptr(hid), ptr(lg))
        return hid, lg

    def prefill_embeds(self):
        out = np.zeros((self.prefill_len(), self.cfg.hidden), dtype=np.float32)
        olib.q3o_session_prefill_embeds(self.h, ptr(out))
        return out

    def trailing(self):
        n = olib.q3o_session_trailing_len(self.h)
        tr = np.zeros((n, self.cfg.hidden), dtype=np.float32); pad = np.zeros(self.cfg.hidden, dtype=np.float32)
        olib.q3o_session_trailing(self.h, ptr(tr), ptr(pad))
        return tr, pad

    def generate(self, capture=False):
        ml = self.effective()[1]
        codes = np.zeros((ml, 16), dtype=np.uint32)
        tl = np.zeros((ml + 1, self.cfg.codec_vocab), dtype=np.float32) if capture else None
        cl = np.zeros((ml, 15, self.cfg.cp_vocab), dtype=np.float32) if capture else None
        n = olib.q3o_session_generate(self.h, ptr(codes), ptr(tl), ptr(cl))
        if capture:
            return codes[:n], tl[:n + 1], cl[:n]
        return codes[:n]

    def talker_step(self, embed):
        e = np.ascontiguousarray(embed, dtype=np.float32)
        hid = np.zeros(self.cfg.hidden, dtype=np.float32); lg = np.zeros(self.cfg.codec_vocab, dtype=np.float32)
        olib.q3o_session_talker_step(self.h, ptr(e), ptr(hid), ptr(lg))
        return hid, lg

    def cp_generate(self, last_hidden, sem_embed):
        lh = np.ascontiguousarray(last_hidden, dtype=np.float32); se = np.ascontiguousarray(sem_embed, dtype=np.float32)
        codes = np.zeros(15, dtype=np.uint32); lg = np.zeros((15, self.cfg.cp_vocab), dtype=np.float32)
        olib.q3o_session_cp_generate(self.h, ptr(lh), ptr(se), ptr(codes), ptr(lg))
        return codes, lg


# ---- speaker-embedding path (oracle/q3_oracle_spk.c) ----
class OSpkConfig(ctypes.Structure):
    _fields_ = [("mel_dim", ctypes.c_int32), ("enc_dim", ctypes.c_int32), ("channels", ctypes.c_int32 * 5),
                ("kernel_sizes", ctypes.c_int32 * 5), ("dilations", ctypes.c_int32 * 5), ("attention_channels", ctypes.c_int32),
                ("res2net_scale", ctypes.c_int32), ("se_channels", ctypes.c_int32), ("sample_rate", ctypes.c_int32)]


olib.q3o_spk_new.restype = vp; olib.q3o_spk_new.argtypes = [ctypes.POINTER(OSpkConfig)]
olib.q3o_spk_free.argtypes = [vp]
olib.q3o_spk_set_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64]
olib.q3o_spk_last_error.restype = ctypes.c_char_p
olib.q3o_hann_window.argtypes = [ci, vp]
olib.q3o_mel_filterbank.argtypes = [ci, ci, ci, ctypes.c_float, ctypes.c_float, vp]
olib.q3o_mel_frames.argtypes = [ci, ci, ci]
olib.q3o_mel_speaker.argtypes = [vp, ci, vp, ci]
olib.q3o_reflect_pad_1d.argtypes = [vp, ci, ci, ci, ci, vp]
olib.q3o_spk_forward.argtypes = [vp, vp, ci, vp, ctypes.POINTER(vp)]
olib.q3o_spk_encode.argtypes = [vp, vp, ci, vp]


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hann_window(n):
    out = np.empty(n, np.float32); olib.q3o_hann_window(n, _fp(out)); return out


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    out = np.empty((n_mels, n_fft // 2 + 1), np.float32)
    olib.q3o_mel_filterbank(sr, n_fft, n_mels, fmin, fmax, _fp(out)); return out


def mel_speaker(samples):
    x = np.ascontiguousarray(samples, np.float32)
    T = olib.q3o_mel_frames(x.size, 1024, 256)
    out = np.empty((128, T), np.float32)
    assert olib.q3o_mel_speaker(_fp(x), x.size, _fp(out), T) == T
    return out


def reflect_pad_1d(x, pl, pr):
    x = np.ascontiguousarray(x, np.float32); C, T = x.shape
    out = np.empty((C, T + pl + pr), np.float32)
    olib.q3o_reflect_pad_1d(_fp(x), C, T, pl, pr, _fp(out)); return out


class OracleSpeakerEncoder:
    def __init__(self, cfg):
        """cfg: qwen3_tts_rs_amd.SpeakerEncoderConfig"""
        c = OSpkConfig()
        c.mel_dim, c.enc_dim = cfg.mel_dim, cfg.enc_dim
        for i in range(5):
            c.channels[i] = cfg.enc_channels[i]; c.kernel_sizes[i] = cfg.enc_kernel_sizes[i]; c.dilations[i] = cfg.enc_dilations[i]
        c.attention_channels, c.res2net_scale, c.se_channels, c.sample_rate = cfg.enc_attention_channels, cfg.enc_res2net_scale, cfg.enc_se_channels, cfg.sample_rate
        self.cfg = cfg
        self._h = olib.q3o_spk_new(ctypes.byref(c))

    def set_tensor(self, name, arr):
        a = np.ascontiguousarray(arr, np.float32)
        assert olib.q3o_spk_set_tensor(self._h, name.encode(), _fp(a), a.size) == 0

    def forward(self, mel, taps=False):
        m = np.ascontiguousarray(mel, np.float32); T = m.shape[1]
        out = np.empty(self.cfg.enc_dim, np.float32)
        c = self.cfg
        tl = None; tp = None
        if taps:
            tl = [np.empty((c.enc_channels[0], T), np.float32)] + [np.empty((c.enc_channels[i], T), np.float32) for i in (1, 2, 3)] + \
                 [np.empty((c.enc_channels[4], T), np.float32), np.empty(2 * c.enc_channels[4], np.float32)]
            tp = (vp * 6)(*[_fp(t) for t in tl])
        rc = olib.q3o_spk_forward(self._h, _fp(m), T, _fp(out), tp)
        if rc != 0:
            raise RuntimeError(olib.q3o_spk_last_error().decode())
        return (out, tl) if taps else out

    def encode(self, samples):
        x = np.ascontiguousarray(samples, np.float32)
        out = np.empty(self.cfg.enc_dim, np.float32)
        if olib.q3o_spk_encode(self._h, _fp(x), x.size, _fp(out)) != 0:
            raise RuntimeError(olib.q3o_spk_last_error().decode())
        return out

    def close(self):
        if self._h and olib is not None:
            olib.q3o_spk_free(self._h); self._h = None


# ---- speech-tokenizer encoder (oracle/q3_oracle_mimi.c) ----
class OMimiConfig(ctypes.Structure):
    _fields_ = [("n_filters", ctypes.c_int32), ("hidden", ctypes.c_int32), ("ratios", ctypes.c_int32 * 4), ("kernel", ctypes.c_int32),
                ("res_kernel", ctypes.c_int32), ("last_kernel", ctypes.c_int32), ("compress", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("n_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("inter", ctypes.c_int32), ("window", ctypes.c_int32),
                ("cb_size", ctypes.c_int32), ("cb_dim", ctypes.c_int32), ("n_q", ctypes.c_int32), ("n_sem", ctypes.c_int32),
                ("norm_eps", ctypes.c_float), ("rope_theta", ctypes.c_float)]


olib.q3o_mimi_new.restype = vp; olib.q3o_mimi_new.argtypes = [ctypes.POINTER(OMimiConfig)]
olib.q3o_mimi_free.argtypes = [vp]
olib.q3o_mimi_last_error.restype = ctypes.c_char_p; olib.q3o_mimi_last_error.argtypes = [vp]
olib.q3o_mimi_set_tensor.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_int64]
olib.q3o_mimi_frames.argtypes = [ctypes.POINTER(OMimiConfig), ctypes.c_int64]
olib.q3o_mimi_encode.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.POINTER(vp)]


def to_omimi_config(cfg) -> OMimiConfig:
    """cfg: qwen3_tts_rs_amd.SpeechEncoderConfig (same field names)"""
    c = OMimiConfig()
    for f, _ in OMimiConfig._fields_:
        v = getattr(cfg, f)
        if f == "ratios":
            c.ratios = (ctypes.c_int32 * 4)(*v)
        else:
            setattr(c, f, v)
    return c


class OracleSpeechEncoder:
    def __init__(self, cfg):
        self.cfg = cfg
        self._c = to_omimi_config(cfg)
        self._h = olib.q3o_mimi_new(ctypes.byref(self._c))

    def set_tensor(self, name, arr):
        a = np.ascontiguousarray(arr, np.float32)
        assert olib.q3o_mimi_set_tensor(self._h, name.encode(), _fp(a), a.size) == 0

    def frames(self, n_samples):
        return olib.q3o_mimi_frames(ctypes.byref(self._c), n_samples)

    def encode(self, samples, taps=False):
        x = np.ascontiguousarray(samples, np.float32)
        T = self.frames(x.size)
        codes = np.zeros((T, self.cfg.n_q), np.uint32)
        tl = None; tp = None
        if taps:
            T25 = x.size
            for r in self.cfg.ratios:
                T25 = -(-T25 // r)
            tl = [np.empty((self.cfg.hidden, T25), np.float32), np.empty((self.cfg.hidden, T25), np.float32), np.empty((self.cfg.hidden, T), np.float32),
                  np.empty((T, self.cfg.n_q), np.float32)]
            tp = (vp * 4)(*[_fp(t) for t in tl])
        n = olib.q3o_mimi_encode(self._h, _fp(x), x.size, _fp(codes), tp)
        if n < 0:
            raise RuntimeError(olib.q3o_mimi_last_error(self._h).decode())
        assert n == T
        return (codes, tl) if taps else codes

    def close(self):
        if self._h and olib is not None:
            olib.q3o_mimi_free(self._h); self._h = None
