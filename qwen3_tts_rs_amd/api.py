"""Python mirror of the reference's public API over the C ABI (include/q3tts.h).

Names, argument meaning and error behaviour follow the reference crate's re-exports
(src/lib.rs:107-117): `Qwen3TTS`, `SynthesisOptions`, `SynthesisTiming`, `StreamingSession`,
`AudioBuffer`, `Speaker`, `Language`, `CODEC_EOS_TOKEN_ID`, `SAMPLES_PER_FRAME`. Differences:
text arrives as token ids (the tokenizer, src/tokenizer/text.rs, is outside the hot path) and the
synthesis calls accept a LIST of utterances, which run as one batch on the GPU (the reference is
batch 1; each sequence behaves exactly like its own batch-1 run).
"""
import ctypes
import enum
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import lib, check, COptions, CRequest, CTiming
from .config import Q3Config
from . import synth

MAX_BATCH = 64                 # Q3_MAX_BATCH (include/q3tts.h): sequences per session
CODEC_EOS_TOKEN_ID = 2150      # lib.rs:1466
SAMPLES_PER_FRAME = 1920       # lib.rs:1469


class Language(enum.Enum):     # talker.rs:59-108
    Chinese = 2055
    English = 2050
    Japanese = 2058
    Korean = 2064
    German = 2053
    French = 2061
    Russian = 2069
    Portuguese = 2071
    Spanish = 2054
    Italian = 2070

    def token_id(self) -> int:
        return self.value

    @staticmethod
    def from_str(s: str) -> "Language":
        table = {"english": "English", "en": "English", "chinese": "Chinese", "zh": "Chinese", "japanese": "Japanese",
                 "ja": "Japanese", "korean": "Korean", "ko": "Korean", "german": "German", "de": "German",
                 "french": "French", "fr": "French", "russian": "Russian", "ru": "Russian", "portuguese": "Portuguese",
                 "pt": "Portuguese", "spanish": "Spanish", "es": "Spanish", "italian": "Italian", "it": "Italian"}
        k = s.lower()
        if k not in table:
            raise ValueError(f"Unknown language: {s}")
        return Language[table[k]]


class Speaker(enum.Enum):      # talker.rs:111-157
    Serena = 3066
    Vivian = 3065
    UncleFu = 3010
    Ryan = 3061
    Aiden = 2861
    OnoAnna = 2873
    Sohee = 2864
    Eric = 2875
    Dylan = 2878

    def token_id(self) -> int:
        return self.value

    def native_language(self) -> Language:
        if self in (Speaker.Ryan, Speaker.Aiden):
            return Language.English
        if self is Speaker.OnoAnna:
            return Language.Japanese
        if self is Speaker.Sohee:
            return Language.Korean
        return Language.Chinese

    @staticmethod
    def from_str(s: str) -> "Speaker":
        table = {"ryan": "Ryan", "serena": "Serena", "vivian": "Vivian", "aiden": "Aiden", "uncle_fu": "UncleFu",
                 "unclefu": "UncleFu", "ono_anna": "OnoAnna", "onoanna": "OnoAnna", "sohee": "Sohee", "eric": "Eric",
                 "dylan": "Dylan"}
        k = s.lower()
        if k not in table:
            raise ValueError(f"Unknown speaker: {s}")
        return Speaker[table[k]]


@dataclass
class SynthesisOptions:        # lib.rs:1786-1836
    max_length: int = 2048
    temperature: float = 0.9
    top_k: int = 50
    top_p: float = 0.9
    repetition_penalty: float = 1.05
    eos_token_id: Optional[int] = CODEC_EOS_TOKEN_ID
    chunk_frames: int = 10
    min_new_tokens: int = 2
    seed: Optional[int] = None

    def to_c(self) -> COptions:
        o = COptions()
        o.temperature = float(self.temperature); o.top_p = float(self.top_p)
        o.repetition_penalty = float(self.repetition_penalty)
        o.seed = 0 if self.seed is None else int(self.seed)
        o.max_length = int(self.max_length); o.top_k = int(self.top_k)
        o.eos_token_id = -1 if self.eos_token_id is None else int(self.eos_token_id)
        o.chunk_frames = int(self.chunk_frames); o.min_new_tokens = int(self.min_new_tokens)
        o.has_seed = 0 if self.seed is None else 1
        return o


@dataclass
class SynthesisTiming:         # lib.rs:138-147
    prefill_ms: float
    generation_ms: float
    generation_frames: int
    decode_ms: float


@dataclass
class AudioBuffer:             # audio/io.rs:28-34
    samples: np.ndarray
    sample_rate: int = 24000

    def __len__(self) -> int:
        return int(self.samples.shape[0])

    def duration(self) -> float:
        return len(self) / float(self.sample_rate)

    def save(self, path: str):                       # AudioBuffer::save → save_wav (audio/io.rs:143-165)
        save_wav(path, self.samples, self.sample_rate)

    @classmethod
    def load(cls, path: str) -> "AudioBuffer":       # AudioBuffer::load → load_wav (audio/io.rs:106-141)
        return load_wav(path)


@dataclass
class Utterance:
    """One request: token ids + conditioning (the argument lists of lib.rs:718-724 / 802-808)."""
    text_ids: Sequence[int]
    speaker: Speaker = Speaker.Ryan
    language: Language = Language.English
    instruct_ids: Optional[Sequence[int]] = None     # voice design
    xvector: Optional[np.ndarray] = None             # voice clone: speaker embedding (x-vector)
    ref_codes: Optional[np.ndarray] = None           # ICL voice clone: reference codec frames [n_ref][16]
    ref_text_ids: Optional[Sequence[int]] = None     # ICL voice clone: reference transcript token ids
    seed: Optional[int] = None                       # overrides options.seed for this sequence
    max_length: Optional[int] = None                 # overrides options.max_length for this sequence (rows of a session end at their own limit)
    options: Optional["SynthesisOptions"] = None     # this sequence's own sampling options (SynthesisOptions is per call in the reference); default: the session's

    def mode(self) -> int:
        if self.instruct_ids is not None:
            return 2
        if self.xvector is not None:
            return 1
        return 0


def fill_request(r, u: "Utterance", options: "SynthesisOptions", keep: list):
    """q3_request of one utterance; the arrays it points to are appended to `keep` (the caller keeps them alive)"""
    r.mode = u.mode()
    t = np.ascontiguousarray(u.text_ids, dtype=np.uint32); keep.append(t)
    r.text_ids = t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_text = len(t)
    if u.instruct_ids is not None:
        ins = np.ascontiguousarray(u.instruct_ids, dtype=np.uint32); keep.append(ins)
        r.instruct_ids = ins.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_instruct = len(ins)
    r.speaker_id = u.speaker.token_id(); r.language_id = u.language.token_id()
    if u.xvector is not None:
        xv = np.ascontiguousarray(u.xvector, dtype=np.float32); keep.append(xv)
        r.xvector = xv.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    if u.ref_codes is not None:          # prepended at decode even without a transcript (lib.rs:1022); ICL needs both
        rc = np.ascontiguousarray(u.ref_codes, dtype=np.uint32).reshape(-1, 16); keep.append(rc)
        r.ref_codes = rc.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_ref = rc.shape[0]
        if u.ref_text_ids is not None:
            rt = np.ascontiguousarray(u.ref_text_ids, dtype=np.uint32); keep.append(rt)
            r.ref_text_ids = rt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)); r.n_ref_text = len(rt)
    o = (u.options or options).to_c()
    o.chunk_frames = options.chunk_frames           # the streaming chunk is a property of the session
    if u.seed is not None:
        o.seed = int(u.seed); o.has_seed = 1
    if u.max_length is not None:
        o.max_length = int(u.max_length)
    r.opts = o


class Session:
    """A batch of utterances on one GPU (q3_session). Owns KV pages, RNG streams, penalty masks."""

    def __init__(self, model: "Qwen3TTS", utts: Sequence[Utterance], options: SynthesisOptions, debug: bool = False,
                 frame_budget: int = 0, prompt_budget: int = 0, kv_bf16: bool = False):
        """frame_budget / prompt_budget > 0: capacity (frames per row / prompt positions per row) for requests swapped in
        later with a larger max_length or a longer prompt (q3_session_create_reserved); the rows of `utts` still end at
        their own limits."""
        self.model = model
        self.B = len(utts)
        self.options = options
        self._keep = []
        self._ref_frames = [0 if u.ref_codes is None else int(np.asarray(u.ref_codes).reshape(-1, 16).shape[0]) for u in utts]
        self._frame_budget = int(frame_budget)
        self._row_limit = [self._limit_of(u) for u in utts]       # frames a row can reach: sizes the PCM buffers of run()
        reqs = (CRequest * self.B)()
        for i, u in enumerate(utts):
            self._fill(reqs[i], u)
        h = ctypes.c_void_p()
        if frame_budget > 0 or prompt_budget > 0:
            check(lib.q3_session_create_reserved(model._h, reqs, self.B, int(frame_budget), int(prompt_budget), ctypes.byref(h)))
        else:
            check(lib.q3_session_create(model._h, reqs, self.B, ctypes.byref(h)))
        self._h = h
        if kv_bf16:      # the reference GPU path's cache dtype (q3_session_set_kv_dtype): half the K/V bytes, not bit-comparable with the F32 oracle
            check(lib.q3_session_set_kv_dtype(self._h, 1))
        if debug:
            check(lib.q3_session_set_debug(self._h, 1))

    def _fill(self, r, u: Utterance):
        fill_request(r, u, self.options, self._keep)

    def _limit_of(self, u: Utterance) -> int:
        lim = u.max_length if u.max_length is not None else (u.options or self.options).max_length
        return max(int(lim), int(self.options.max_length), self._frame_budget)

    def close(self):
        if getattr(self, "_h", None):
            lib.q3_session_free(self._h); self._h = None

    __del__ = close

    def replace(self, b: int, utt: Utterance):
        """Continuous batching (q3_session_replace): row b — normally a finished one whose codes / PCM have been fetched — starts
        over with `utt` at its frame 0 while the other rows keep going; they are bit-for-bit unaffected."""
        r = CRequest()
        keep = []                                   # the C side copies the request: nothing to keep beyond the call
        fill_request(r, utt, self.options, keep)
        check(lib.q3_session_replace(self._h, int(b), ctypes.byref(r)))
        del keep
        if b < len(self._row_limit):
            self._row_limit[b] = self._limit_of(utt)
        self._ref_frames[b] = 0 if utt.ref_codes is None else int(np.asarray(utt.ref_codes).reshape(-1, 16).shape[0])

    def next_chunk_row(self, b: int) -> Tuple[Optional[AudioBuffer], bool]:
        """StreamingSession::next_chunk for row b of a multi-sequence session: (chunk or None, done)."""
        spf = self.model.config.samples_per_frame
        buf = np.zeros(max(self.options.chunk_frames, 1) * spf, dtype=np.float32)
        n = ctypes.c_size_t(); d = ctypes.c_int()
        check(lib.q3_session_next_chunk_row(self._h, int(b), buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(n), ctypes.byref(d)))
        return (AudioBuffer(buf[:n.value].copy()) if n.value else None), bool(d.value)

    def prefill(self):
        check(lib.q3_session_prefill(self._h))

    def generate(self, n_frames: int, use_graph: bool = True):
        check(lib.q3_session_generate(self._h, int(n_frames), 1 if use_graph else 0))

    def frames(self, b: int = 0) -> Tuple[int, bool]:
        n = ctypes.c_int(); d = ctypes.c_int()
        check(lib.q3_session_frames(self._h, b, ctypes.byref(n), ctypes.byref(d)))
        return n.value, bool(d.value)

    def codes(self, b: int = 0) -> np.ndarray:
        n, _ = self.frames(b)
        out = np.zeros((max(n, 1), 16), dtype=np.uint32)
        got = ctypes.c_int()
        check(lib.q3_session_codes(self._h, b, out.ctypes.data_as(ctypes.c_void_p), out.shape[0], ctypes.byref(got)))
        return out[:got.value]

    def decode(self, b: int = 0, f0: int = 0, f1: Optional[int] = None) -> np.ndarray:
        n, _ = self.frames(b)
        f1 = n if f1 is None else f1
        spf = self.model.config.samples_per_frame
        extra = 0 if self._ref_frames is None else self._ref_frames[b]
        out = np.zeros(max((f1 - f0 + extra) * spf, 1), dtype=np.float32)
        got = ctypes.c_size_t()
        check(lib.q3_session_decode(self._h, b, f0, f1, out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(got)))
        return out[:got.value]

    def run(self, use_graph: bool = True) -> Tuple[List[AudioBuffer], SynthesisTiming]:
        """synthesize_with_timing (lib.rs:425-501) for the batch."""
        spf = self.model.config.samples_per_frame
        capl = [(self._row_limit[i] + self._ref_frames[i]) * spf for i in range(self.B)]      # a row's own limit may exceed the session's
        bufs = [np.zeros(c, dtype=np.float32) for c in capl]
        ptrs = (ctypes.c_void_p * self.B)(*[b.ctypes.data_as(ctypes.c_void_p) for b in bufs])
        caps = (ctypes.c_size_t * self.B)(*capl)
        ns = (ctypes.c_size_t * self.B)()
        t = CTiming()
        check(lib.q3_session_run(self._h, 1 if use_graph else 0, ptrs, caps, ns, ctypes.byref(t)))
        audio = [AudioBuffer(bufs[i][:ns[i]].copy()) for i in range(self.B)]
        return audio, SynthesisTiming(t.prefill_ms, t.generation_ms, t.generation_frames, t.decode_ms)

    def run_timing_only(self, use_graph: bool = True, pcm_out=None) -> SynthesisTiming:
        """`run` without building AudioBuffers. pcm_out = None: the samples stay in HBM; pcm_out = [(address, capacity in
        samples)] per row (e.g. rows of one pinned host tensor, reused step after step): every row's samples are copied to
        the host inside the call, as `synthesize` hands them back (lib.rs:718-784)."""
        ns = (ctypes.c_size_t * self.B)()
        t = CTiming()
        ptrs = caps = None
        if pcm_out is not None:
            assert len(pcm_out) == self.B
            ptrs = (ctypes.c_void_p * self.B)(*[int(a) for a, _ in pcm_out])
            caps = (ctypes.c_size_t * self.B)(*[int(c) for _, c in pcm_out])
        check(lib.q3_session_run(self._h, 1 if use_graph else 0, ptrs, caps, ns, ctypes.byref(t)))
        self.last_samples = [int(x) for x in ns]
        return SynthesisTiming(t.prefill_ms, t.generation_ms, t.generation_frames, t.decode_ms)

    # ---- stage taps (parity tests) ----
    def prefill_len(self, b: int = 0) -> Tuple[int, int]:
        p = ctypes.c_int(); t = ctypes.c_int()
        check(lib.q3_session_prefill_len(self._h, b, ctypes.byref(p), ctypes.byref(t)))
        return p.value, t.value

    def get(self, what: int, shape, b: int = 0, dtype=np.float32) -> np.ndarray:
        out = np.zeros(shape, dtype=dtype)
        check(lib.q3_session_get(self._h, what, b, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        return out

    def talker_step(self, embeds: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        cfg = self.model.config
        e = np.ascontiguousarray(embeds, dtype=np.float32).reshape(self.B, cfg.hidden)
        hid = np.zeros((self.B, cfg.hidden), dtype=np.float32); lg = np.zeros((self.B, cfg.codec_vocab), dtype=np.float32)
        check(lib.q3_talker_step(self._h, e.ctypes.data_as(ctypes.c_void_p), hid.ctypes.data_as(ctypes.c_void_p),
                                 lg.ctypes.data_as(ctypes.c_void_p)))
        return hid, lg

    def cp_generate(self, last_hidden: np.ndarray, sem_embed: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        cfg = self.model.config
        lh = np.ascontiguousarray(last_hidden, dtype=np.float32).reshape(self.B, cfg.hidden)
        se = np.ascontiguousarray(sem_embed, dtype=np.float32).reshape(self.B, cfg.hidden)
        codes = np.zeros((self.B, 15), dtype=np.uint32); lg = np.zeros((self.B, 15, cfg.cp_vocab), dtype=np.float32)
        check(lib.q3_cp_generate(self._h, lh.ctypes.data_as(ctypes.c_void_p), se.ctypes.data_as(ctypes.c_void_p),
                                 codes.ctypes.data_as(ctypes.c_void_p), lg.ctypes.data_as(ctypes.c_void_p)))
        return codes, lg

    def set_profile(self, on: bool):
        check(lib.q3_session_set_profile(self._h, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> Tuple[float, float, int]:
        ms = ctypes.c_double(); by = ctypes.c_double(); n = ctypes.c_long()
        check(lib.q3_session_profile_read(self._h, ctypes.byref(ms), ctypes.byref(by), ctypes.byref(n), 1 if reset else 0))
        return ms.value, by.value, n.value

    def profile_shapes(self, reset: bool = True) -> List[Tuple[int, ...]]:
        """distinct GEMV launches since set_profile(True): (M, N, K, epi, rms 0/1, reserved 0, tiled, count)"""
        n = ctypes.c_int()
        check(lib.q3_session_profile_shapes(self._h, None, 0, ctypes.byref(n), 0))
        buf = (ctypes.c_int * (8 * max(n.value, 1)))()
        check(lib.q3_session_profile_shapes(self._h, buf, n.value, ctypes.byref(n), 1 if reset else 0))
        return [tuple(buf[i * 8:(i + 1) * 8]) for i in range(n.value)]

    def submit_info(self) -> Tuple[int, int]:
        """(path, packets per frame): 0 nothing captured / eager, 1 hipGraphLaunch, 2 own AQL queue with HIP's fences, 4 own AQL
        queue with fence-free boundaries between the write-through kernels (the default), 3 own queue without any fence (probe)
        (include/q3tts.h: q3_session_submit_info; environment Q3_AQL)."""
        p = ctypes.c_int(); n = ctypes.c_int()
        check(lib.q3_session_submit_info(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def submit_fences(self) -> Tuple[int, int]:
        """(packets per frame without their acquire fence, without their release fence): q3_session_submit_fences"""
        a = ctypes.c_int(); r = ctypes.c_int()
        check(lib.q3_session_submit_fences(self._h, ctypes.byref(a), ctypes.byref(r)))
        return a.value, r.value

    def frame_bytes(self, kv_len: int) -> Tuple[float, float]:
        w = ctypes.c_double(); k = ctypes.c_double()
        check(lib.q3_session_frame_bytes(self._h, kv_len, ctypes.byref(w), ctypes.byref(k)))
        return w.value, k.value


class Batcher:
    """Continuous batcher (q3_batcher): requests of any prompt kind, length and options queue up and run through the rows of
    one session; `step` fills free rows, runs a few frames of the shared frame graph and collects finished rows. The
    scheduling loop is native — this class only marshals requests and results."""
    QUEUED, RUNNING, DONE, FAILED = 0, 1, 2, 3

    def __init__(self, model: "Qwen3TTS", slots: int = 8, frame_budget: int = 2048, prompt_budget: int = 0,
                 options: Optional[SynthesisOptions] = None):
        """slots: rows of the session; frame_budget: largest max_length a request may ask for; prompt_budget: prefill
        positions a row can hold (at least 16 = CustomVoice / x-vector prompts; VoiceDesign: instruct length + 16; ICL:
        reference frames + 16); options: defaults for utterances without their own."""
        self.model = model
        self.options = options or SynthesisOptions()
        h = ctypes.c_void_p()
        check(lib.q3_batcher_create(model._h, int(slots), int(frame_budget), int(prompt_budget), ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.q3_batcher_free(self._h); self._h = None

    __del__ = close

    def submit(self, utt: Utterance, want_pcm: bool = True) -> int:
        """Queue one request; returns its ticket. The request is copied by the library."""
        keep = []                                   # alive until the call returns: the library copies the request
        r = CRequest(); fill_request(r, utt, self.options, keep)
        t = ctypes.c_int64()
        check(lib.q3_batcher_submit(self._h, ctypes.byref(r), 1 if want_pcm else 0, ctypes.byref(t)))
        return int(t.value)

    def step(self, n_frames: int = 32, use_graph: bool = True) -> Tuple[int, int, int]:
        """One scheduling round: (rows running, requests queued, tickets finished in this call)."""
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.q3_batcher_step(self._h, int(n_frames), 1 if use_graph else 0, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def poll(self, ticket: int) -> Tuple[int, int, int]:
        st, n, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        check(lib.q3_batcher_poll(self._h, int(ticket), ctypes.byref(st), ctypes.byref(n), ctypes.byref(ns)))
        return st.value, n.value, ns.value

    def fetch(self, ticket: int) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """(codes [n][16] u32, PCM or None) of a finished ticket, which is released; a failed ticket raises its error."""
        st, n, ns = self.poll(ticket)
        codes = np.zeros((n, 16), np.uint32); pcm = np.zeros(ns, np.float32)
        check(lib.q3_batcher_fetch(self._h, int(ticket), codes.ctypes.data_as(ctypes.c_void_p), n,
                                   pcm.ctypes.data_as(ctypes.c_void_p) if ns else None, ns))
        return codes, (pcm if ns else None)

    def run_all(self, utts: Sequence[Utterance], want_pcm: bool = True, poll_frames: int = 32, use_graph: bool = True):
        """Submit everything, step until the queue is drained; results in request order (a failed request raises)."""
        tickets = [self.submit(u, want_pcm) for u in utts]
        while True:
            running, queued, _ = self.step(poll_frames, use_graph)
            if running == 0 and queued == 0:
                break
        return [self.fetch(t) for t in tickets]


class StreamingSession:
    """StreamingSession (lib.rs:1484-1782): iterate to receive AudioBuffer chunks of up to
    chunk_frames*1920 samples; each chunk is decoded as an independent utterance (lib.rs:1755-1758)."""

    def __init__(self, model: "Qwen3TTS", utt: Utterance, options: SynthesisOptions, continuous: bool = False):
        """continuous=True: chunks are decoded with left context and concatenate to exactly the non-streaming audio
        (q3_session_set_stream_mode 1); the default reproduces the reference's context-free chunk decode."""
        self._s = Session(model, [utt], options)
        if continuous:
            check(lib.q3_session_set_stream_mode(self._s._h, 1))
        self._done = False
        self._spf = model.config.samples_per_frame
        self._chunk = options.chunk_frames

    def next_chunk(self) -> Optional[AudioBuffer]:
        if self._done:
            return None
        buf = np.zeros(self._chunk * self._spf, dtype=np.float32)
        n = ctypes.c_size_t(); d = ctypes.c_int()
        check(lib.q3_session_next_chunk(self._s._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(n), ctypes.byref(d)))
        if d.value:
            self._done = True
        if n.value == 0:
            return None
        return AudioBuffer(buf[:n.value].copy())

    def frames_generated(self) -> int:
        return self._s.frames(0)[0]

    def is_done(self) -> bool:
        return self._done

    def __iter__(self):
        return self

    def __next__(self) -> AudioBuffer:
        c = self.next_chunk()
        if c is None:
            raise StopIteration
        return c


class ModelType(enum.Enum):    # config.rs:176-194
    Base = 0
    CustomVoice = 1
    VoiceDesign = 2


class Qwen3TTS:
    """Qwen3TTS facade (lib.rs:154-173). Construct with `from_pretrained` (model directory: config.json +
    safetensors, lib.rs:180-262), `from_tensors` (name → array, the from_weights path, lib.rs:267) or
    `from_synthetic`."""

    def __init__(self, config: Q3Config, device: int = 0, _handle=None):
        self.config = config
        self.device_index = device
        self.model_type: Optional[ModelType] = None     # None = loaded without config.json (lib.rs:383-386)
        if _handle is not None:
            self._h = _handle
            return
        h = ctypes.c_void_p()
        c = config.to_c()
        check(lib.q3_model_create(ctypes.byref(c), device, ctypes.byref(h)))
        self._h = h

    @classmethod
    def from_pretrained(cls, model_dir: str, device: int = 0) -> "Qwen3TTS":
        """Load `<dir>/config.json`, `<dir>/model.safetensors` and `<dir>/speech_tokenizer/model.safetensors`
        (lib.rs:180-262) through the C++ loader (q3_model_load). Tokenization stays with the caller."""
        h = ctypes.c_void_p(); mt = ctypes.c_int(-1)
        check(lib.q3_model_load(str(model_dir).encode(), device, ctypes.byref(h), ctypes.byref(mt)))
        from .config import CConfig
        c = CConfig()
        check(lib.q3_model_config(h, ctypes.byref(c)))
        m = cls(Q3Config.from_c(c), device, _handle=h)
        m.model_type = ModelType(mt.value) if mt.value >= 0 else None
        # Base checkpoints carry `speaker_encoder.*` (lib.rs:233-249): attach the encoder when the keys are there
        import os
        from .speaker import SpeakerEncoder, SpeakerEncoderConfig
        st = os.path.join(str(model_dir), "model.safetensors")
        probe = ctypes.c_int()
        if device >= 0 and lib.q3_safetensors_info(st.encode(), b"speaker_encoder.fc.weight", ctypes.byref(probe), None, 0, None) == 0:
            cfg_path = os.path.join(str(model_dir), "config.json")
            scfg = SpeakerEncoderConfig.from_json(cfg_path)[0] if os.path.exists(cfg_path) else SpeakerEncoderConfig(enc_dim=m.config.hidden)
            m.attach_speaker_encoder(SpeakerEncoder.from_safetensors(st, scfg, device))
        # the speech tokenizer file carries the Mimi encoder under `encoder.*` (lib.rs:251-258, encoder_12hz.rs:54-71)
        from .speech_encoder import SpeechEncoder
        tok = os.path.join(str(model_dir), "speech_tokenizer", "model.safetensors")
        if not os.path.exists(tok):
            tok = os.path.join(os.path.dirname(os.path.abspath(str(model_dir))), "speech_tokenizer", "model.safetensors")
        if device >= 0 and os.path.exists(tok) and \
                lib.q3_safetensors_info(tok.encode(), b"encoder.downsample.conv.weight", ctypes.byref(probe), None, 0, None) == 0:
            try:        # non-fatal, as try_load_speech_encoder (lib.rs:1362-1388): without it only ICL cloning is unavailable
                m.attach_speech_encoder(SpeechEncoder.from_safetensors(tok, None, device))
            except _lib.Q3Error:
                pass
        return m

    # ---- voice cloning front end (lib.rs:1049-1190) ----
    speaker_encoder = None       # SpeakerEncoder, attached by from_pretrained for Base checkpoints / attach_speaker_encoder

    def attach_speaker_encoder(self, enc):
        if enc.config.enc_dim != self.config.hidden:
            raise ValueError(f"speaker embedding dim {enc.config.enc_dim} != talker hidden size {self.config.hidden}")
        self.speaker_encoder = enc

    def supports_voice_cloning(self) -> bool:          # lib.rs:390-396
        return self.model_type in (None, ModelType.Base)

    def has_speaker_encoder(self) -> bool:
        return self.speaker_encoder is not None

    speech_encoder = None        # SpeechEncoder (Mimi), attached by from_pretrained when speech_tokenizer/model.safetensors has encoder.* keys

    def attach_speech_encoder(self, enc):
        self.speech_encoder = enc

    def has_speech_encoder(self) -> bool:              # lib.rs:1049-1051
        return self.speech_encoder is not None

    def create_voice_clone_prompt(self, ref_audio: "AudioBuffer", ref_text_ids=None, ref_codes=None):
        """create_voice_clone_prompt (lib.rs:1132-1190). x_vector_only when `ref_text_ids` is None; with a transcript the
        reference audio is also encoded to codec frames by the speech encoder (Encoder12Hz, encoder_12hz.rs:119-144) for
        ICL. `ref_codes` (optional) overrides that encoder with frames computed elsewhere."""
        from .speaker import VoiceClonePrompt
        if self.speaker_encoder is None:
            hint = {ModelType.CustomVoice: " CustomVoice models use preset speakers (synthesize_with_voice), not voice cloning. "
                                           "Use a Base model for voice cloning.",
                    ModelType.VoiceDesign: " VoiceDesign models use text-described voices, not voice cloning. "
                                           "Use a Base model for voice cloning."}.get(
                self.model_type, " Ensure model weights contain `speaker_encoder.*` keys (only Base models include a speaker encoder).")
            raise _lib.Q3Error(3, "Speaker encoder not available." + hint)
        if ref_audio.sample_rate != 24000:        # both encoders assume 24 kHz input (lib.rs:1156-1166)
            ref_audio = resample_to_24k(ref_audio)
        emb = self.speaker_encoder.encode(ref_audio.samples, ref_audio.sample_rate)
        if ref_text_ids is None:
            return VoiceClonePrompt(emb)
        if ref_codes is None:
            if self.speech_encoder is None:         # lib.rs:1172-1178
                raise _lib.Q3Error(7, "ICL voice cloning requires a speech encoder, but it was not loaded. Ensure the speech tokenizer "
                                      "weights contain encoder keys, or use x_vector_only mode by passing ref_text=None.")
            ref_codes = self.speech_encoder.encode(ref_audio.samples, ref_audio.sample_rate)      # [T_frames, 16]
        return VoiceClonePrompt(emb, np.ascontiguousarray(ref_codes, dtype=np.uint32), np.asarray(ref_text_ids, dtype=np.uint32))

    def synthesize_voice_clone_prompt(self, text_ids, prompt, language: "Language", options=None):
        """synthesize_voice_clone (lib.rs:1202-1262) with a VoiceClonePrompt."""
        return self.synthesize_voice_clone(text_ids, prompt.speaker_embedding, language, options,
                                           ref_codes=prompt.ref_codes, ref_text_ids=prompt.ref_text_ids)

    def supports_preset_speakers(self) -> bool:        # lib.rs:398-404 (permissive when unknown)
        return self.model_type in (None, ModelType.CustomVoice)

    def supports_voice_design(self) -> bool:           # lib.rs:409-411
        return self.model_type == ModelType.VoiceDesign

    def close(self):
        if getattr(self, "_h", None):
            lib.q3_model_free(self._h); self._h = None

    __del__ = close

    def set_tensor(self, name: str, arr: np.ndarray, dtype: int):
        a = np.ascontiguousarray(arr)
        check(lib.q3_model_set_tensor(self._h, name.encode(), dtype, a.ctypes.data_as(ctypes.c_void_p), a.size))

    def finalize(self):
        check(lib.q3_model_finalize(self._h))

    @classmethod
    def from_synthetic(cls, config: Q3Config, device: int = 0, seed: int = synth.DEFAULT_SEED, sink=None) -> "Qwen3TTS":
        """`sink(name, array, dtype)` (optional) also receives every tensor — used by tests to feed the
        CPU oracle the identical checkpoint."""
        m = cls(config, device)
        for name, arr, dt in synth.synthetic_checkpoint(config, m._h, seed):
            m.set_tensor(name, arr, dt)
            if sink is not None:
                sink(name, arr, dt)
        m.finalize()
        return m

    @classmethod
    def from_tensors(cls, config: Q3Config, tensors, device: int = 0) -> "Qwen3TTS":
        m = cls(config, device)
        for name, (arr, dt) in tensors.items():
            m.set_tensor(name, arr, dt)
        m.finalize()
        return m

    def kv_pool_limit(self, max_pages: int):
        """Cap of the model's KV page pool (q3_model_kv_pool_limit; 0 = HBM is the limit). A session that needs a page beyond
        it fails with Q3_KV_OVERFLOW — the reference's cache-overflow bail (kv_cache.rs:293-300)."""
        check(lib.q3_model_kv_pool_limit(self._h, int(max_pages)))

    def set_codec_planes(self, planes: int):
        """bf16 planes per f32 operand in the vocoder's matrix-core convs (q3_model_set_codec_planes): 3 = exact f32 products
        (default), 2 = hi + mid planes only (PCM within 1e-4 RMS of the reference instead of 2.5e-5; 30 % less vocoder time)."""
        check(lib.q3_model_set_codec_planes(self._h, int(planes)))

    def kv_pool_trim(self) -> int:
        """Slabs of the KV page pool that hold no page in use go back to the device (q3_model_kv_pool_trim); bytes freed."""
        n = ctypes.c_size_t()
        check(lib.q3_model_kv_pool_trim(self._h, ctypes.byref(n)))
        return n.value

    def kv_pool_info(self) -> dict:
        """Page geometry and occupancy of the model's KV pool in f32-equivalent pages — a bf16 session's page counts half
        (q3_model_kv_pool_info)."""
        pp = ctypes.c_int(); pb = ctypes.c_size_t(); tot = ctypes.c_int(); use = ctypes.c_int(); peak = ctypes.c_int()
        check(lib.q3_model_kv_pool_info(self._h, ctypes.byref(pp), ctypes.byref(pb), ctypes.byref(tot), ctypes.byref(use), ctypes.byref(peak)))
        return {"page_positions": pp.value, "page_bytes": pb.value, "pages_total": tot.value, "pages_in_use": use.value, "pages_peak": peak.value}

    def arena(self) -> Tuple[int, int]:
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        check(lib.q3_model_arena(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def mark_loaded(self):
        check(lib.q3_model_mark_loaded(self._h))

    # ---- synthesis API (lib.rs:416-501, 718-870, 1070-1110) ----
    def session(self, utts: Sequence[Utterance], options: Optional[SynthesisOptions] = None, debug: bool = False, kv_bf16: bool = False) -> Session:
        return Session(self, utts, options or SynthesisOptions(), debug, kv_bf16=kv_bf16)

    def synthesize(self, text_ids: Sequence[int], options: Optional[SynthesisOptions] = None) -> AudioBuffer:
        return self.synthesize_with_voice(text_ids, Speaker.Ryan, Language.English, options)

    def synthesize_with_voice(self, text_ids, speaker: Speaker, language: Language, options=None) -> AudioBuffer:
        audio, _ = self.synthesize_with_timing(text_ids, speaker, language, options)
        return audio

    def synthesize_with_timing(self, text_ids, speaker: Speaker, language: Language, options=None):
        s = self.session([Utterance(text_ids, speaker, language)], options)
        try:
            audio, timing = s.run()
        finally:
            s.close()
        return audio[0], timing

    def synthesize_voice_design(self, text_ids, instruct_ids, language: Language, options=None) -> AudioBuffer:
        s = self.session([Utterance(text_ids, language=language, instruct_ids=instruct_ids)], options)
        try:
            audio, _ = s.run()
        finally:
            s.close()
        return audio[0]

    def synthesize_voice_clone(self, text_ids, xvector, language: Language, options=None, ref_codes=None, ref_text_ids=None):
        """synthesize_voice_clone_debug (lib.rs:897-1046): x-vector-only, or ICL when ref_codes + ref_text_ids are given.
        Returns (AudioBuffer, codes)."""
        s = self.session([Utterance(text_ids, language=language, xvector=xvector, ref_codes=ref_codes, ref_text_ids=ref_text_ids)], options)
        try:
            s.prefill(); s.generate(max(s._row_limit))
            return AudioBuffer(s.decode(0)), s.codes(0)
        finally:
            s.close()

    @staticmethod
    def prefill_shape(u: "Utterance") -> Tuple[int, int, int, int]:
        """What fixes an utterance's prefill length: rows of one shape are prefilled together (q3_session_create groups a
        ragged batch by it)."""
        icl = u.ref_codes is not None and u.ref_text_ids is not None
        n_ins = len(u.instruct_ids) if u.instruct_ids is not None else 0
        n_icl = (len(np.asarray(u.ref_codes).reshape(-1, 16)) + 1) if icl else 0
        return (u.mode(), n_ins, 1 if (len(u.text_ids) > 0 and not icl) else 0, n_icl)

    def synthesize_batch(self, utts: Sequence[Utterance], options=None):
        """Batch of arbitrary requests (any mix of prompt kinds and lengths): up to Q3_MAX_BATCH (64) of them per session — a
        session prefills rows of equal prefill length together and decodes all rows in one captured frame graph
        (q3_session_create on a ragged batch) —, results in request order. The timing is the sum over the sessions."""
        audio: List[Optional[AudioBuffer]] = [None] * len(utts)
        tot = SynthesisTiming(0.0, 0.0, 0, 0.0)
        for k in range(0, len(utts), MAX_BATCH):
            part = list(range(k, min(k + MAX_BATCH, len(utts))))
            s = self.session([utts[i] for i in part], options)
            try:
                a, t = s.run()
            finally:
                s.close()
            for j, i in enumerate(part):
                audio[i] = a[j]
            tot = SynthesisTiming(tot.prefill_ms + t.prefill_ms, tot.generation_ms + t.generation_ms,
                                  tot.generation_frames + t.generation_frames, tot.decode_ms + t.decode_ms)
        return audio, tot

    def synthesize_continuous(self, utts: Sequence[Utterance], options=None, slots: int = 8, poll_frames: int = 32,
                              decode: bool = True, use_graph: bool = True):
        """Continuous batching over ONE session of `slots` rows: the first `slots` requests start together; whenever a row ends
        (EOS, or its own max_length) its codes are collected and the next waiting request is swapped into the row
        (q3_session_replace) while the other rows keep running — no row idles until the slowest one is done. The requests
        may be of any prompt kind and length (a session prefills rows of equal prefill length together). Returns (codes per request, PCM per request or None, frames generated, wall seconds
        of the generation loop)."""
        import time as _time
        o = options or SynthesisOptions()
        utts = list(utts)
        n0 = min(slots, len(utts))
        budget = max((u.max_length if u.max_length is not None else (u.options or o).max_length) for u in utts)
        # capacity for the longest request and the longest prompt, whenever they arrive (prompt positions: instruct + role /
        # codec overlay rows + ICL reference frames; 16 covers the fixed part of every prompt kind, talker.rs:451-627)
        prompt = max((0 if u.instruct_ids is None else len(u.instruct_ids)) + (0 if u.ref_codes is None else int(np.asarray(u.ref_codes).reshape(-1, 16).shape[0])) + 16 for u in utts)
        s = Session(self, utts[:n0], o, frame_budget=budget, prompt_budget=prompt)
        owner = list(range(n0)); nxt = n0
        codes: List[Optional[np.ndarray]] = [None] * len(utts); pcm: List[Optional[np.ndarray]] = [None] * len(utts)
        frames = 0
        t0 = _time.perf_counter()
        try:
            s.prefill()
            while any(i is not None for i in owner):
                s.generate(poll_frames, use_graph=use_graph)
                for b in range(n0):
                    i = owner[b]
                    if i is None:
                        continue
                    n, done = s.frames(b)
                    if not done:
                        continue
                    codes[i] = s.codes(b); frames += int(codes[i].shape[0])
                    if decode:
                        pcm[i] = s.decode(b)
                    if nxt < len(utts):
                        s.replace(b, utts[nxt]); owner[b] = nxt; nxt += 1
                    else:
                        owner[b] = None
            wall = _time.perf_counter() - t0
        finally:
            s.close()
        return codes, pcm, frames, wall

    def synthesize_streaming(self, text_ids, speaker: Speaker, language: Language, options=None,
                             continuous: bool = False) -> StreamingSession:
        return StreamingSession(self, Utterance(text_ids, speaker, language), options or SynthesisOptions(), continuous)

    def synthesize_voice_design_streaming(self, text_ids, instruct_ids, language: Language, options=None,
                                          continuous: bool = False) -> StreamingSession:
        """synthesize_voice_design_streaming (lib.rs:1095-1128)."""
        return StreamingSession(self, Utterance(text_ids, language=language, instruct_ids=instruct_ids), options or SynthesisOptions(), continuous)

    def synthesize_voice_clone_debug(self, text_ids, prompt, language: Language, options=None):
        """synthesize_voice_clone_debug (lib.rs:897-1046): (AudioBuffer, FrameCodes) for a VoiceClonePrompt."""
        return self.synthesize_voice_clone(text_ids, prompt.speaker_embedding, language, options,
                                           ref_codes=prompt.ref_codes, ref_text_ids=prompt.ref_text_ids)

    def device(self) -> str:                            # lib.rs:1325-1327
        return f"hip:{self.device_index}"

    @classmethod
    def from_pretrained_with_tokenizer(cls, model_dir: str, tokenizer_dir: Optional[str], device: int = 0):
        """from_pretrained_with_tokenizer (lib.rs:192-262): (model, TextTokenizer) — tokenization stays on the host side."""
        from .text import TextTokenizer
        return cls.from_pretrained(model_dir, device), TextTokenizer.from_pretrained(model_dir, tokenizer_dir)

    def decode_codes(self, codes: np.ndarray, taps=None) -> AudioBuffer:
        """decode_codes (lib.rs:881-890): codes [n][16] u32."""
        c = np.ascontiguousarray(codes, dtype=np.uint32).reshape(-1, 16)
        out = np.zeros(c.shape[0] * self.config.samples_per_frame, dtype=np.float32)
        tp = None
        if taps is not None:
            tp = (ctypes.c_void_p * 10)(*[t.ctypes.data_as(ctypes.c_void_p) if t is not None else None for t in taps])
        check(lib.q3_decode_codes(self._h, c.ctypes.data_as(ctypes.c_void_p), c.shape[0], out.ctypes.data_as(ctypes.c_void_p), tp))
        return AudioBuffer(out)

    def frame_embed(self, sem_token: int, codes15, text_add: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(codes15, dtype=np.uint32); t = np.ascontiguousarray(text_add, dtype=np.float32)
        out = np.zeros(self.config.hidden, dtype=np.float32)
        check(lib.q3_frame_embed(self._h, int(sem_token), c.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p),
                                 out.ctypes.data_as(ctypes.c_void_p)))
        return out


def pcm16(samples: np.ndarray) -> np.ndarray:
    """`(clamp(x, -1, 1) * 32767) as i16` (audio/io.rs:158-160)."""
    a = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    out = np.empty(a.size, dtype=np.int16)
    check(lib.q3_pcm16_from_f32(a.ctypes.data_as(ctypes.c_void_p), a.size, out.ctypes.data_as(ctypes.c_void_p)))
    return out


def save_wav(path: str, samples: np.ndarray, sample_rate: int = 24000):
    """save_wav (audio/io.rs:143-165): mono PCM16."""
    a = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    check(lib.q3_wav_write_pcm16(str(path).encode(), a.ctypes.data_as(ctypes.c_void_p), a.size, int(sample_rate)))


def load_wav(path: str) -> AudioBuffer:
    """load_wav (audio/io.rs:106-141): PCM/float WAV → mono f32."""
    n = ctypes.c_int64(); rate = ctypes.c_uint32()
    check(lib.q3_wav_read(str(path).encode(), None, 0, ctypes.byref(n), ctypes.byref(rate)))
    out = np.empty(n.value, dtype=np.float32)
    check(lib.q3_wav_read(str(path).encode(), out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(n), ctypes.byref(rate)))
    return AudioBuffer(out, int(rate.value))


def resample(audio: AudioBuffer, target_rate: int) -> AudioBuffer:
    """audio::resample (audio/resample.rs:164-171): windowed-sinc resampling (q3_resample); a copy when rates match."""
    x = np.ascontiguousarray(audio.samples, dtype=np.float32)
    n = ctypes.c_int64()
    check(lib.q3_resample(x.ctypes.data_as(ctypes.c_void_p), x.size, audio.sample_rate, target_rate, None, 0, ctypes.byref(n)))
    out = np.empty(n.value, np.float32)
    check(lib.q3_resample(x.ctypes.data_as(ctypes.c_void_p), x.size, audio.sample_rate, target_rate,
                          out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(n)))
    return AudioBuffer(out, target_rate)


def resample_to_24k(audio: AudioBuffer) -> AudioBuffer:      # audio/resample.rs:174-176
    return resample(audio, 24000)


def save_codes_binary(path: str, codes: np.ndarray):
    """save_codes_binary (bin/generate_audio.rs:788-801): [n_frames][16] codes as i64 LE, frame-major."""
    a = np.ascontiguousarray(codes, dtype=np.uint32)
    assert a.ndim == 2
    check(lib.q3_codes_write_bin(str(path).encode(), a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1]))


def load_codes_binary(path: str, n_groups: int = 16) -> np.ndarray:
    n = ctypes.c_int()
    check(lib.q3_codes_read_bin(str(path).encode(), None, 0, n_groups, ctypes.byref(n)))
    out = np.empty((n.value, n_groups), dtype=np.uint32)
    check(lib.q3_codes_read_bin(str(path).encode(), out.ctypes.data_as(ctypes.c_void_p), n.value, n_groups, ctypes.byref(n)))
    return out


def save_audio_binary(path: str, samples: np.ndarray):
    """save_audio_binary (bin/generate_audio.rs:804-813): f32 LE."""
    a = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    check(lib.q3_audio_write_bin(str(path).encode(), a.ctypes.data_as(ctypes.c_void_p), a.size))


def load_audio_binary(path: str) -> np.ndarray:
    """the reader of save_audio_binary's format (generate_audio.rs:880-886): f32 LE samples."""
    n = ctypes.c_int64()
    check(lib.q3_audio_read_bin(str(path).encode(), None, 0, ctypes.byref(n)))
    out = np.empty(n.value, dtype=np.float32)
    if n.value:
        check(lib.q3_audio_read_bin(str(path).encode(), out.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n)))
    return out


def codes_to_tensor(codes: np.ndarray) -> np.ndarray:
    """codes_to_tensor (lib.rs:1417-1431): [n][16] u32 → [1][16][n] i64."""
    c = np.ascontiguousarray(codes, dtype=np.uint32).reshape(-1, 16)
    out = np.zeros((16, c.shape[0]), dtype=np.int64)
    lib.q3_codes_to_tensor(c.ctypes.data_as(ctypes.c_void_p), c.shape[0], out.ctypes.data_as(ctypes.c_void_p))
    return out.reshape(1, 16, c.shape[0])


# ---- standalone ops (parity tests) ----
def fused_residual_rmsnorm(x: np.ndarray, res: np.ndarray, w: np.ndarray, eps: float, device: int = 0):
    """FusedRmsNorm::forward_residual (fused_ops.rs:49-96) on the GPU. f32 arrays, or uint16 = bf16 bits."""
    rows, cols = x.shape
    dt = 1 if x.dtype == np.uint16 else 0
    x = np.ascontiguousarray(x); res = np.ascontiguousarray(res); w = np.ascontiguousarray(w)
    normed = np.zeros_like(x); summ = np.zeros_like(x)
    check(lib.q3_fused_residual_rmsnorm(device, dt, x.ctypes.data_as(ctypes.c_void_p), res.ctypes.data_as(ctypes.c_void_p),
                                        w.ctypes.data_as(ctypes.c_void_p), rows, cols, eps,
                                        normed.ctypes.data_as(ctypes.c_void_p), summ.ctypes.data_as(ctypes.c_void_p)))
    return normed, summ


def linear(x: np.ndarray, w_bf16: np.ndarray, bias: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
    M, K = x.shape; N = w_bf16.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float32); w = np.ascontiguousarray(w_bf16, dtype=np.uint16)
    y = np.zeros((M, N), dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    check(lib.q3_linear(device, x.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p),
                        None if b is None else b.ctypes.data_as(ctypes.c_void_p), M, N, K, y.ctypes.data_as(ctypes.c_void_p)))
    return y


def sample(logits: np.ndarray, u: np.ndarray, options: SynthesisOptions, seen: Optional[np.ndarray] = None,
           token_count: int = -1, device: int = 0) -> np.ndarray:
    """apply_generation_penalties + sample on the GPU (token_count < 0: plain `sample`, sampling.rs:140)."""
    lg = np.ascontiguousarray(logits, dtype=np.float32); rows, vocab = lg.shape
    uu = np.ascontiguousarray(u, dtype=np.float32)
    out = np.zeros(rows, dtype=np.uint32)
    o = options.to_c()
    sn = None if seen is None else np.ascontiguousarray(seen, dtype=np.uint8)
    check(lib.q3_sample(device, lg.ctypes.data_as(ctypes.c_void_p), None if sn is None else sn.ctypes.data_as(ctypes.c_void_p),
                        uu.ctypes.data_as(ctypes.c_void_p), rows, vocab, ctypes.byref(o), token_count,
                        out.ctypes.data_as(ctypes.c_void_p)))
    return out


def bench_linear(M: int, N: int, K: int, epi: int = 0, rms=False, tiled: int = -1, iters: int = 200,
                 n_copies: int = 0, device: int = 0) -> float:
    """µs per launch of one GEMV shape: mean over 5 hipGraph replays of `iters` launches cycling over HBM-resident weight
    copies. rms: fused input RMSNorm (norm weight applied in the kernel) or none."""
    nbytes = N * K * 2 * (2 if epi == 3 else 1)
    if n_copies <= 0:
        n_copies = max(2, int(600e6 // nbytes))
    us = ctypes.c_double()
    check(lib.q3_bench_linear(device, M, N, K, epi, 1 if rms else 0, tiled, iters, n_copies, ctypes.byref(us)))
    return us.value


def auto_device() -> int:
    """auto_device (lib.rs:1854-1926): this build has exactly one backend — an MI355X. No CPU path."""
    n = lib.q3_device_count()
    if n <= 0:
        raise RuntimeError("no HIP device visible: the MI355X-native build has no CPU fallback")
    return 0
