"""ctypes binding of libq3tts.so (include/q3tts.h). The product has NO CPU fallback: if the HIP
library is missing this module raises, and every compute entry point fails loudly without a GPU."""
import ctypes
import os

from .config import CConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("Q3TTS_LIB") or os.path.join(_HERE, "libq3tts.so")   # env override: kernel-ablation builds (dev aid)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the gfx950 HIP library is not built. Run "
        "`python -c 'import __graft_entry__ as g; g.build()'` (or qwen3_tts_rs_amd/csrc/build.sh).")

lib = ctypes.CDLL(LIB_PATH)


class COptions(ctypes.Structure):
    _fields_ = [
        ("temperature", ctypes.c_double), ("top_p", ctypes.c_double), ("repetition_penalty", ctypes.c_double),
        ("seed", ctypes.c_uint64), ("max_length", ctypes.c_int32), ("top_k", ctypes.c_int32),
        ("eos_token_id", ctypes.c_int32), ("chunk_frames", ctypes.c_int32), ("min_new_tokens", ctypes.c_int32),
        ("has_seed", ctypes.c_int32),
    ]


class CRequest(ctypes.Structure):
    _fields_ = [
        ("mode", ctypes.c_int32),
        ("text_ids", ctypes.POINTER(ctypes.c_uint32)), ("n_text", ctypes.c_int32),
        ("instruct_ids", ctypes.POINTER(ctypes.c_uint32)), ("n_instruct", ctypes.c_int32),
        ("speaker_id", ctypes.c_uint32), ("language_id", ctypes.c_uint32),
        ("xvector", ctypes.POINTER(ctypes.c_float)),
        ("opts", COptions),
        ("ref_codes", ctypes.POINTER(ctypes.c_uint32)), ("n_ref", ctypes.c_int32),
        ("ref_text_ids", ctypes.POINTER(ctypes.c_uint32)), ("n_ref_text", ctypes.c_int32),
    ]


class CSpkConfig(ctypes.Structure):      # q3_spk_config (SpeakerEncoderConfig, config.rs:100-174)
    _fields_ = [("mel_dim", ctypes.c_int32), ("enc_dim", ctypes.c_int32), ("channels", ctypes.c_int32 * 5),
                ("kernel_sizes", ctypes.c_int32 * 5), ("dilations", ctypes.c_int32 * 5), ("attention_channels", ctypes.c_int32),
                ("res2net_scale", ctypes.c_int32), ("se_channels", ctypes.c_int32), ("sample_rate", ctypes.c_int32)]


class CMimiConfig(ctypes.Structure):     # q3_mimi_config (mimi::Config::v0_1(Some(16)), encoder_12hz.rs:73)
    _fields_ = [("n_filters", ctypes.c_int32), ("hidden", ctypes.c_int32), ("ratios", ctypes.c_int32 * 4), ("kernel", ctypes.c_int32),
                ("res_kernel", ctypes.c_int32), ("last_kernel", ctypes.c_int32), ("compress", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("n_heads", ctypes.c_int32), ("head_dim", ctypes.c_int32), ("inter", ctypes.c_int32), ("window", ctypes.c_int32),
                ("cb_size", ctypes.c_int32), ("cb_dim", ctypes.c_int32), ("n_q", ctypes.c_int32), ("n_sem", ctypes.c_int32),
                ("norm_eps", ctypes.c_float), ("rope_theta", ctypes.c_float)]


class CTiming(ctypes.Structure):
    _fields_ = [("prefill_ms", ctypes.c_double), ("generation_ms", ctypes.c_double), ("decode_ms", ctypes.c_double),
                ("generation_frames", ctypes.c_int32)]


c_void_p, c_int, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
P = ctypes.POINTER

# every symbol include/q3tts.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "q3_abi_version": (c_int, []),
    "q3_last_error": (c_char_p, []),
    "q3_device_count": (c_int, []),
    "q3_model_create": (c_int, [P(CConfig), c_int, P(c_void_p)]),
    "q3_model_free": (None, [c_void_p]),
    "q3_model_set_tensor": (c_int, [c_void_p, c_char_p, c_int, c_void_p, ctypes.c_int64]),
    "q3_model_n_tensors": (c_int, [c_void_p]),
    "q3_model_tensor_info": (c_int, [c_void_p, c_int, P(c_char_p), P(ctypes.c_int64), P(c_int)]),
    "q3_model_arena": (c_int, [c_void_p, P(c_void_p), P(ctypes.c_size_t)]),
    "q3_model_kv_pool_limit": (c_int, [c_void_p, c_int]),
    "q3_model_set_codec_planes": (c_int, [c_void_p, c_int]),
    "q3_model_kv_pool_trim": (c_int, [c_void_p, P(ctypes.c_size_t)]),
    "q3_model_kv_pool_info": (c_int, [c_void_p, P(c_int), P(ctypes.c_size_t), P(c_int), P(c_int), P(c_int)]),
    "q3_model_mark_loaded": (c_int, [c_void_p]),
    "q3_model_finalize": (c_int, [c_void_p]),
    "q3_synth_fill": (c_int, [ctypes.c_uint64, c_char_p, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int64, c_void_p]),
    "q3_session_create": (c_int, [c_void_p, P(CRequest), c_int, P(c_void_p)]),
    "q3_session_create_reserved": (c_int, [c_void_p, P(CRequest), c_int, c_int, c_int, P(c_void_p)]),
    "q3_session_free": (None, [c_void_p]),
    "q3_session_prefill": (c_int, [c_void_p]),
    "q3_session_generate": (c_int, [c_void_p, c_int, c_int]),
    "q3_session_frames": (c_int, [c_void_p, c_int, P(c_int), P(c_int)]),
    "q3_session_codes": (c_int, [c_void_p, c_int, c_void_p, c_int, P(c_int)]),
    "q3_session_decode": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_size_t, P(ctypes.c_size_t)]),
    "q3_session_set_kv_dtype": (c_int, [c_void_p, c_int]),
    "q3_session_run": (c_int, [c_void_p, c_int, P(c_void_p), P(ctypes.c_size_t), P(ctypes.c_size_t), P(CTiming)]),
    "q3_session_next_chunk": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, P(ctypes.c_size_t), P(c_int)]),
    "q3_session_next_chunk_row": (c_int, [c_void_p, c_int, c_void_p, ctypes.c_size_t, P(ctypes.c_size_t), P(c_int)]),
    "q3_session_replace": (c_int, [c_void_p, c_int, c_void_p]),
    "q3_batcher_create": (c_int, [c_void_p, c_int, c_int, c_int, P(c_void_p)]),
    "q3_batcher_free": (None, [c_void_p]),
    "q3_batcher_submit": (c_int, [c_void_p, c_void_p, c_int, P(ctypes.c_int64)]),
    "q3_batcher_step": (c_int, [c_void_p, c_int, c_int, P(c_int), P(c_int), P(c_int)]),
    "q3_batcher_poll": (c_int, [c_void_p, ctypes.c_int64, P(c_int), P(c_int), P(ctypes.c_size_t)]),
    "q3_batcher_fetch": (c_int, [c_void_p, ctypes.c_int64, c_void_p, c_int, c_void_p, ctypes.c_size_t]),
    "q3_session_set_debug": (c_int, [c_void_p, c_int]),
    "q3_session_prefill_len": (c_int, [c_void_p, c_int, P(c_int), P(c_int)]),
    "q3_session_get": (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.c_size_t]),
    "q3_talker_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "q3_cp_generate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "q3_frame_embed": (c_int, [c_void_p, ctypes.c_uint32, c_void_p, c_void_p, c_void_p]),
    "q3_sample": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, P(COptions), c_int, c_void_p]),
    "q3_rng_seed": (None, [ctypes.c_uint64, P(ctypes.c_uint64)]),
    "q3_rng_next": (ctypes.c_float, [P(ctypes.c_uint64)]),
    "q3_fused_residual_rmsnorm": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_float, c_void_p, c_void_p]),
    "q3_linear": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "q3_decode_codes": (c_int, [c_void_p, c_void_p, c_int, c_void_p, P(c_void_p)]),
    "q3_codes_to_tensor": (None, [c_void_p, c_int, c_void_p]),
    "q3_session_set_profile": (c_int, [c_void_p, c_int]),
    "q3_session_profile_read": (c_int, [c_void_p, P(ctypes.c_double), P(ctypes.c_double), P(ctypes.c_long), c_int]),
    "q3_session_profile_shapes": (c_int, [c_void_p, P(c_int), c_int, P(c_int), c_int]),
    "q3_bench_linear": (c_int, [c_int] * 9 + [P(ctypes.c_double)]),
    "q3_session_stream": (c_int, [c_void_p, P(c_void_p)]),
    "q3_session_frame_bytes": (c_int, [c_void_p, c_int, P(ctypes.c_double), P(ctypes.c_double)]),
    "q3_session_submit_info": (c_int, [c_void_p, P(c_int), P(c_int)]),
    "q3_session_submit_fences": (c_int, [c_void_p, P(c_int), P(c_int)]),
    "q3_session_set_stream_mode": (c_int, [c_void_p, c_int]),
    "q3_model_config": (c_int, [c_void_p, P(CConfig)]),
    "q3_config_default": (c_int, [c_int, P(CConfig)]),
    "q3_config_from_json": (c_int, [c_char_p, P(CConfig), P(c_int)]),
    "q3_model_load": (c_int, [c_char_p, c_int, P(c_void_p), P(c_int)]),
    "q3_model_load_safetensors": (c_int, [c_void_p, c_char_p, P(c_int)]),
    "q3_safetensors_info": (c_int, [c_char_p, c_char_p, P(c_int), P(ctypes.c_int64), c_int, P(c_int)]),
    "q3_pcm16_from_f32": (c_int, [c_void_p, ctypes.c_int64, c_void_p]),
    "q3_wav_write_pcm16": (c_int, [c_char_p, c_void_p, ctypes.c_int64, ctypes.c_uint32]),
    "q3_wav_read": (c_int, [c_char_p, c_void_p, ctypes.c_int64, P(ctypes.c_int64), P(ctypes.c_uint32)]),
    "q3_codes_write_bin": (c_int, [c_char_p, c_void_p, c_int, c_int]),
    "q3_codes_read_bin": (c_int, [c_char_p, c_void_p, c_int, c_int, P(c_int)]),
    "q3_audio_write_bin": (c_int, [c_char_p, c_void_p, ctypes.c_int64]),
    "q3_audio_read_bin": (c_int, [c_char_p, c_void_p, ctypes.c_int64, P(ctypes.c_int64)]),
    "q3_resample": (c_int, [c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint32, c_void_p, ctypes.c_int64, P(ctypes.c_int64)]),
    "q3_dp_unique_id": (c_int, [c_void_p]),
    "q3_dp_init": (c_int, [c_int, c_int, c_void_p, c_int, P(c_void_p)]),
    "q3_dp_free": (None, [c_void_p]),
    "q3_dp_info": (c_int, [c_void_p, P(c_int), P(c_int)]),
    "q3_dp_broadcast_weights": (c_int, [c_void_p, c_void_p, c_int]),
    "q3_dp_allgather_f64": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "q3_spk_config_default": (c_int, [P(CSpkConfig)]),
    "q3_spk_config_from_json": (c_int, [c_char_p, P(CSpkConfig), P(c_int)]),
    "q3_spk_create": (c_int, [P(CSpkConfig), c_int, P(c_void_p)]),
    "q3_spk_free": (None, [c_void_p]),
    "q3_spk_get_config": (c_int, [c_void_p, P(CSpkConfig)]),
    "q3_spk_n_tensors": (c_int, [c_void_p]),
    "q3_spk_tensor_info": (c_int, [c_void_p, c_int, P(c_char_p), P(ctypes.c_int64)]),
    "q3_spk_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_int, ctypes.c_int64]),
    "q3_spk_finalize": (c_int, [c_void_p]),
    "q3_spk_load_safetensors": (c_int, [c_void_p, c_char_p]),
    "q3_spk_mel_frames": (c_int, [ctypes.c_int64]),
    "q3_spk_mel": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64, P(c_int)]),
    "q3_spk_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, P(c_void_p)]),
    "q3_spk_encode": (c_int, [c_void_p, c_void_p, ctypes.c_int64, ctypes.c_uint32, c_void_p]),
    "q3_mimi_config_default": (c_int, [P(CMimiConfig)]),
    "q3_mimi_create": (c_int, [P(CMimiConfig), c_int, P(c_void_p)]),
    "q3_mimi_free": (None, [c_void_p]),
    "q3_mimi_get_config": (c_int, [c_void_p, P(CMimiConfig)]),
    "q3_mimi_n_tensors": (c_int, [c_void_p]),
    "q3_mimi_tensor_info": (c_int, [c_void_p, c_int, P(c_char_p), P(ctypes.c_int64)]),
    "q3_mimi_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_int, ctypes.c_int64]),
    "q3_mimi_finalize": (c_int, [c_void_p]),
    "q3_mimi_load_safetensors": (c_int, [c_void_p, c_char_p]),
    "q3_mimi_frames": (c_int, [P(CMimiConfig), ctypes.c_int64]),
    "q3_mimi_encode": (c_int, [c_void_p, c_void_p, ctypes.c_int64, ctypes.c_uint32, c_void_p, c_int, P(c_int), P(c_void_p)]),
}

for _name, (_res, _args) in SYMBOLS.items():
    _f = getattr(lib, _name)      # AttributeError here = header/library mismatch
    _f.restype = _res
    _f.argtypes = _args


class Q3Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"q3 status {status}: {msg}")
        self.status = status


def check(status):
    if status != 0:
        raise Q3Error(status, lib.q3_last_error().decode("utf-8", "replace"))
