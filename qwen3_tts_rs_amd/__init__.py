"""qwen3_tts_rs_amd — MI355X-native (gfx950) hot path of Qwen3-TTS behind the reference's
generation/session API. See DESIGN.md / INTEGRATION.md; the C ABI is include/q3tts.h."""
from .config import Q3Config, qwen3_tts_0_6b, qwen3_tts_1_7b, tiny, tiny_same_width
from .api import (Qwen3TTS, Session, Batcher, StreamingSession, SynthesisOptions, SynthesisTiming, AudioBuffer, Utterance,
                  Speaker, Language, CODEC_EOS_TOKEN_ID, SAMPLES_PER_FRAME, codes_to_tensor, auto_device,
                  fused_residual_rmsnorm, linear, sample)
from .speaker import SpeakerEncoder, SpeakerEncoderConfig, VoiceClonePrompt, tiny_speaker_config
from .speech_encoder import SpeechEncoder, SpeechEncoderConfig, tiny_speech_config

__all__ = ["Q3Config", "qwen3_tts_0_6b", "qwen3_tts_1_7b", "tiny", "tiny_same_width", "Qwen3TTS", "Session",
           "StreamingSession", "SynthesisOptions", "SynthesisTiming", "AudioBuffer", "Utterance", "Speaker", "Language",
           "CODEC_EOS_TOKEN_ID", "SAMPLES_PER_FRAME", "codes_to_tensor", "auto_device", "fused_residual_rmsnorm",
           "linear", "sample", "SpeakerEncoder", "SpeakerEncoderConfig", "VoiceClonePrompt", "tiny_speaker_config",
           "SpeechEncoder", "SpeechEncoderConfig", "tiny_speech_config"]
