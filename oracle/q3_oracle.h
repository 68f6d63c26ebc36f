/*
 * q3_oracle.h — CPU F32 restatement of the reference's candle-CPU hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline. The product (qwen3_tts_rs_amd) never links it.
 *
 * PARITY STATUS: "parity unpinned" at the numeric level against the reference binary — the
 * reference is Rust on candle 0.9 (Cargo.toml:33-37), neither cargo/rustc nor candle's sources
 * exist in this image, and the reference repo ships no golden vectors for this path
 * (tests/reference_validation.rs early-returns without test_data/). What IS
 * pinned: every property / known-answer test the reference's own unit tests hold for this path
 * (sampling.rs:441-770, generation/tts.rs:76-120, lib.rs:2031-2118, causal_conv.rs:123-138,
 * causal_trans_conv.rs:163-198, decoder_12hz.rs:714-722, fused_ops.rs:269-313) — see
 * tests/test_oracle_reference_kats.py — plus an independent numpy restatement
 * (tests/np_reference.py) written from the same reference file list.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * repo root).
 */
#ifndef Q3_ORACLE_H
#define Q3_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shape constants: talker.rs:176-290 (TalkerConfig), code_predictor.rs:48-113,
 * decoder_12hz.rs:14-67 (Decoder12HzConfig). */
typedef struct q3o_config {
    int32_t text_vocab;   /* 151936 */
    int32_t text_dim;     /* 2048 */
    int32_t hidden;       /* 2048 (1.7B) / 1024 (0.6B) */
    int32_t inter;        /* 6144 / 3072 */
    int32_t n_layers;     /* 28 */
    int32_t n_heads;      /* 16 */
    int32_t n_kv_heads;   /* 8 */
    int32_t head_dim;     /* 128 */
    int32_t codec_vocab;  /* 3072 */
    int32_t cp_hidden;    /* 1024 */
    int32_t cp_inter;     /* 3072 */
    int32_t cp_layers;    /* 5 */
    int32_t cp_heads;     /* 16 */
    int32_t cp_kv_heads;  /* 8 */
    int32_t cp_vocab;     /* 2048 */
    int32_t n_groups;     /* 16 */
    float   rms_eps;      /* 1e-6 */
    float   rope_theta;   /* 1e6 */
    int32_t dec_cb_dim;   /* 256  codebook inner dim */
    int32_t dec_q_dim;    /* 512  quantizer output dim */
    int32_t dec_latent;   /* 1024 */
    int32_t dec_hidden;   /* 512 */
    int32_t dec_layers;   /* 8 */
    int32_t dec_heads;    /* 16 */
    int32_t dec_head_dim; /* 64 */
    int32_t dec_inter;    /* 1024 */
    int32_t dec_cb_size;  /* 2048 */
    int32_t dec_dim;      /* 1536 */
    int32_t dec_up_ratios[2]; /* 2,2 */
    int32_t dec_up_rates[4];  /* 8,5,4,3 */
    float   dec_eps;      /* 1e-5 */
    float   dec_theta;    /* 1e4 */
} q3o_config;

/* SynthesisOptions (lib.rs:1786-1836) */
typedef struct q3o_options {
    double  temperature;        /* 0.9  (f64 in the reference) */
    double  top_p;              /* 0.9 */
    double  repetition_penalty; /* 1.05 */
    uint64_t seed;
    int32_t max_length;         /* 2048 */
    int32_t top_k;              /* 50 */
    int32_t eos_token_id;       /* 2150; -1 = None */
    int32_t chunk_frames;       /* 10 */
    int32_t min_new_tokens;     /* 2 */
    int32_t has_seed;           /* 0 = None (unsupported by parity harnesses) */
} q3o_options;

enum { Q3O_MODE_CUSTOM_VOICE = 0, Q3O_MODE_VOICE_CLONE = 1, Q3O_MODE_VOICE_DESIGN = 2 };

typedef struct q3o_request {
    int32_t mode;
    const uint32_t* text_ids;     int32_t n_text;
    const uint32_t* instruct_ids; int32_t n_instruct;   /* voice design */
    uint32_t speaker_id;          /* codec id of the speaker token (custom voice) */
    uint32_t language_id;         /* codec id of the language token */
    const float* xvector;         /* [hidden] speaker embedding (voice clone) */
    q3o_options opts;
    /* ICL voice clone (lib.rs:897-1046): reference codec frames [n_ref][16] + reference text ids */
    const uint32_t* ref_codes;    int32_t n_ref;
    const uint32_t* ref_text_ids; int32_t n_ref_text;
} q3o_request;

typedef struct q3o_model q3o_model;
typedef struct q3o_session q3o_session;

const char* q3o_last_error(void);
void q3o_set_threads(int n);
int q3o_get_threads(void);   /* threads the parallel regions use (default min(cores, 32)) */

q3o_model* q3o_model_new(const q3o_config* cfg);
void q3o_model_free(q3o_model* m);
/* copies n f32 values under `name` (reference safetensors names) */
int q3o_model_set_tensor(q3o_model* m, const char* name, const float* data, int64_t n);
/* resolve all tensors; which: 1 = talker+code predictor, 2 = decoder, 3 = both */
int q3o_model_finalize(q3o_model* m, int which);

/* ---- standalone pieces ---- */
void q3o_linear(const float* x, const float* w, const float* b, float* y, int M, int N, int K);
void q3o_rms_norm(const float* x, const float* w, float* y, int rows, int cols, float eps);
void q3o_fused_residual_rmsnorm(const float* x, const float* res, const float* w, int rows, int cols,
                                float eps, float* normed, float* sum);
void q3o_rope_table(float theta, int head_dim, int pos0, int n_pos, float* cos_out, float* sin_out);
void q3o_rng_seed(uint64_t seed, uint64_t* state);
float q3o_rng_next(uint64_t* state);
void q3o_build_suppression_mask(int vocab, int eos_id, uint8_t* mask);
void q3o_apply_penalties(float* logits, int vocab, const uint8_t* seen, double rep_penalty,
                         int token_count, int min_new_tokens, int eos_id);
void q3o_top_k_filter(float* logits, int vocab, int k);
void q3o_top_p_filter(float* logits, int vocab, double p);
uint32_t q3o_sample(const float* logits, int vocab, double temperature, int top_k, double top_p,
                    uint64_t* rng_state);
void q3o_codes_to_tensor(const uint32_t* frames, int n_frames, int64_t* out /*[16][n]*/);

/* ---- session: prefill + generation (lib.rs:718-784, 530-656) ---- */
q3o_session* q3o_session_new(q3o_model* m, const q3o_request* req);
void q3o_session_free(q3o_session* s);
int q3o_session_prefill_len(const q3o_session* s);
/* options actually used by generate (ICL adjusts repetition_penalty / max_length, lib.rs:913-929) */
void q3o_session_effective(const q3o_session* s, double* repetition_penalty, int* max_length);
/* state right after prefill: normed last hidden [hidden], logits [codec_vocab] */
void q3o_session_prefill_out(const q3o_session* s, float* last_hidden, float* logits);
/* prefill input embeddings [prefill_len][hidden] (for stage tests) */
void q3o_session_prefill_embeds(const q3o_session* s, float* out);
int q3o_session_trailing_len(const q3o_session* s);
void q3o_session_trailing(const q3o_session* s, float* trailing /*[T_tr][hidden]*/, float* pad /*[hidden]*/);
/* full generate_codes loop; codes_out [max_length][16]; optional debug captures (may be NULL):
 * talker_logits [max_length+1][codec_vocab] (raw logits fed to each sampling decision, index 0 =
 * prefill logits), cp_logits [max_length][15][cp_vocab]. Returns number of frames. */
int q3o_session_generate(q3o_session* s, uint32_t* codes_out, float* talker_logits, float* cp_logits);
/* teacher-forced single steps */
void q3o_session_talker_step(q3o_session* s, const float* input_embed, float* hidden_out, float* logits_out);
void q3o_session_cp_generate(q3o_session* s, const float* last_hidden, const float* sem_embed,
                             uint32_t* codes15, float* cp_logits /*[15][cp_vocab] or NULL*/);
/* frame glue (lib.rs:612-622): step_input = sem + sum(acoustic) + text_addition */
void q3o_frame_embed(q3o_model* m, uint32_t sem_token, const uint32_t* codes15, const float* text_add, float* out);

/* ---- codec decoder (decoder_12hz.rs:411-505) ---- */
int q3o_decode(q3o_model* m, const int64_t* codes /*[16][T]*/, int T, float* pcm /*[1920*T]*/);
/* stage taps for tests: returns malloc-free copy into caller buffers; stage ids below */
enum { Q3O_DEC_QUANT = 0, Q3O_DEC_PRECONV = 1, Q3O_DEC_PRETRANS = 2, Q3O_DEC_UP0 = 3, Q3O_DEC_UP1 = 4,
       Q3O_DEC_INIT = 5, Q3O_DEC_BLK0 = 6, Q3O_DEC_BLK1 = 7, Q3O_DEC_BLK2 = 8, Q3O_DEC_BLK3 = 9, Q3O_DEC_N = 10 };
int q3o_decode_taps(q3o_model* m, const int64_t* codes, int T, float* pcm, float** taps /*[Q3O_DEC_N] or NULL entries*/);
void q3o_causal_conv1d(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L, int k, int dil, int groups);
void q3o_causal_trans_conv1d(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L, int k, int stride);
void q3o_snake_beta(const float* x, const float* alpha, const float* beta, float* y, int C, int L);

/* ---- speaker-embedding path (q3_oracle_spk.c): mel.rs:47-59,135-324; speaker.rs:24-469 ---- */
typedef struct q3o_spk_config {
    int32_t mel_dim;            /* 128 */
    int32_t enc_dim;            /* 1024 */
    int32_t channels[5];        /* 512,512,512,512,1536 */
    int32_t kernel_sizes[5];    /* 5,3,3,3,1 */
    int32_t dilations[5];       /* 1,2,3,4,1 */
    int32_t attention_channels; /* 128 */
    int32_t res2net_scale;      /* 8 */
    int32_t se_channels;        /* 128 */
    int32_t sample_rate;        /* 24000 */
} q3o_spk_config;
typedef struct q3o_spk q3o_spk;
q3o_spk* q3o_spk_new(const q3o_spk_config* cfg);
void q3o_spk_free(q3o_spk* s);
int q3o_spk_set_tensor(q3o_spk* s, const char* name, const float* data, int64_t n);
const char* q3o_spk_last_error(void);
void q3o_hann_window(int len, float* out);
void q3o_mel_filterbank(int sample_rate, int n_fft, int n_mels, float fmin, float fmax, float* out /*[n_mels][n_fft/2+1]*/);
int q3o_mel_frames(int n_samples, int n_fft, int hop);
/* log-mel for the speaker encoder, out [128][T]; returns T or -1 */
int q3o_mel_speaker(const float* samples, int n, float* mel, int cap_frames);
void q3o_reflect_pad_1d(const float* x, int C, int T, int pl, int pr, float* out);
/* taps (NULL or 6 pointers, NULL entries skipped): 0 blocks.0 out, 1-3 SE-Res2Net outs, 4 MFA out, 5 pooled [2*C4] */
int q3o_spk_forward(q3o_spk* s, const float* mel /*[mel_dim][T]*/, int T, float* out /*[enc_dim]*/, float** taps);
int q3o_spk_encode(q3o_spk* s, const float* samples, int n, float* out);

/* ---- speech-tokenizer encoder (q3_oracle_mimi.c): encoder_12hz.rs:34-144 over candle-transformers' Mimi = the published
 * Mimi encoder (HF transformers models/mimi/modeling_mimi.py) ---- */
typedef struct q3o_mimi_config {
    int32_t n_filters;      /* 64 */
    int32_t hidden;         /* 512 */
    int32_t ratios[4];      /* encoder order: 4, 5, 6, 8 */
    int32_t kernel;         /* 7 */
    int32_t res_kernel;     /* 3 */
    int32_t last_kernel;    /* 3 */
    int32_t compress;       /* 2 */
    int32_t n_layers;       /* 8 */
    int32_t n_heads;        /* 8 */
    int32_t head_dim;       /* 64 */
    int32_t inter;          /* 2048 */
    int32_t window;         /* 250 */
    int32_t cb_size;        /* 2048 */
    int32_t cb_dim;         /* 256 */
    int32_t n_q;            /* 16 */
    int32_t n_sem;          /* 1 */
    float norm_eps;         /* 1e-5 */
    float rope_theta;       /* 1e4 */
} q3o_mimi_config;
typedef struct q3o_mimi q3o_mimi;
q3o_mimi* q3o_mimi_new(const q3o_mimi_config* cfg);
void q3o_mimi_free(q3o_mimi* m);
const char* q3o_mimi_last_error(const q3o_mimi* m);
int q3o_mimi_set_tensor(q3o_mimi* m, const char* name, const float* data, int64_t n);     /* names as in speech_tokenizer/model.safetensors ("encoder.…") */
int q3o_mimi_frames(const q3o_mimi_config* cfg, int64_t n_samples);
/* codes [T][n_q]; taps NULL or 4 pointers (NULL entries skipped): SEANet out [hidden][T25], transformer out [hidden][T25], downsampled
 * [hidden][T], per-decision squared-distance margins [T][n_q] */
int q3o_mimi_encode(q3o_mimi* m, const float* samples, int64_t n, uint32_t* codes, float** taps);

#ifdef __cplusplus
}
#endif
#endif
