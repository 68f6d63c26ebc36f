/*
 * q3tts.h — C ABI of the MI355X-native Qwen3-TTS hot path (libq3tts.so, gfx950).
 *
 * This is the drop-in boundary for the reference's generation/session API: each entry point
 * names the reference interface (file:line, relative to the TrevorS/qwen3-tts-rs repo root) it
 * replaces. A Rust shim (`extern "C"` block) re-exposes the reference's
 * `Qwen3TTS` / `SynthesisOptions` / `StreamingSession` names on top of these symbols; the Python
 * mirror in qwen3_tts_rs_amd/api.py does the same over ctypes.
 *
 * Conventions: plain pointers and sizes only; every function returns q3_status (0 = OK) and
 * never throws/aborts across the ABI; `q3_last_error()` returns the thread-local message
 * (the reference's anyhow::Error text). All `*_host` pointers are host memory; device memory is
 * owned by the library (model arena, per-session KV pages) except where a function says
 * "device pointer". One process drives one GPU; the model handle is immutable after
 * q3_model_finalize and may be shared by any number of sessions on that GPU.
 */
#ifndef Q3TTS_H
#define Q3TTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3_ABI_VERSION 1
#define Q3_MAX_BATCH 64      /* sequences per session (rows of the decode GEMVs: 16-column MFMA tiles x 4) */

typedef enum q3_status {
    Q3_OK = 0,
    Q3_INVALID_ARG = 1,
    Q3_IO = 2,
    Q3_MISSING_WEIGHT = 3,   /* lib.rs:226-231, decoder_12hz.rs:176-181 ("Missing weight: ...") */
    Q3_KV_OVERFLOW = 4,      /* kv_cache.rs:293-300 */
    Q3_HIP_ERROR = 5,
    Q3_RCCL_ERROR = 6,
    Q3_UNSUPPORTED = 7,
    Q3_OOM = 8
} q3_status;

/* Shape constants parsed from config.json by the reference (config.rs:238-336 →
 * talker.rs:176-290 TalkerConfig, code_predictor.rs:48-113, decoder_12hz.rs:14-67). */
typedef struct q3_config {
    int32_t text_vocab;   /* 151936 */
    int32_t text_dim;     /* 2048 */
    int32_t hidden;       /* 2048 (1.7B) / 1024 (0.6B) */
    int32_t inter;        /* 6144 / 3072 */
    int32_t n_layers;     /* 28 */
    int32_t n_heads;      /* 16 */
    int32_t n_kv_heads;   /* 8 */
    int32_t head_dim;     /* 128 (only value supported by the gfx950 kernels) */
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
    int32_t dec_cb_dim;   /* 256 */
    int32_t dec_q_dim;    /* 512 */
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
} q3_config;

/* SynthesisOptions (lib.rs:1786-1836); f64 fields are f64 in the reference too. */
typedef struct q3_options {
    double   temperature;        /* 0.9 */
    double   top_p;              /* 0.9 */
    double   repetition_penalty; /* 1.05 */
    uint64_t seed;               /* used when has_seed != 0 */
    int32_t  max_length;         /* 2048  (max_new_tokens) */
    int32_t  top_k;              /* 50 */
    int32_t  eos_token_id;       /* 2150 (CODEC_EOS_TOKEN_ID, lib.rs:1466); -1 = None */
    int32_t  chunk_frames;       /* 10 */
    int32_t  min_new_tokens;     /* 2 */
    int32_t  has_seed;           /* 0 = None: seeded from the wall clock (sampling.rs:66-82) */
} q3_options;

enum { Q3_MODE_CUSTOM_VOICE = 0, Q3_MODE_VOICE_CLONE = 1, Q3_MODE_VOICE_DESIGN = 2 };
enum { Q3_DTYPE_F32 = 0, Q3_DTYPE_BF16 = 1 };

/* One utterance. Replaces the argument lists of synthesize_with_voice (lib.rs:718-724),
 * synthesize_voice_design (lib.rs:802-808) and the x-vector-only branch of
 * synthesize_voice_clone (lib.rs:897-951); token ids come from the caller's tokenizer
 * (text.rs:210-217 stays on the Rust side). */
typedef struct q3_request {
    int32_t mode;
    const uint32_t* text_ids;     int32_t n_text;
    const uint32_t* instruct_ids; int32_t n_instruct;  /* voice design: ChatML-framed instruct */
    uint32_t speaker_id;          /* Speaker::token_id (talker.rs:145-157) */
    uint32_t language_id;         /* Language::token_id (talker.rs:94-107) */
    const float* xvector;         /* [hidden] speaker embedding (voice clone) or NULL */
    q3_options opts;
    /* ICL voice clone (VoiceClonePrompt.ref_codes / ref_text_ids, lib.rs:127-134, 897-1046): reference codec
     * frames [n_ref][16] (from the caller's speech encoder) and the reference transcript's token ids. With
     * both present (mode = voice clone) the talker prefill is extended by the ICL block of
     * build_icl_prompt (talker.rs:646-710, streaming overlay), repetition_penalty is floored at 1.5 and
     * max_length capped at max(75, 6·n_text) (lib.rs:913-929).
     * ref_codes ALONE (with or without the transcript) already makes the full-utterance decode of a prefilled session
     * (q3_session_decode(b, 0, n_frames), q3_session_run) prepend the reference frames and cut their share of the samples
     * off again, as lib.rs:1022-1041 does for any prompt that carries codes; before q3_session_prefill (no frames, the
     * reference frames not yet on the device) the call decodes the empty range. */
    const uint32_t* ref_codes;    int32_t n_ref;
    const uint32_t* ref_text_ids; int32_t n_ref_text;
} q3_request;

/* SynthesisTiming (lib.rs:138-147) */
typedef struct q3_timing {
    double prefill_ms, generation_ms, decode_ms;
    int32_t generation_frames;
} q3_timing;

typedef struct q3_model q3_model;
typedef struct q3_session q3_session;

int         q3_abi_version(void);
const char* q3_last_error(void);
/* number of visible HIP devices, or -1 (replaces auto_device / parse_device, lib.rs:1854-1926) */
int         q3_device_count(void);

/* ---------------- model (Qwen3TTS::from_weights, lib.rs:267-368) ---------------- */
/* device = HIP device index; device = -1 creates a manifest-only handle (tensor names/shapes, no GPU) */
q3_status q3_model_create(const q3_config* cfg, int device, q3_model** out);
void      q3_model_free(q3_model* m);
/* Upload one checkpoint tensor under its safetensors name. `dtype` is the
 * SOURCE dtype; talker/code-predictor matrices are stored bf16 in HBM (the checkpoint's native
 * dtype, lib.rs:1394-1396), norms/biases and all decoder tensors f32 (lib.rs:344-353). */
q3_status q3_model_set_tensor(q3_model* m, const char* name, int dtype, const void* data_host, int64_t n);
/* the shape constants the handle was built with (talker.config(), talker.rs:843-846) */
q3_status q3_model_config(const q3_model* m, q3_config* out);
/* Names the model expects: i in [0, q3_model_n_tensors); returns name, element count, stored dtype */
int       q3_model_n_tensors(const q3_model* m);
q3_status q3_model_tensor_info(const q3_model* m, int i, const char** name, int64_t* n, int* stored_dtype);
/* Weight arena for the one data-parallel collective (RCCL broadcast from rank 0): device pointer
 * + size; after the broadcast non-root ranks call q3_model_mark_loaded then q3_model_finalize. */
q3_status q3_model_arena(q3_model* m, void** dev_ptr, size_t* bytes);
q3_status q3_model_mark_loaded(q3_model* m);
/* Paged talker KV. The reference allocates one cache per call, max_new_tokens + 256 positions long, and bails when it
 * overflows (kv_cache.rs:234-310, overflow :293-300; sized at lib.rs:450). Here every session of a model draws PAGES
 * (128 positions x every layer and KV head, K and V) from one pool per model as its rows grow, and returns them when a
 * row is replaced or the session ends: a 4k-position prompt and a ten-position prompt share the memory, and
 * q3_session_replace relinks the prefilled pages into the row instead of copying them.
 * q3_model_kv_pool_limit: the most pages the model's sessions may hold at once (0 = no limit but HBM); a session that
 * needs a page beyond it fails with Q3_KV_OVERFLOW before it runs (the reference's overflow bail). ONE budget covers f32 and
 * bf16 sessions (q3_session_set_kv_dtype): pages are counted in f32 equivalents, a bf16 page is half of one — the f32 pages a
 * bf16 session's prompt is prefilled into count in full until they are converted. The native batcher (q3_batcher_*) admits a
 * request only when its worst case (prompt + max_length) fits beside what the running rows may still take, so under a limit a
 * request waits in the queue instead of failing mid-generation; a request that cannot fit even alone fails on its ticket.
 * q3_model_kv_pool_info: page geometry and occupancy in f32-equivalent pages (any pointer may be NULL).
 * q3_model_kv_pool_trim: slabs (32 pages, ~1 GB at 28 layers x 8 KV heads) none of whose pages is held go back to the device;
 * the first f32 slab is allocated by q3_model_finalize so that it is not on the first request's time to first audio. */
q3_status q3_model_kv_pool_limit(q3_model* m, int max_pages);
q3_status q3_model_kv_pool_trim(q3_model* m, size_t* bytes_freed);
/* Vocoder arithmetic. The codec decoder's convs (decoder_block.rs:81-92, 240-247; F32 in the reference on every device)
 * run on the bf16 matrix cores with each f32 operand split into bf16 planes. planes = 3 (default): hi + mid + lo, six
 * products per multiply-accumulate — every f32 product exact, PCM within 2.5e-5 RMS of the reference CPU path.
 * planes = 2: hi + mid, three products — operands carry 16-17 mantissa bits (more than TF32's 11), PCM within 1e-4 RMS
 * (the path's tolerance is 1e-3), the 640-frame decode 20.6 -> 14.4 ms. Codec token ids never depend on it. Applies to
 * decodes started after the call; anything but 2 or 3 is Q3_INVALID_ARG. */
q3_status q3_model_set_codec_planes(q3_model* m, int planes);
q3_status q3_model_kv_pool_info(q3_model* m, int* page_positions, size_t* page_bytes, int* pages_total, int* pages_in_use, int* pages_peak);
/* Verify every tensor is present ("Missing weight: <name>"), derive codebooks
 * (decoder_12hz.rs:189-225) and RoPE tables. */
q3_status q3_model_finalize(q3_model* m);
/* Deterministic synthetic tensor generator (SURVEY.md Appendix B rules; host side, no GPU):
 * value_i = offset + scale * z_i, z ~ approx N(0,1) from a counter hash of (seed, name, i);
 * dtype BF16 writes uint16 (round-to-nearest-even), F32 writes float. */
q3_status q3_synth_fill(uint64_t seed, const char* name, int dtype, float scale, float offset,
                        int64_t n, void* out_host);

/* ---------------- session = one batch of utterances on one GPU ----------------
 * Owns KV pages, RNG streams, penalty masks (the fields of StreamingSession, lib.rs:1484-1508).
 * batch > 1 has no reference counterpart (reference batch = 1): every sequence behaves exactly
 * as its own batch-1 run. The sequences may be of ANY mix of prompt kinds and lengths (CustomVoice lib.rs:718-784,
 * VoiceDesign 802-870, x-vector / ICL voice clone 897-1046): rows of one prefill length are prefilled together (the text
 * length is free — it rides along as trailing text — so all CustomVoice requests share one batched prefill); a RAGGED
 * batch is prefilled in groups of equal prefill length by q3_session_prefill, each group on the side, and every row is then
 * moved into the session the way q3_session_replace does — decode runs in one captured frame graph over all rows. A
 * malformed request of a ragged batch is reported by q3_session_prefill (equal-length batches: by q3_session_create);
 * debug / profiling sessions (q3_session_set_debug) need rows of one prefill length.
 * Every request keeps its own q3_options — temperature, top-k / top-p, repetition penalty, min_new_tokens, EOS id, seed,
 * max_length: one sampler row per sequence on the device; only chunk_frames must be the same for all of them. */
q3_status q3_session_create(q3_model* m, const q3_request* reqs, int batch, q3_session** out);
/* The same with capacity for later arrivals (q3_session_replace): code buffers and the pre-drawn PCG streams are sized for
 * max(frame_budget, the largest max_length of `reqs`) frames per row, the KV extent of a row for
 * max(prompt_budget, the batch's prefill length) prompt positions + those frames. The rows of `reqs` behave exactly as
 * under q3_session_create (each ends at its own max_length). No reference counterpart (one utterance per call). */
q3_status q3_session_create_reserved(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget,
                                     q3_session** out);
void      q3_session_free(q3_session* s);
/* prefill_custom_voice / _voice_clone / _voice_design + run_prefill_layers (talker.rs:451-627,
 * 823-841), build_trailing_text (lib.rs:508-519) and the first sampling decision
 * (lib.rs:558-571). */
q3_status q3_session_prefill(q3_session* s);
/* generate_codes frame loop (lib.rs:580-652): run up to n_frames more frames for every live
 * sequence; returns when they are done on the device. use_graph != 0: the frame is captured once
 * (hipGraph) and replayed per frame — as packets on the library's own AQL queue (the default;
 * q3_session_submit_info) or through hipGraphLaunch; 0: eager launches. Same kernels, same bits. */
q3_status q3_session_generate(q3_session* s, int n_frames, int use_graph);
/* frames emitted so far for sequence b (stops at the frame whose semantic token is EOS,
 * lib.rs:581-585) and whether it hit EOS / max_length */
q3_status q3_session_frames(q3_session* s, int b, int* n_frames, int* done);
/* gpu_frames_to_frame_codes (lib.rs:676-690): [n][16] u32, frame-major */
q3_status q3_session_codes(q3_session* s, int b, uint32_t* codes_host, int cap_frames, int* n_frames);
/* decode_codes over frames [f0, f1) of sequence b (lib.rs:881-890; streaming decodes each chunk
 * context-free, lib.rs:1755-1758): writes (f1-f0)*1920 f32 samples */
q3_status q3_session_decode(q3_session* s, int b, int f0, int f1, float* pcm_host, size_t cap, size_t* n_samples);
/* synthesize_with_timing (lib.rs:425-501) for the whole batch: prefill → generate → decode.
 * pcm_host[b] receives up to cap[b] samples (NULL = skip copy-out); n_samples[b] is set. */
q3_status q3_session_run(q3_session* s, int use_graph, float** pcm_host, const size_t* cap, size_t* n_samples,
                         q3_timing* timing);
/* StreamingSession::next_chunk (lib.rs:1650-1759) for sequence 0 of a batch-1 session: generates
 * up to chunk_frames frames, decodes them as an independent utterance; *done=1 with
 * *n_samples=0 when finished. */
q3_status q3_session_next_chunk(q3_session* s, float* pcm_host, size_t cap, size_t* n_samples, int* done);
/* The same for sequence b of a session with several sequences (one StreamingSession per row, lib.rs:1484-1541): the rows
 * advance in lockstep, so the first row asked generates the chunk's frames for all of them and the others only run their
 * vocoder. An error leaves the row's position untouched: the call can be repeated (lib.rs:1775-1781). */
q3_status q3_session_next_chunk_row(q3_session* s, int b, float* pcm_host, size_t cap, size_t* n_samples, int* done);
/* Continuous batching: replace row b of a PREFILLED session — normally one whose sequence has ended (q3_session_frames:
 * done; fetch its codes / PCM first) — by a new request, which then starts at its frame 0 while the other rows go on. The
 * reference keeps all per-utterance state per call (KV caches, SamplingContext, penalty mask, trailing text:
 * lib.rs:743-756; StreamingSession lib.rs:1484-1541); here it is row b's slice of the session's device state, refilled
 * from a one-row prefill of `req`. The request carries its own q3_options — sampler settings, seed, EOS id, max_length
 * (SynthesisOptions is per call in the reference, lib.rs:1786-1836; only chunk_frames must equal the session's); it must
 * fit the row (max_length <= the session's largest; text rows <= max(1024, the longest of the original batch); prompt +
 * max_length within the row's KV extent). Any mode fits any session (an ICL request's repetition-penalty floor of 1.5,
 * lib.rs:1154-1160, is resolved into its own row). Each row of a session stops at its own opts.max_length;
 * q3_session_generate returns early once every row is done. Other rows are bit-for-bit unaffected. */
q3_status q3_session_replace(q3_session* s, int b, const q3_request* req);
/* ---------------- continuous batcher: a queue of requests through the rows of one session ----------------
 * The serving loop around q3_session_replace, native: requests of any prompt kind, length and options are queued; a step
 * fills free rows from the queue, runs up to n_frames frames of the shared frame graph and collects the rows that ended.
 * No thread of its own — the host calls q3_batcher_step from its loop; calls may interleave freely, one thread at a time.
 * Each request's codes / PCM are those of its own batch-1 run (per-call state of the reference: lib.rs:743-756). No
 * reference counterpart (one utterance per call). */
typedef struct q3_batcher q3_batcher;
enum { Q3_TICKET_QUEUED = 0, Q3_TICKET_RUNNING = 1, Q3_TICKET_DONE = 2, Q3_TICKET_FAILED = 3 };
/* slots = rows of the session (1..64); frame_budget = largest max_length a request may ask for; prompt_budget = prefill
 * positions a row can hold (at least 16, which covers CustomVoice and x-vector prompts — their text rides along as trailing
 * text; VoiceDesign needs its instruct length + 16, ICL its reference frames + 16): see q3_session_create_reserved. The
 * session is opened lazily, on idle rows, when the first request arrives. */
q3_status q3_batcher_create(q3_model* m, int slots, int frame_budget, int prompt_budget, q3_batcher** out);
void      q3_batcher_free(q3_batcher* b);
/* Queue a request (deep copy: the caller's arrays may go away). want_pcm != 0: the finished row is vocoded
 * (the samples of q3_session_decode, ICL prompts included); else only its codes are kept. Since round 6 the vocoder of a
 * finished row runs on a worker thread and stream of the batcher while its row is refilled and the frames go on: the ticket
 * reads RUNNING until its samples have landed (q3_batcher_poll), and q3_batcher_fetch waits for them. */
q3_status q3_batcher_submit(q3_batcher* b, const q3_request* req, int want_pcm, int64_t* ticket);
/* One scheduling round. A request that cannot be placed (prompt or text longer than a row holds) fails alone: its ticket
 * carries the status and message. n_finished counts tickets whose generation ended (or that failed) in this call; with
 * want_pcm the samples of such a ticket may still be with the decode worker (it polls RUNNING until they land). A call
 * that returns n_running == 0 and n_queued == 0 has waited for the worker: every such ticket polls DONE afterwards. */
q3_status q3_batcher_step(q3_batcher* b, int n_frames, int use_graph, int* n_running, int* n_queued, int* n_finished);
/* state (Q3_TICKET_*); n_frames: frames of a finished ticket, frames run so far of a running one; n_samples: PCM samples held */
q3_status q3_batcher_poll(q3_batcher* b, int64_t ticket, int* state, int* n_frames, size_t* n_samples);
/* Copy a finished ticket's results out ([n_frames][16] u32; n_samples f32) and release it. A FAILED ticket returns its
 * status (message in q3_last_error) and is released too. */
q3_status q3_batcher_fetch(q3_batcher* b, int64_t ticket, uint32_t* codes_host, int cap_frames, float* pcm_host, size_t cap_samples);

/* Chunk decode mode of q3_session_next_chunk. 0 (default) = each chunk decoded as an independent utterance, exactly
 * as the reference does (lib.rs:1755-1758: audible seams, every chunk restarts from zero padding). 1 = continuous:
 * the vocoder's front runs over all frames so far and its convolutional stack over the chunk plus 12 frames of left
 * context, so the concatenated chunks are sample-identical to the non-streaming decode (SURVEY.md §8(f) rank 1,
 * "improved overlap mode"). */
q3_status q3_session_set_stream_mode(q3_session* s, int mode);

/* ---------------- stage-level entry points (parity tests; the reference's
 * tests/reference_validation.rs stages) ---------------- */
enum {
    Q3_GET_PREFILL_EMBEDS = 0,  /* [prefill_len][hidden] f32 */
    Q3_GET_LAST_HIDDEN = 1,     /* [hidden] normed last hidden (talker.rs:729-735) */
    Q3_GET_LOGITS = 2,          /* [codec_vocab] raw logits of the latest talker step */
    Q3_GET_TRAILING = 3,        /* [T_tr][hidden] */
    Q3_GET_PAD_EMBED = 4,       /* [hidden] */
    Q3_GET_LOGITS_HIST = 5,     /* [frames+1][codec_vocab] (session created with debug capture) */
    Q3_GET_CP_LOGITS = 6,       /* [15][cp_vocab] of the latest code-predictor run */
    Q3_GET_TOKEN = 7,           /* u32 current semantic token */
    Q3_GET_CP_LOGITS_HIST = 8   /* [frames][15][cp_vocab] (debug capture) */
};
/* K/V dtype of the talker cache, before q3_session_prefill: Q3_DTYPE_F32 (default — the parity contract is the reference's CPU
 * F32 path) or Q3_DTYPE_BF16, the dtype of the reference GPU path's cache (kv_cache.rs:234-310; KVCache::new(..., dtype) at
 * lib.rs:450): the prompt is prefilled in f32 and converted once, decode steps append and read bf16 — half the K/V bytes per
 * frame; results are no longer bit-comparable with the F32 oracle. Needs the paged cache. */
q3_status q3_session_set_kv_dtype(q3_session* s, int dtype);
q3_status q3_session_set_debug(q3_session* s, int capture_logits);
q3_status q3_session_prefill_len(q3_session* s, int b, int* prefill_len, int* trailing_len);
q3_status q3_session_get(q3_session* s, int what, int b, void* out_host, size_t bytes);
/* generate_step_with_embed (talker.rs:716-736), teacher-forced: embeds_host [batch][hidden] */
q3_status q3_talker_step(q3_session* s, const float* embeds_host, float* hidden_host, float* logits_host);
/* generate_acoustic_codes (code_predictor.rs:320-416), teacher-forced: inputs [batch][hidden] */
q3_status q3_cp_generate(q3_session* s, const float* last_hidden_host, const float* sem_embed_host,
                         uint32_t* codes15_host, float* cp_logits_host /*[batch][15][cp_vocab] or NULL*/);
/* lib.rs:612-622 frame glue: sem + ((e0+e1)+..+e14) + text_add → [hidden] */
q3_status q3_frame_embed(q3_model* m, uint32_t sem_token, const uint32_t* codes15, const float* text_add_host,
                         float* out_host);
/* apply_generation_penalties_gpu + sample (lib.rs:1271-1322; sampling.rs:140-319) on device for
 * `rows` independent rows: logits_host [rows][vocab], seen_host [rows][vocab] u8 (may be NULL),
 * u_host [rows] uniform draws (SamplingContext::rand_f32) → tokens_host [rows] */
q3_status q3_sample(int device, const float* logits_host, const uint8_t* seen_host, const float* u_host,
                    int rows, int vocab, const q3_options* opts, int token_count, uint32_t* tokens_host);
/* PCG-XSH-RR stream of SamplingContext (sampling.rs:32-51, 84-94); host side */
void      q3_rng_seed(uint64_t seed, uint64_t* state);
float     q3_rng_next(uint64_t* state);
/* FusedRmsNorm::forward_residual (fused_ops.rs:49-96; kernels/fused_residual_rmsnorm.cu:39-90):
 * returns (rms_norm(x+res)*w, x+res); dtype F32 or BF16 (storage), f32 math */
q3_status q3_fused_residual_rmsnorm(int device, int dtype, const void* x_host, const void* res_host,
                                    const void* w_host, int rows, int cols, float eps,
                                    void* normed_host, void* sum_host);
/* y = x·Wᵀ (+b) with bf16 weights / f32 activations — the GEMV family used by every projection
 * (candle Linear, transformer.rs:224-227): x [M][K] f32, w [N][K] bf16 */
q3_status q3_linear(int device, const float* x_host, const uint16_t* w_bf16_host, const float* bias_host,
                    int M, int N, int K, float* y_host);
/* Qwen3TTS::decode_codes (lib.rs:881-890) / Decoder12Hz::decode (decoder_12hz.rs:411-505):
 * frames [n][16] u32 → n*1920 f32 samples. taps (optional, [Q3_DEC_N] host pointers or NULL)
 * receive stage outputs for the stage-by-stage validation the reference does in
 * tests/reference_validation.rs:1755-2400. */
enum { Q3_DEC_QUANT = 0, Q3_DEC_PRECONV = 1, Q3_DEC_PRETRANS = 2, Q3_DEC_UP0 = 3, Q3_DEC_UP1 = 4,
       Q3_DEC_INIT = 5, Q3_DEC_BLK0 = 6, Q3_DEC_BLK1 = 7, Q3_DEC_BLK2 = 8, Q3_DEC_BLK3 = 9, Q3_DEC_N = 10 };
q3_status q3_decode_codes(q3_model* m, const uint32_t* frames_host, int n_frames, float* pcm_host,
                          float** taps_host);
/* codes_to_tensor (lib.rs:1417-1431): [n][16] u32 → [16][n] i64 (host helper) */
void      q3_codes_to_tensor(const uint32_t* frames, int n_frames, int64_t* out);

/* ---------------- measurement hooks (bench.py) ---------------- */
/* per-kernel-class accumulated GPU time since the last reset, measured with hipEvents on the
 * session stream when profiling is enabled (q3_session_set_profile) */
q3_status q3_session_set_profile(q3_session* s, int enable);
/* accumulated since the last reset: GPU milliseconds, algorithmic weight bytes and launch count of
 * the bf16 GEMV family (the dominant kernel) */
q3_status q3_session_profile_read(q3_session* s, double* ms, double* bytes, long* launches, int reset);
/* the distinct GEMV launches of the frame loop since profiling was enabled (bench.py's roofline inventory): rows of 8 ints
 * {M, N, K, epilogue, fused input RMSNorm 0/1, reserved (0), tiling, count} — exactly the arguments q3_bench_linear
 * replays; rows == NULL: only *n_rows */
q3_status q3_session_profile_shapes(q3_session* s, int* rows, int cap_rows, int* n_rows, int reset);
/* µs per launch of one GEMV shape: `iters` launches over `n_copies` distinct weight buffers (HBM-resident
 * stream, not Infinity-Cache hits) replayed from one hipGraph and timed with HIP events on that stream.
 * tiled (the inventory's tiling column): 1 = 16-row tiles, 2 = 4-row tiles, 3 = 16-row tiles with the K range split over
 * two workgroups (order-independent atomic reduction), 0 = first-generation row-major kernel, -1 = the engine's
 * choice among the unsplit kernels. Used by bench.py for the roofline of the dominant kernel and by tools/bench_kernels.py. */
q3_status q3_bench_linear(int device, int M, int N, int K, int epi, int rms, int tiled, int iters, int n_copies,
                          double* avg_us);
/* raw stream handle (hipStream_t) the session launches on */
q3_status q3_session_stream(q3_session* s, void** stream);
/* How this session's captured frame is replayed: *path = 0 nothing captured yet / eager launches, 1 = hipGraphLaunch (Q3_AQL=0),
 * 2 = the library's own AQL queue with HIP's packet headers (agent-scope fences at every kernel boundary; bit-identical to path
 * 1; Q3_AQL=1), 4 = own AQL queue, the boundaries between the frame's write-through kernels without those fences (the default
 * since round 6: same codes, ~4.5 % less time per frame; the first and last packet of every frame keep their fences), 3 = own
 * queue without ANY boundary fence — a measurement probe that gives WRONG results (state that crosses frames moves with plain
 * accesses), reachable only with Q3_AQL=2 plus the explicit opt-in Q3_AQL_UNSAFE=1 (a warning is printed);
 * *nodes = dispatch packets per frame (0 on paths 0 / 1). A graph the converter cannot take stays on path 1.
 * The reference replays nothing — every op is an eager candle launch (src/lib.rs:580-652); new here (environment: Q3_AQL). */
q3_status q3_session_submit_info(q3_session* s, int* path, int* nodes);
/* How many of the frame's dispatch packets go out WITHOUT their agent-scope acquire / release fence (paths 3 / 4 of
 * q3_session_submit_info; 0 / 0 on the other paths and before the frame is captured). On path 4 these are the packets of the kernel
 * families that move their data write-through; the first packet of a frame always acquires and the last always releases.
 * New here (the parity tests hold the policy to it). */
q3_status q3_session_submit_fences(q3_session* s, int* acquire_free, int* release_free);
/* Algorithmic HBM bytes of one frame for this session's batch at KV length L */
q3_status q3_session_frame_bytes(q3_session* s, int kv_len, double* weight_bytes, double* kv_bytes);

/* ---------------- on-disk formats (q3_io.cpp) ----------------
 * ModelType (config.rs:176-194); UNKNOWN = loaded without a usable config.json (lib.rs:383-389) */
enum { Q3_MODEL_UNKNOWN = -1, Q3_MODEL_BASE = 0, Q3_MODEL_CUSTOM_VOICE = 1, Q3_MODEL_VOICE_DESIGN = 2 };
/* TalkerConfig::default / ::custom_voice + CodePredictorConfig::default + Decoder12HzConfig::default
 * (talker.rs:176-290, code_predictor.rs:48-113, decoder_12hz.rs:14-67): variant 0 = 0.6B, 1 = 1.7B */
q3_status q3_config_default(int variant, q3_config* out);
/* ParsedModelConfig::from_file (config.rs:238-336): same keys, same unwrap_or defaults; decoder fields
 * are Decoder12HzConfig::default (the reference does not read them from config.json either, lib.rs:345) */
q3_status q3_config_from_json(const char* path, q3_config* out, int* model_type);
/* Qwen3TTS::from_pretrained minus the tokenizer (lib.rs:180-262): <dir>/config.json (optional; weight
 * inspection fallback of lib.rs:371-381), <dir>/model.safetensors, <dir>/speech_tokenizer/model.safetensors
 * (or the parent directory's), every expected tensor uploaded, q3_model_finalize run. Tensors the hot
 * path does not use (speaker_encoder.*, encoder.*) are skipped. device = -1: manifest-only handle. */
q3_status q3_model_load(const char* model_dir, int device, q3_model** out, int* model_type);
/* load_weights (lib.rs:1390-1396) into an existing handle: uploads every tensor of the file the model
 * expects (BF16 / F32 / F16 / F64 sources); n_loaded may be NULL */
q3_status q3_model_load_safetensors(q3_model* m, const char* path, int* n_loaded);
/* header lookup of one tensor: stored dtype (Q3_DTYPE_* or -1), rank and up to cap_dims dims */
q3_status q3_safetensors_info(const char* path, const char* name, int* dtype, int64_t* shape, int cap_dims, int* n_dims);
/* save_wav (audio/io.rs:143-165): mono PCM16, `(clamp(x,-1,1) * 32767) as i16` */
q3_status q3_pcm16_from_f32(const float* samples_host, int64_t n, int16_t* out_host);
q3_status q3_wav_write_pcm16(const char* path, const float* samples_host, int64_t n, uint32_t sample_rate);
/* load_wav (audio/io.rs:106-141): int PCM scaled by 2^(bits-1), float as is, channels averaged to mono.
 * out_host = NULL queries *n_samples / *sample_rate only */
q3_status q3_wav_read(const char* path, float* out_host, int64_t cap, int64_t* n_samples, uint32_t* sample_rate);
/* save_codes_binary / save_audio_binary (bin/generate_audio.rs:788-813): i64 LE frame-major codes, f32 LE samples */
q3_status q3_codes_write_bin(const char* path, const uint32_t* codes_host, int n_frames, int n_groups);
q3_status q3_codes_read_bin(const char* path, uint32_t* codes_host, int cap_frames, int n_groups, int* n_frames);
q3_status q3_audio_write_bin(const char* path, const float* samples_host, int64_t n);
/* the same dump read back — how `--compare` loads the other side's audio (generate_audio.rs:880-886); out_host == NULL or
 * cap == 0: only *n_samples */
q3_status q3_audio_read_bin(const char* path, float* out_host, int64_t cap, int64_t* n_samples);
/* audio::resample / resample_to_24k (audio/resample.rs:17-176): windowed-sinc low-pass with rubato's parameters
 * (sinc_len 128, cutoff 0.95, BlackmanHarris2), output i at input time i * sr_in / sr_out; out_host = NULL queries
 * *n_out = round(n * sr_out / sr_in). Host arithmetic (once per reference clip). */
q3_status q3_resample(const float* in_host, int64_t n, uint32_t sr_in, uint32_t sr_out, float* out_host, int64_t cap, int64_t* n_out);

/* ---------------- speaker encoder (q3_speaker.hip): x-vector voice cloning ----------------
 * SpeakerEncoder (models/speaker.rs:345-469) behind create_voice_clone_prompt (lib.rs:1132-1190): 24 kHz mono
 * reference audio -> log-mel (audio/mel.rs:47-59, 135-227) -> ECAPA-TDNN -> [enc_dim] embedding, the `xvector`
 * of q3_request. Config = SpeakerEncoderConfig (models/config.rs:100-174), weights = `speaker_encoder.*` of a
 * Base checkpoint's model.safetensors (kept f32 on the device; bf16/f16 sources are widened on upload).
 * The ICL half of the prompt (reference codes) comes from the speech encoder below (q3_mimi_*). */
typedef struct q3_spk_config {
    int32_t mel_dim;            /* 128 */
    int32_t enc_dim;            /* 1024 (0.6B) / 2048 (1.7B): the talker's hidden size */
    int32_t channels[5];        /* 512,512,512,512,1536 */
    int32_t kernel_sizes[5];    /* 5,3,3,3,1 */
    int32_t dilations[5];       /* 1,2,3,4,1 */
    int32_t attention_channels; /* 128 */
    int32_t res2net_scale;      /* 8 */
    int32_t se_channels;        /* 128 */
    int32_t sample_rate;        /* 24000 */
} q3_spk_config;
typedef struct q3_speaker_encoder q3_speaker_encoder;
/* SpeakerEncoderConfig::default (config.rs:132-174) */
q3_status q3_spk_config_default(q3_spk_config* out);
/* `speaker_encoder_config` object of config.json (config.rs:233, serde defaults per field); *present = 0 and the
 * defaults when the key is absent (CustomVoice / VoiceDesign checkpoints) */
q3_status q3_spk_config_from_json(const char* path, q3_spk_config* out, int* present);
/* SpeakerEncoder::new (speaker.rs:362-428): allocates the weight arena for the config's tensor manifest */
q3_status q3_spk_create(const q3_spk_config* cfg, int device, q3_speaker_encoder** out);
void q3_spk_free(q3_speaker_encoder* e);
q3_status q3_spk_get_config(const q3_speaker_encoder* e, q3_spk_config* out);
int q3_spk_n_tensors(const q3_speaker_encoder* e);
q3_status q3_spk_tensor_info(const q3_speaker_encoder* e, int i, const char** name, int64_t* n);
/* data_host: n elements of src_dtype (Q3_DTYPE_F32 / Q3_DTYPE_BF16) */
q3_status q3_spk_set_tensor(q3_speaker_encoder* e, const char* name, const void* data_host, int src_dtype, int64_t n);
/* every tensor present -> packs the matrix-core weight images, builds window / DFT / mel-filter tables */
q3_status q3_spk_finalize(q3_speaker_encoder* e);
/* uploads every `speaker_encoder.*` tensor of a safetensors file and finalizes; Q3_MISSING_WEIGHT with the
 * reference's hint (lib.rs:1137-1153) when the file has none */
q3_status q3_spk_load_safetensors(q3_speaker_encoder* e, const char* path);
/* frames the mel front end yields for n samples: (n + 2*384 - 1024) / 256 + 1 (mel.rs:168-199) */
int q3_spk_mel_frames(int64_t n_samples);
/* MelSpectrogram::compute_for_speaker_encoder (mel.rs:135-166): mel_host [mel_dim][T], T = q3_spk_mel_frames(n) */
q3_status q3_spk_mel(q3_speaker_encoder* e, const float* samples_host, int64_t n, float* mel_host, int64_t cap_floats, int* n_frames);
/* SpeakerEncoder::forward (speaker.rs:443-469) for one mel [mel_dim][T]; taps (test hook): NULL or 6 host pointers
 * (NULL entries skipped): blocks.0 out, the three SE-Res2Net outs [C][T], MFA out [C4][T], pooled [2*C4] */
q3_status q3_spk_forward(q3_speaker_encoder* e, const float* mel_host, int T, float* out_host, float** taps_host);
/* SpeakerEncoder::encode (speaker.rs:431-438) = mel + forward; sample_rate must be 24000 (the reference resamples
 * first, lib.rs:1156-1166 — q3_resample before calling) */
q3_status q3_spk_encode(q3_speaker_encoder* e, const float* samples_host, int64_t n, uint32_t sample_rate, float* out_host);

/* ---------------- speech-tokenizer encoder (q3_mimi.hip): ICL reference codes from raw audio ----------------
 * Encoder12Hz (models/codec/encoder_12hz.rs:34-144) behind create_voice_clone_prompt's ICL branch (lib.rs:1172-1178):
 * 24 kHz mono reference audio -> SEANet encoder -> 8-layer transformer -> stride-2 conv -> split residual VQ ->
 * frames of 16 codebook indices at 12.5 Hz, the `ref_codes` of q3_request. The reference instantiates candle-transformers'
 * Mimi (`mimi::Config::v0_1(Some(16))`, encoder_12hz.rs:73) over the HF-format `encoder.*` keys of
 * speech_tokenizer/model.safetensors; this is the published Mimi encoder for that format (weights kept f32 on the device). */
typedef struct q3_mimi_config {
    int32_t n_filters;      /* 64   SEANet base width */
    int32_t hidden;         /* 512  SEANet output / transformer width */
    int32_t ratios[4];      /* 4, 5, 6, 8: strides of the four down-sampling stages, in encoder order (24 kHz -> 25 Hz) */
    int32_t kernel;         /* 7 */
    int32_t res_kernel;     /* 3 */
    int32_t last_kernel;    /* 3 */
    int32_t compress;       /* 2 */
    int32_t n_layers;       /* 8 */
    int32_t n_heads;        /* 8 */
    int32_t head_dim;       /* 64 */
    int32_t inter;          /* 2048 */
    int32_t window;         /* 250: causal attention context in frames */
    int32_t cb_size;        /* 2048 */
    int32_t cb_dim;         /* 256 */
    int32_t n_q;            /* 16 codebooks: */
    int32_t n_sem;          /* 1 semantic + 15 acoustic */
    float norm_eps;         /* 1e-5 */
    float rope_theta;       /* 1e4 */
} q3_mimi_config;
typedef struct q3_speech_encoder q3_speech_encoder;
/* mimi::Config::v0_1(Some(16)) (encoder_12hz.rs:73) */
q3_status q3_mimi_config_default(q3_mimi_config* out);
/* Encoder12Hz::from_weights (encoder_12hz.rs:54-117): allocates the weight arena for the config's tensor manifest; device -1 = manifest only */
q3_status q3_mimi_create(const q3_mimi_config* cfg, int device, q3_speech_encoder** out);
void q3_mimi_free(q3_speech_encoder* e);
q3_status q3_mimi_get_config(const q3_speech_encoder* e, q3_mimi_config* out);
int q3_mimi_n_tensors(const q3_speech_encoder* e);
q3_status q3_mimi_tensor_info(const q3_speech_encoder* e, int i, const char** name, int64_t* n);
/* data_host: n elements of src_dtype (Q3_DTYPE_F32 / Q3_DTYPE_BF16), checkpoint layout (the strided convs are re-laid on upload) */
q3_status q3_mimi_set_tensor(q3_speech_encoder* e, const char* name, const void* data_host, int src_dtype, int64_t n);
q3_status q3_mimi_finalize(q3_speech_encoder* e);
/* Encoder12Hz::from_safetensors (encoder_12hz.rs:45-48): every `encoder.*` tensor of speech_tokenizer/model.safetensors, then
 * finalize; Q3_MISSING_WEIGHT "No encoder keys found …" (encoder_12hz.rs:68-70) when the file has none */
q3_status q3_mimi_load_safetensors(q3_speech_encoder* e, const char* path);
/* frames for n samples: ceil over the four strides, then ceil(/2) */
int q3_mimi_frames(const q3_mimi_config* cfg, int64_t n_samples);
/* Encoder12Hz::encode (encoder_12hz.rs:119-144): codes_host [T][n_q] u32 (frame-major, like q3_request.ref_codes);
 * codes_host == NULL: only *n_frames. sample_rate must be 24000 (q3_resample first). taps_host (test hook): NULL or 3 host
 * pointers (NULL entries skipped): SEANet out [hidden][T25], transformer out [hidden][T25], down-sampled [hidden][T] */
q3_status q3_mimi_encode(q3_speech_encoder* e, const float* samples_host, int64_t n, uint32_t sample_rate, uint32_t* codes_host,
                         int cap_frames, int* n_frames, float** taps_host);

/* ---------------- data parallelism without torch.distributed (q3_dp.cpp) ----------------
 * one process per GPU, utterance i -> rank i mod N, exactly one collective on the data path — the
 * broadcast of rank 0's weight arena over RCCL (xGMI) — plus an all-gather of a few doubles for end-of-job timing. The
 * reference has no multi-GPU code to replace (per-call KV / RNG / masks, lib.rs:744-756, make utterances independent);
 * these entry points are what its Rust host would call where bench.py uses torch.distributed. RCCL is dlopen'ed on the
 * first call. Rendezvous = the host ships the 128-byte id from rank 0 to the other ranks by its own means
 * (file, TCP, MPI, environment). */
#define Q3_DP_ID_BYTES 128
typedef struct q3_dp_comm q3_dp_comm;
/* ncclGetUniqueId: call on rank 0, hand the bytes to every rank */
q3_status q3_dp_unique_id(void* id_out /* Q3_DP_ID_BYTES */);
/* ncclCommInitRank on `device` (collective: every rank of the job must call it) */
q3_status q3_dp_init(int rank, int world, const void* id, int device, q3_dp_comm** out);
void q3_dp_free(q3_dp_comm* c);
q3_status q3_dp_info(const q3_dp_comm* c, int* rank, int* world);
/* ncclBroadcast of the whole weight arena from `root` (collective); non-root ranks are marked loaded and must then
 * call q3_model_finalize themselves. world == 1: the broadcast degenerates to a self-copy RCCL performs in place. */
q3_status q3_dp_broadcast_weights(q3_dp_comm* c, q3_model* m, int root);
/* ncclAllGather of n doubles per rank: out_host [world][n] (timings, frame counts) */
q3_status q3_dp_allgather_f64(q3_dp_comm* c, const double* in_host, int n, double* out_host);

#ifdef __cplusplus
}
#endif
#endif /* Q3TTS_H */
