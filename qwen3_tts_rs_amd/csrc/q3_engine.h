// q3_engine.h — what the translation units of the engine share (not part of the C ABI; include/q3tts.h is).
// Round 6: q3_engine.hip (3 800 lines: model, arena, KV pool, session, graph capture, batcher, test entry points) was split into
//   q3_model.hip     errors, synthetic tensors, the weight manifest / arena, q3_model_* (create, set_tensor, finalize, KV pool API)
//   q3_codec_run.hip device-memory cache, the vocoder pipeline (codec_decode_dev) and q3_decode_codes
//   q3_session.hip   sessions: KV paging, the talker / code-predictor step, frame capture + own-queue submission, prefill, generate,
//                    streaming chunks, q3_session_run / decode / get
//   q3_batcher.hip   continuous batching: q3_session_replace (side prefill + transplant) and the native batcher q3_batcher_*
//   q3_testapi.hip   low-level entry points of the parity tests and of bench.py's roofline replays
// Together they are the host side of libq3tts.so: weight arena, sessions (KV pages, RNG streams, penalty masks), the per-frame
// launch sequence (captured once, replayed from the library's own AQL queue), the codec-decoder pipeline and the C ABI declared in
// include/q3tts.h. Layer map of the reference they replace (paths relative to the reference repo):
//   src/lib.rs:530-656 generate_codes, 425-501 synthesize_with_timing, 718-784 / 1484-1541 per-call state, 1484-1782 StreamingSession
//   src/models/talker.rs:451-627 prefill builders, 716-736 generate_step_with_embed
//   src/models/code_predictor.rs:320-416 generate_acoustic_codes
//   src/models/codec/decoder_12hz.rs:411-505 decode
// No behaviour change: the functions moved as they were; the ones another unit calls lost their `static` and are declared below
// (hidden visibility: the .so exports exactly what include/q3tts.h declares).
#pragma once
#include "../../include/q3tts.h"
#include "q3_kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "q3_internal.h"
#include "q3_aql.h"

using namespace q3;
#define Q3_HIDDEN __attribute__((visibility("hidden")))

// ------------------------------------------------------------------------------------------------
// errors (q3_model.hip): thread-local message behind q3_last_error()
// ------------------------------------------------------------------------------------------------
Q3_HIDDEN q3_status set_err(q3_status st, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// Every host thread that calls into the engine runs its HIP calls in RELAXED stream-capture mode: a session captures its frame
// (kernel launches only, on its own non-blocking stream) while other threads — the batcher's prefill worker, a server's other
// sessions — allocate, hipMemcpy and synchronise; with the default (global) thread mode this HIP runtime fails THOSE calls with
// "operation not permitted when stream is capturing" whatever mode the capture itself was begun in.
static inline void q3_relax_capture_mode() {
    static thread_local bool done = false;
    if (!done) { hipStreamCaptureMode m = hipStreamCaptureModeRelaxed; (void)hipThreadExchangeStreamCaptureMode(&m); done = true; }
}
// ... and that is not enough on this runtime: an operation on the LEGACY (null) stream — a synchronous hipMemcpy, the zero-fills of
// DevPool, hipDeviceSynchronize — issued by one thread while another thread's stream is capturing fails with "operation would
// make the legacy stream depend on a capturing blocking stream" (and invalidates the capture), non-blocking capturing stream and
// relaxed modes notwithstanding (tests/test_frame_submission.py: two sessions on two threads). So the two exclude each other:
// a capture (q3_session.hip frame_capture: ~1 ms of kernel launches, once per session) holds this lock exclusively, every
// legacy-stream operation of the engine holds it shared (the q3_hip* wrappers below; the engine's units call nothing else).
#include "q3_capture_lock.h"
#define HIPC(expr)                                                                                        \
    do {                                                                                                  \
        q3_relax_capture_mode();                                                                          \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return set_err(Q3_HIP_ERROR, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                                     \
    } while (0)
#define Q3C(expr)                         \
    do {                                  \
        q3_status s_ = (expr);            \
        if (s_ != Q3_OK) return s_;       \
    } while (0)

// token ids (talker.rs:30-55)
enum { IM_START = 151644, ASSISTANT = 77091, NEWLINE = 198, TTS_PAD = 151671, TTS_BOS = 151672, TTS_EOS = 151673 };
enum { CODEC_PAD = 2148, CODEC_BOS = 2149, CODEC_EOS = 2150, CODEC_THINK = 2154, CODEC_THINK_BOS = 2156, CODEC_THINK_EOS = 2157 };

static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32_host(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
enum SlotKind { SK_PLAIN = 0, SK_TRANSCONV = 1, SK_TILED = 2 };
struct Slot {
    std::string name; int64_t n = 0; int stored = Q3_DTYPE_F32; size_t offset = 0; bool loaded = false;
    int kind = SK_PLAIN; int tc_cin = 0, tc_cout = 0, tc_k = 0, tc_stride = 0;
    int rows = 0, cols = 0;      // SK_TILED: logical [rows][cols]; 16-row-tile image at `offset`,
    size_t offset2 = 0; bool dual = false;   // plus a 4-row-tile image at `offset2` when the projection is narrow
};
// a GEMV weight: 16-row-tile image (t1) and, for narrow projections, a 4-row-tile image (t2)
struct TW { const uint16_t* t1 = nullptr; const uint16_t* t2 = nullptr; };
struct LayerW {
    const float *in_ln, *q_norm, *k_norm, *post_ln;
    TW qkv, o, gate, up, down;
};

struct DecLayerW { const float *in_ln, *q, *k, *v, *o, *attn_scale, *post_ln, *gate, *up, *down, *mlp_scale; };
struct ResUnitW { const float *a1, *ib1, *c1w, *c1b, *a2, *ib2, *c2w, *c2b; };
struct DecBlockW { const float *a, *ib, *tw, *tb; ResUnitW res[3]; int cin, cout, rate; };
struct UpW { const float *tw, *tb, *dww, *dwb, *nw, *nb, *p1w, *p1b, *p2w, *p2b, *gamma; int ratio; };

// ------------------------------------------------------------------------------------------------
// Paged talker KV (north_star: "in-place paged KV in 288 GB HBM3E"; replaces the per-call preallocated cache of
// kv_cache.rs:234-310 and its overflow bail :293-300). One pool per model: pages of KV_PAGE_POS positions x every layer
// and KV head (q3_kernels.h), carved from 32-page slabs that are hipMalloc'ed on demand and kept for the model's lifetime. Sessions
// take pages as their rows cross page boundaries and hand them back when a row is replaced or the session ends, so a
// 4k-position prompt and a ten-position prompt draw on the same memory, and a continuous-batching swap RELINKS the
// prefilled pages of the side session into the row instead of copying extents. Pages are never cleared: the attention
// kernels read a position only after it was written.
// ------------------------------------------------------------------------------------------------
// One page budget for both pools of a model (q3_model_kv_pool_limit), in UNITS of half an f32 page: an f32 page costs 2, a
// bf16 page (same geometry, 2-byte elements) 1 — so the documented limit, the Q3_KV_OVERFLOW bail and the occupancy figures hold
// for bf16 sessions and for the f32 pages their prompts are prefilled into alike.
struct KvBudget {
    std::mutex mu;
    long limit = 0, used = 0, peak = 0;            // units; limit 0 = bounded by HBM only
    bool fits(long units) { std::lock_guard<std::mutex> g(mu); return limit <= 0 || used + units <= limit; }
};
struct KvPool {
    // Slab layout (layer-major, so that ONE layer's K/V of every page of a slab sits in one contiguous run — the attention
    // launch of a layer touches SLAB_SLOTS x 512 KB = 16 MB runs instead of one 64 KB run per (page, head) spread 29 MB
    // apart, which cost ~1 us of address translation per launch at B = 8: the frame was 0.9 % slower than with contiguous
    // extents):   slab[n_layers][2 (K, V)][SLAB_SLOTS][nkv][KV_PAGE_POS][HEAD_DIM] f32 (+ padding, below).
    // A page = one slot of a slab, named by the address of its layer-0 K run; layer l is `l * layer_stride()` floats further,
    // V `v_delta()` floats behind K.
    static constexpr int SLAB_SLOTS = 32;
    std::mutex mu;
    size_t run_floats = 0; int n_layers = 0;       // run = nkv * KV_PAGE_POS * HEAD_DIM ELEMENTS (one layer's K of one page)
    size_t elem_bytes = sizeof(float);              // 4, or 2 for the pool of bf16 sessions (same geometry in elements)
    std::vector<void*> slabs; std::vector<float*> free_pages;
    int total = 0, in_use = 0;                      // pages of this pool
    KvBudget* budget = nullptr; int unit = 2;       // the model's shared budget and what one page of this pool costs of it
    // K -> V and layer -> layer distances are kept OFF powers of two (17 KB of padding behind every region): a lane asks for
    // the K row and the V row of a position together, and at exactly 16 MB apart the two requests meet in the same memory
    // channel (k_attn_fused 7.6 vs 6.9 us per launch at B = 8 against the contiguous caches, whose distance is arbitrary)
    static constexpr size_t PAD_FLOATS = 17 * 256;
    size_t page_bytes() const { return (size_t)2 * n_layers * run_floats * elem_bytes; }
    size_t v_delta() const { return (size_t)SLAB_SLOTS * run_floats + PAD_FLOATS; }
    size_t layer_stride() const { return 2 * v_delta(); }
    size_t slab_bytes() const { return (size_t)n_layers * layer_stride() * elem_bytes; }
    // n pages or none: hipErrorOutOfMemory when the limit (q3_model_kv_pool_limit) or the device says no
    hipError_t take(int n, std::vector<float*>& out) {
        std::lock_guard<std::mutex> g(mu);
        if (n <= 0) return hipSuccess;
        {
            std::lock_guard<std::mutex> gb(budget->mu);
            if (budget->limit > 0 && budget->used + (long)n * unit > budget->limit) return hipErrorOutOfMemory;
        }
        while ((int)free_pages.size() < n) {          // whole slabs (~1 GB at 28 layers x 8 KV heads), kept for the model's lifetime: no hipMalloc in steady state
            void* slab = nullptr;
            if (hipMalloc(&slab, slab_bytes()) != hipSuccess) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
            slabs.push_back(slab);
            for (int i = SLAB_SLOTS - 1; i >= 0; --i) free_pages.push_back((float*)((char*)slab + (size_t)i * run_floats * elem_bytes));
            total += SLAB_SLOTS;
        }
        for (int i = 0; i < n; ++i) { out.push_back(free_pages.back()); free_pages.pop_back(); }
        in_use += n;
        std::lock_guard<std::mutex> gb(budget->mu);      // (sessions of one model may run on several host threads: two takers can
        budget->used += (long)n * unit;                  //  overshoot the limit by one request between the check and here; the
        if (budget->used > budget->peak) budget->peak = budget->used;      // limit is an admission bound, not a hard allocator wall)
        return hipSuccess;
    }
    void give(std::vector<float*>& pages) {
        std::lock_guard<std::mutex> g(mu);
        for (float* p : pages) free_pages.push_back(p);
        in_use -= (int)pages.size();
        { std::lock_guard<std::mutex> gb(budget->mu); budget->used -= (long)pages.size() * unit; }
        pages.clear();
    }
    // slabs none of whose pages is held go back to the device (q3_model_kv_pool_trim); returns the bytes freed
    size_t trim() {
        std::lock_guard<std::mutex> g(mu);
        size_t freed = 0;
        for (size_t i = 0; i < slabs.size();) {
            char* lo = (char*)slabs[i]; char* hi = lo + (size_t)SLAB_SLOTS * run_floats * elem_bytes;      // the slab's layer-0 K runs name its pages
            int n_free = 0;
            for (float* p : free_pages) n_free += ((char*)p >= lo && (char*)p < hi) ? 1 : 0;
            if (n_free < SLAB_SLOTS) { ++i; continue; }
            free_pages.erase(std::remove_if(free_pages.begin(), free_pages.end(), [&](float* p) { return (char*)p >= lo && (char*)p < hi; }), free_pages.end());
            (void)hipFree(slabs[i]); slabs.erase(slabs.begin() + (long)i);
            total -= SLAB_SLOTS; freed += slab_bytes();
        }
        return freed;
    }
    // the first slab ahead of the first request (q3_model_finalize): its hipMalloc (~1 GB) is then not on a session's time to first audio
    hipError_t prewarm() {
        std::vector<float*> one;
        {
            std::lock_guard<std::mutex> g(mu);
            if (!slabs.empty()) return hipSuccess;
        }
        const hipError_t e = take(1, one);
        if (e == hipSuccess) give(one);
        return e;
    }
    ~KvPool() { for (void* s : slabs) (void)hipFree(s); }
};

struct q3_model {
    q3_config cfg{};
    int device = 0;
    int codec_planes = 3;  // q3_model_set_codec_planes: 3 = f32-exact bf16x3 products in the vocoder's convs, 2 = the two leading planes
    KvBudget kv_budget;    // one limit / occupancy for both pools below, in half-f32-page units
    KvPool kv_pool;
    KvPool kv_pool16;      // pages of bf16 sessions (q3_session_set_kv_dtype): the same geometry with 2-byte elements
    // sessions hold pages, streams and weights of their model: the handle itself is ONE reference and every live session another;
    // whoever drops the count to zero destroys the model (q3_model_free with sessions still alive only gives up the handle's
    // reference — a host that tears down in the wrong order must not crash — and nobody touches *m after its own decrement)
    std::atomic<int> refs{1};
    std::vector<Slot> slots;
    std::unordered_map<std::string, int> index;
    char* arena = nullptr; size_t arena_bytes = 0;
    bool finalized = false;
    // derived device buffers
    float *rope_cos = nullptr, *rope_sin = nullptr; int rope_len = 0;      // talker/CP (theta, hd 128)
    float* derived = nullptr;                                              // codebooks + snake tables
    // bf16x3-packed copies of the vocoder's conv / linear weights (launch_pack_conv_w), keyed by the f32 pointer
    std::unordered_map<const float*, const void*> wpk; void* wpk_arena = nullptr;
    const void* pk(const float* w) const { auto it = wpk.find(w); return it == wpk.end() ? nullptr : it->second; }
    // frame-loop streams of freed sessions, reused by the next q3_session_create: creating a priority stream costs 1.6 ms and
    // destroying one 1.1 ms — 8 % of a streaming session's time to first audio, more than its whole prefill
    std::mutex stream_mu; std::vector<hipStream_t> idle_streams;
    const float* first_cb = nullptr; const float** rest_cbs_dev = nullptr; // device array of 15 pointers
    const uint16_t** cp_embs_dev = nullptr;                                // device array of 15 pointers
    // 1.7B: small_to_mtp_projection applied once to every row of the 15 acoustic embedding tables and of the talker's
    // codec embedding (code_predictor.rs:337-345, 386-396 project the gathered row on every pass): f32 [rows][cp_hidden]
    float* proj_tabs = nullptr; const float* cp_proj[15] = {}; const float* sem_proj = nullptr;
    // layer-0 q|k|v of every such row (input RMSNorm + qkv projection of code-predictor layer 0): f32 [rows][qkv dim]
    float* qkv0_tabs = nullptr; const float* cp_qkv0[15] = {}; const float* sem_qkv0 = nullptr;
    // resolved pointers
    const uint16_t *text_emb, *codec_emb;
    TW fc1w, fc2w, codec_head, mtp_w;
    const float *fc1b, *fc2b, *norm, *mtp_b, *cp_norm;
    std::vector<LayerW> tl, cl;
    std::vector<const uint16_t*> cp_emb; std::vector<TW> cp_head;
    const float *first_proj, *rest_proj, *pre_w, *pre_b, *inp_w, *inp_b, *outp_w, *outp_b, *dec_norm;
    std::vector<DecLayerW> dl;
    UpW up[2]; const float *init_w, *init_b; DecBlockW blk[4];
    const float *fin_a, *fin_ib, *fin_w, *fin_b;
};

static inline int up16(int v) { return (v + 15) & ~15; }
static inline int up32(int v) { return (v + 31) & ~31; }
static inline int up4(int v) { return (v + 3) & ~3; }
static inline int up128(int v) { return (v + 127) & ~127; }
// Narrow projections (fewer than 4096 output rows) keep BOTH tilings resident — HBM capacity is not the
// constraint, launch latency is: the 4-row-tile kernel fills the chip for M <= 2 tokens (and for N <= 1024 up
// to M = 8), the 16-row-tile kernel is cheaper per token for larger batches.
static inline bool dual_tiled(int n_total) { return n_total < 4096; }
static inline int kpad_for(int mode, int K) { return mode == 2 ? up128(K) : up32(K); }
static inline size_t tiled_elems(int mode, int rows, int cols) {
    return mode == 2 ? (size_t)up4(rows) * up128(cols) : (size_t)up16(rows) * up32(cols);
}

// short K with many rows (code-predictor / 0.6B gate-up 3072 x 1024, lm_head 2048 x 1024): the 16-row kernel already has
// one whole-slice group per wave and >= 128 workgroups, and beats the 4-row tiles even at M = 1 (gate/up 5.3 vs 6.5 us)
static inline bool short_k_wide(int N, int K) { return K <= 1024 && N >= 2048; }
static inline int pick_mode(const TW& w, int M, int N, int K) {
    static const bool no4 = getenv("Q3_GEMV_NO_MFMA4") != nullptr;     // tuning aid: 16-row tiles wherever both images exist (M > 2)
    if (w.t2 && !w.t1) return 2;
    if (M > 16 && w.t1) return 1;             // wide batches: k_gemv_wide works on the 16-row tiles
    if (no4 && w.t1 && M > 2) return 1;
    if (w.t2 && !short_k_wide(N, K) && (M <= 2 || (N <= 1024 && M <= 8))) return 2;
    return 1;
}
static inline void set_w(LinArgs& a, const TW& w, int M, int N, int K) {
    a.tiled = pick_mode(w, M, N, K); a.W = a.tiled == 2 ? w.t2 : w.t1; a.Kpad = kpad_for(a.tiled, K);
}
static inline void set_w2(LinArgs& a, const TW& w, const TW& w2, int M, int N, int K) {
    set_w(a, w, M, N, K); a.W2 = a.tiled == 2 ? w2.t2 : w2.t1;
}
// session-lifetime device blocks from the per-device size-class cache (q3_codec_run.hip)
Q3_HIDDEN hipError_t dev_malloc(void** p, size_t bytes);
Q3_HIDDEN void dev_free(void* p);

struct DevPool {
    std::vector<void*> ptrs;
    // lazy: the zero-fills of a run of allocations are only ISSUED (null stream, in order with the synchronous hipMemcpy
    // uploads that may follow) and settle() waits for all of them once — a session is ~45 buffers, and a memset + a device
    // synchronisation each was 2-3 ms of every q3_session_create (the side session of a continuous-batching swap)
    bool lazy = false;
    hipError_t settle() { return q3_null_stream_sync(); }
    template <typename T> hipError_t alloc(T** p, size_t count) {
        void* q = nullptr;
        hipError_t e = dev_malloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return e;
        ptrs.push_back(q); *p = (T*)q;
        // zero-fill on the null stream and WAIT for it: the users launch on non-blocking streams, which the null stream
        // does not order against — a memset still in flight would wipe what their first kernels write (seen once the
        // blocks started coming from the cache instead of a slow hipMalloc)
        return q3_null_stream_memset(q, (count ? count : 1) * sizeof(T), !lazy);
    }
    void release_all() {
        if (lazy) (void)q3_null_stream_sync();      // a creation that failed midway: no zero-fill may outlive its block's ownership
        for (void* p : ptrs) dev_free(p);
        ptrs.clear(); lazy = false;
    }
    ~DevPool() { release_all(); }
};

// ------------------------------------------------------------------------------------------------
// codec decoder pipeline
// ------------------------------------------------------------------------------------------------
// Left context (frames) the convolutional stack needs for its output to be independent of where it was started:
// propagating the causal receptive field back from the PCM (final k7: 6 samples; per block 6*(1+3+9) = 78 samples
// of residual units + 1 input sample of the 2-tap polyphase transposed conv; decoder.0 k7; two ConvNeXt dwconv7
// + 1-tap transposed convs) gives 29 @640/frame -> 28 @160 -> 23 @32 -> 14 @4 -> 20 -> 26 @4 -> 13 @2 -> 19 -> 10 frames.
constexpr int CODEC_CTX_FRAMES = 12;

struct CodecWS {
    int cap_frames = 0;        // frames the convolutional stack can take in one call
    int cap_front = 0;         // frames the quantiser / pre-transformer front can take (>= cap_frames)
    float *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *bufF = nullptr, *bufD = nullptr, *bufE = nullptr;
    float *cs = nullptr, *sn = nullptr;
    uint32_t* frames = nullptr; float* pcm = nullptr;
    void release() {
        dev_free(bufA); dev_free(bufB); dev_free(bufC); dev_free(bufF); dev_free(bufD); dev_free(bufE); dev_free(cs); dev_free(sn); dev_free(frames); dev_free(pcm);
        bufA = bufB = bufC = bufF = bufD = bufE = cs = sn = pcm = nullptr; frames = nullptr; cap_frames = 0; cap_front = 0;
    }
};

// ------------------------------------------------------------------------------------------------
// session
// ------------------------------------------------------------------------------------------------
struct LmBuf { float *X, *SUM, *QKV, *Q, *ATT, *ACT, *PART; };
struct LmDims { int H, I, nh, nkv, layers; float eps; };

// a request with its arrays owned (queued tickets of the batcher; the rows of a ragged first batch until they are prefilled)
struct BatReq {
    q3_request r{}; std::vector<uint32_t> text, instruct, ref_codes, ref_text; std::vector<float> xvec;
    void own(const q3_request& q, int hidden) {
        r = q;
        text.assign(q.text_ids, q.text_ids + (q.text_ids ? q.n_text : 0));
        instruct.assign(q.instruct_ids, q.instruct_ids + (q.instruct_ids ? q.n_instruct : 0));
        ref_codes.assign(q.ref_codes, q.ref_codes + (q.ref_codes ? (size_t)q.n_ref * 16 : 0));
        ref_text.assign(q.ref_text_ids, q.ref_text_ids + (q.ref_text_ids ? q.n_ref_text : 0));
        if (q.xvector) xvec.assign(q.xvector, q.xvector + hidden);
        fix();
    }
    void fix() {      // pointers into this object's own storage (after a move of the object)
        r.text_ids = text.empty() ? nullptr : text.data(); r.n_text = (int32_t)text.size();
        r.instruct_ids = instruct.empty() ? nullptr : instruct.data(); r.n_instruct = (int32_t)instruct.size();
        r.ref_codes = ref_codes.empty() ? nullptr : ref_codes.data(); r.n_ref = (int32_t)(ref_codes.size() / 16);
        r.ref_text_ids = ref_text.empty() ? nullptr : ref_text.data(); r.n_ref_text = (int32_t)ref_text.size();
        r.xvector = xvec.empty() ? nullptr : xvec.data();
    }
};
struct SeqInfo {
    q3_request req; std::vector<uint32_t> text, instruct, ref_codes, ref_text; std::vector<float> xvec; bool icl = false;
    int prefill_len = 0, trailing_len = 0, row_base = 0, n_rows = 0, trail_base = 0, pad_row = 0;
    int n_frames = 0; bool done = false;
    // rows end at different frames: a row generates at most `limit` frames (its own max_length) counted from session frame
    // `start_run` (0, or the session's frame count when the row was swapped in: q3_session_replace)
    int start_run = 0, limit = 0, stream_pos = 0;
    bool idle = false;      // frozen by session_idle_row: holds one page (the one its frozen position lies in), takes no more
};

struct ProfAcc { double ms = 0; double bytes = 0; long launches = 0; };
struct ProfShape { int M, N, K, epi, rms, produce, tiled, count; };

struct q3_session {
    q3_model* m = nullptr; int B = 0;
    hipStream_t stream = nullptr; bool owns_stream = true;     // the side session of a swap borrows its host's stream
    DevPool pool;
    std::vector<SeqInfo> seq;
    q3_options opts{};
    int max_frames = 0, max_seq = 0, prefill_len = 0, n_splits = 1;
    int* limit = nullptr;                 // [B] per-row frame limits on the device (SampleArgs::limit)
    SampleRow* sample_rows = nullptr;     // [B] per-row sampling options on the device (SampleArgs::rows)
    int row_cap = 0, repl_base = 0;       // text-row slots of replacement rows: slot b = rows repl_base + b*row_cap .. (q3_session_replace)
    LmBuf tb{}, cb{};
    float *LASTH = nullptr, *LOGITS = nullptr, *CP_IN = nullptr, *CP_LOGITS = nullptr;
    float* wide_ws = nullptr; size_t wide_ws_bytes = 0;       // slice sums of the wide-session GEMM (B > 16; q3_kernels_wide.hip)
    float *kcache = nullptr, *vcache = nullptr, *ckcache = nullptr, *cvcache = nullptr;
    size_t kv_layer_stride = 0, ckv_layer_stride = 0;
    // paged talker KV (the default; Q3_KV_CONTIGUOUS=1 keeps one extent per row: A/B aid): kv_table[b][KV_MAX_PAGES] page
    // pointers on the device (what the attention kernels read), kv_rows[b] = the pages row b holds, in position order
    bool paged = false; unsigned long long* kv_table = nullptr; std::vector<std::vector<float*>> kv_rows;
    // ragged first batch (q3_session_create with rows of different prefill lengths / prompt kinds): the session was opened on
    // idle rows, these are the real requests; q3_session_prefill prefills them in groups of equal prefill length and moves each
    // row in (transplant_row), exactly as a continuous-batching swap would
    std::vector<BatReq> ragged;
    int kv_overflow_row = -1;             // the row whose page request the pool refused (kv_reserve_row): the batcher fails that row alone
    // bf16 K/V (opt-in, q3_session_set_kv_dtype; the reference GPU path's cache dtype): the prompt is prefilled into f32 pages
    // as always, converted once into pages of the bf16 pool (kv_in_bf16 from then on), and the decode attention reads / appends bf16
    bool kv_bf16 = false, kv_in_bf16 = false; unsigned long long* kv_conv = nullptr;      // kv_conv: [2][B * KV_MAX_PAGES] page lists of the conversion launch
    float *rows = nullptr, *embeds = nullptr, *xvec = nullptr; int n_rows_total = 0;
    // prefill scratch, session-lifetime (no hipMalloc / hipFree and no extra stream syncs on the time-to-first-audio path)
    uint32_t* ids_dev = nullptr; int *tr_dev = nullptr, *ci_dev = nullptr; float *proj_e = nullptr, *proj_h = nullptr;
    uint32_t* ref_codes_dev = nullptr;
    int *trail_base = nullptr, *trail_len = nullptr, *pad_row = nullptr;
    uint32_t* tok = nullptr; uint8_t* seen = nullptr; int *frame_idx = nullptr, *pos = nullptr, *token_count = nullptr;
    float* U = nullptr; uint32_t* codes = nullptr;
    float* logits_hist = nullptr; float* cp_logits_hist = nullptr; bool debug = false;
    bool prefilled = false; int frames_run = 0;
    hipGraphExec_t graph_exec = nullptr; hipGraph_t graph = nullptr;
    // the captured frame as a packet program on the library's own AQL queue (q3_aql.cpp); nullptr: frames replay through
    // hipGraphLaunch.  aql_mode: 0 = off, 1 = HIP's header policy (agent-scope fences on every packet), 2 = no fences
    q3::AqlProgram* aql = nullptr; int aql_mode = 0; bool aql_tried = false, aql_failed = false;
    bool precapture = false;       // q3_session_prefill captures the frame while the prompt's kernels run (set by the callers that will replay it)
    CodecWS cws;
    // overlapped segment decode (q3_session_run): vocoder segments run on their own stream while the frame loop continues
    hipStream_t dec_stream = nullptr; hipEvent_t dec_ev = nullptr;
    std::vector<CodecWS> par_ws; std::vector<hipStream_t> par_streams;     // q3_session_run: utterances vocoded side by side
    CodecWS seg_ws;                                          // q3_session_run: segments vocoded beside the frame loop
    std::vector<uint32_t> codes_host; bool codes_host_valid = false;
    int stream_pos = 0;    // streaming: frames already decoded
    int stream_mode = 0;   // 0 = context-free chunk decode (reference behaviour), 1 = continuous (left context re-run: seamless)
    bool profile = false; ProfAcc prof_linear;
    bool legacy_attn = getenv("Q3_LEGACY_ATTN") != nullptr;   // A/B aid: three-kernel attention path
    bool proj_tables = getenv("Q3_NO_PROJ_TABLES") == nullptr;   // A/B aid: set to project the gathered embedding on every pass
    bool qkv_tables = getenv("Q3_NO_QKV_TABLES") == nullptr;     // A/B aid: set to run the layer-0 qkv GEMV on every pass
    bool ksplit = getenv("Q3_NO_KSPLIT") == nullptr;          // A/B aid: set to keep o-proj / down-proj on the unsplit kernels
    bool cp_attn = getenv("Q3_NO_CP_ATTN") == nullptr;        // A/B aid: set to run the code predictor on the generic k_attn_fused
    bool no_chunk = getenv("Q3_NO_CHUNK") != nullptr;         // A/B aid: one position per prefill step, 16-pass code predictor
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events; std::vector<double> prof_event_bytes;
    std::vector<hipEvent_t> prof_pool; size_t prof_pool_next = 0;
    std::vector<ProfShape> prof_shapes;
#ifdef Q3_TRACE
    // development builds only (q3_kernels.h, Q3_TRACE): per-node stamp slices of the captured frame graph
    unsigned long long* trace_buf = nullptr; int trace_node = 0, trace_cap = 0;
    struct TraceMeta { int kind, a, b, c, d, e, f; };       // kind 0 linear (M, N, K, epi, rms, tiled) / 1 attn_cp / 2 attn_fused / 3 attn_merge (B, nh, splits, pos)
    std::vector<TraceMeta> trace_meta;
    unsigned long long* trace_next(int kind, int a, int b, int c, int d = 0, int e = 0, int f = 0) {
        if (!trace_buf || trace_node >= trace_cap) return nullptr;
        trace_meta.push_back({kind, a, b, c, d, e, f});
        return trace_buf + (size_t)(trace_node++) * TRACE_NODE;
    }
#endif
    ~q3_session();
};


// ------------------------------------------------------------------------------------------------
// functions one unit defines and another calls
// ------------------------------------------------------------------------------------------------
// q3_model.hip
Q3_HIDDEN void model_destroy(q3_model* m);
Q3_HIDDEN void retile_bf16(const uint16_t* src, int N, int K, uint16_t* dst, int mode);
// q3_codec_run.hip
Q3_HIDDEN q3_status codec_reserve(const q3_model* m, CodecWS& ws, int T, int Tf = 0);
Q3_HIDDEN int samples_per_frame(const q3_config& c);
Q3_HIDDEN q3_status codec_decode_dev(const q3_model* m, CodecWS& ws, int T, hipStream_t st, float** taps, int c0 = 0);
// q3_session.hip
Q3_HIDDEN hipError_t sync_frames(q3_session* s);
Q3_HIDDEN q3_status kv_reserve_row(q3_session* s, int b, int n_pos);
Q3_HIDDEN q3_status kv_reserve_frames(q3_session* s, int frames);
Q3_HIDDEN void kv_release_row(q3_session* s, int b);
Q3_HIDDEN LmDims talker_dims(const q3_config& c);
Q3_HIDDEN LmDims cp_dims(const q3_config& c);
Q3_HIDDEN q3_status talker_step(q3_session* s, const int* pos_dev, int pos_static, bool with_head, int rows_per_seq = 1);
Q3_HIDDEN SampleRow sample_row(const q3_options& o);
Q3_HIDDEN void request_shape(const q3_request& r, int* prefill_len, int* limit);
Q3_HIDDEN long row_worst_units(int prefill_len, int limit, bool bf16);
Q3_HIDDEN q3_request idle_request(int chunk_frames);
Q3_HIDDEN q3_status session_create(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out,
                                   hipStream_t borrow = nullptr);
Q3_HIDDEN q3_status lm_layer(q3_session* s, const LmDims& d, const LayerW& w, LmBuf& b, float* kc, float* vc, int max_seq,
                             const int* pos_dev, int pos_static, int n_splits, int rows_per_seq = 1, bool skip_qkv = false,
                             const CpGatherArgs* fold = nullptr, int paged_layer = -1);
// q3_batcher.hip
Q3_HIDDEN q3_status transplant_row(q3_session* s, int b, q3_session* side, int j, int limit);
Q3_HIDDEN q3_status transplant_check(q3_session* s, q3_session* side, int j, int limit_req, int* limit_out);
