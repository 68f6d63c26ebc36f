// q3_engine.hip — host side of libq3tts.so: weight arena, sessions (KV pages, RNG streams, penalty
// masks), the per-frame launch sequence (optionally replayed as one hipGraph), the codec-decoder
// pipeline and the C ABI declared in include/q3tts.h.
//
// Layer map of the reference this file replaces (paths relative to the reference repo):
//   src/lib.rs:530-656 generate_codes, 425-501 synthesize_with_timing, 1484-1782 StreamingSession
//   src/models/talker.rs:451-627 prefill builders, 716-736 generate_step_with_embed
//   src/models/code_predictor.rs:320-416 generate_acoustic_codes
//   src/models/codec/decoder_12hz.rs:411-505 decode
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

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static q3_status set_err(q3_status st, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return st;
}
extern "C" q3_status q3i_set_err(q3_status st, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return st;
}
#define HIPC(expr)                                                                                        \
    do {                                                                                                  \
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

extern "C" int q3_abi_version(void) { return Q3_ABI_VERSION; }
extern "C" const char* q3_last_error(void) { return g_err; }
extern "C" int q3_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

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
// synthetic tensor generator (host)
// ------------------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static inline uint64_t fnv1a64(const char* s) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001b3ULL; }
    return h;
}
extern "C" q3_status q3_synth_fill(uint64_t seed, const char* name, int dtype, float scale, float offset, int64_t n,
                                   void* out_host) {
    if (!name || !out_host || n < 0) return set_err(Q3_INVALID_ARG, "q3_synth_fill: bad argument");
    const uint64_t key = splitmix64(seed ^ fnv1a64(name));
    // Irwin-Hall(4) of 16-bit uniforms: exact integer sum, std = 65536/sqrt(3)
    const float c = (float)((double)scale / (65536.0 / 1.7320508075688772));
    float* of = (float*)out_host; uint16_t* ob = (uint16_t*)out_host;
    auto body = [=](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint64_t z = splitmix64(key + (uint64_t)i * 0x9E3779B97F4A7C15ULL);
            const int s = (int)(z & 0xffff) + (int)((z >> 16) & 0xffff) + (int)((z >> 32) & 0xffff) + (int)((z >> 48) & 0xffff) - 131070;
            const float v = offset + (float)s * c;
            if (dtype == Q3_DTYPE_BF16) ob[i] = f32_to_bf16_host(v);
            else of[i] = v;
        }
    };
    // plain std::thread fan-out (no OpenMP runtime inside the product library)
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 2 || n < (1 << 20)) { body(0, n); return Q3_OK; }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t lo = (int64_t)t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) th.emplace_back(body, lo, hi);
    }
    for (auto& t : th) t.join();
    return Q3_OK;
}

// PCG stream (sampling.rs:32-51, 84-94)
extern "C" void q3_rng_seed(uint64_t seed, uint64_t* state) { *state = seed * 2685821657736338717ULL + 1442695040888963407ULL; }
extern "C" float q3_rng_next(uint64_t* state) {
    const uint64_t old = *state;
    *state = old * 6364136223846793005ULL + 1442695040888963407ULL;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
    const uint32_t out = (xs >> rot) | (xs << ((32 - rot) & 31));
    return (float)out / (float)UINT32_MAX;
}
extern "C" void q3_codes_to_tensor(const uint32_t* frames, int n_frames, int64_t* out) {
    for (int f = 0; f < n_frames; ++f)
        for (int q = 0; q < 16; ++q) out[(size_t)q * n_frames + f] = (int64_t)frames[(size_t)f * 16 + q];
}

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
    // sessions hold pages, streams and weights of their model: q3_model_free with sessions still alive only marks the model,
    // the last q3_session_free destroys it (a host that tears down in the wrong order must not crash)
    std::atomic<int> live_sessions{0}; std::atomic<bool> zombie{false}; std::atomic<bool> claimed{false};     // claimed: someone is destroying it
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

static void add_slot(q3_model* m, const std::string& name, int64_t n, int stored, bool align = true) {
    Slot s; s.name = name; s.n = n; s.stored = stored;
    size_t off = m->arena_bytes;
    if (align) off = (off + 255) & ~(size_t)255;
    s.offset = off;
    m->arena_bytes = off + (size_t)n * (stored == Q3_DTYPE_BF16 ? 2 : 4);
    m->index[name] = (int)m->slots.size();
    m->slots.push_back(s);
}
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
// GEMV weight [rows][cols] bf16, stored MFMA-tiled (q3_kernels_gemv.hip); element count reported to the
// caller stays rows*cols (the checkpoint's), the arena holds the padded tiled image.
static void add_tiled(q3_model* m, const std::string& name, int rows, int cols, bool dual, bool align = true) {
    Slot s; s.name = name; s.n = (int64_t)rows * cols; s.stored = Q3_DTYPE_BF16; s.kind = SK_TILED; s.rows = rows; s.cols = cols; s.dual = dual;
    size_t off = m->arena_bytes;
    if (align) off = (off + 255) & ~(size_t)255;
    s.offset = off;
    m->arena_bytes = off + tiled_elems(1, rows, cols) * 2;
    m->index[name] = (int)m->slots.size();
    m->slots.push_back(s);
}
// second (4-row-tile) images of a group of tensors, laid out back to back (fused QKV needs them contiguous)
static void add_alt_images(q3_model* m, std::initializer_list<std::string> names) {
    bool first = true;
    for (const auto& nme : names) {
        Slot& s = m->slots[m->index[nme]];
        if (!s.dual) continue;
        size_t off = m->arena_bytes;
        if (first) off = (off + 255) & ~(size_t)255;
        first = false;
        s.offset2 = off;
        m->arena_bytes = off + tiled_elems(2, s.rows, s.cols) * 2;
    }
}
static void add_layer_slots(q3_model* m, const std::string& p, int H, int I, int nh, int nkv, int hd) {
    add_slot(m, p + ".input_layernorm.weight", H, Q3_DTYPE_F32);
    // q,k,v rows are stored back to back so the fused QKV GEMV sees one [QD+2KD][H] matrix
    const bool dq = dual_tiled((nh + 2 * nkv) * hd);
    add_tiled(m, p + ".self_attn.q_proj.weight", nh * hd, H, dq);
    add_tiled(m, p + ".self_attn.k_proj.weight", nkv * hd, H, dq, false);
    add_tiled(m, p + ".self_attn.v_proj.weight", nkv * hd, H, dq, false);
    add_alt_images(m, {p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"});
    add_tiled(m, p + ".self_attn.o_proj.weight", H, nh * hd, dual_tiled(H));
    add_alt_images(m, {p + ".self_attn.o_proj.weight"});
    add_slot(m, p + ".self_attn.q_norm.weight", hd, Q3_DTYPE_F32);
    add_slot(m, p + ".self_attn.k_norm.weight", hd, Q3_DTYPE_F32);
    add_slot(m, p + ".post_attention_layernorm.weight", H, Q3_DTYPE_F32);
    add_tiled(m, p + ".mlp.gate_proj.weight", I, H, dual_tiled(I));
    add_tiled(m, p + ".mlp.up_proj.weight", I, H, dual_tiled(I));
    add_tiled(m, p + ".mlp.down_proj.weight", H, I, dual_tiled(H));
    add_alt_images(m, {p + ".mlp.gate_proj.weight"}); add_alt_images(m, {p + ".mlp.up_proj.weight"}); add_alt_images(m, {p + ".mlp.down_proj.weight"});
}
static std::string fmt(const char* f, ...) {
    char b[256]; va_list ap; va_start(ap, f); vsnprintf(b, sizeof b, f, ap); va_end(ap); return b;
}

// tensor manifest: names/shapes of SURVEY.md Appendix B (talker.rs:380-405, code_predictor.rs:163-205,
// decoder_12hz.rs:191-381)
static void build_manifest(q3_model* m) {
    const q3_config& c = m->cfg;
    const int H = c.hidden, TD = c.text_dim, CH = c.cp_hidden;
    add_slot(m, "talker.model.text_embedding.weight", (int64_t)c.text_vocab * TD, Q3_DTYPE_BF16);
    add_tiled(m, "talker.text_projection.linear_fc1.weight", TD, TD, dual_tiled(TD)); add_alt_images(m, {"talker.text_projection.linear_fc1.weight"});
    add_slot(m, "talker.text_projection.linear_fc1.bias", TD, Q3_DTYPE_F32);
    add_tiled(m, "talker.text_projection.linear_fc2.weight", H, TD, dual_tiled(H)); add_alt_images(m, {"talker.text_projection.linear_fc2.weight"});
    add_slot(m, "talker.text_projection.linear_fc2.bias", H, Q3_DTYPE_F32);
    add_slot(m, "talker.model.codec_embedding.weight", (int64_t)c.codec_vocab * H, Q3_DTYPE_BF16);
    for (int i = 0; i < c.n_layers; ++i)
        add_layer_slots(m, fmt("talker.model.layers.%d", i), H, c.inter, c.n_heads, c.n_kv_heads, c.head_dim);
    add_slot(m, "talker.model.norm.weight", H, Q3_DTYPE_F32);
    add_tiled(m, "talker.codec_head.weight", c.codec_vocab, H, dual_tiled(c.codec_vocab)); add_alt_images(m, {"talker.codec_head.weight"});
    if (H != CH) {
        add_tiled(m, "talker.code_predictor.small_to_mtp_projection.weight", CH, H, dual_tiled(CH)); add_alt_images(m, {"talker.code_predictor.small_to_mtp_projection.weight"});
        add_slot(m, "talker.code_predictor.small_to_mtp_projection.bias", CH, Q3_DTYPE_F32);
    }
    for (int g = 0; g < c.n_groups - 1; ++g)
        add_slot(m, fmt("talker.code_predictor.model.codec_embedding.%d.weight", g), (int64_t)c.cp_vocab * H, Q3_DTYPE_BF16);
    for (int i = 0; i < c.cp_layers; ++i)
        add_layer_slots(m, fmt("talker.code_predictor.model.layers.%d", i), CH, c.cp_inter, c.cp_heads, c.cp_kv_heads, c.head_dim);
    add_slot(m, "talker.code_predictor.model.norm.weight", CH, Q3_DTYPE_F32);
    for (int g = 0; g < c.n_groups - 1; ++g)
        { const std::string nm = fmt("talker.code_predictor.lm_head.%d.weight", g); add_tiled(m, nm, c.cp_vocab, CH, dual_tiled(c.cp_vocab)); add_alt_images(m, {nm}); }
    // decoder (all f32)
    const int CB = c.dec_cb_size, CD = c.dec_cb_dim, Q = c.dec_q_dim, LAT = c.dec_latent, DH = c.dec_hidden;
    const int QD = c.dec_heads * c.dec_head_dim, DI = c.dec_inter;
    auto F = [&](const std::string& n, int64_t cnt) { add_slot(m, n, cnt, Q3_DTYPE_F32); };
    F("decoder.quantizer.rvq_first.vq.layers.0._codebook.embedding_sum", (int64_t)CB * CD);
    F("decoder.quantizer.rvq_first.vq.layers.0._codebook.cluster_usage", CB);
    for (int i = 0; i < 15; ++i) {
        F(fmt("decoder.quantizer.rvq_rest.vq.layers.%d._codebook.embedding_sum", i), (int64_t)CB * CD);
        F(fmt("decoder.quantizer.rvq_rest.vq.layers.%d._codebook.cluster_usage", i), CB);
    }
    F("decoder.quantizer.rvq_first.output_proj.weight", (int64_t)Q * CD);
    F("decoder.quantizer.rvq_rest.output_proj.weight", (int64_t)Q * CD);
    F("decoder.pre_conv.conv.weight", (int64_t)LAT * Q * 3);
    F("decoder.pre_conv.conv.bias", LAT);
    F("decoder.pre_transformer.input_proj.weight", (int64_t)DH * LAT);
    F("decoder.pre_transformer.input_proj.bias", DH);
    F("decoder.pre_transformer.output_proj.weight", (int64_t)LAT * DH);
    F("decoder.pre_transformer.output_proj.bias", LAT);
    F("decoder.pre_transformer.norm.weight", DH);
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = fmt("decoder.pre_transformer.layers.%d", i);
        F(p + ".input_layernorm.weight", DH);
        F(p + ".self_attn.q_proj.weight", (int64_t)QD * DH);
        F(p + ".self_attn.k_proj.weight", (int64_t)QD * DH);
        F(p + ".self_attn.v_proj.weight", (int64_t)QD * DH);
        F(p + ".self_attn.o_proj.weight", (int64_t)DH * QD);
        F(p + ".self_attn_layer_scale.scale", DH);
        F(p + ".post_attention_layernorm.weight", DH);
        F(p + ".mlp.gate_proj.weight", (int64_t)DI * DH);
        F(p + ".mlp.up_proj.weight", (int64_t)DI * DH);
        F(p + ".mlp.down_proj.weight", (int64_t)DH * DI);
        F(p + ".mlp_layer_scale.scale", DH);
    }
    auto TC = [&](const std::string& n, int cin, int cout, int k, int stride) {
        add_slot(m, n, (int64_t)cin * cout * k, Q3_DTYPE_F32);
        Slot& s = m->slots.back(); s.kind = SK_TRANSCONV; s.tc_cin = cin; s.tc_cout = cout; s.tc_k = k; s.tc_stride = stride;
    };
    for (int i = 0; i < 2; ++i) {
        const std::string p = fmt("decoder.upsample.%d", i);
        const int r = c.dec_up_ratios[i];
        TC(p + ".0.conv.weight", LAT, LAT, r, r);
        F(p + ".0.conv.bias", LAT);
        F(p + ".1.dwconv.conv.weight", (int64_t)LAT * 7);
        F(p + ".1.dwconv.conv.bias", LAT);
        F(p + ".1.norm.weight", LAT);
        F(p + ".1.norm.bias", LAT);
        F(p + ".1.pwconv1.weight", (int64_t)4 * LAT * LAT);
        F(p + ".1.pwconv1.bias", 4 * LAT);
        F(p + ".1.pwconv2.weight", (int64_t)4 * LAT * LAT);
        F(p + ".1.pwconv2.bias", LAT);
        F(p + ".1.gamma", LAT);
    }
    const int D = c.dec_dim;
    F("decoder.decoder.0.conv.weight", (int64_t)D * LAT * 7);
    F("decoder.decoder.0.conv.bias", D);
    int cin = D;
    for (int b = 0; b < 4; ++b) {
        const int r = c.dec_up_rates[b], cout = cin / 2;
        const std::string p = fmt("decoder.decoder.%d.block", b + 1);
        F(p + ".0.alpha", cin); F(p + ".0.beta", cin);
        TC(p + ".1.conv.weight", cin, cout, 2 * r, r);
        F(p + ".1.conv.bias", cout);
        for (int u = 0; u < 3; ++u) {
            const std::string q = fmt("%s.%d", p.c_str(), u + 2);
            F(q + ".act1.alpha", cout); F(q + ".act1.beta", cout);
            F(q + ".conv1.conv.weight", (int64_t)cout * cout * 7); F(q + ".conv1.conv.bias", cout);
            F(q + ".act2.alpha", cout); F(q + ".act2.beta", cout);
            F(q + ".conv2.conv.weight", (int64_t)cout * cout); F(q + ".conv2.conv.bias", cout);
        }
        cin = cout;
    }
    F("decoder.decoder.5.alpha", cin); F("decoder.decoder.5.beta", cin);
    F("decoder.decoder.6.conv.weight", (int64_t)cin * 7);
    F("decoder.decoder.6.conv.bias", 1);
}

static q3_status check_config(const q3_config& c) {
    // config.json is untrusted input (u64 values cast to int32 by the parser): every dimension positive and bounded
    // before any size is computed from it
    const struct { const char* name; int v, lo, hi; } dims[] = {
        {"text_vocab", c.text_vocab, 1, 1 << 22}, {"text_dim", c.text_dim, 32, 1 << 15}, {"hidden", c.hidden, 32, 1 << 15}, {"inter", c.inter, 32, 1 << 17},
        {"n_layers", c.n_layers, 1, 256}, {"n_heads", c.n_heads, 1, 256}, {"n_kv_heads", c.n_kv_heads, 1, 256},
        {"cp_hidden", c.cp_hidden, 32, 1 << 15}, {"cp_inter", c.cp_inter, 32, 1 << 17}, {"cp_layers", c.cp_layers, 1, 64},
        {"cp_heads", c.cp_heads, 1, 256}, {"cp_kv_heads", c.cp_kv_heads, 1, 256}, {"cp_vocab", c.cp_vocab, 2, 4096},
        {"dec_cb_dim", c.dec_cb_dim, 1, 256}, {"dec_q_dim", c.dec_q_dim, 1, 1 << 14}, {"dec_latent", c.dec_latent, 1, 1 << 14},
        {"dec_hidden", c.dec_hidden, 1, 1 << 14}, {"dec_layers", c.dec_layers, 1, 64}, {"dec_heads", c.dec_heads, 1, 256},
        {"dec_inter", c.dec_inter, 1, 1 << 16}, {"dec_cb_size", c.dec_cb_size, 2, 4096}, {"dec_dim", c.dec_dim, 16, 1 << 14},
        {"dec_up_ratios[0]", c.dec_up_ratios[0], 1, 16}, {"dec_up_ratios[1]", c.dec_up_ratios[1], 1, 16},
        {"dec_up_rates[0]", c.dec_up_rates[0], 1, 32}, {"dec_up_rates[1]", c.dec_up_rates[1], 1, 32},
        {"dec_up_rates[2]", c.dec_up_rates[2], 1, 32}, {"dec_up_rates[3]", c.dec_up_rates[3], 1, 32}};
    for (const auto& d : dims)
        if (d.v < d.lo || d.v > d.hi) return set_err(Q3_UNSUPPORTED, "config: %s = %d is outside [%d, %d]", d.name, d.v, d.lo, d.hi);
    if (c.dec_dim % 16) return set_err(Q3_UNSUPPORTED, "config: dec_dim %d must be a multiple of 16 (four halvings)", c.dec_dim);
    if (!(c.rms_eps > 0.0f) || !(c.dec_eps > 0.0f) || !(c.rope_theta > 1.0f) || !(c.dec_theta > 1.0f))
        return set_err(Q3_UNSUPPORTED, "config: eps / rope theta out of range");
    if (c.head_dim != HEAD_DIM) return set_err(Q3_UNSUPPORTED, "head_dim %d unsupported (kernels are built for 128)", c.head_dim);
    if (c.dec_head_dim != 64) return set_err(Q3_UNSUPPORTED, "decoder head_dim %d unsupported (64)", c.dec_head_dim);
    if (c.hidden % 32 || c.inter % 32 || c.text_dim % 32 || c.cp_hidden % 32 || c.cp_inter % 32)
        return set_err(Q3_UNSUPPORTED, "hidden/intermediate sizes must be multiples of 32");
    if (c.n_groups != 16) return set_err(Q3_UNSUPPORTED, "n_groups must be 16");
    if (c.codec_vocab > 4096 || c.codec_vocab < 1024) return set_err(Q3_UNSUPPORTED, "codec_vocab must be in [1024, 4096]");
    const int nrep = c.n_heads / (c.n_kv_heads ? c.n_kv_heads : 1), crep = c.cp_heads / (c.cp_kv_heads ? c.cp_kv_heads : 1);
    if ((nrep != 1 && nrep != 2 && nrep != 4) || (crep != 1 && crep != 2 && crep != 4))
        return set_err(Q3_UNSUPPORTED, "heads/kv_heads ratio must be 1, 2 or 4");
    if (c.dec_cb_dim > 256) return set_err(Q3_UNSUPPORTED, "dec_cb_dim > 256");
    return Q3_OK;
}

extern "C" q3_status q3_model_create(const q3_config* cfg, int device, q3_model** out) {
    if (!cfg || !out) return set_err(Q3_INVALID_ARG, "q3_model_create: null argument");
    Q3C(check_config(*cfg));
    if (device == -1) {   // manifest-only handle (no GPU): names/shapes for tools and CPU tests
        std::unique_ptr<q3_model> mm(new q3_model());
        mm->cfg = *cfg; mm->device = -1;
        build_manifest(mm.get());
        *out = mm.release();
        return Q3_OK;
    }
    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return set_err(Q3_INVALID_ARG, "device %d not available (%d visible)", device, ndev);
    HIPC(hipSetDevice(device));
    std::unique_ptr<q3_model> m(new q3_model());
    m->kv_pool.run_floats = (size_t)cfg->n_kv_heads * KV_PAGE_POS * HEAD_DIM; m->kv_pool.n_layers = cfg->n_layers;
    m->kv_pool16.run_floats = m->kv_pool.run_floats; m->kv_pool16.n_layers = cfg->n_layers; m->kv_pool16.elem_bytes = 2;
    m->kv_pool.budget = &m->kv_budget; m->kv_pool.unit = 2; m->kv_pool16.budget = &m->kv_budget; m->kv_pool16.unit = 1;
    m->cfg = *cfg; m->device = device;
    build_manifest(m.get());
    HIPC(hipMalloc((void**)&m->arena, m->arena_bytes));
    HIPC(hipMemset(m->arena, 0, m->arena_bytes));
    *out = m.release();
    return Q3_OK;
}

static void model_destroy(q3_model* m);
extern "C" void q3_model_free(q3_model* m) {
    if (!m) return;
    // zombie first, THEN look at the count: a session freed between a check and a later store would find zombie unset, leave,
    // and nobody would destroy the model. With this order either the last session sees zombie (its fetch_sub comes after the
    // store) and destroys the model, or this thread sees the count at zero — `claimed` makes sure only one of them does.
    m->zombie.store(true);
    if (m->live_sessions.load() > 0) return;
    if (m->claimed.exchange(true)) return;
    model_destroy(m);
}
static void model_destroy(q3_model* m) {
    if (m->device < 0) { delete m; return; }
    hipSetDevice(m->device);
    hipFree(m->arena); hipFree(m->rope_cos); hipFree(m->rope_sin); hipFree(m->derived); hipFree(m->wpk_arena);
    hipFree((void*)m->rest_cbs_dev); hipFree((void*)m->cp_embs_dev); hipFree(m->proj_tabs); hipFree(m->qkv0_tabs);
    for (hipStream_t st : m->idle_streams) (void)hipStreamDestroy(st);
    delete m;
}

// Paged KV pool of the model (KvPool above). limit: the most pages sessions may hold at once (0 = HBM is the limit); a session
// that needs a page beyond it fails with Q3_KV_OVERFLOW — the reference's KV-overflow bail (kv_cache.rs:293-300).
extern "C" q3_status q3_model_set_codec_planes(q3_model* m, int planes) {
    if (!m) return set_err(Q3_INVALID_ARG, "q3_model_set_codec_planes: null model");
    if (planes != 2 && planes != 3) return set_err(Q3_INVALID_ARG, "q3_model_set_codec_planes: %d (2 or 3)", planes);
    m->codec_planes = planes;
    return Q3_OK;
}

extern "C" q3_status q3_model_kv_pool_limit(q3_model* m, int max_pages) {
    if (!m || m->device < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_limit: no device model");
    if (max_pages < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_limit: negative limit");
    std::lock_guard<std::mutex> g(m->kv_budget.mu);
    if (max_pages > 0 && 2L * max_pages < m->kv_budget.used)
        return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_limit: %ld pages (f32 equivalents) are in use", (m->kv_budget.used + 1) / 2);
    m->kv_budget.limit = 2L * max_pages;
    return Q3_OK;
}
// Pages are counted in f32 equivalents: a page of a bf16 session (q3_session_set_kv_dtype) is half of one, rounded up in the
// totals below — one budget covers both pools.
extern "C" q3_status q3_model_kv_pool_info(q3_model* m, int* page_positions, size_t* page_bytes, int* pages_total, int* pages_in_use, int* pages_peak) {
    if (!m || m->device < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_info: no device model");
    int tot = 0;
    { std::lock_guard<std::mutex> g(m->kv_pool.mu); tot += m->kv_pool.total; }
    { std::lock_guard<std::mutex> g(m->kv_pool16.mu); tot += (m->kv_pool16.total + 1) / 2; }
    std::lock_guard<std::mutex> g(m->kv_budget.mu);
    if (page_positions) *page_positions = KV_PAGE_POS;
    if (page_bytes) *page_bytes = m->kv_pool.page_bytes();
    if (pages_total) *pages_total = tot;
    if (pages_in_use) *pages_in_use = (int)((m->kv_budget.used + 1) / 2);
    if (pages_peak) *pages_peak = (int)((m->kv_budget.peak + 1) / 2);
    return Q3_OK;
}
// Slabs of either pool none of whose pages is held go back to the device (a server that has seen one very long prompt need not
// keep its ~1 GB slabs for the model's lifetime). Safe beside running sessions: held pages pin their slab.
extern "C" q3_status q3_model_kv_pool_trim(q3_model* m, size_t* bytes_freed) {
    if (!m || m->device < 0) return set_err(Q3_INVALID_ARG, "q3_model_kv_pool_trim: no device model");
    HIPC(hipSetDevice(m->device));
    const size_t n = m->kv_pool.trim() + m->kv_pool16.trim();
    if (bytes_freed) *bytes_freed = n;
    return Q3_OK;
}

extern "C" q3_status q3_model_config(const q3_model* m, q3_config* out) {
    if (!m || !out) return set_err(Q3_INVALID_ARG, "q3_model_config: null argument");
    *out = m->cfg;
    return Q3_OK;
}
extern "C" int q3_model_n_tensors(const q3_model* m) { return m ? (int)m->slots.size() : 0; }
extern "C" q3_status q3_model_tensor_info(const q3_model* m, int i, const char** name, int64_t* n, int* stored_dtype) {
    if (!m || i < 0 || i >= (int)m->slots.size()) return set_err(Q3_INVALID_ARG, "tensor index out of range");
    if (name) *name = m->slots[i].name.c_str();
    if (n) *n = m->slots[i].n;
    if (stored_dtype) *stored_dtype = m->slots[i].stored;
    return Q3_OK;
}

// row-major [N][K] bf16 → MFMA tiles, zero padded.
//   mode 1: [N↑16/16][K↑32/32][lane][8],  lane = (k-group << 4) | row      (16 rows x 32 k per KiB)
//   mode 2: [N↑4/4][K↑128/128][lane][8],  lane = (k-group << 2) | row      (4 rows x 128 k per KiB)
static void retile_bf16(const uint16_t* src, int N, int K, uint16_t* dst, int mode) {
    const int RT = mode == 2 ? 4 : 16, KS = mode == 2 ? 128 : 32, RB = mode == 2 ? 2 : 4;
    const int T = (N + RT - 1) / RT, S = (K + KS - 1) / KS;
    auto body = [=](int t0, int t1) {
        for (int t = t0; t < t1; ++t)
            for (int s = 0; s < S; ++s)
                for (int l = 0; l < 64; ++l) {
                    const int n = t * RT + (l & (RT - 1)), k0 = s * KS + (l >> RB) * 8;
                    uint16_t* d = dst + (((size_t)t * S + s) * 64 + l) * 8;
                    for (int e = 0; e < 8; ++e) d[e] = (n < N && k0 + e < K) ? src[(size_t)n * K + k0 + e] : (uint16_t)0;
                }
    };
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 2 || (size_t)N * K < (1u << 20)) { body(0, T); return; }
    std::vector<std::thread> th;
    const int per = (T + (int)nt - 1) / (int)nt;
    for (unsigned i = 0; i < nt; ++i) {
        const int a = (int)i * per, b = a + per < T ? a + per : T;
        if (a < b) th.emplace_back(body, a, b);
    }
    for (auto& x : th) x.join();
}

extern "C" q3_status q3_model_set_tensor(q3_model* m, const char* name, int dtype, const void* data, int64_t n) {
    if (!m || !name || !data) return set_err(Q3_INVALID_ARG, "q3_model_set_tensor: null argument");
    auto it = m->index.find(name);
    if (it == m->index.end()) return set_err(Q3_INVALID_ARG, "unknown tensor name: %s", name);
    Slot& s = m->slots[it->second];
    if (n != s.n) return set_err(Q3_INVALID_ARG, "tensor %s has %lld elements, expected %lld", name, (long long)n, (long long)s.n);
    if (m->device < 0) return set_err(Q3_UNSUPPORTED, "manifest-only model handle (device -1) holds no weights");
    HIPC(hipSetDevice(m->device));
    const size_t bytes = (size_t)n * (s.stored == Q3_DTYPE_BF16 ? 2 : 4);
    std::vector<char> tmp;
    const void* src = data;
    size_t up_bytes = bytes;
    if (s.kind == SK_TILED) {
        std::vector<uint16_t> w((size_t)n);
        if (dtype == Q3_DTYPE_BF16) memcpy(w.data(), data, (size_t)n * 2);
        else if (dtype == Q3_DTYPE_F32) for (int64_t i = 0; i < n; ++i) w[(size_t)i] = f32_to_bf16_host(((const float*)data)[i]);
        else return set_err(Q3_INVALID_ARG, "unsupported source dtype %d", dtype);
        up_bytes = tiled_elems(1, s.rows, s.cols) * 2;
        tmp.resize(up_bytes);
        retile_bf16(w.data(), s.rows, s.cols, (uint16_t*)tmp.data(), 1);
        src = tmp.data();
        if (s.dual) {
            std::vector<uint16_t> t2(tiled_elems(2, s.rows, s.cols));
            retile_bf16(w.data(), s.rows, s.cols, t2.data(), 2);
            HIPC(hipMemcpy(m->arena + s.offset2, t2.data(), t2.size() * 2, hipMemcpyHostToDevice));
        }
    } else if (s.kind == SK_TRANSCONV) {
        // [cin][cout][k] → per-phase causal-conv weights [stride][cout][cin][taps]
        std::vector<float> w((size_t)n);
        if (dtype == Q3_DTYPE_F32) memcpy(w.data(), data, (size_t)n * 4);
        else for (int64_t i = 0; i < n; ++i) w[(size_t)i] = bf16_to_f32_host(((const uint16_t*)data)[i]);
        const int cin = s.tc_cin, cout = s.tc_cout, k = s.tc_k, st = s.tc_stride, taps = k / st;
        tmp.resize(bytes);
        float* o = (float*)tmp.data();
        for (int ph = 0; ph < st; ++ph)
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int tp = 0; tp < taps; ++tp) {
                        // tap tp multiplies x[j - (taps-1-tp)]  ⇒  kernel index ph + (taps-1-tp)*stride
                        const int kk = ph + (taps - 1 - tp) * st;
                        o[(((size_t)ph * cout + co) * cin + ci) * taps + tp] = w[((size_t)ci * cout + co) * k + kk];
                    }
        src = tmp.data();
    } else if (s.stored == Q3_DTYPE_BF16 && dtype == Q3_DTYPE_F32) {
        tmp.resize(bytes);
        uint16_t* o = (uint16_t*)tmp.data(); const float* f = (const float*)data;
        for (int64_t i = 0; i < n; ++i) o[i] = f32_to_bf16_host(f[i]);
        src = tmp.data();
    } else if (s.stored == Q3_DTYPE_F32 && dtype == Q3_DTYPE_BF16) {
        tmp.resize(bytes);
        float* o = (float*)tmp.data(); const uint16_t* h = (const uint16_t*)data;
        for (int64_t i = 0; i < n; ++i) o[i] = bf16_to_f32_host(h[i]);
        src = tmp.data();
    } else if (dtype != Q3_DTYPE_F32 && dtype != Q3_DTYPE_BF16) {
        return set_err(Q3_INVALID_ARG, "unsupported source dtype %d", dtype);
    }
    HIPC(hipMemcpy(m->arena + s.offset, src, up_bytes, hipMemcpyHostToDevice));
    s.loaded = true;
    m->finalized = false;
    return Q3_OK;
}

extern "C" q3_status q3_model_arena(q3_model* m, void** dev_ptr, size_t* bytes) {
    if (!m) return set_err(Q3_INVALID_ARG, "null model");
    if (dev_ptr) *dev_ptr = m->arena;
    if (bytes) *bytes = m->arena_bytes;
    return Q3_OK;
}
extern "C" q3_status q3_model_mark_loaded(q3_model* m) {
    if (!m) return set_err(Q3_INVALID_ARG, "null model");
    for (auto& s : m->slots) s.loaded = true;
    return Q3_OK;
}

template <typename T>
static const T* P(const q3_model* m, const std::string& name) {
    auto it = m->index.find(name);
    if (it == m->index.end()) return nullptr;
    return (const T*)(m->arena + m->slots[it->second].offset);
}
static TW PT(const q3_model* m, const std::string& name) {
    TW w;
    auto it = m->index.find(name);
    if (it == m->index.end()) return w;
    const Slot& s = m->slots[it->second];
    w.t1 = (const uint16_t*)(m->arena + s.offset);
    if (s.dual) w.t2 = (const uint16_t*)(m->arena + s.offset2);
    return w;
}
static void resolve_layer(const q3_model* m, LayerW& L, const std::string& p) {
    L.in_ln = P<float>(m, p + ".input_layernorm.weight");
    L.qkv = PT(m, p + ".self_attn.q_proj.weight");
    L.o = PT(m, p + ".self_attn.o_proj.weight");
    L.q_norm = P<float>(m, p + ".self_attn.q_norm.weight");
    L.k_norm = P<float>(m, p + ".self_attn.k_norm.weight");
    L.post_ln = P<float>(m, p + ".post_attention_layernorm.weight");
    L.gate = PT(m, p + ".mlp.gate_proj.weight");
    L.up = PT(m, p + ".mlp.up_proj.weight");
    L.down = PT(m, p + ".mlp.down_proj.weight");
}

extern "C" q3_status q3_model_finalize(q3_model* m) {
    if (!m) return set_err(Q3_INVALID_ARG, "null model");
    if (m->device < 0) return set_err(Q3_UNSUPPORTED, "manifest-only model handle (device -1) cannot be finalized");
    for (auto& s : m->slots)
        if (!s.loaded) return set_err(Q3_MISSING_WEIGHT, "Missing weight: %s", s.name.c_str());
    HIPC(hipSetDevice(m->device));
    const q3_config& c = m->cfg;
    m->text_emb = P<uint16_t>(m, "talker.model.text_embedding.weight");
    m->fc1w = PT(m, "talker.text_projection.linear_fc1.weight");
    m->fc1b = P<float>(m, "talker.text_projection.linear_fc1.bias");
    m->fc2w = PT(m, "talker.text_projection.linear_fc2.weight");
    m->fc2b = P<float>(m, "talker.text_projection.linear_fc2.bias");
    m->codec_emb = P<uint16_t>(m, "talker.model.codec_embedding.weight");
    m->norm = P<float>(m, "talker.model.norm.weight");
    m->codec_head = PT(m, "talker.codec_head.weight");
    m->mtp_w = PT(m, "talker.code_predictor.small_to_mtp_projection.weight");
    m->mtp_b = P<float>(m, "talker.code_predictor.small_to_mtp_projection.bias");
    m->cp_norm = P<float>(m, "talker.code_predictor.model.norm.weight");
    m->tl.resize(c.n_layers); m->cl.resize(c.cp_layers);
    for (int i = 0; i < c.n_layers; ++i) resolve_layer(m, m->tl[i], fmt("talker.model.layers.%d", i));
    for (int i = 0; i < c.cp_layers; ++i) resolve_layer(m, m->cl[i], fmt("talker.code_predictor.model.layers.%d", i));
    m->cp_emb.resize(15); m->cp_head.resize(15);
    for (int g = 0; g < 15; ++g) {
        m->cp_emb[g] = P<uint16_t>(m, fmt("talker.code_predictor.model.codec_embedding.%d.weight", g));
        m->cp_head[g] = PT(m, fmt("talker.code_predictor.lm_head.%d.weight", g));
    }
    if (!m->cp_embs_dev) HIPC(hipMalloc((void**)&m->cp_embs_dev, 15 * sizeof(void*)));
    HIPC(hipMemcpy((void*)m->cp_embs_dev, m->cp_emb.data(), 15 * sizeof(void*), hipMemcpyHostToDevice));
    if (m->mtp_w.t1) {
        // Pre-projected embedding tables (1.7B): 15 x [cp_vocab][CH] + [codec_vocab][CH] f32 (138 MB). Built with the very
        // GEMV launches the frame loop would use (8 gathered rows per launch), so a table row is exactly what the
        // per-pass projection of that row computes at a batch of 8.
        const int Hh = c.hidden, CHh = c.cp_hidden;
        const size_t total = ((size_t)15 * c.cp_vocab + c.codec_vocab) * CHh;
        if (!m->proj_tabs) HIPC(hipMalloc((void**)&m->proj_tabs, total * 4));
        float* xin = nullptr; uint32_t* ids = nullptr;
        HIPC(hipMalloc((void**)&xin, (size_t)8 * Hh * 4)); HIPC(hipMalloc((void**)&ids, 8 * 4));
        float* cur = m->proj_tabs;
        hipError_t e = hipSuccess;
        for (int tbl = 0; tbl <= 15 && e == hipSuccess; ++tbl) {
            const uint16_t* emb = tbl < 15 ? m->cp_emb[tbl] : m->codec_emb;
            const int rows = tbl < 15 ? c.cp_vocab : c.codec_vocab;
            if (tbl < 15) m->cp_proj[tbl] = cur; else m->sem_proj = cur;
            for (int r0 = 0; r0 < rows && e == hipSuccess; r0 += 8) {
                const int M = (rows - r0) < 8 ? (rows - r0) : 8;
                uint32_t h[8]; for (int i = 0; i < 8; ++i) h[i] = (uint32_t)(r0 + (i < M ? i : 0));
                e = hipMemcpyAsync(ids, h, sizeof h, hipMemcpyHostToDevice, 0);
                if (e == hipSuccess) e = launch_gather_rows_bf16(emb, ids, xin, M, Hh, 0);
                LinArgs a;
                a.N = CHh; a.K = Hh; set_w(a, m->mtp_w, 8, CHh, Hh); a.x = xin; a.ldx = Hh; a.bias = m->mtp_b; a.y = cur + (size_t)r0 * CHh; a.ldy = CHh;
                a.M = M; a.epi = EPI_NONE;
                if (e == hipSuccess) e = launch_linear(a, 0);
                if (e == hipSuccess) e = hipStreamSynchronize(0);      // `h` is reused by the next iteration
            }
            cur += (size_t)rows * CHh;
        }
        hipFree(xin); hipFree(ids);
        if (e != hipSuccess) return set_err(Q3_HIP_ERROR, "projection tables: %s", hipGetErrorString(e));
    }
    if (c.cp_hidden % 4 == 0 && (m->mtp_w.t1 ? m->proj_tabs != nullptr : c.cp_hidden == c.hidden)) {
        // Layer-0 q|k|v tables: the first code-predictor layer sees only (projected) embedding rows on passes >= 1, so
        // RMSNorm + qkv of every possible row is computed once, again with the launches the frame loop would use.
        const int CHh = c.cp_hidden, QKVD = (c.cp_heads + 2 * c.cp_kv_heads) * HEAD_DIM;
        const size_t total = ((size_t)15 * c.cp_vocab + c.codec_vocab) * QKVD;
        if (!m->qkv0_tabs) HIPC(hipMalloc((void**)&m->qkv0_tabs, total * 4));
        float* xin = nullptr; uint32_t* ids = nullptr;
        HIPC(hipMalloc((void**)&xin, (size_t)8 * CHh * 4)); HIPC(hipMalloc((void**)&ids, 8 * 4));
        float* cur = m->qkv0_tabs;
        hipError_t e = hipSuccess;
        for (int tbl = 0; tbl <= 15 && e == hipSuccess; ++tbl) {
            const int rows = tbl < 15 ? c.cp_vocab : c.codec_vocab;
            const float* prow = m->mtp_w.t1 ? (tbl < 15 ? m->cp_proj[tbl] : m->sem_proj) : nullptr;
            const uint16_t* emb = tbl < 15 ? m->cp_emb[tbl] : m->codec_emb;
            if (tbl < 15) m->cp_qkv0[tbl] = cur; else m->sem_qkv0 = cur;
            for (int r0 = 0; r0 < rows && e == hipSuccess; r0 += 8) {
                const int M = (rows - r0) < 8 ? (rows - r0) : 8;
                const float* x = prow ? prow + (size_t)r0 * CHh : xin;
                if (!prow) {
                    uint32_t h[8]; for (int i = 0; i < 8; ++i) h[i] = (uint32_t)(r0 + (i < M ? i : 0));
                    e = hipMemcpyAsync(ids, h, sizeof h, hipMemcpyHostToDevice, 0);
                    if (e == hipSuccess) e = launch_gather_rows_bf16(emb, ids, xin, M, CHh, 0);
                    if (e == hipSuccess) e = hipStreamSynchronize(0);
                }
                LinArgs a;
                a.N = QKVD; a.K = CHh; set_w(a, m->cl[0].qkv, 8, QKVD, CHh); a.x = x; a.ldx = CHh; a.norm_w = m->cl[0].in_ln; a.eps = c.rms_eps;
                a.y = cur + (size_t)r0 * QKVD; a.ldy = QKVD; a.M = M; a.epi = EPI_NONE;
                if (e == hipSuccess) e = launch_linear(a, 0);
                if (!prow && e == hipSuccess) e = hipStreamSynchronize(0);
            }
            cur += (size_t)rows * QKVD;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(0);
        hipFree(xin); hipFree(ids);
        if (e != hipSuccess) return set_err(Q3_HIP_ERROR, "layer-0 qkv tables: %s", hipGetErrorString(e));
    }

    // RoPE tables on the host with libm (bit-identical to the CPU oracle): transformer.rs:78-92, 133-175
    if (!m->rope_cos) {
        m->rope_len = 8192;
        std::vector<float> cs((size_t)m->rope_len * 64), sn((size_t)m->rope_len * 64);
        for (int i = 0; i < 64; ++i) {
            const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)HEAD_DIM);
            for (int p = 0; p < m->rope_len; ++p) {
                const float f = (float)p * inv;
                cs[(size_t)p * 64 + i] = cosf(f); sn[(size_t)p * 64 + i] = sinf(f);
            }
        }
        HIPC(hipMalloc((void**)&m->rope_cos, cs.size() * 4));
        HIPC(hipMalloc((void**)&m->rope_sin, sn.size() * 4));
        HIPC(hipMemcpy(m->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(m->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    }

    // decoder pointers + derived tensors (normalised codebooks, snake tables)
    const int CB = c.dec_cb_size, CD = c.dec_cb_dim;
    size_t n_snake = 0;
    { int cin = c.dec_dim; for (int b = 0; b < 4; ++b) { n_snake += cin; cin /= 2; n_snake += (size_t)cin * 6; } n_snake += cin; }
    const size_t derived_floats = (size_t)16 * CB * CD + 2 * n_snake;
    if (!m->derived) HIPC(hipMalloc((void**)&m->derived, derived_floats * 4));
    float* cursor = m->derived;
    std::vector<const float*> rest(15);
    auto CBOOK = [&](const std::string& p) -> const float* {
        float* dst = cursor; cursor += (size_t)CB * CD;
        launch_norm_codebook(P<float>(m, p + "._codebook.embedding_sum"), P<float>(m, p + "._codebook.cluster_usage"), dst, CB, CD, 0);
        return dst;
    };
    m->first_cb = CBOOK("decoder.quantizer.rvq_first.vq.layers.0");
    for (int i = 0; i < 15; ++i) rest[i] = CBOOK(fmt("decoder.quantizer.rvq_rest.vq.layers.%d", i));
    if (!m->rest_cbs_dev) HIPC(hipMalloc((void**)&m->rest_cbs_dev, 15 * sizeof(void*)));
    HIPC(hipMemcpy((void*)m->rest_cbs_dev, rest.data(), 15 * sizeof(void*), hipMemcpyHostToDevice));
    auto SNAKE = [&](const std::string& pa, const std::string& pb, int C, const float*& a, const float*& ib) {
        float* da = cursor; cursor += C; float* di = cursor; cursor += C;
        launch_snake_tables(P<float>(m, pa), P<float>(m, pb), da, di, C, 0);
        a = da; ib = di;
    };
    m->first_proj = P<float>(m, "decoder.quantizer.rvq_first.output_proj.weight");
    m->rest_proj = P<float>(m, "decoder.quantizer.rvq_rest.output_proj.weight");
    m->pre_w = P<float>(m, "decoder.pre_conv.conv.weight"); m->pre_b = P<float>(m, "decoder.pre_conv.conv.bias");
    m->inp_w = P<float>(m, "decoder.pre_transformer.input_proj.weight"); m->inp_b = P<float>(m, "decoder.pre_transformer.input_proj.bias");
    m->outp_w = P<float>(m, "decoder.pre_transformer.output_proj.weight"); m->outp_b = P<float>(m, "decoder.pre_transformer.output_proj.bias");
    m->dec_norm = P<float>(m, "decoder.pre_transformer.norm.weight");
    m->dl.resize(c.dec_layers);
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = fmt("decoder.pre_transformer.layers.%d", i);
        DecLayerW& L = m->dl[i];
        L.in_ln = P<float>(m, p + ".input_layernorm.weight");
        L.q = P<float>(m, p + ".self_attn.q_proj.weight"); L.k = P<float>(m, p + ".self_attn.k_proj.weight");
        L.v = P<float>(m, p + ".self_attn.v_proj.weight"); L.o = P<float>(m, p + ".self_attn.o_proj.weight");
        L.attn_scale = P<float>(m, p + ".self_attn_layer_scale.scale");
        L.post_ln = P<float>(m, p + ".post_attention_layernorm.weight");
        L.gate = P<float>(m, p + ".mlp.gate_proj.weight"); L.up = P<float>(m, p + ".mlp.up_proj.weight");
        L.down = P<float>(m, p + ".mlp.down_proj.weight"); L.mlp_scale = P<float>(m, p + ".mlp_layer_scale.scale");
    }
    for (int i = 0; i < 2; ++i) {
        const std::string p = fmt("decoder.upsample.%d", i);
        UpW& U = m->up[i]; U.ratio = c.dec_up_ratios[i];
        U.tw = P<float>(m, p + ".0.conv.weight"); U.tb = P<float>(m, p + ".0.conv.bias");
        U.dww = P<float>(m, p + ".1.dwconv.conv.weight"); U.dwb = P<float>(m, p + ".1.dwconv.conv.bias");
        U.nw = P<float>(m, p + ".1.norm.weight"); U.nb = P<float>(m, p + ".1.norm.bias");
        U.p1w = P<float>(m, p + ".1.pwconv1.weight"); U.p1b = P<float>(m, p + ".1.pwconv1.bias");
        U.p2w = P<float>(m, p + ".1.pwconv2.weight"); U.p2b = P<float>(m, p + ".1.pwconv2.bias");
        U.gamma = P<float>(m, p + ".1.gamma");
    }
    m->init_w = P<float>(m, "decoder.decoder.0.conv.weight"); m->init_b = P<float>(m, "decoder.decoder.0.conv.bias");
    int cin = c.dec_dim;
    for (int b = 0; b < 4; ++b) {
        DecBlockW& B = m->blk[b];
        const std::string p = fmt("decoder.decoder.%d.block", b + 1);
        B.cin = cin; B.cout = cin / 2; B.rate = c.dec_up_rates[b];
        SNAKE(p + ".0.alpha", p + ".0.beta", cin, B.a, B.ib);
        B.tw = P<float>(m, p + ".1.conv.weight"); B.tb = P<float>(m, p + ".1.conv.bias");
        for (int u = 0; u < 3; ++u) {
            const std::string q = fmt("%s.%d", p.c_str(), u + 2);
            ResUnitW& R = B.res[u];
            SNAKE(q + ".act1.alpha", q + ".act1.beta", B.cout, R.a1, R.ib1);
            R.c1w = P<float>(m, q + ".conv1.conv.weight"); R.c1b = P<float>(m, q + ".conv1.conv.bias");
            SNAKE(q + ".act2.alpha", q + ".act2.beta", B.cout, R.a2, R.ib2);
            R.c2w = P<float>(m, q + ".conv2.conv.weight"); R.c2b = P<float>(m, q + ".conv2.conv.bias");
        }
        cin = B.cout;
    }
    SNAKE("decoder.decoder.5.alpha", "decoder.decoder.5.beta", cin, m->fin_a, m->fin_ib);
    m->fin_w = P<float>(m, "decoder.decoder.6.conv.weight"); m->fin_b = P<float>(m, "decoder.decoder.6.conv.bias");
    // bf16x3 copies of every vocoder conv / linear weight the matrix-core kernel can take (cout % 32 == 0, cin % 16 == 0):
    // (pointer, cout, cin, taps, phases); transposed convs are already stored per phase [stride][cout][cin][taps]
    {
        struct PW { const float* w; int cout, cin, k, phases; };
        std::vector<PW> list;
        const int CDm = c.dec_cb_dim, Qm = c.dec_q_dim, LATm = c.dec_latent, DHm = c.dec_hidden, QDm = c.dec_heads * c.dec_head_dim, DIm = c.dec_inter;
        list.push_back({m->first_proj, Qm, CDm, 1, 1}); list.push_back({m->rest_proj, Qm, CDm, 1, 1});
        list.push_back({m->pre_w, LATm, Qm, 3, 1});
        list.push_back({m->inp_w, DHm, LATm, 1, 1}); list.push_back({m->outp_w, LATm, DHm, 1, 1});
        for (auto& L : m->dl) {
            list.push_back({L.q, QDm, DHm, 1, 1}); list.push_back({L.k, QDm, DHm, 1, 1}); list.push_back({L.v, QDm, DHm, 1, 1});
            list.push_back({L.o, DHm, QDm, 1, 1}); list.push_back({L.gate, DIm, DHm, 1, 1}); list.push_back({L.up, DIm, DHm, 1, 1});
            list.push_back({L.down, DHm, DIm, 1, 1});
        }
        for (int i = 0; i < 2; ++i) {
            list.push_back({m->up[i].tw, LATm, LATm, 1, m->up[i].ratio});
            list.push_back({m->up[i].p1w, 4 * LATm, LATm, 1, 1}); list.push_back({m->up[i].p2w, LATm, 4 * LATm, 1, 1});
        }
        list.push_back({m->init_w, c.dec_dim, LATm, 7, 1});
        for (int b = 0; b < 4; ++b) {
            const DecBlockW& B = m->blk[b];
            list.push_back({B.tw, B.cout, B.cin, 2, B.rate});
            for (int u = 0; u < 3; ++u) { list.push_back({B.res[u].c1w, B.cout, B.cout, 7, 1}); list.push_back({B.res[u].c2w, B.cout, B.cout, 1, 1}); }
        }
        size_t total = 0;
        for (auto& e : list) if (e.cout % 32 == 0 && e.cin % 16 == 0) total += packed_conv_w_bytes(e.cout, e.cin, e.k) * (size_t)e.phases;
        m->wpk.clear();
        if (total) {
            if (!m->wpk_arena) HIPC(hipMalloc(&m->wpk_arena, total));
            char* cur = (char*)m->wpk_arena;
            for (auto& e : list) {
                if (e.cout % 32 || e.cin % 16) continue;
                const size_t per = packed_conv_w_bytes(e.cout, e.cin, e.k);
                m->wpk[e.w] = cur;
                for (int ph = 0; ph < e.phases; ++ph)
                    HIPC(launch_pack_conv_w(e.w + (size_t)ph * e.cout * e.cin * e.k, cur + (size_t)ph * per, e.cout, e.cin, e.k, 0));
                cur += per * (size_t)e.phases;
            }
        }
    }
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    // the f32 page pool's first slab now, not inside the first session's prefill (its ~1 GB hipMalloc sat on the first request's
    // time to first audio); a failure here is not fatal — the first session will report it. Q3_KV_NO_PREWARM=1: lazily, as before
    if (!getenv("Q3_KV_NO_PREWARM")) { if (m->kv_pool.prewarm() != hipSuccess) (void)hipGetLastError(); }
    m->finalized = true;
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// device buffer helpers
// ------------------------------------------------------------------------------------------------
// Session-lifetime buffers (KV pages, workspaces, vocoder scratch) come from a small per-device cache of exact-size
// blocks: a server creates sessions of a few recurring shapes, and hipMalloc / hipFree of multi-GB blocks on the
// request path costs driver time that varies from box to box (VRAM clearing, page-table work) — up to 150 ms per
// 16-utterance session was seen in otherwise identical runs. Blocks are returned only by owners that have
// synchronised the streams that used them (q3_session_free, prefill_gemm); at most Q3_DEV_CACHE_MB (default 32768)
// MB stay cached per process, beyond that blocks go back to the driver. HBM is 288 GB: capacity is not the constraint.
namespace {
struct DevCache {
    std::mutex mu;
    std::unordered_multimap<uint64_t, void*> free_blocks;      // key = device << 48 | bytes
    std::unordered_map<void*, uint64_t> live;                  // blocks handed out by get()
    size_t cached = 0, cap = 0;
    DevCache() { const char* e = getenv("Q3_DEV_CACHE_MB"); cap = (size_t)(e ? atol(e) : 32768) << 20; }
    static uint64_t key(int dev, size_t bytes) { return ((uint64_t)dev << 48) | (uint64_t)bytes; }
    // size classes: 16 per octave (<= 6.25 % slack), so that the side sessions of a continuous-batching server — one KV extent
    // per prompt length — reuse each other's blocks instead of leaving one cached block per distinct length
    static size_t size_class(size_t bytes) {
        if (bytes <= 4096) return (bytes + 255) & ~(size_t)255;
        size_t p2 = 1; while ((p2 << 1) <= bytes) p2 <<= 1;
        const size_t step = p2 >> 4;
        return (bytes + step - 1) / step * step;
    }
    hipError_t get(void** p, size_t bytes) {
        bytes = size_class(bytes);
        int dev = 0; (void)hipGetDevice(&dev);
        const uint64_t k = key(dev, bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_blocks.find(k);
            if (it != free_blocks.end()) { *p = it->second; free_blocks.erase(it); cached -= bytes; live[*p] = k; return hipSuccess; }
        }
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {                                  // out of memory: give the cache back and retry once
            trim(0);
            e = hipMalloc(p, bytes);
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> g(mu);
        live[*p] = k;
        return hipSuccess;
    }
    void put(void* p) {
        if (!p) return;
        uint64_t k = 0; bool known = false;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live.find(p);
            if (it != live.end()) { k = it->second; known = true; live.erase(it); }
            const size_t bytes = (size_t)(k & 0xffffffffffffull);
            if (known && cap && cached + bytes <= cap) { free_blocks.emplace(k, p); cached += bytes; return; }
        }
        (void)hipFree(p);
    }
    void trim(size_t keep) {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto it = free_blocks.begin(); it != free_blocks.end() && cached > keep;) {
                cached -= (size_t)(it->first & 0xffffffffffffull); drop.push_back(it->second); it = free_blocks.erase(it);
            }
        }
        for (void* p : drop) (void)hipFree(p);
    }
};
DevCache& dev_cache() { static DevCache* c = new DevCache(); return *c; }    // leaked on purpose: outlives every static destructor
}  // namespace
static inline hipError_t dev_malloc(void** p, size_t bytes) { return dev_cache().get(p, bytes ? bytes : 4); }
static inline void dev_free(void* p) { dev_cache().put(p); }

struct DevPool {
    std::vector<void*> ptrs;
    // lazy: the zero-fills of a run of allocations are only ISSUED (null stream, in order with the synchronous hipMemcpy
    // uploads that may follow) and settle() waits for all of them once — a session is ~45 buffers, and a memset + a device
    // synchronisation each was 2-3 ms of every q3_session_create (the side session of a continuous-batching swap)
    bool lazy = false;
    hipError_t settle() { return hipStreamSynchronize(nullptr); }
    template <typename T> hipError_t alloc(T** p, size_t count) {
        void* q = nullptr;
        hipError_t e = dev_malloc(&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return e;
        ptrs.push_back(q); *p = (T*)q;
        // zero-fill on the null stream and WAIT for it: the users launch on non-blocking streams, which the null stream
        // does not order against — a memset still in flight would wipe what their first kernels write (seen once the
        // blocks started coming from the cache instead of a slow hipMalloc)
        hipError_t m = hipMemsetAsync(q, 0, (count ? count : 1) * sizeof(T), nullptr);
        if (m != hipSuccess) return m;
        return lazy ? hipSuccess : hipStreamSynchronize(nullptr);
    }
    void release_all() {
        if (lazy) (void)hipStreamSynchronize(nullptr);      // a creation that failed midway: no zero-fill may outlive its block's ownership
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

// T = frames through the convolutional stack in one call, Tf = frames through the front (Tf >= T)
static q3_status codec_reserve(const q3_model* m, CodecWS& ws, int T, int Tf = 0) {
    if (Tf < T) Tf = T;
    if (T <= ws.cap_frames && Tf <= ws.cap_front) return Q3_OK;
    if (T < ws.cap_frames) T = ws.cap_frames;
    if (Tf < ws.cap_front) Tf = ws.cap_front;
    if (ws.bufA) HIPC(hipDeviceSynchronize());        // growing: the old blocks go back to the cache, nothing may still be using them
    ws.release();
    const q3_config& c = m->cfg;
    int up = 1; for (int i = 0; i < 2; ++i) up *= c.dec_up_ratios[i];
    // largest [C][L] activation per frame
    size_t per = (size_t)4 * c.dec_latent * up;                 // ConvNeXt hidden 4*LAT × (T*up)
    { size_t L = up; int C = c.dec_dim; per = per > (size_t)C * L ? per : (size_t)C * L;
      for (int b = 0; b < 4; ++b) { L *= c.dec_up_rates[b]; C /= 2; if ((size_t)C * L > per) per = (size_t)C * L; } }
    const int QDm = c.dec_heads * c.dec_head_dim;
    // front: A holds q|k|v|attn-out (4*QD rows), B gate|up (2*DI) or the quantiser output, C the latent — per front frame
    size_t per_front = (size_t)4 * QDm;
    if ((size_t)2 * c.dec_inter > per_front) per_front = (size_t)2 * c.dec_inter;
    if ((size_t)2 * c.dec_cb_dim > per_front) per_front = (size_t)2 * c.dec_cb_dim;
    if ((size_t)c.dec_latent > per_front) per_front = (size_t)c.dec_latent;
    if ((size_t)c.dec_q_dim > per_front) per_front = (size_t)c.dec_q_dim;
    size_t n = per * (size_t)T;
    if (per_front * (size_t)Tf > n) n = per_front * (size_t)Tf;
    HIPC(dev_malloc((void**)&ws.bufA, n * 4)); HIPC(dev_malloc((void**)&ws.bufB, n * 4)); HIPC(dev_malloc((void**)&ws.bufC, n * 4));
    HIPC(dev_malloc((void**)&ws.bufF, n * 4));
    const size_t small = (size_t)Tf * (size_t)(QDm > c.dec_latent ? QDm : c.dec_latent);
    HIPC(dev_malloc((void**)&ws.bufD, small * 4)); HIPC(dev_malloc((void**)&ws.bufE, small * 4));
    HIPC(dev_malloc((void**)&ws.cs, (size_t)Tf * 32 * 4)); HIPC(dev_malloc((void**)&ws.sn, (size_t)Tf * 32 * 4));
    HIPC(dev_malloc((void**)&ws.frames, (size_t)Tf * 16 * 4));
    size_t total_up = up; for (int b = 0; b < 4; ++b) total_up *= c.dec_up_rates[b];
    HIPC(dev_malloc((void**)&ws.pcm, (size_t)T * total_up * 4));
    // RoPE table of the pre-transformer (decoder_12hz.rs:541-553), host libm
    std::vector<float> cs((size_t)Tf * 32), sn((size_t)Tf * 32);
    for (int i = 0; i < 32; ++i) {
        const float inv = 1.0f / powf(c.dec_theta, (float)(2 * i) / (float)c.dec_head_dim);
        for (int t = 0; t < Tf; ++t) { const float f = (float)t * inv; cs[(size_t)t * 32 + i] = cosf(f); sn[(size_t)t * 32 + i] = sinf(f); }
    }
    HIPC(hipMemcpy(ws.cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ws.sn, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    ws.cap_frames = T; ws.cap_front = Tf;
    return Q3_OK;
}

static int samples_per_frame(const q3_config& c) {
    int u = 1; for (int i = 0; i < 2; ++i) u *= c.dec_up_ratios[i]; for (int i = 0; i < 4; ++i) u *= c.dec_up_rates[i];
    return u;
}

static thread_local const q3_model* tl_codec_model = nullptr;     // set by codec_decode_dev: packed-weight lookup of the helpers below
static const void* packed_of(const float* w) { return tl_codec_model ? tl_codec_model->pk(w) : nullptr; }
// bf16 planes per operand in the vocoder's matrix-core convs (q3_model_set_codec_planes): only codec_decode_dev sets 2,
// for its own launches — the encoders that share the kernels (speaker / speech tokenizer) always run the exact products
static thread_local int tl_codec_planes = 3;
struct CodecPlanesScope { explicit CodecPlanesScope(int p) { tl_codec_planes = p; } ~CodecPlanesScope() { tl_codec_planes = 3; } };
static hipError_t conv1(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L, hipStream_t st,
                        const float* resid = nullptr, const float* scale = nullptr, int act = 0,
                        const float* sa = nullptr, const float* sib = nullptr) {
    ConvArgs a; a.x = x; a.w = w; a.b = b; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.k = 1; a.dil = 1;
    a.resid = resid; a.scale = scale; a.act = act; a.snake_a = sa; a.snake_b = sib; a.wpk = packed_of(w); a.planes = tl_codec_planes;
    return launch_conv1d(a, st);
}
static hipError_t convk(const float* x, const float* w, const float* b, float* y, int cin, int cout, int L, int k, int dil,
                        hipStream_t st, const float* sa = nullptr, const float* sib = nullptr, int act = 0) {
    ConvArgs a; a.x = x; a.w = w; a.b = b; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.k = k; a.dil = dil;
    a.snake_a = sa; a.snake_b = sib; a.act = act; a.wpk = packed_of(w); a.planes = tl_codec_planes;
    return launch_conv1d(a, st);
}

// frames already on device in ws.frames; result in ws.pcm. taps: host pointers or nullptr.
// c0 = 0: whole-utterance decode, ws.pcm = [T*spf]. c0 > 0 (segment decode): the front (quantiser, pre_conv,
// pre-transformer: everything with unbounded left context, and cheap) runs over all T frames, the convolutional
// stack only over frames [c0, T), and ws.pcm = [(T-c0)*spf]; samples of frames >= c0 + CODEC_CTX_FRAMES are
// identical to the whole-utterance decode (every kernel sums each output in a position-independent order).
static q3_status codec_decode_dev(const q3_model* m, CodecWS& ws, int T, hipStream_t st, float** taps, int c0 = 0) {
    const q3_config& c = m->cfg;
    tl_codec_model = m;
    // Q3_CODEC_PLANES=2|3: A/B aid, overrides q3_model_set_codec_planes for the vocoder only (never the encoders)
    static const int env_planes = [] { const char* e = getenv("Q3_CODEC_PLANES"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 3) ? v : 0; }();
    const int NPL = env_planes ? env_planes : m->codec_planes;
    const CodecPlanesScope planes_scope(NPL);
    const int CD = c.dec_cb_dim, Q = c.dec_q_dim, LAT = c.dec_latent, DH = c.dec_hidden, QD = c.dec_heads * c.dec_head_dim, DI = c.dec_inter;
    auto TAP = [&](int id, const float* dev, size_t n) -> q3_status {
        if (taps && taps[id]) { HIPC(hipStreamSynchronize(st)); HIPC(hipMemcpy(taps[id], dev, n * 4, hipMemcpyDeviceToHost)); }
        return Q3_OK;
    };
    float *A = ws.bufA, *B = ws.bufB, *C = ws.bufC, *D = ws.bufD, *E = ws.bufE;
    // D1 quantiser: E1 = A[0..256T), E2 = A[256T..512T) → quantized in B [Q][T]
    float* e1 = A; float* e2 = A + (size_t)CD * T;
    HIPC(launch_rvq_embed(ws.frames, T, m->first_cb, m->rest_cbs_dev, e1, e2, CD, c.dec_cb_size, st));
    HIPC(conv1(e1, m->first_proj, nullptr, B, CD, Q, T, st));
    HIPC(conv1(e2, m->rest_proj, nullptr, B, CD, Q, T, st, B));
    Q3C(TAP(Q3_DEC_QUANT, B, (size_t)Q * T));
    // D2 pre_conv → C [LAT][T]
    HIPC(convk(B, m->pre_w, m->pre_b, C, Q, LAT, T, 3, 1, st));
    Q3C(TAP(Q3_DEC_PRECONV, C, (size_t)LAT * T));
    // D3 pre-transformer. hidden Hd = D [DH][T]
    float* Hd = D;
    HIPC(conv1(C, m->inp_w, m->inp_b, Hd, LAT, DH, T, st));
    float* Nn = E;                                   // [DH][T]
    float* q = A; float* k = A + (size_t)QD * T; float* v = A + (size_t)2 * QD * T; float* ao = A + (size_t)3 * QD * T;
    float* g = B; float* u = B + (size_t)DI * T;
    const float scale = (float)pow((double)c.dec_head_dim, -0.5);
    for (int l = 0; l < c.dec_layers; ++l) {
        const DecLayerW& L = m->dl[l];
        HIPC(launch_rmsnorm_c(Hd, L.in_ln, Nn, DH, T, c.dec_eps, st));
        // q | k | v (and gate | up below) as ONE launch when their packed weights sit back to back in the arena (they are
        // packed in this order at finalize: concatenating A-operand tiles along the output channels is just adjacency) —
        // 3 x 160 workgroups of 29 us each become one grid of 480; per output the same arithmetic
        // Only the bf16x3 kernel reads the packed images; its fallbacks (Q3_CONV_F32 / Q3_CONV_VALU A/B switches, shapes
        // outside its divisibility rules) read the f32 tensor of ONE projection, so the fusion is tied to that path and to
        // the f32 tensors being adjacent as well (a fallback would otherwise run past the end of L.q / L.gate).
        auto adjacent = [&](const float* w0, const float* w1, int cout, int cin) {
            const char* p0 = (const char*)m->pk(w0); const char* p1 = (const char*)m->pk(w1);
            return p0 && p1 && p1 == p0 + packed_conv_w_bytes(cout, cin, 1) && w1 == w0 + (size_t)cout * cin &&
                   cout % 32 == 0 && cin % 16 == 0;
        };
        static const bool no_fuse_qkv = getenv("Q3_CODEC_NO_QKV_FUSE") != nullptr || getenv("Q3_CONV_F32") != nullptr ||
                                        getenv("Q3_CONV_VALU") != nullptr;      // A/B aids
        if (!no_fuse_qkv && adjacent(L.q, L.k, QD, DH) && adjacent(L.k, L.v, QD, DH)) {
            HIPC(conv1(Nn, L.q, nullptr, q, DH, 3 * QD, T, st));
        } else {
            HIPC(conv1(Nn, L.q, nullptr, q, DH, QD, T, st));
            HIPC(conv1(Nn, L.k, nullptr, k, DH, QD, T, st));
            HIPC(conv1(Nn, L.v, nullptr, v, DH, QD, T, st));
        }
        HIPC(launch_rope_c(q, k, ws.cs, ws.sn, c.dec_heads, c.dec_head_dim, T, st));
        HIPC(launch_attn_c(q, k, v, ao, c.dec_heads, c.dec_head_dim, T, scale, st));
        HIPC(conv1(ao, L.o, nullptr, Hd, QD, DH, T, st, Hd, L.attn_scale));
        HIPC(launch_rmsnorm_c(Hd, L.post_ln, Nn, DH, T, c.dec_eps, st));
        if (!no_fuse_qkv && adjacent(L.gate, L.up, DI, DH)) {
            HIPC(conv1(Nn, L.gate, nullptr, g, DH, 2 * DI, T, st));
        } else {
            HIPC(conv1(Nn, L.gate, nullptr, g, DH, DI, T, st));
            HIPC(conv1(Nn, L.up, nullptr, u, DH, DI, T, st));
        }
        HIPC(launch_silu_mul(g, u, g, (int64_t)DI * T, st));
        HIPC(conv1(g, L.down, nullptr, Hd, DI, DH, T, st, Hd, L.mlp_scale));
    }
    HIPC(launch_rmsnorm_c(Hd, m->dec_norm, Nn, DH, T, c.dec_eps, st));
    HIPC(conv1(Nn, m->outp_w, m->outp_b, C, DH, LAT, T, st));      // C [LAT][T]
    Q3C(TAP(Q3_DEC_PRETRANS, C, (size_t)LAT * T));
    // D4 upsample stages: cur in C (segment decode: the latent columns [c0, T) copied out to F)
    float* cur = C; float* o1 = A; float* o2 = B;
    int L = T;
    if (c0 > 0) {
        HIPC(launch_copy_rows(C + c0, T, ws.bufF, T - c0, LAT, T - c0, st));
        cur = ws.bufF; L = T - c0;
    }
    for (int i = 0; i < 2; ++i) {
        const UpW& U = m->up[i];
        float* upo = (cur == C) ? A : C;            // transconv output [LAT][L*r]
        HIPC(launch_transconv1d_taps(cur, U.tw, U.tb, upo, LAT, LAT, L, U.ratio, 1, nullptr, nullptr, st, nullptr, nullptr, nullptr, m->pk(U.tw), NPL));
        L *= U.ratio;
        // dwconv → LN → pw1+GELU → pw2·gamma + residual (in place into upo)
        float* dw = (upo == A) ? C : A;             // [LAT][L]
        HIPC(launch_dwconv7(upo, U.dww, U.dwb, dw, LAT, L, st));
        float* ln = dw + (size_t)LAT * L;           // second half of that buffer
        HIPC(launch_layernorm_c(dw, U.nw, U.nb, ln, LAT, L, 1e-6f, st));
        HIPC(conv1(ln, U.p1w, U.p1b, B, LAT, 4 * LAT, L, st, nullptr, nullptr, 1));
        HIPC(conv1(B, U.p2w, U.p2b, upo, 4 * LAT, LAT, L, st, upo, U.gamma));
        cur = upo;
        Q3C(TAP(Q3_DEC_UP0 + i, cur, (size_t)LAT * L));
    }
    (void)o1; (void)o2;
    // D5-D9. SnakeBeta is applied by the PRODUCER's epilogue (each element activated once, not once per
    // consuming output-channel tile): every tensor below exists as "raw" (residual / tap) and/or "act"
    // (= snake of the next consumer).
    int Cc = c.dec_dim;
    float* pool4[4] = {A, B, C, ws.bufF};
    auto other = [&](std::initializer_list<const float*> used) -> float* {
        for (float* p : pool4) { bool u = false; for (const float* q : used) if (q == p) u = true; if (!u) return p; }
        return nullptr;
    };
    // decoder.0 (k=7): raw only if tapped; activated with block 0's snake
    float* xact = other({cur});
    {
        ConvArgs a; a.x = cur; a.w = m->init_w; a.wpk = m->pk(m->init_w); a.b = m->init_b; a.cin = LAT; a.cout = Cc; a.L = L; a.k = 7; a.dil = 1; a.planes = NPL;
        a.post_a = m->blk[0].a; a.post_ib = m->blk[0].ib;
        if (taps && taps[Q3_DEC_INIT]) { float* raw = other({cur, xact}); a.y = raw; a.y2 = xact; HIPC(launch_conv1d(a, st)); Q3C(TAP(Q3_DEC_INIT, raw, (size_t)Cc * L)); }
        else { a.y = xact; HIPC(launch_conv1d(a, st)); }
    }
    static const int dils[3] = {1, 3, 9};
    for (int b = 0; b < 4; ++b) {
        const DecBlockW& Bk = m->blk[b];
        // transposed conv: raw Y (residual of unit 0) + YA = snake(act1 of unit 0)
        float* Y = other({xact});
        float* YA = other({xact, Y});
        HIPC(launch_transconv1d_taps(xact, Bk.tw, Bk.tb, Y, Bk.cin, Bk.cout, L, Bk.rate, 2, nullptr, nullptr, st, Bk.res[0].a1, Bk.res[0].ib1, YA, m->pk(Bk.tw), NPL));
        L *= Bk.rate; Cc = Bk.cout;
        float* T2 = other({Y, YA});
        for (int uu = 0; uu < 3; ++uu) {
            const ResUnitW& R = Bk.res[uu];
            const float* nxt_a = uu < 2 ? Bk.res[uu + 1].a1 : (b < 3 ? m->blk[b + 1].a : m->fin_a);
            const float* nxt_ib = uu < 2 ? Bk.res[uu + 1].ib1 : (b < 3 ? m->blk[b + 1].ib : m->fin_ib);
            {   // 96 / 192 channels: the whole unit in one launch (raw tensor updated in place, activated copy into T2)
                ResUnitArgs r{};
                r.xa = YA; r.y = Y; r.ya = T2; r.w1pk = m->pk(R.c1w); r.w2pk = m->pk(R.c2w); r.b1 = R.c1b; r.b2 = R.c2b;
                r.mid_a = R.a2; r.mid_ib = R.ib2; r.post_a = nxt_a; r.post_ib = nxt_ib; r.C = Cc; r.L = L; r.dil = dils[uu]; r.planes = NPL;
                const hipError_t e = launch_resunit(r, st);
                if (e == hipSuccess) { float* t = YA; YA = T2; T2 = t; continue; }
                if (e != hipErrorNotSupported) HIPC(e);
            }
            {   // conv7 (dilated) on the activated input; output activated with act2
                ConvArgs a; a.x = YA; a.w = R.c1w; a.wpk = m->pk(R.c1w); a.b = R.c1b; a.y = T2; a.cin = Cc; a.cout = Cc; a.L = L; a.k = 7; a.dil = dils[uu]; a.planes = NPL;
                a.post_a = R.a2; a.post_ib = R.ib2;
                HIPC(launch_conv1d(a, st));
            }
            {   // conv1 + residual: raw → Y (in place), activated → YA for the next consumer
                ConvArgs a; a.x = T2; a.w = R.c2w; a.wpk = m->pk(R.c2w); a.b = R.c2b; a.y = Y; a.y2 = YA; a.cin = Cc; a.cout = Cc; a.L = L; a.k = 1; a.dil = 1; a.resid = Y; a.planes = NPL;
                if (uu < 2) { a.post_a = Bk.res[uu + 1].a1; a.post_ib = Bk.res[uu + 1].ib1; }
                else if (b < 3) { a.post_a = m->blk[b + 1].a; a.post_ib = m->blk[b + 1].ib; }
                else { a.post_a = m->fin_a; a.post_ib = m->fin_ib; }
                HIPC(launch_conv1d(a, st));
            }
        }
        Q3C(TAP(Q3_DEC_BLK0 + b, Y, (size_t)Cc * L));
        xact = YA;
    }
    // D9 final conv on the activated tensor + clamp
    HIPC(convk(xact, m->fin_w, m->fin_b, ws.pcm, Cc, 1, L, 7, 1, st, nullptr, nullptr, 2));
    return Q3_OK;
}

extern "C" q3_status q3_decode_codes(q3_model* m, const uint32_t* frames_host, int n_frames, float* pcm_host, float** taps_host) {
    if (!m || !m->finalized) return set_err(Q3_INVALID_ARG, "model not finalized");
    if (n_frames < 0 || (n_frames > 0 && (!frames_host || !pcm_host))) return set_err(Q3_INVALID_ARG, "q3_decode_codes: bad argument");
    if (n_frames == 0) return Q3_OK;
    for (int f = 0; f < n_frames; ++f)
        for (int g = 1; g < 16; ++g)
            if (frames_host[(size_t)f * 16 + g] >= (uint32_t)m->cfg.dec_cb_size)
                return set_err(Q3_INVALID_ARG, "code %u out of range for codebook %d (frame %d)", frames_host[(size_t)f * 16 + g], g, f);
    HIPC(hipSetDevice(m->device));
    CodecWS ws;
    q3_status st = codec_reserve(m, ws, n_frames);
    if (st == Q3_OK) {
        hipError_t e = hipMemcpy(ws.frames, frames_host, (size_t)n_frames * 16 * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = set_err(Q3_HIP_ERROR, "hipMemcpy frames: %s", hipGetErrorString(e));
    }
    if (st == Q3_OK) st = codec_decode_dev(m, ws, n_frames, 0, taps_host);
    if (st == Q3_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(pcm_host, ws.pcm, (size_t)n_frames * samples_per_frame(m->cfg) * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = set_err(Q3_HIP_ERROR, "decode: %s", hipGetErrorString(e));
    }
    ws.release();
    return st;
}

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
    CodecWS seg_ws; float* pcm_all = nullptr; size_t pcm_all_floats = 0;
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

// Everything this session has in flight has landed: the frames on the library's own AQL queue (q3_aql.cpp; not ordered with any
// HIP stream) and whatever rides the session's stream. Every host-side wait of the engine goes through here.
static hipError_t sync_frames(q3_session* s) {
    if (s->aql) {
        std::string why;
        if (!q3::aql_wait(s->aql, &why)) { s->aql_failed = true; set_err(Q3_HIP_ERROR, "AQL frame submission: %s", why.c_str()); return hipErrorUnknown; }
    }
    return hipStreamSynchronize(s->stream);
}

static hipError_t run_linear(q3_session* s, const LinArgs& a_in) {
    LinArgs a = a_in;
    a.ws = s->wide_ws; a.ws_bytes = s->wide_ws_bytes;
#ifdef Q3_TRACE
    a.trace = s->trace_next(0, a.M, a.N, a.K, a.epi, a.norm_w ? 1 : 0, a.tiled);
#endif
    if (!s->profile) return launch_linear(a, s->stream);
    // profiling: bracket the launch with event records (inside graph capture these become event-record
    // nodes, so the timestamps are taken on the GPU timeline without host launch latency in between)
    hipEvent_t e0, e1;
    if (s->prof_pool_next + 2 <= s->prof_pool.size()) { e0 = s->prof_pool[s->prof_pool_next++]; e1 = s->prof_pool[s->prof_pool_next++]; }
    else {
        hipError_t e = hipEventCreate(&e0); if (e != hipSuccess) return e;
        e = hipEventCreate(&e1); if (e != hipSuccess) return e;
        s->prof_pool.push_back(e0); s->prof_pool.push_back(e1); s->prof_pool_next = s->prof_pool.size();
    }
    hipError_t e = hipEventRecord(e0, s->stream); if (e != hipSuccess) return e;
    e = launch_linear(a, s->stream);
    hipError_t e2 = hipEventRecord(e1, s->stream);
    s->prof_events.push_back({e0, e1});
    s->prof_event_bytes.push_back((double)a.N * a.K * 2.0 * (a.epi == EPI_SWIGLU ? 2.0 : 1.0));
    {   // launch inventory (q3_session_profile_shapes): M, N, K, epilogue, fused input norm, reserved, tiling
        ProfShape ps{a.M, a.N, a.K, a.epi, a.norm_w ? 1 : 0, 0, a.ksplit == 2 ? 3 : a.tiled, 1};
        bool found = false;
        for (auto& q : s->prof_shapes)
            if (q.M == ps.M && q.N == ps.N && q.K == ps.K && q.epi == ps.epi && q.rms == ps.rms && q.produce == ps.produce && q.tiled == ps.tiled) { q.count += 1; found = true; break; }
        if (!found) s->prof_shapes.push_back(ps);
    }
    return e != hipSuccess ? e : e2;
}

// Paged KV bookkeeping (KvPool): row b gets the pages for positions [0, n_pos) it does not hold yet; the new table entries
// are queued on the session's stream ahead of the kernels that read them (hipMemcpyAsync stages pageable sources before it
// returns; kv_rows[b] never reallocates: reserved to KV_MAX_PAGES at creation).
static q3_status kv_reserve_row(q3_session* s, int b, int n_pos) {
    if (!s->paged) return Q3_OK;
    if (n_pos > KV_MAX_PAGES * KV_PAGE_POS) return set_err(Q3_KV_OVERFLOW, "%d positions exceed a row's page table (%d)", n_pos, KV_MAX_PAGES * KV_PAGE_POS);
    std::vector<float*>& row = s->kv_rows[(size_t)b];
    const int need = (n_pos + KV_PAGE_POS - 1) / KV_PAGE_POS, have = (int)row.size();
    if (need <= have) return Q3_OK;
    KvPool& pool = s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool;
    if (pool.take(need - have, row) != hipSuccess)
    {
        s->kv_overflow_row = b;
        return set_err(Q3_KV_OVERFLOW, "KV page pool exhausted: row %d needs %d more %s page(s) of %d positions (budget: %ld of %ld half-pages in use)",
                       b, need - have, s->kv_in_bf16 ? "bf16" : "f32", KV_PAGE_POS, s->m->kv_budget.used, s->m->kv_budget.limit);
    }
    static_assert(sizeof(float*) == sizeof(unsigned long long), "page table entries are 64-bit pointers");
    HIPC(hipMemcpyAsync(s->kv_table + (size_t)b * KV_MAX_PAGES + have, row.data() + have, (size_t)(need - have) * 8, hipMemcpyHostToDevice, s->stream));
    return Q3_OK;
}
// pages for what the next `frames` frames of every row can touch: frame f of a row writes position prefill_len + f, a row
// that reached its limit keeps rewriting position prefill_len + limit (k_sample freezes its counters)
static q3_status kv_reserve_frames(q3_session* s, int frames) {
    if (!s->paged) return Q3_OK;
    s->kv_overflow_row = -1;
    for (int b = 0; b < s->B; ++b) {
        const SeqInfo& q = s->seq[(size_t)b];
        if (q.idle) continue;
        int upto = s->frames_run - q.start_run + frames;
        if (upto > q.limit) upto = q.limit;
        if (upto < 0) upto = 0;
        Q3C(kv_reserve_row(s, b, q.prefill_len + upto + 1));
    }
    return Q3_OK;
}
static void kv_release_row(q3_session* s, int b) {        // the caller has drained every stream that may still touch the row
    if (s->paged && !s->kv_rows[(size_t)b].empty()) (s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool).give(s->kv_rows[(size_t)b]);
}

// bf16 sessions: every row's f32 pages -> as many pages of the bf16 pool (k_kv_pages_to_bf16), the table rewritten, the f32
// pages returned. The caller has drained the stream; this drains it again before the f32 pages go back.
static q3_status kv_convert_to_bf16(q3_session* s) {
    if (!s->paged) return set_err(Q3_UNSUPPORTED, "bf16 K/V needs the paged cache");
    const q3_config& c = s->m->cfg;
    std::vector<unsigned long long> src, dst; std::vector<std::vector<float*>> fresh((size_t)s->B);
    q3_status st = Q3_OK;
    for (int b = 0; b < s->B && st == Q3_OK; ++b) {
        const int n = (int)s->kv_rows[(size_t)b].size();
        if (s->m->kv_pool16.take(n, fresh[(size_t)b]) != hipSuccess) { st = set_err(Q3_KV_OVERFLOW, "KV page pool (bf16) exhausted: row %d needs %d page(s)", b, n); break; }
        for (int i = 0; i < n; ++i) { src.push_back((unsigned long long)s->kv_rows[(size_t)b][(size_t)i]); dst.push_back((unsigned long long)fresh[(size_t)b][(size_t)i]); }
    }
    if (st != Q3_OK) { for (auto& f : fresh) if (!f.empty()) s->m->kv_pool16.give(f); return st; }
    const int n = (int)src.size();
    hipError_t e = hipMemcpyAsync(s->kv_conv, src.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->kv_conv + (size_t)s->B * KV_MAX_PAGES, dst.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = launch_kv_pages_to_bf16(s->kv_conv, s->kv_conv + (size_t)s->B * KV_MAX_PAGES, n, c.n_layers, c.n_kv_heads, s->m->kv_pool.layer_stride(), s->m->kv_pool.v_delta(), s->stream);
    if (e == hipSuccess) e = sync_frames(s);
    if (e != hipSuccess) { for (auto& f : fresh) if (!f.empty()) s->m->kv_pool16.give(f); return set_err(Q3_HIP_ERROR, "K/V conversion to bf16: %s", hipGetErrorString(e)); }
    for (int b = 0; b < s->B; ++b) {
        s->m->kv_pool.give(s->kv_rows[(size_t)b]);
        s->kv_rows[(size_t)b].assign(fresh[(size_t)b].begin(), fresh[(size_t)b].end());
        if (!s->kv_rows[(size_t)b].empty())
            HIPC(hipMemcpyAsync(s->kv_table + (size_t)b * KV_MAX_PAGES, s->kv_rows[(size_t)b].data(), s->kv_rows[(size_t)b].size() * 8, hipMemcpyHostToDevice, s->stream));
    }
    HIPC(sync_frames(s));
    s->kv_in_bf16 = true;
    return Q3_OK;
}

// one DecoderLayer (transformer.rs:442-467) for the single new token of every sequence
static q3_status lm_layer(q3_session* s, const LmDims& d, const LayerW& w, LmBuf& b, float* kc, float* vc, int max_seq,
                          const int* pos_dev, int pos_static, int n_splits, int rows_per_seq = 1, bool skip_qkv = false,
                          const CpGatherArgs* fold = nullptr,       // fold: this layer's attention does the pass's gather
                          int paged_layer = -1) {                   // >= 0: the talker's paged cache, this layer's index (kc / vc / max_seq unused)
    const q3_model* m = s->m;
    // B = number of activation ROWS of this step: one per sequence, or rows_per_seq consecutive positions per
    // sequence (chunked prefill, the code predictor's 2-token first pass)
    const int QD = d.nh * HEAD_DIM, KD = d.nkv * HEAD_DIM, B = s->B * rows_per_seq;
    LinArgs a;
    a.N = QD + 2 * KD; a.K = d.H; set_w(a, w.qkv, B, a.N, a.K); a.x = b.X; a.ldx = d.H; a.norm_w = w.in_ln; a.eps = d.eps;
    a.y = b.QKV; a.ldy = QD + 2 * KD; a.M = B; a.epi = EPI_NONE;
    // Wide sessions: the q|k|v GEMM hands its K-slice sums straight to the attention kernel, which adds them on the 3 x 128
    // values it needs (AttnArgs::qkv_part) — the slice-sum launch in between (93 per frame at 1.7B, 5.2 us + gap each) is gone.
    // Only for the one-launch decode attention (static one-row steps); Q3_WIDE_NO_QKV_FUSE=1: off (A/B aid)
    static const bool qkv_fuse = getenv("Q3_WIDE_NO_QKV_FUSE") == nullptr;
    WidePartial wp{nullptr, nullptr, 0};
    if (!skip_qkv) {       // skip: the caller already filled b.QKV (code predictor layer 0, table rows)
        bool fused = false;
        if (qkv_fuse && B >= gemm_wide_min_rows() && rows_per_seq == 1 && !s->legacy_attn && !s->profile && !s->debug && s->wide_ws) {
            LinArgs a2 = a; a2.ws = s->wide_ws; a2.ws_bytes = s->wide_ws_bytes;
            const hipError_t e = launch_gemm_wide_partial(a2, s->stream, &wp);
            if (e == hipSuccess && wp.S <= 8) fused = true;
            else if (e != hipSuccess && e != hipErrorNotSupported) HIPC(e);
            else if (e == hipSuccess) { wp = WidePartial{nullptr, nullptr, 0}; }       // more than eight slices: let the slice-sum launch redo it
        }
        if (!fused) { wp = WidePartial{nullptr, nullptr, 0}; HIPC(run_linear(s, a)); }
    }
    AttnArgs t{};
    t.qkv = b.QKV; t.ld_qkv = QD + 2 * KD; t.q_norm_w = w.q_norm; t.k_norm_w = w.k_norm; t.eps = d.eps;
    t.rope_cos = m->rope_cos; t.rope_sin = m->rope_sin; t.pos_dev = pos_dev; t.pos_static = pos_static;
    t.kcache = kc; t.vcache = vc; t.max_seq = max_seq; t.qbuf = b.Q; t.part = b.PART; t.out = b.ATT; t.ld_out = QD;
    t.B = B; t.nh = d.nh; t.nkv = d.nkv; t.n_splits = n_splits; t.rows_per_seq = rows_per_seq;
    if (paged_layer >= 0) {
        t.kv_pages = s->kv_table; t.kv_layer_off = (size_t)paged_layer * s->m->kv_pool.layer_stride();
        t.kv_vdelta = s->m->kv_pool.v_delta();
        t.kv_row_pages = (max_seq + KV_PAGE_POS - 1) / KV_PAGE_POS;
        t.kv_bf16 = s->kv_in_bf16 ? 1 : 0;
    }
    if (wp.part) { t.qkv_part = wp.part; t.qkv_ssq = wp.ssq; t.qkv_S = wp.S; t.qkv_K = d.H; t.qkv_eps = d.eps; }
    // Split-K projections (LinArgs::ksplit, k_gemv_sk2): o-proj and down-proj with N <= 2048 and K >= 2048 at 3 .. 16 rows (wide
    // sessions: blocks of 16 rows, see below)
    // run as two K halves that meet in the output through order-independent atomic adds. The output buffer must hold zeros:
    // SUM is cleared by this layer's attention launch (its last reader was the previous down-proj), X by the gate/up
    // launch (its last reader is this layer's o-proj, as the residual).
    const bool first2 = !s->legacy_attn && rows_per_seq == 2 && !pos_dev && pos_static == 0 && paged_layer < 0;
    const bool attn3 = !first2 && (s->legacy_attn || rows_per_seq > 1);
    // Wide sessions (B > 16, round 3): the same kernel over blocks of 16 rows (grid plane z) — one launch instead of the
    // split-K GEMM + slice-sum pair for exactly the narrow outputs where the second launch hurt most (B = 64, code
    // predictor o / down: 9.4 + 4.9 us -> one launch). Q3_WIDE_NO_SK2=1: off (A/B aid)
    static const bool wide_sk2 = getenv("Q3_WIDE_NO_SK2") == nullptr;
    const bool sk_rows = s->ksplit && B >= 3 && (B <= 16 || (wide_sk2 && B <= Q3_MAX_BATCH && rows_per_seq == 1)) && d.H <= 2048 && d.H % 4 == 0;
    // (round 6: also behind the code predictor's 2-token first pass — 16 rows at B = 8 — whose o-projection ran on k_gemv_lds with
    // 64 workgroups: 5.8 us against 3.6 as two K halves; Q3_FIRST2_NO_SK=1: the old kernel, A/B aid)
    static const bool first2_sk = getenv("Q3_FIRST2_NO_SK") == nullptr;
    const bool o_sk = sk_rows && (!first2 || first2_sk) && !attn3 && w.o.t1 && QD >= 2048 && up32(QD) / 32 >= 16;
    // (beyond 32 rows the 25 MB talker down-proj is re-read by every 16-row block — 32.0 us at B = 64 against 16.5 + 4.8 for
    // the GEMM pair — while the smaller matrices win: code predictor o 14.6 -> 6.8, down 15.7 -> 9.2, talker o 14.4 -> 11.7 us)
    const bool dn_big = B > 32 && (size_t)d.I * d.H * 2 > ((size_t)12 << 20);
    const bool dn_sk = sk_rows && !dn_big && w.down.t1 && w.gate.t1 && d.I >= 2048 && up32(d.I) / 32 >= 16;
    if (t.kv_bf16 && (first2 || attn3)) return set_err(Q3_UNSUPPORTED, "multi-row talker steps are not available once a session's K/V is bf16");
    if (first2) {
        if (fold) {                                 // pass-1 gather folded: row 2b+1 = table row tok[b]
            t.g_tok = fold->tok; t.g_qkv_tab = fold->qkv_tab; t.g_proj_tab = fold->proj_tab; t.g_proj_dim = fold->proj_dim;
            t.g_x = fold->out; t.g_ldx = fold->ld_out;
        }
        if (o_sk) { t.zero = b.SUM; t.zero_n = B * d.H; }
        HIPC(launch_attn_first2(t, s->stream));    // the code predictor's 2-token first pass
    } else if (attn3) {     // rows of one sequence depend on each other's K/V: three launches
        HIPC(launch_qknorm_rope_kv(t, s->stream));
        HIPC(launch_attn_decode(t, s->stream));
        HIPC(launch_attn_merge(t, s->stream));
    } else {
        if (fold) {
            t.g_logits = fold->cp_logits; t.g_vocab = fold->cp_vocab; t.g_qkv_tab = fold->qkv_tab;
            t.g_proj_tab = fold->proj_tab; t.g_proj_dim = fold->proj_dim; t.g_x = fold->out; t.g_ldx = fold->ld_out;
            t.g_codes = fold->codes; t.g_frame_idx = fold->frame_idx; t.g_max_frames = fold->max_frames; t.g_code_slot = fold->pass - 1;
        }
        if (o_sk) { t.zero = b.SUM; t.zero_n = B * d.H; }
        if (s->cp_attn && attn_cp_ok(t)) {      // <= 16 positions, static position: the code predictor
#ifdef Q3_TRACE
            t.trace = s->trace_next(1, t.B, t.nh, 1, t.pos_static, t.g_logits ? 1 : 0);
#endif
            HIPC(launch_attn_cp(t, s->stream));
        } else {
#ifdef Q3_TRACE
            t.trace = s->trace_next(2, t.B, t.nh, n_splits, t.pos_static, t.g_logits ? 1 : 0);
#endif
            HIPC(launch_attn_fused(t, s->stream));
#ifdef Q3_TRACE
            if (n_splits > 1) t.trace = s->trace_next(3, t.B, t.nh, n_splits, t.pos_static);
#endif
            if (n_splits > 1) HIPC(launch_attn_merge(t, s->stream));
        }
    }
    auto force16 = [&](LinArgs& l, const TW& tw) { l.tiled = 1; l.W = tw.t1; l.Kpad = kpad_for(1, l.K); l.ksplit = 2; };
    LinArgs o;
    o.N = d.H; o.K = QD; set_w(o, w.o, B, o.N, o.K); o.x = b.ATT; o.ldx = QD; o.resid = b.X; o.ldr = d.H; o.y = b.SUM; o.ldy = d.H; o.M = B; o.epi = EPI_RESID;
    if (o_sk) force16(o, w.o);
    HIPC(run_linear(s, o));
    LinArgs g;
    g.N = d.I; g.K = d.H; set_w2(g, w.gate, w.up, B, g.N, g.K); g.x = b.SUM; g.ldx = d.H; g.norm_w = w.post_ln; g.eps = d.eps;
    g.y = b.ACT; g.ldy = d.I; g.M = B; g.epi = EPI_SWIGLU;
    if (dn_sk) { g.zero = b.X; g.zero_n = B * d.H; }
    HIPC(run_linear(s, g));
    LinArgs dn;
    dn.N = d.H; dn.K = d.I; set_w(dn, w.down, B, dn.N, dn.K); dn.x = b.ACT; dn.ldx = d.I; dn.resid = b.SUM; dn.ldr = d.H; dn.y = b.X; dn.ldy = d.H; dn.M = B; dn.epi = EPI_RESID;
    if (dn_sk) force16(dn, w.down);
    HIPC(run_linear(s, dn));
    return Q3_OK;
}

static bool d_nh_ok(const q3_config& c) {     // GEMM prefill needs N % 64 == 0 for every projection and a 1- or 2-way GQA ratio
    const int qkv = (c.n_heads + 2 * c.n_kv_heads) * HEAD_DIM, rep = c.n_heads / (c.n_kv_heads ? c.n_kv_heads : 1);
    return qkv % 64 == 0 && c.hidden % 64 == 0 && c.inter % 64 == 0 && (rep == 1 || rep == 2);
}
static LmDims talker_dims(const q3_config& c) { return LmDims{c.hidden, c.inter, c.n_heads, c.n_kv_heads, c.n_layers, c.rms_eps}; }
static LmDims cp_dims(const q3_config& c) { return LmDims{c.cp_hidden, c.cp_inter, c.cp_heads, c.cp_kv_heads, c.cp_layers, c.rms_eps}; }

// talker layers on the contents of tb.X at position pos (device array or static); with_head: final
// norm → LASTH and codec_head → LOGITS (talker.rs:716-736)
static q3_status talker_step(q3_session* s, const int* pos_dev, int pos_static, bool with_head, int rows_per_seq = 1) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const LmDims d = talker_dims(c);
    for (int i = 0; i < c.n_layers; ++i)
        Q3C(lm_layer(s, d, m->tl[i], s->tb, s->paged ? nullptr : s->kcache + (size_t)i * s->kv_layer_stride,
                     s->paged ? nullptr : s->vcache + (size_t)i * s->kv_layer_stride,
                     s->max_seq, pos_dev, pos_static, s->n_splits, rows_per_seq, false, nullptr, s->paged ? i : -1));
    if (with_head) {
        // final norm of each sequence's LAST row of the step
        HIPC(launch_rmsnorm(s->tb.X + (size_t)(rows_per_seq - 1) * c.hidden, rows_per_seq * c.hidden, m->norm, s->LASTH, c.hidden, s->B,
                            c.hidden, c.rms_eps, s->stream));
        LinArgs h;
        h.N = c.codec_vocab; h.K = c.hidden; set_w(h, m->codec_head, s->B, h.N, h.K); h.x = s->LASTH; h.ldx = c.hidden; h.y = s->LOGITS; h.ldy = c.codec_vocab;
        h.M = s->B; h.epi = EPI_NONE;
        HIPC(run_linear(s, h));
    }
    return Q3_OK;
}

// generate_acoustic_codes (code_predictor.rs:320-416) as 16 single-token passes: pass 0 = talker
// hidden (pos 0), pass 1 = semantic embedding (pos 1) → lm_head[0]; pass p = embedding of code p-2
// (pos p) → lm_head[p-1]. (The reference runs passes 0 and 1 as one 2-token causal prefill; for
// causal attention that is the same computation.) Codes 0..13 are recorded by the next pass's
// gather, code 14 by frame_embed / the caller.
static q3_status cp_run(q3_session* s) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const LmDims d = cp_dims(c);
    const int H = c.hidden, CH = c.cp_hidden, V = c.cp_vocab, B = s->B;
    const int n_pass = c.n_groups;   // 16
    // First pass as the reference does it (code_predictor.rs:337-367): the talker hidden state and the semantic
    // embedding go through the layers TOGETHER as a 2-token causal prefill (rows 2b, 2b+1), when 2B rows fit.
    const bool two = !s->no_chunk && 2 * B <= 16;
    for (int p = two ? 1 : 0; p < n_pass; ++p) {
        const int rows = (two && p == 1) ? 2 : 1;
        CpGatherArgs g{};
        g.pass = p; g.last_hidden = s->LASTH; g.H = H; g.codec_emb = m->codec_emb; g.tok = s->tok;
        g.cp_emb = p >= 2 ? m->cp_emb[p - 2] : nullptr;
        g.cp_logits = p >= 2 ? s->CP_LOGITS + (size_t)(p - 2) * B * V : nullptr;
        g.cp_vocab = V; g.codes = s->codes; g.frame_idx = s->frame_idx; g.max_frames = s->max_frames; g.B = B;
        float* dst = m->mtp_w.t1 ? s->CP_IN : s->cb.X; const int ld = m->mtp_w.t1 ? H : CH;
        // 1.7B with pre-projected tables: only the talker hidden state (pass 0) still goes through the 2048 -> 1024
        // projection at run time; every embedding row arrives already projected (14 GEMV launches less per frame)
        const bool tabs = m->mtp_w.t1 && m->proj_tabs && s->proj_tables;
        auto project = [&](int M, int ldy, const float* x_in = nullptr) -> q3_status {
            LinArgs a;
            a.N = CH; a.K = H; set_w(a, m->mtp_w, M, CH, H); a.x = x_in ? x_in : s->CP_IN; a.ldx = H; a.bias = m->mtp_b; a.y = s->cb.X; a.ldy = ldy;
            a.M = M; a.epi = EPI_NONE;
            HIPC(run_linear(s, a));
            return Q3_OK;
        };
        // layer-0 q|k|v of a table row comes from the table too (both model sizes): the layer-0 qkv GEMV then only runs
        // for the rows that carry the talker hidden state
        const bool qt = m->qkv0_tabs && s->qkv_tables && (tabs || !m->mtp_w.t1);
        const int QKVD = (d.nh + 2 * d.nkv) * HEAD_DIM;
        bool skip0 = false, fold0 = false;
        auto qkv_rows0 = [&](int ldx, int ldy) -> q3_status {      // run-time layer-0 qkv of the B pass-0 rows
            LinArgs a;
            a.N = QKVD; a.K = CH; set_w(a, m->cl[0].qkv, B, QKVD, CH); a.x = s->cb.X; a.ldx = ldx; a.norm_w = m->cl[0].in_ln; a.eps = d.eps;
            a.y = s->cb.QKV; a.ldy = ldy; a.M = B; a.epi = EPI_NONE;
            HIPC(run_linear(s, a));
            return Q3_OK;
        };
        if (tabs) {
            if (rows == 2) {
                // B rows of talker hidden: projected straight from LASTH (the copy to CP_IN was a launch of its own)
                static const bool no_fold2 = getenv("Q3_CP_NO_FOLD") != nullptr;
                if (no_fold2) { g.pass = 0; g.out = s->CP_IN; g.ld_out = H; HIPC(launch_cp_gather(g, s->stream)); }
                Q3C(project(B, 2 * CH, no_fold2 ? s->CP_IN : s->LASTH));           // -> cb.X rows 2b
                g.pass = 1; g.out = s->cb.X + CH; g.ld_out = 2 * CH; g.proj_tab = m->sem_proj; g.proj_dim = CH;   // rows 2b+1
                if (qt) { g.qkv_tab = m->sem_qkv0; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV + QKVD; g.ld_qkv_out = 2 * QKVD; }
                // the semantic row (table row tok[b]) is read by the 2-token attention itself
                fold0 = qt && !s->legacy_attn && !no_fold2 && CH % 4 == 0;
                if (!fold0) HIPC(launch_cp_gather(g, s->stream));
                if (qt) { Q3C(qkv_rows0(2 * CH, 2 * QKVD)); skip0 = true; }
            } else if (p == 0) {
                g.out = s->CP_IN; g.ld_out = H;
                HIPC(launch_cp_gather(g, s->stream));
                Q3C(project(B, CH));
            } else {
                g.out = s->cb.X; g.ld_out = CH; g.proj_tab = p == 1 ? m->sem_proj : m->cp_proj[p - 2]; g.proj_dim = CH;
                if (qt) { g.qkv_tab = p == 1 ? m->sem_qkv0 : m->cp_qkv0[p - 2]; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV; g.ld_qkv_out = QKVD; skip0 = true; }
                // passes >= 2 with both tables: no launch of its own — the layer-0 attention re-derives the argmax and
                // reads the table rows itself (14 launches less per frame; Q3_CP_NO_FOLD=1: the separate launch, A/B aid)
                static const bool no_fold = getenv("Q3_CP_NO_FOLD") != nullptr;
                fold0 = qt && p >= 2 && !s->legacy_attn && !no_fold && CH % 4 == 0;
                if (!fold0) HIPC(launch_cp_gather(g, s->stream));
            }
        } else {
        if (rows == 2) {
            // row 2b = talker hidden (pass-0 source), row 2b+1 = semantic embedding (pass-1 source)
            g.pass = 0; g.out = dst; g.ld_out = 2 * ld;
            HIPC(launch_cp_gather(g, s->stream));
            g.pass = 1; g.out = dst + ld;
            if (qt) { g.qkv_tab = m->sem_qkv0; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV + QKVD; g.ld_qkv_out = 2 * QKVD; }
            HIPC(launch_cp_gather(g, s->stream));
            if (qt) { Q3C(qkv_rows0(2 * CH, 2 * QKVD)); skip0 = true; }
        } else {
            g.out = dst; g.ld_out = ld;
            if (qt && p >= 1) { g.qkv_tab = p == 1 ? m->sem_qkv0 : m->cp_qkv0[p - 2]; g.qkv_dim = QKVD; g.qkv_out = s->cb.QKV; g.ld_qkv_out = QKVD; skip0 = true; }
            HIPC(launch_cp_gather(g, s->stream));
        }
        if (m->mtp_w.t1) Q3C(project(B * rows, CH));
        }
        for (int i = 0; i < c.cp_layers; ++i)
            Q3C(lm_layer(s, d, m->cl[i], s->cb, s->ckcache + (size_t)i * s->ckv_layer_stride, s->cvcache + (size_t)i * s->ckv_layer_stride,
                         n_pass + 1, nullptr, rows == 2 ? 0 : p, 1, rows, i == 0 && skip0, (i == 0 && fold0) ? &g : nullptr));
        if (p >= 1) {
            LinArgs h;
            h.N = V; h.K = CH; set_w(h, m->cp_head[p - 1], B, V, CH); h.x = s->cb.X + (size_t)(rows - 1) * CH; h.ldx = rows * CH;
            h.norm_w = m->cp_norm; h.eps = c.rms_eps;
            h.y = s->CP_LOGITS + (size_t)(p - 1) * B * V; h.ldy = V; h.M = B; h.epi = EPI_NONE;
            HIPC(run_linear(s, h));
        }
    }
    return Q3_OK;
}

// sampling options of one request -> the sampler's per-row record (sampling.rs:140-319 branch conditions)
static SampleRow sample_row(const q3_options& o) {
    SampleRow r{};
    r.apply_temp = (o.temperature != 1.0 && o.temperature > 0.0) ? 1 : 0;
    r.inv_temp = (float)(1.0 / o.temperature);
    r.greedy = o.temperature < 0.01 ? 1 : 0;
    r.top_k = o.top_k; r.use_top_p = (o.top_p < 1.0 && o.top_p > 0.0) ? 1 : 0; r.top_p = (float)o.top_p;
    r.use_rep = (o.repetition_penalty != 1.0 && !(fabs(o.repetition_penalty - 1.0) < 1e-9)) ? 1 : 0;
    r.rep_pen = (float)o.repetition_penalty; r.rep_inv = 1.0f / (float)o.repetition_penalty;
    r.eos_id = o.eos_token_id; r.min_new_tokens = o.min_new_tokens;
    return r;
}

static void fill_sample_args(q3_session* s, SampleArgs& a) {
    const q3_config& c = s->m->cfg;
    memset(&a, 0, sizeof a);
    a.logits = s->LOGITS; a.ld = c.codec_vocab; a.seen = s->seen; a.u = s->U; a.u_stride = s->max_frames + 2; a.limit = s->limit;
    a.draw_idx = s->token_count; a.tok = s->tok; a.token_count = s->token_count; a.frame_idx = s->frame_idx; a.pos = s->pos;
    a.vocab = c.codec_vocab; a.B = s->B;
    const SampleRow r = sample_row(s->opts);          // scalar fields = the first request's (every row reads its own through a.rows)
    a.apply_temp = r.apply_temp; a.inv_temp = r.inv_temp; a.greedy = r.greedy; a.top_k = r.top_k; a.use_top_p = r.use_top_p; a.top_p = r.top_p;
    a.use_rep = r.use_rep; a.rep_pen = r.rep_pen; a.rep_inv = r.rep_inv; a.eos_id = r.eos_id; a.min_new_tokens = r.min_new_tokens;
    a.rows = s->sample_rows;
    a.codec_eos = CODEC_EOS; a.use_suppress = 1;
    if (s->debug && s->logits_hist) { a.logits_hist = s->logits_hist; a.hist_stride_b = (s->max_frames + 1) * c.codec_vocab; a.hist_cap = s->max_frames + 1; }
}

// one frame of generate_codes (lib.rs:580-652)
static q3_status frame_launch(q3_session* s) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
#ifdef Q3_TRACE
    s->trace_node = 0; s->trace_meta.clear();        // every frame (and the capture) re-uses the same slices
#endif
    Q3C(cp_run(s));
    FrameEmbedArgs f{};
    f.codec_emb = m->codec_emb; f.tok = s->tok;
    for (int g = 0; g < 15; ++g) f.cp_embs[g] = g < (int)m->cp_emb.size() ? m->cp_emb[(size_t)g] : nullptr;
    f.cp_logits_last = s->CP_LOGITS + (size_t)14 * s->B * c.cp_vocab; f.cp_vocab = c.cp_vocab;
    f.codes = s->codes; f.frame_idx = s->frame_idx; f.max_frames = s->max_frames;
    f.text_rows = s->rows; f.trail_base = s->trail_base; f.trail_len = s->trail_len; f.pad_row = s->pad_row;
    f.out = s->tb.X; f.H = c.hidden; f.B = s->B; f.n_acoustic = c.n_groups - 1;
    HIPC(launch_frame_embed(f, s->stream));
    if (s->debug && s->cp_logits_hist) {
        // capture is host-indexed: only valid outside graph replay (debug sessions never use graphs)
        HIPC(hipMemcpyAsync(s->cp_logits_hist + (size_t)s->frames_run * 15 * s->B * c.cp_vocab, s->CP_LOGITS,
                            (size_t)15 * s->B * c.cp_vocab * 4, hipMemcpyDeviceToDevice, s->stream));
    }
    Q3C(talker_step(s, s->pos, 0, true));
    SampleArgs a; fill_sample_args(s, a); a.advance = 1;
    HIPC(launch_sample(a, s->stream));
    return Q3_OK;
}


static q3_status session_create(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out,
                                hipStream_t borrow = nullptr);
// prefill positions and frame limit of a request as session_create will resolve them (talker.rs:451-491 / 511-564 / 585-627; the
// ICL length cap of lib.rs:913-929) — for callers that must know a row's worst-case KV extent before a session exists
static void request_shape(const q3_request& r, int* prefill_len, int* limit) {
    const bool icl = r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes && r.ref_text_ids;
    const int n_ins = r.mode == Q3_MODE_VOICE_DESIGN ? r.n_instruct : 0;
    const int overlay = r.mode == Q3_MODE_VOICE_DESIGN ? 5 : 6;
    *prefill_len = n_ins + 3 + overlay + ((r.n_text > 0 && !icl) ? 1 : 0) + (icl ? r.n_ref + 1 : 0);
    int lim = r.opts.max_length;
    if (icl) { int cap = 6 * r.n_text; if (cap < 75) cap = 75; if (lim > cap) lim = cap; }
    *limit = lim;
}
// worst-case cost of a row against the model's KV budget (KvBudget units): every page prompt + limit + 1 positions can reach; a
// bf16 row holds f32 pages for its prompt and, while they are converted, bf16 pages beside them
static long row_worst_units(int prefill_len, int limit, bool bf16) {
    const long full = (prefill_len + limit + 1 + KV_PAGE_POS - 1) / KV_PAGE_POS, pre = (prefill_len + 1 + KV_PAGE_POS - 1) / KV_PAGE_POS;
    return bf16 ? std::max(3 * pre, full) : 2 * full;
}
// a request that only occupies a row: a one-token CustomVoice prompt with a one-frame limit, built from fixed, known-valid values
// (ten prefill positions). Rows of the batcher's session before a ticket enters them; rows of a ragged first batch before prefill.
static q3_request idle_request(int chunk_frames) {
    static const uint32_t one_tok[1] = {0};
    q3_request d{};
    d.mode = Q3_MODE_CUSTOM_VOICE; d.text_ids = one_tok; d.n_text = 1;
    d.speaker_id = 0; d.language_id = 0;                                       // codec token 0: valid in every vocabulary
    d.opts.temperature = 0.9; d.opts.top_p = 0.9; d.opts.repetition_penalty = 1.05; d.opts.top_k = 50;      // SynthesisOptions::default (lib.rs:1786-1836)
    d.opts.eos_token_id = -1; d.opts.min_new_tokens = 2; d.opts.max_length = 1; d.opts.has_seed = 1; d.opts.seed = 0;
    d.opts.chunk_frames = chunk_frames;
    return d;
}
// Sessions take rows of ANY mix of prompt kinds and lengths (round 5; BASELINE config[3] on a Base checkpoint mixes x-vector and
// ICL prompts, lib.rs:718-784, 802-870, 897-1046 are per-call in the reference). Rows of one prefill length are prefilled
// together by the session itself (the fast path: one batched prefill). A RAGGED batch is opened on idle rows sized for the
// longest prompt / largest limit; q3_session_prefill then prefills the rows in groups of equal prefill length — each group a
// batched side prefill — and moves every row in the way a continuous-batching swap does (transplant_row), so every row carries
// the bits of its own run and decode proceeds in one captured frame graph over all rows.
static q3_status session_create_any(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out) {
    if (!m || !reqs || !out) return set_err(Q3_INVALID_ARG, "q3_session_create: null argument");
    if (batch < 1 || batch > Q3_MAX_BATCH) return set_err(Q3_UNSUPPORTED, "batch %d unsupported (1..%d sequences per session)", batch, Q3_MAX_BATCH);
    bool ragged = false; int S0 = 0, L0 = 0, Smax = 0, Lmax = 0, rows_max = 0;
    for (int b = 0; b < batch; ++b) {
        const q3_request& r = reqs[b];
        if (r.n_text < 0 || r.n_instruct < 0 || r.n_ref < 0 || r.n_ref_text < 0) return set_err(Q3_INVALID_ARG, "bad token id arrays");
        int S = 0, L = 0; request_shape(r, &S, &L);
        if (b == 0) { S0 = S; L0 = L; }
        ragged = ragged || S != S0;
        Smax = std::max(Smax, S); Lmax = std::max(Lmax, L);
        rows_max = std::max(rows_max, (r.mode == Q3_MODE_VOICE_DESIGN ? r.n_instruct : 0) + 5 + r.n_ref_text + r.n_text + 1);
        if (r.opts.chunk_frames != reqs[0].opts.chunk_frames) return set_err(Q3_UNSUPPORTED, "all requests of a batch must share chunk_frames (the streaming chunk is a property of the session)");
    }
    (void)L0;
    if (!ragged) return session_create(m, reqs, batch, frame_budget, prompt_budget, out);
    if (Lmax < 1) return set_err(Q3_INVALID_ARG, "max_length must be >= 1");
    const int fb = std::max(frame_budget, Lmax);
    int pb = std::max(std::max(prompt_budget, Smax), 16);
    if (rows_max > pb + 1024) pb = rows_max - 1024;          // a row's text-row slot is prompt_budget + 1024 rows
    std::vector<q3_request> idle((size_t)batch, idle_request(reqs[0].opts.chunk_frames));
    q3_session* s = nullptr;
    Q3C(session_create(m, idle.data(), batch, fb, pb, &s));
    s->ragged.resize((size_t)batch);
    for (int b = 0; b < batch; ++b) s->ragged[(size_t)b].own(reqs[b], m->cfg.hidden);
    *out = s;
    return Q3_OK;
}
extern "C" q3_status q3_session_create(q3_model* m, const q3_request* reqs, int batch, q3_session** out) {
    return session_create_any(m, reqs, batch, 0, 0, out);
}
extern "C" q3_status q3_session_create_reserved(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out) {
    if (frame_budget < 0 || prompt_budget < 0) return set_err(Q3_INVALID_ARG, "q3_session_create_reserved: negative budget");
    return session_create_any(m, reqs, batch, frame_budget, prompt_budget, out);
}
static q3_status session_create(q3_model* m, const q3_request* reqs, int batch, int frame_budget, int prompt_budget, q3_session** out,
                                hipStream_t borrow) {
    if (!m || !reqs || !out) return set_err(Q3_INVALID_ARG, "q3_session_create: null argument");
    if (!m->finalized) return set_err(Q3_INVALID_ARG, "model not finalized");
    if (batch < 1 || batch > Q3_MAX_BATCH) return set_err(Q3_UNSUPPORTED, "batch %d unsupported (1..%d sequences per session)", batch, Q3_MAX_BATCH);
    HIPC(hipSetDevice(m->device));
    const q3_config& c = m->cfg;
    std::unique_ptr<q3_session> s(new q3_session());
    s->m = m; s->B = batch; s->opts = reqs[0].opts;
    m->live_sessions.fetch_add(1);
    if (s->opts.max_length < 1) return set_err(Q3_INVALID_ARG, "max_length must be >= 1");
    s->seq.resize(batch);
    int rows = 0;
    for (int b = 0; b < batch; ++b) {
        const q3_request& r = reqs[b];
        const bool icl_req = r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes && r.ref_text_ids;
        if (r.opts.chunk_frames != s->opts.chunk_frames) return set_err(Q3_UNSUPPORTED, "all requests of a batch must share chunk_frames (the streaming chunk is a property of the session)");
        if (!(r.opts.temperature >= 0.0) || r.opts.repetition_penalty <= 0.0) return set_err(Q3_INVALID_ARG, "bad sampling options");
        if (r.mode < 0 || r.mode > 2) return set_err(Q3_INVALID_ARG, "bad mode %d", r.mode);
        if (r.n_text < 0 || r.n_instruct < 0 || (r.n_text > 0 && !r.text_ids) || (r.n_instruct > 0 && !r.instruct_ids))
            return set_err(Q3_INVALID_ARG, "bad token id arrays");
        if (r.mode == Q3_MODE_VOICE_CLONE && !r.xvector) return set_err(Q3_INVALID_ARG, "voice clone needs an x-vector");
        SeqInfo& q = s->seq[b];
        q.req = r;
        q.text.assign(r.text_ids, r.text_ids + r.n_text);
        for (uint32_t id : q.text) if (id >= (uint32_t)c.text_vocab) return set_err(Q3_INVALID_ARG, "text id %u out of range", id);
        if (r.mode == Q3_MODE_VOICE_DESIGN) q.instruct.assign(r.instruct_ids, r.instruct_ids + r.n_instruct);
        for (uint32_t id : q.instruct) if (id >= (uint32_t)c.text_vocab) return set_err(Q3_INVALID_ARG, "instruct id %u out of range", id);
        if (r.language_id >= (uint32_t)c.codec_vocab || (r.mode == Q3_MODE_CUSTOM_VOICE && r.speaker_id >= (uint32_t)c.codec_vocab))
            return set_err(Q3_INVALID_ARG, "speaker/language id out of range");
        if (r.xvector) q.xvec.assign(r.xvector, r.xvector + c.hidden);
        q.icl = r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes && r.ref_text_ids;
        if (r.mode == Q3_MODE_VOICE_CLONE && r.n_ref > 0 && r.ref_codes) {
            // reference frames are prepended at decode whenever the prompt carries them — also without a reference
            // transcript, when the prefill stays x-vector-only (lib.rs:1022: `if let Some(ref_codes) = &prompt.ref_codes`)
            q.ref_codes.assign(r.ref_codes, r.ref_codes + (size_t)r.n_ref * 16);
            for (int f = 0; f < r.n_ref; ++f) {
                if (q.ref_codes[(size_t)f * 16] >= (uint32_t)c.codec_vocab) return set_err(Q3_INVALID_ARG, "reference semantic code out of range");
                for (int g = 1; g < 16; ++g) if (q.ref_codes[(size_t)f * 16 + g] >= (uint32_t)c.cp_vocab) return set_err(Q3_INVALID_ARG, "reference acoustic code out of range");
            }
        }
        if (q.icl) {
            if (r.n_ref_text < 0) return set_err(Q3_INVALID_ARG, "bad reference text");
            q.ref_text.assign(r.ref_text_ids, r.ref_text_ids + r.n_ref_text);
            for (uint32_t id : q.ref_text) if (id >= (uint32_t)c.text_vocab) return set_err(Q3_INVALID_ARG, "reference text id %u out of range", id);
            // lib.rs:913-929 (must be identical for every sequence of the batch, checked below through s->opts)
            q.req.opts.repetition_penalty = r.opts.repetition_penalty < 1.5 ? 1.5 : r.opts.repetition_penalty;
            int cap = 6 * r.n_text; if (cap < 75) cap = 75;
            if (q.req.opts.max_length > cap) q.req.opts.max_length = cap;
            if (b == 0) s->opts = q.req.opts;
        }
        const int n_ins = (int)q.instruct.size();
        const int overlay = r.mode == Q3_MODE_VOICE_DESIGN ? 5 : 6;
        const int n_icl = q.icl ? r.n_ref + 1 : 0;                              // streaming overlay: icl_len = n_codec
        const int n_text_all = q.icl ? r.n_ref_text + r.n_text + 1 : 0;         // [ref_text, text, tts_eos]
        q.prefill_len = n_ins + 3 + overlay + ((r.n_text > 0 && !q.icl) ? 1 : 0) + n_icl;
        q.trailing_len = q.icl ? (n_text_all > n_icl ? n_text_all - n_icl : 1) : (r.n_text > 1 ? r.n_text - 1 : 0) + 1;
        q.row_base = rows;
        q.n_rows = n_ins + 5 + (q.icl ? r.n_ref_text : 0) + r.n_text + 1;     // instruct, role×3, pad, bos, [ref_text…], text…, eos
        rows += q.n_rows;
        if (q.prefill_len != s->seq[0].prefill_len)
            return set_err(Q3_UNSUPPORTED, "all sequences of a batch must have the same prefill length (%d vs %d)", q.prefill_len, s->seq[0].prefill_len);
    }
    s->prefill_len = s->seq[0].prefill_len;
    s->n_rows_total = rows;
    s->max_frames = frame_budget > 1 ? frame_budget : 1;          // room for rows that arrive later with a larger limit (continuous batching)
    for (auto& q : s->seq) {
        if (q.req.opts.max_length < 1) return set_err(Q3_INVALID_ARG, "max_length must be at least 1");
        q.limit = q.req.opts.max_length; q.start_run = 0;
        if (q.limit > s->max_frames) s->max_frames = q.limit;
        if (q.n_rows > s->row_cap) s->row_cap = q.n_rows;
    }
    s->opts.max_length = s->max_frames;
    if (s->row_cap < 1024) s->row_cap = 1024;      // replacement slots hold any text up to ~1000 tokens (8 MB per row at H = 2048), longer if the batch had one
    if (prompt_budget > 0 && s->row_cap < prompt_budget + 1024) s->row_cap = prompt_budget + 1024;      // ... or the caller announced longer prompts (instruct / reference text rows are projected rows too)
    s->repl_base = rows;
    // KV sized for what the path needs (prefill + frames), not the reference's max_new_tokens+256 (lib.rs:450)
    s->max_seq = (prompt_budget > s->prefill_len ? prompt_budget : s->prefill_len) + s->max_frames + 1;   // prompt_budget: later rows with longer prompts
    if (s->max_seq > m->rope_len) return set_err(Q3_KV_OVERFLOW, "sequence length %d exceeds the RoPE table (%d)", s->max_seq, m->rope_len);
    {
        static const int ns_env = [] { const char* e = getenv("Q3_ATTN_SPLITS"); return e ? atoi(e) : 0; }();   // tuning aid
        // ~2 attention workgroups per CU over a 640-frame utterance. With the three-deep unconditional K/V requests of
        // k_attn_fused (round 2) a workgroup walks its keys without a round trip per key and fewer, longer key ranges
        // win: B = 8: 3.908 / 3.724 / 3.651 / 3.609 / 3.635 / 3.812 ms/frame at 1 / 2 / 4 / 8 / 16 / 32 splits (before:
        // 4.40 / 3.99 / 3.77 / 3.67 / 3.68); B = 1 stays at 16 (2.794 vs 2.800 at 8)
        int ns = ns_env > 0 ? ns_env : (batch <= 2 ? 1024 : 512) / (batch * c.n_kv_heads);     // (B <= 2: up to the long-context cap below)
        // long contexts (a 4k-position prompt: 38 MB of f32 K/V per layer) want more than 16 workgroups per KV head to
        // stream them (B = 1 at 4.1k positions: 3.90 -> 3.55 ms/frame); short sessions keep the cheaper 16-way merge
        const int cap = ns_env > 0 ? MAX_SPLITS : (s->max_seq > 2048 ? MAX_SPLITS : 16);
        if (ns < 1) ns = 1; if (ns > cap) ns = cap; s->n_splits = ns;
    }
    {   // the frame loop is a chain of ~600 short dependent kernels per frame: give its queue the highest priority so
        // that its workgroups are dispatched ahead of the vocoder segments running beside it (q3_session_run)
        if (borrow) { s->stream = borrow; s->owns_stream = false; }      // q3_session_replace: no second queue for a one-row prefill
        else {
            {
                std::lock_guard<std::mutex> g(m->stream_mu);
                if (!m->idle_streams.empty()) { s->stream = m->idle_streams.back(); m->idle_streams.pop_back(); }
            }
            if (!s->stream) {
                int least = 0, greatest = 0;
                HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIPC(hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, greatest));
            }
        }
    }
    const int B = batch, H = c.hidden, CH = c.cp_hidden;
    s->pool.lazy = true;
    auto alloc_lm = [&](LmBuf& b, const LmDims& d, int nsplit) -> hipError_t {
        const int QD = d.nh * HEAD_DIM, KD = d.nkv * HEAD_DIM;
        const size_t R = batch > 16 ? (size_t)up16(batch) : 16;     // rows: up to 16 for multi-row steps (B <= 16), else one row per sequence
        hipError_t e;
        if ((e = s->pool.alloc(&b.X, R * d.H)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.SUM, R * d.H)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.QKV, R * (QD + 2 * KD))) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.Q, R * QD)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.ATT, R * QD)) != hipSuccess) return e;
        if ((e = s->pool.alloc(&b.ACT, R * d.I)) != hipSuccess) return e;
        return s->pool.alloc(&b.PART, R * d.nh * nsplit * PART_STRIDE);
    };
    HIPC(alloc_lm(s->tb, talker_dims(c), s->n_splits));
    HIPC(alloc_lm(s->cb, cp_dims(c), 1));
    if (B > 16) {     // wide sessions: workspace of the split-K GEMM, sized for the largest projection of either network
        size_t need = 0;
        auto upd = [&](int N, int K, int epi) { if (N % 128 == 0 && K % 128 == 0) { const size_t b = gemm_wide_ws_bytes(B, N, K, epi); if (b > need) need = b; } };
        const int QDt = c.n_heads * HEAD_DIM, KDt = c.n_kv_heads * HEAD_DIM, QDc = c.cp_heads * HEAD_DIM, KDc = c.cp_kv_heads * HEAD_DIM;
        upd(QDt + 2 * KDt, H, EPI_NONE); upd(H, QDt, EPI_RESID); upd(c.inter, H, EPI_SWIGLU); upd(H, c.inter, EPI_RESID); upd(c.codec_vocab, H, EPI_NONE);
        upd(QDc + 2 * KDc, c.cp_hidden, EPI_NONE); upd(c.cp_hidden, QDc, EPI_RESID); upd(c.cp_inter, c.cp_hidden, EPI_SWIGLU); upd(c.cp_hidden, c.cp_inter, EPI_RESID);
        upd(c.cp_vocab, c.cp_hidden, EPI_NONE); upd(c.cp_hidden, H, EPI_NONE);
        if (need) { HIPC(s->pool.alloc(&s->wide_ws, need / 4)); s->wide_ws_bytes = need; }
    }
    HIPC(s->pool.alloc(&s->LASTH, (size_t)B * H));
    HIPC(s->pool.alloc(&s->LOGITS, (size_t)B * c.codec_vocab));
    HIPC(s->pool.alloc(&s->CP_IN, (size_t)(B > 16 ? up16(B) : 16) * H));
    HIPC(s->pool.alloc(&s->CP_LOGITS, (size_t)15 * B * c.cp_vocab));
    s->kv_layer_stride = (size_t)B * c.n_kv_heads * s->max_seq * HEAD_DIM;
    {   // talker KV: pages from the model's pool as the rows grow (default), or one extent per row sized for the worst case
        const char* e = getenv("Q3_KV_CONTIGUOUS");          // read per session: the paged-vs-contiguous test flips it
        s->paged = !(e && atoi(e) != 0);
        if (s->paged) {
            if (s->max_seq > KV_MAX_PAGES * KV_PAGE_POS) return set_err(Q3_KV_OVERFLOW, "sequence length %d exceeds a row's page table (%d)", s->max_seq, KV_MAX_PAGES * KV_PAGE_POS);
            HIPC(s->pool.alloc(&s->kv_table, (size_t)B * KV_MAX_PAGES));
            HIPC(s->pool.alloc(&s->kv_conv, (size_t)2 * B * KV_MAX_PAGES));
            s->kv_rows.resize((size_t)B);
            for (auto& r : s->kv_rows) r.reserve(KV_MAX_PAGES);
        } else {
            HIPC(s->pool.alloc(&s->kcache, s->kv_layer_stride * c.n_layers));
            HIPC(s->pool.alloc(&s->vcache, s->kv_layer_stride * c.n_layers));
        }
    }
    s->ckv_layer_stride = (size_t)B * c.cp_kv_heads * (c.n_groups + 1) * HEAD_DIM;
    // K and V of the code predictor in ONE block: k_attn_cp takes their distance as a 32-bit float count, and two blocks of the
    // size-class cache can lie further apart than that (the launch then silently fell back to k_attn_fused: 70 nodes per
    // frame 1.1 us slower each — seen in a profiling run of round 4)
    HIPC(s->pool.alloc(&s->ckcache, 2 * s->ckv_layer_stride * c.cp_layers + 64));
    s->cvcache = s->ckcache + s->ckv_layer_stride * c.cp_layers + 64;
    HIPC(s->pool.alloc(&s->rows, ((size_t)rows + (size_t)B * s->row_cap) * H));      // + one replacement slot per row (q3_session_replace)
    HIPC(s->pool.alloc(&s->limit, B));
    HIPC(s->pool.alloc(&s->sample_rows, B));
    {
        std::vector<int> lim(B); std::vector<SampleRow> sr(B);
        for (int b = 0; b < B; ++b) { lim[b] = s->seq[b].limit; sr[b] = sample_row(s->seq[b].req.opts); }
        HIPC(hipMemcpy(s->limit, lim.data(), B * 4, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(s->sample_rows, sr.data(), B * sizeof(SampleRow), hipMemcpyHostToDevice));
    }
    HIPC(s->pool.alloc(&s->embeds, (size_t)B * s->prefill_len * H));
    HIPC(s->pool.alloc(&s->xvec, (size_t)B * H));
    HIPC(s->pool.alloc(&s->ids_dev, (size_t)rows)); HIPC(s->pool.alloc(&s->tr_dev, (size_t)B * s->prefill_len)); HIPC(s->pool.alloc(&s->ci_dev, (size_t)B * s->prefill_len));
    HIPC(s->pool.alloc(&s->proj_e, (size_t)rows * c.text_dim)); HIPC(s->pool.alloc(&s->proj_h, (size_t)rows * c.text_dim));
    HIPC(s->pool.alloc(&s->trail_base, B)); HIPC(s->pool.alloc(&s->trail_len, B)); HIPC(s->pool.alloc(&s->pad_row, B));
    HIPC(s->pool.alloc(&s->tok, B)); HIPC(s->pool.alloc(&s->seen, (size_t)B * c.codec_vocab));
    HIPC(s->pool.alloc(&s->frame_idx, B)); HIPC(s->pool.alloc(&s->pos, B)); HIPC(s->pool.alloc(&s->token_count, B));
    HIPC(s->pool.alloc(&s->U, (size_t)B * (s->max_frames + 2)));      // one draw per sampled token (max_frames + 1) + a spare for a frozen row
    HIPC(s->pool.alloc(&s->codes, (size_t)B * s->max_frames * 16));
    // RNG: one PCG stream per sequence, one draw per sampled token
    std::vector<float> U((size_t)B * (s->max_frames + 2), 0.0f);
    for (int b = 0; b < B; ++b) {
        uint64_t st;
        const q3_options& o = reqs[b].opts;
        uint64_t seed = o.seed;
        if (!o.has_seed) seed = (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count() + 0x9E37ULL * b;
        q3_rng_seed(seed, &st);
        for (int i = 0; i <= s->max_frames; ++i) U[(size_t)b * (s->max_frames + 2) + i] = q3_rng_next(&st);
    }
    HIPC(hipMemcpy(s->U, U.data(), U.size() * 4, hipMemcpyHostToDevice));
    HIPC(s->pool.settle());                 // every zero-fill has landed before a kernel on the session's own stream can run
    s->pool.lazy = false;
    *out = s.release();
    return Q3_OK;
}

// Every exit path (q3_session_free, a failed q3_session_create) goes through here: streams are drained BEFORE the pool's
// blocks return to the device cache (member destructors run after this body), so no later session is handed memory a
// queued kernel still writes.
q3_session::~q3_session() {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (dec_stream) (void)hipStreamSynchronize(dec_stream);
    for (auto st : par_streams) (void)hipStreamSynchronize(st);
    if (aql) q3::aql_program_destroy(aql);               // waits for its outstanding replays
    for (int b = 0; b < (int)kv_rows.size(); ++b) kv_release_row(this, b);      // every stream that touched them is idle
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (auto& ev : prof_pool) (void)hipEventDestroy(ev);
    for (auto st : par_streams) (void)hipStreamDestroy(st);
    cws.release(); seg_ws.release();
    for (auto& w : par_ws) w.release();
    if (pcm_all) dev_free(pcm_all);
    if (dec_ev) (void)hipEventDestroy(dec_ev);
    if (dec_stream) (void)hipStreamDestroy(dec_stream);
    if (stream && owns_stream) {                   // synchronised above: idle, handed to the next session of this model
        std::lock_guard<std::mutex> g(m->stream_mu);
        if (m->idle_streams.size() < 16) { m->idle_streams.push_back(stream); stream = nullptr; }
    }
    if (stream && owns_stream) (void)hipStreamDestroy(stream);
    pool.release_all();                                          // the model's device must still be current for these
    if (m->live_sessions.fetch_sub(1) == 1 && m->zombie.load() && !m->claimed.exchange(true)) model_destroy(m);
}

extern "C" void q3_session_free(q3_session* s) { delete s; }

// K/V dtype of the talker cache (before q3_session_prefill): Q3_DTYPE_F32 (default: the parity contract is the reference's CPU
// F32 path) or Q3_DTYPE_BF16 — the reference GPU path's cache dtype (kv_cache.rs:234-310): half the K/V bytes per frame, results
// no longer bit-comparable with the F32 oracle. Needs the paged cache.
extern "C" q3_status q3_session_set_kv_dtype(q3_session* s, int dtype) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->prefilled) return set_err(Q3_INVALID_ARG, "q3_session_set_kv_dtype must be called before prefill");
    if (dtype != Q3_DTYPE_F32 && dtype != Q3_DTYPE_BF16) return set_err(Q3_INVALID_ARG, "q3_session_set_kv_dtype: unknown dtype %d", dtype);
    if (dtype == Q3_DTYPE_BF16 && !s->paged) return set_err(Q3_UNSUPPORTED, "bf16 K/V needs the paged cache (Q3_KV_CONTIGUOUS is set)");
    s->kv_bf16 = dtype == Q3_DTYPE_BF16;
    return Q3_OK;
}

extern "C" q3_status q3_session_set_debug(q3_session* s, int capture) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->prefilled) return set_err(Q3_INVALID_ARG, "set_debug must be called before prefill");
    s->debug = capture != 0;
    if (s->debug && !s->logits_hist) {
        const q3_config& c = s->m->cfg;
        HIPC(hipSetDevice(s->m->device));
        HIPC(s->pool.alloc(&s->logits_hist, (size_t)s->B * (s->max_frames + 1) * c.codec_vocab));
        HIPC(s->pool.alloc(&s->cp_logits_hist, (size_t)s->max_frames * 15 * s->B * c.cp_vocab));
    }
    return Q3_OK;
}
extern "C" q3_status q3_session_set_profile(q3_session* s, int enable) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    s->profile = enable != 0;
    return Q3_OK;
}
extern "C" q3_status q3_session_stream(q3_session* s, void** stream) {
    if (!s || !stream) return set_err(Q3_INVALID_ARG, "null argument");
    *stream = (void*)s->stream;
    return Q3_OK;
}
extern "C" q3_status q3_session_prefill_len(q3_session* s, int b, int* prefill_len, int* trailing_len) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    if (!s->ragged.empty()) {                    // a ragged batch before its prefill: the rows are still idle placeholders
        int S = 0, L = 0; request_shape(s->ragged[(size_t)b].r, &S, &L);
        if (prefill_len) *prefill_len = S;
        if (trailing_len) *trailing_len = -1;
        return Q3_OK;
    }
    if (prefill_len) *prefill_len = s->seq[b].prefill_len;
    if (trailing_len) *trailing_len = s->seq[b].trailing_len;
    return Q3_OK;
}

// text projection (talker.rs:316-320) of `n` gathered rows: fc2(silu(fc1(e)+b1))+b2, 8 rows a time
static q3_status text_project(q3_session* s, const uint32_t* ids_dev, int n, float* out_rows) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const int TD = c.text_dim, H = c.hidden;
    if (n > s->n_rows_total) return set_err(Q3_INVALID_ARG, "text_project: %d rows exceed the session's %d", n, s->n_rows_total);
    float *e = s->proj_e, *h = s->proj_h;      // session scratch: everything below is queued on s->stream, nothing waits
    q3_status st = Q3_OK;
    hipError_t er = launch_gather_rows_bf16(m->text_emb, ids_dev, e, n, TD, s->stream);
    if (er == hipSuccess && n >= 48 && TD % 64 == 0 && H % 64 == 0 && !s->no_chunk) {
        // TextProjection (talker.rs:316-320) over all n token rows as two GEMMs instead of n/8 GEMV passes
        GemmArgs g1; g1.W = m->fc1w.t1; g1.x = e; g1.ldx = TD; g1.bias = m->fc1b; g1.y = h; g1.ldy = TD;
        g1.M = n; g1.N = TD; g1.K = TD; g1.Kpad = (TD + 31) / 32 * 32; g1.epi = EPI_SILU;
        er = launch_lm_gemm(g1, s->stream);
        if (er == hipSuccess) {
            GemmArgs g2; g2.W = m->fc2w.t1; g2.x = h; g2.ldx = TD; g2.bias = m->fc2b; g2.y = out_rows; g2.ldy = H;
            g2.M = n; g2.N = H; g2.K = TD; g2.Kpad = (TD + 31) / 32 * 32; g2.epi = EPI_NONE;
            er = launch_lm_gemm(g2, s->stream);
        }
    } else
    for (int r0 = 0; r0 < n && er == hipSuccess; r0 += 8) {
        const int M = (n - r0) < 8 ? (n - r0) : 8;
        LinArgs a;
        a.N = TD; a.K = TD; set_w(a, m->fc1w, M, TD, TD); a.x = e + (size_t)r0 * TD; a.ldx = TD; a.bias = m->fc1b; a.y = h + (size_t)r0 * TD; a.ldy = TD; a.M = M; a.epi = EPI_SILU;
        er = launch_linear(a, s->stream);
        if (er != hipSuccess) break;
        LinArgs b2;
        b2.N = H; b2.K = TD; set_w(b2, m->fc2w, M, H, TD); b2.x = h + (size_t)r0 * TD; b2.ldx = TD; b2.bias = m->fc2b; b2.y = out_rows + (size_t)r0 * H; b2.ldy = H; b2.M = M; b2.epi = EPI_NONE;
        er = launch_linear(b2, s->stream);
    }
    if (er != hipSuccess) st = set_err(Q3_HIP_ERROR, "text projection: %s", hipGetErrorString(er));
    return st;
}

// run_prefill_layers (talker.rs:823-841) for long prompts: chunks of up to 128 positions per sequence go through every
// layer as GEMMs over the decode path's tiled weight image + a query-blocked causal attention (q3_kernels_prefill.hip).
// Leaves the KV cache filled for positions [0, S) and LASTH / LOGITS of the last position, like the chunked decode-step
// schedule it replaces for S >= 48.
static q3_status prefill_gemm(q3_session* s, int S_all, int S, bool with_head) {      // positions [0, S) of the S_all-position prompts
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const LmDims d = talker_dims(c);
    const int B = s->B, H = d.H, QD = d.nh * HEAD_DIM, KD = d.nkv * HEAD_DIM, I = d.I;
    // positions per sequence per pass: up to 4224 activation rows per launch (a whole 4k-position prompt in one pass:
    // 33 M-tiles x N/128 workgroups per GEMM, 520 attention workgroups; 2048-row passes: 113 ms instead of 99 for the
    // 4105-position prefill of the 1.7B model), at least one 128-row tile per sequence
    static const int rows_env = [] { const char* e = getenv("Q3_PREFILL_ROWS"); return e ? atoi(e) : 4224; }();
    int C = rows_env / B; C = C < 128 ? 128 : (C / 128) * 128;
    const int max_rows = B * (S < C ? S : C);
    struct SyncedPool : DevPool { hipStream_t st; explicit SyncedPool(hipStream_t s_) : st(s_) {} ~SyncedPool() { (void)hipStreamSynchronize(st); } };
    SyncedPool tmp(s->stream);       // on EVERY return path the stream is drained before the blocks go back to the cache
    float *X, *QKV, *Qb, *ATT, *SUM, *ACT, *DEN;
    HIPC(tmp.alloc(&X, (size_t)max_rows * H)); HIPC(tmp.alloc(&QKV, (size_t)max_rows * (QD + 2 * KD)));
    HIPC(tmp.alloc(&Qb, (size_t)max_rows * QD)); HIPC(tmp.alloc(&ATT, (size_t)max_rows * QD));
    HIPC(tmp.alloc(&SUM, (size_t)max_rows * H)); HIPC(tmp.alloc(&ACT, (size_t)max_rows * I)); HIPC(tmp.alloc(&DEN, (size_t)max_rows));
    auto kp = [](int K) { return (K + 31) / 32 * 32; };
    // every GEMM input is split once into its three exact bf16 terms (launch_split_rows) instead of once per workgroup
    // column inside the GEMM; Q3_GEMM_NO_PLANES=1 keeps the in-kernel split (A/B aid)
    // long prompts: every query block's key range is halved over two workgroups and merged (k_attn_prefill_t)
    static const bool no_split = getenv("Q3_PREFILL_ATTN_NOSPLIT") != nullptr;
    static const bool gen2_attn = getenv("Q3_PREFILL_ATTN_GEN2") != nullptr || getenv("Q3_PREFILL_ATTN_VALU") != nullptr;
    const bool kv_split = !no_split && !gen2_attn && S >= 1024;
    float* PART = nullptr;
    if (kv_split) HIPC(tmp.alloc(&PART, (size_t)max_rows * d.nh * 2 * PART_STRIDE));
    // bf16-matrix-core attention (k_attn_prefill_x3) over per-layer bf16x3 planes of the cached K/V; Q3_PREFILL_ATTN_X3=0
    // keeps the f32-MFMA generation (A/B aid, read per call)
    const char* x3e = getenv("Q3_PREFILL_ATTN_X3");
    const bool attn_x3 = !gen2_attn && !(x3e && atoi(x3e) == 0) && S >= 256;
    unsigned char* KVP = nullptr;
    const int kvp_tiles = (S + 31) / 32;
    if (attn_x3) HIPC(tmp.alloc(&KVP, (size_t)B * d.nkv * kvp_tiles * KVP_TILE_BYTES));
    static const bool no_planes = getenv("Q3_GEMM_NO_PLANES") != nullptr;
    const bool planes = !no_planes && H % 8 == 0 && QD % 8 == 0 && I % 8 == 0;
    const int kmax = std::max(kp(H), std::max(kp(QD), kp(I)));
    const size_t plane_elems = (size_t)((max_rows + 127) / 128 * 128) * kmax;      // whole 128-row tiles (GemmArgs::xp)
    uint16_t* XP = nullptr;
    if (planes) HIPC(tmp.alloc(&XP, plane_elems * 3));
    auto split = [&](GemmArgs& g) -> hipError_t {
        if (!planes) return hipSuccess;
        g.xp = XP; g.xp_plane = plane_elems;
        return launch_split_rows(g.x, g.ldx, g.norm_w, XP, plane_elems, g.M, g.K, g.Kpad, s->stream);
    };
    // split-K workspace of the third GEMM geometry: small row counts only (that is where a GEMM's grid underfills the chip)
    float* SKW = nullptr; size_t skw_bytes = 0;
    if (planes && max_rows <= 1024) { skw_bytes = (size_t)8 * max_rows * (QD + 2 * KD > 2 * I / 4 ? QD + 2 * KD : 2 * I / 4) * sizeof(float); HIPC(tmp.alloc(&SKW, skw_bytes / sizeof(float))); }
    int ch = 0;
    for (int t0 = 0; t0 < S; t0 += C) {
        ch = (S - t0) < C ? (S - t0) : C;
        const int rows = B * ch;
        for (int b = 0; b < B; ++b)
            HIPC(launch_copy_rows(s->embeds + ((size_t)b * S_all + t0) * H, H, X + (size_t)b * ch * H, H, ch, H, s->stream));
        for (int i = 0; i < d.layers; ++i) {
            const LayerW& w = m->tl[i];
            HIPC(launch_row_den(X, H, DEN, rows, H, d.eps, s->stream));
            GemmArgs g; g.W = w.qkv.t1; g.x = X; g.ldx = H; g.norm_w = w.in_ln; g.den = DEN; g.y = QKV; g.ldy = QD + 2 * KD;
            g.M = rows; g.N = QD + 2 * KD; g.K = H; g.Kpad = kp(H); g.epi = EPI_NONE;
            g.splitk_ws = SKW; g.splitk_ws_bytes = skw_bytes;
            HIPC(split(g)); HIPC(launch_lm_gemm(g, s->stream));
            AttnArgs t{};
            t.qkv = QKV; t.ld_qkv = QD + 2 * KD; t.q_norm_w = w.q_norm; t.k_norm_w = w.k_norm; t.eps = d.eps;
            t.rope_cos = m->rope_cos; t.rope_sin = m->rope_sin; t.pos_dev = nullptr; t.pos_static = t0;
            if (s->paged) {
                t.kv_pages = s->kv_table; t.kv_layer_off = (size_t)i * m->kv_pool.layer_stride(); t.kv_vdelta = m->kv_pool.v_delta();
            } else { t.kcache = s->kcache + (size_t)i * s->kv_layer_stride; t.vcache = s->vcache + (size_t)i * s->kv_layer_stride; }
            t.max_seq = s->max_seq; t.qbuf = Qb; t.part = nullptr; t.out = ATT; t.ld_out = QD;
            t.B = rows; t.nh = d.nh; t.nkv = d.nkv; t.n_splits = 1; t.rows_per_seq = ch;
            HIPC(launch_qknorm_rope_kv(t, s->stream));
            if (attn_x3) {
                HIPC(launch_kv_planes(t, B * d.nkv, t0 + ch, kvp_tiles, KVP, s->stream));
                t.kvp = KVP; t.kvp_tiles = kvp_tiles;
            }
            if (kv_split) { t.part = PART; t.n_splits = 2; }
            HIPC(launch_attn_prefill(t, s->stream));
            if (kv_split) HIPC(launch_attn_merge(t, s->stream));
            GemmArgs o; o.W = w.o.t1; o.x = ATT; o.ldx = QD; o.resid = X; o.ldr = H; o.y = SUM; o.ldy = H;
            o.M = rows; o.N = H; o.K = QD; o.Kpad = kp(QD); o.epi = EPI_RESID;
            o.splitk_ws = SKW; o.splitk_ws_bytes = skw_bytes;
            HIPC(split(o)); HIPC(launch_lm_gemm(o, s->stream));
            HIPC(launch_row_den(SUM, H, DEN, rows, H, d.eps, s->stream));
            GemmArgs gu; gu.W = w.gate.t1; gu.W2 = w.up.t1; gu.x = SUM; gu.ldx = H; gu.norm_w = w.post_ln; gu.den = DEN; gu.y = ACT; gu.ldy = I;
            gu.M = rows; gu.N = I; gu.K = H; gu.Kpad = kp(H); gu.epi = EPI_SWIGLU;
            gu.splitk_ws = SKW; gu.splitk_ws_bytes = skw_bytes;
            HIPC(split(gu)); HIPC(launch_lm_gemm(gu, s->stream));
            GemmArgs dn; dn.W = w.down.t1; dn.x = ACT; dn.ldx = I; dn.resid = SUM; dn.ldr = H; dn.y = X; dn.ldy = H;
            dn.M = rows; dn.N = H; dn.K = I; dn.Kpad = kp(I); dn.epi = EPI_RESID;
            dn.splitk_ws = SKW; dn.splitk_ws_bytes = skw_bytes;
            HIPC(split(dn)); HIPC(launch_lm_gemm(dn, s->stream));
        }
    }
    if (!with_head) { HIPC(sync_frames(s)); return Q3_OK; }      // the caller runs the remaining positions
    // head on each sequence's last position (row b*ch + ch-1 of the last chunk): final norm -> LASTH, codec_head -> LOGITS
    HIPC(launch_rmsnorm(X + (size_t)(ch - 1) * H, ch * H, m->norm, s->LASTH, H, B, H, c.rms_eps, s->stream));
    LinArgs h;
    h.N = c.codec_vocab; h.K = H; set_w(h, m->codec_head, B, h.N, h.K); h.x = s->LASTH; h.ldx = H; h.y = s->LOGITS; h.ldy = c.codec_vocab;
    h.M = B; h.epi = EPI_NONE;
    HIPC(run_linear(s, h));
    HIPC(sync_frames(s));      // tmp buffers are freed on return
    return Q3_OK;
}

static q3_status transplant_row(q3_session* s, int b, q3_session* side, int j, int limit);
static q3_status transplant_check(q3_session* s, q3_session* side, int j, int limit_req, int* limit_out);
// q3_session_prefill of a ragged first batch (session_create_any): the idle rows never run — every row's state comes from a side
// session. Rows are grouped by prefill length in row order; a group of G rows is one batched side prefill.
static q3_status prefill_ragged(q3_session* s) {
    if (s->debug || s->profile) return set_err(Q3_UNSUPPORTED, "debug / profiling sessions need rows of one prefill length");
    const int B = s->B;
    s->kv_in_bf16 = s->kv_bf16;                          // the idle rows hold no pages: nothing to convert
    s->prefilled = true; s->frames_run = 0; s->codes_host_valid = false;      // transplant_row stamps rows with start_run = frames_run
    std::vector<char> placed((size_t)B, 0);
    for (int b0 = 0; b0 < B; ++b0) {
        if (placed[(size_t)b0]) continue;
        int S0 = 0, L = 0; request_shape(s->ragged[(size_t)b0].r, &S0, &L);
        std::vector<int> rows; std::vector<q3_request> reqs; std::vector<int> limits;
        for (int b = b0; b < B; ++b) {
            int S = 0, Lb = 0; request_shape(s->ragged[(size_t)b].r, &S, &Lb);
            if (placed[(size_t)b] || S != S0) continue;
            q3_request r = s->ragged[(size_t)b].r;
            // the row's RESOLVED limit (an ICL row's max_length is capped at max(75, 6 * n_text), talker.rs:646-710 / lib.rs:897-1046) is
            // what session_create_any sized max_frames for — the raw max_length of an ICL row may well exceed it
            if (r.opts.max_length < 1 || Lb < 1 || Lb > s->max_frames) { s->prefilled = false; return set_err(Q3_INVALID_ARG, "row %d: max_length %d (resolved %d) outside 1..%d", b, r.opts.max_length, Lb, s->max_frames); }
            limits.push_back(Lb);
            r.opts.max_length = s->max_frames;           // the side session draws the row's PCG stream with the host session's stride
            rows.push_back(b); reqs.push_back(r); placed[(size_t)b] = 1;
        }
        q3_session* side_raw = nullptr;
        q3_status st = session_create(s->m, reqs.data(), (int)reqs.size(), 0, 0, &side_raw, s->stream);
        std::unique_ptr<q3_session> side(side_raw);
        if (st == Q3_OK) { side->kv_bf16 = s->kv_bf16; }
        std::vector<int> lim(rows.size(), 0);
        for (size_t j = 0; j < rows.size() && st == Q3_OK; ++j) st = transplant_check(s, side.get(), (int)j, limits[j], &lim[j]);
        if (st == Q3_OK) st = q3_session_prefill(side.get());
        if (st == Q3_OK && sync_frames(s) != hipSuccess) st = set_err(Q3_HIP_ERROR, "ragged prefill: stream");
        for (size_t j = 0; j < rows.size() && st == Q3_OK; ++j) st = transplant_row(s, rows[j], side.get(), (int)j, lim[j]);
        if (st != Q3_OK) { s->prefilled = false; return st; }
    }
    s->ragged.clear();
    return Q3_OK;
}

static q3_status frame_capture(q3_session* s, bool stream_busy);
extern "C" q3_status q3_session_prefill(q3_session* s) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->prefilled) return set_err(Q3_INVALID_ARG, "session already prefilled");
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    HIPC(hipSetDevice(m->device));
    if (!s->ragged.empty()) return prefill_ragged(s);
    const int B = s->B, H = c.hidden, S = s->prefill_len;
    for (int b = 0; b < B; ++b) Q3C(kv_reserve_row(s, b, S + 1));      // paged KV: the prompt's positions and the first frame's
    // 1. ids to project, per sequence: [instruct…, IM_START, ASSISTANT, NEWLINE, TTS_PAD, TTS_BOS, text…, TTS_EOS]
    std::vector<uint32_t> ids; ids.reserve(s->n_rows_total);
    std::vector<int> text_row((size_t)B * S, -1), codec_id((size_t)B * S, -1);
    std::vector<int> trail_base(B), trail_len(B), pad_row(B);
    std::vector<float> xv((size_t)B * H, 0.0f);
    for (int b = 0; b < B; ++b) {
        SeqInfo& q = s->seq[b];
        const int n_ins = (int)q.instruct.size(), n_text = (int)q.text.size(), base = q.row_base;
        for (uint32_t id : q.instruct) ids.push_back(id);
        ids.push_back(IM_START); ids.push_back(ASSISTANT); ids.push_back(NEWLINE); ids.push_back(TTS_PAD); ids.push_back(TTS_BOS);
        if (q.icl) for (uint32_t id : q.ref_text) ids.push_back(id);       // ICL: [ref_text, target_text, tts_eos] (talker.rs:656-663)
        for (uint32_t id : q.text) ids.push_back(id);
        ids.push_back(TTS_EOS);
        const int n_ref_text = q.icl ? (int)q.ref_text.size() : 0;
        const int r_role = base + n_ins, r_pad = r_role + 3, r_bos = r_pad + 1, r_text = r_bos + 1, r_eos = r_text + n_ref_text + n_text;
        const int n_icl = q.icl ? (int)(q.ref_codes.size() / 16) + 1 : 0, n_text_all = n_ref_text + n_text + 1;
        q.pad_row = r_pad;
        if (q.icl) q.trail_base = n_text_all > n_icl ? r_text + n_icl : r_pad;       // talker.rs:692-708
        else q.trail_base = n_text > 1 ? r_text + 1 : r_eos;          // build_trailing_text (lib.rs:508-519)
        trail_base[b] = q.trail_base; trail_len[b] = q.trailing_len; pad_row[b] = q.pad_row;
        // prefill positions (talker.rs:451-491 / 511-564 / 585-627)
        int* tr = &text_row[(size_t)b * S]; int* ci = &codec_id[(size_t)b * S];
        int p = 0;
        for (int i = 0; i < n_ins; ++i) tr[p++] = base + i;
        for (int i = 0; i < 3; ++i) tr[p++] = r_role + i;
        const bool vd = q.req.mode == Q3_MODE_VOICE_DESIGN;
        int codec[7]; int nc;
        if (vd) { int t[6] = {CODEC_THINK, CODEC_THINK_BOS, (int)q.req.language_id, CODEC_THINK_EOS, CODEC_PAD, CODEC_BOS}; memcpy(codec, t, sizeof t); nc = 6; }
        else { int t[7] = {CODEC_THINK, CODEC_THINK_BOS, (int)q.req.language_id, CODEC_THINK_EOS, (int)q.req.speaker_id, CODEC_PAD, CODEC_BOS}; memcpy(codec, t, sizeof t); nc = 7; }
        const int overlay = nc - 1;
        for (int i = 0; i < overlay; ++i) {
            tr[p] = (i == overlay - 1) ? r_bos : r_pad;
            ci[p] = (q.req.mode == Q3_MODE_VOICE_CLONE && i == 4) ? -2 : codec[i];
            ++p;
        }
        if (n_text > 0 && !q.icl) { tr[p] = r_text; ci[p] = codec[nc - 1]; ++p; }
        for (int i = 0; i < n_icl; ++i) {      // ICL block: text (or tts_pad) row + codec_bos / Σ16 reference-frame embeddings
            tr[p] = i < n_text_all ? r_text + i : r_pad;
            ci[p] = i == 0 ? CODEC_BOS : -3 - (i - 1);
            ++p;
        }
        if (!q.xvec.empty()) memcpy(&xv[(size_t)b * H], q.xvec.data(), (size_t)H * 4);
    }
    uint32_t* ids_dev = s->ids_dev; int *tr_dev = s->tr_dev, *ci_dev = s->ci_dev;
    if ((int)ids.size() != s->n_rows_total) return set_err(Q3_INVALID_ARG, "prefill: row count mismatch (%zu vs %d)", ids.size(), s->n_rows_total);
    // reference frames of ICL sequences (also needed later by the ICL decode)
    std::vector<size_t> ref_off(B, 0); size_t ref_total = 0;
    for (int b = 0; b < B; ++b) { ref_off[b] = ref_total; ref_total += s->seq[b].ref_codes.size(); }
    if (ref_total && !s->ref_codes_dev) {
        HIPC(s->pool.alloc(&s->ref_codes_dev, ref_total));
        for (int b = 0; b < B; ++b)
            if (!s->seq[b].ref_codes.empty())
                HIPC(hipMemcpy(s->ref_codes_dev + ref_off[b], s->seq[b].ref_codes.data(), s->seq[b].ref_codes.size() * 4, hipMemcpyHostToDevice));
    }
    // uploads ride the session stream (the host vectors live until the synchronisation that ends this function)
    HIPC(hipMemcpyAsync(ids_dev, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(tr_dev, text_row.data(), text_row.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(ci_dev, codec_id.data(), codec_id.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->xvec, xv.data(), xv.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->trail_base, trail_base.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->trail_len, trail_len.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->pad_row, pad_row.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    q3_status st = text_project(s, ids_dev, (int)ids.size(), s->rows);
    if (st == Q3_OK) {
        hipError_t e = hipSuccess;
        for (int b = 0; b < B && e == hipSuccess; ++b)
            e = launch_assemble_rows(s->rows, tr_dev + (size_t)b * S, m->codec_emb, ci_dev + (size_t)b * S, s->xvec + (size_t)b * H,
                                     s->embeds + (size_t)b * S * H, S, H, s->stream,
                                     s->ref_codes_dev ? s->ref_codes_dev + ref_off[b] : nullptr, m->cp_embs_dev);
        if (e != hipSuccess) st = set_err(Q3_HIP_ERROR, "prefill assembly: %s", hipGetErrorString(e));
    }
    if (st != Q3_OK) { (void)sync_frames(s); return st; }
    // 2. run_prefill_layers (talker.rs:823-841): causal attention ⇒ token-by-token decode steps
    //    and the GEMV kernels take up to 16 rows for the price of one, so each weight pass carries a CHUNK of
    //    16/B consecutive positions per sequence (q3_kernels.h AttnArgs::rows_per_seq). Bit-identical to the
    //    one-position-at-a-time schedule (rows are independent in the GEMV; attention sees the same K/V).
    static const int gemm_min = [] { const char* e = getenv("Q3_PREFILL_GEMM_MIN"); return e ? atoi(e) : 48; }();   // 0 disables the GEMM path
    const bool tiles_ok = (d_nh_ok(c));
    const int chunk = s->no_chunk ? 1 : (16 / B > 0 ? 16 / B : 1);
    // The GEMM path works in 128-position tiles and its grids are sized to fill the chip in whole rounds (4096
    // positions: 256 / 512 / 1536 workgroups of 256 CUs' worth); a few positions past the last full tile would cost every
    // GEMM another round (4105 positions, one sequence: +9 ... +50 % per GEMM). Up to Q3_PREFILL_TAIL (default 32)
    // trailing positions of a >= 1024-position prompt therefore go through the decode-step schedule below instead (about
    // 1 ms per 16 rows at 4k context), which appends to the same KV cache. The rule looks at the PROMPT only — never at the
    // batch — so which kernels compute a given position does not depend on how many sequences are prefilled together
    // (a batch pays ceil(B * r / 16) passes for it; positions below the cut always take the GEMM, whose bits are
    // batch-invariant; the decode-step kernels pick their tiling by the row count of a pass, like any decode step).
    static const int tail_max = [] { const char* e = getenv("Q3_PREFILL_TAIL"); return e ? atoi(e) : 32; }();
    int t_begin = 0;
    if (!s->no_chunk && !s->debug && gemm_min > 0 && S >= gemm_min && tiles_ok) {
        const int r = S % 128;
        const int Sg = (S >= 1024 && r > 0 && r <= tail_max) ? S - r : S;
        Q3C(prefill_gemm(s, S, Sg, Sg == S));
        t_begin = Sg;
    }
    for (int t0 = t_begin; t0 < S; t0 += chunk) {
        const int ch = (S - t0) < chunk ? (S - t0) : chunk;
        for (int b = 0; b < B; ++b)
            HIPC(launch_copy_rows(s->embeds + ((size_t)b * S + t0) * H, H, s->tb.X + (size_t)b * ch * H, H, ch, H, s->stream));
        Q3C(talker_step(s, nullptr, t0, t0 + ch >= S, ch));
    }
    // 3. first sampling decision (lib.rs:558-571)
    std::vector<int> posv(B, S), zero(B, 0);
    HIPC(hipMemcpyAsync(s->pos, posv.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->frame_idx, zero.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->token_count, zero.data(), B * 4, hipMemcpyHostToDevice, s->stream));
    SampleArgs a; fill_sample_args(s, a); a.advance = 0;
    HIPC(launch_sample(a, s->stream));
    // Callers that are going to replay the frame (q3_session_run, q3_session_next_chunk) have it captured and converted HERE, while
    // the prompt's kernels run: ~2 ms of host work that used to sit between the prefill and the first frame (time to first audio).
    // (bf16-KV sessions capture later: which attention kernel the frame holds depends on the conversion below.)
    if (s->precapture && !s->kv_bf16 && !s->debug && !s->profile) Q3C(frame_capture(s, true));
    HIPC(sync_frames(s));
    if (s->kv_bf16 && !s->kv_in_bf16) Q3C(kv_convert_to_bf16(s));       // the prompt's K/V moves into pages of the bf16 pool, once
    s->prefilled = true; s->frames_run = 0; s->codes_host_valid = false;
    return Q3_OK;
}

static q3_status refresh_codes(q3_session* s) {
    if (s->codes_host_valid) return Q3_OK;
    HIPC(sync_frames(s));
    s->codes_host.resize((size_t)s->B * s->max_frames * 16);
    for (int b = 0; b < s->B; ++b) {
        int ran = s->frames_run - s->seq[b].start_run;
        if (ran > s->seq[b].limit) ran = s->seq[b].limit;
        if (ran > 0)
            HIPC(hipMemcpy(&s->codes_host[(size_t)b * s->max_frames * 16], s->codes + (size_t)b * s->max_frames * 16,
                           (size_t)ran * 16 * 4, hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> tok(s->B);
    HIPC(hipMemcpy(tok.data(), s->tok, s->B * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < s->B; ++b) {
        SeqInfo& q = s->seq[b];
        int n = s->frames_run - q.start_run; bool done = false;      // frames this row has run (rows swapped in later started later)
        if (n > q.limit) n = q.limit;
        if (n < 0) n = 0;
        const int eos = q.req.opts.eos_token_id;                     // per row (SampleRow)
        if (eos >= 0) {
            const int ran = n;
            for (int f = 0; f < ran; ++f)
                if ((int)s->codes_host[((size_t)b * s->max_frames + f) * 16] == eos) { n = f; done = true; break; }
            if (!done && ran < q.limit && (int)tok[b] == eos) done = true;     // EOS sampled for the next frame
        }
        if (n >= q.limit) done = true;
        q.n_frames = n; q.done = done;
    }
    s->codes_host_valid = true;
    return Q3_OK;
}

static bool all_done(q3_session* s) { for (auto& q : s->seq) if (!q.done) return false; return true; }
// frames the session still has to run for its longest-remaining row (lockstep sessions: max_frames - frames_run)
static int session_remaining(const q3_session* s) {
    int r = 0;
    for (const auto& q : s->seq) { const int left = q.limit - (s->frames_run - q.start_run); if (left > r) r = left; }
    return r;
}

// Which kernels of the frame keep to the activation-transport rule of q3_kernels.h (write-through + drained stores, L1-bypassing
// loads of everything an earlier node of the same frame wrote, nothing of it through the scalar cache): their packets go out
// without the agent-scope acquire / release fences (q3_aql.h). Anything not named here keeps HIP's fences — a kernel added to
// the frame later is safe by default. Development switches: Q3_AQL_T_ACQ=0 / Q3_AQL_T_REL=0 keep that half of every boundary,
// Q3_AQL_T_ONLY=<substring+substring> restricts the rule to kernels whose name holds one of the substrings (bisecting).
static void frame_fence_policy(const char* name, int* acquire, int* release) {
    static const char* const families[] = {"k_gemv_mfmaI", "k_gemv_sk2I", "k_gemv_gu24I", "k_gemv_ldsI", "k_gemv_mfma4I",
                                           "k_attn_cpI", "k_attn_fusedI", "k_attn_mergeI", "k_attn_first2I"};
    const bool drop_acq = !(getenv("Q3_AQL_T_ACQ") && atoi(getenv("Q3_AQL_T_ACQ")) == 0);
    const bool drop_rel = !(getenv("Q3_AQL_T_REL") && atoi(getenv("Q3_AQL_T_REL")) == 0);
    const std::string only = getenv("Q3_AQL_T_ONLY") ? getenv("Q3_AQL_T_ONLY") : "";
    bool conv = false;
    for (const char* f : families) conv = conv || strstr(name, f) != nullptr;
    if (conv && !only.empty()) {
        bool hit = false; size_t i = 0;
        while (i <= only.size()) {
            const size_t j = only.find_first_of(",+", i); const std::string t = only.substr(i, j == std::string::npos ? std::string::npos : j - i);
            if (!t.empty() && strstr(name, t.c_str())) hit = true;
            if (j == std::string::npos) break;
            i = j + 1;
        }
        conv = hit;
    }
    if (!conv) return;
    if (drop_acq) *acquire = 0;
    if (drop_rel) *release = 0;
}

// The frame is captured ONCE per session (frame_launch under stream capture: every per-frame quantity lives in device memory, so
// the graph is static) and turned into the packet program of the library's own AQL queue (q3_aql.cpp); only a graph the converter
// cannot take — or Q3_AQL=0 — is instantiated for hipGraphLaunch. `stream_busy`: the caller has work queued on the session's
// stream that must NOT be waited for here (q3_session_prefill captures while the prompt's kernels run: capture, conversion and
// the kernarg upload are host work, ~2 ms that used to sit between the prefill and the first frame of every session).
// Q3_AQL unset / 3 (round 6: the default; the whole -m gpu suite runs through it): boundaries between kernels that keep to the
// activation-transport rule (q3_kernels.h) without HIP's agent-scope fences. Q3_AQL=0: hipGraphLaunch; 1: own queue with HIP's
// fences on every packet (bit-identical to 0); 2: probe. Why a graph stayed on hipGraphLaunch: Q3_AQL_VERBOSE=1.
static q3_status frame_capture(q3_session* s, bool stream_busy) {
    // (another host thread's allocations or null-stream work can invalidate a capture in progress on this HIP runtime, thread-local
    // capture mode or not — the batcher's prefill worker, a server opening sessions on several threads: the capture is repeated)
    for (int attempt = 0; !s->graph; ++attempt) {
        if (!stream_busy) HIPC(sync_frames(s));
        {
            const hipError_t eb = hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal);
            if (eb != hipSuccess && stream_busy) { (void)hipGetLastError(); return Q3_OK; }      // not now: q3_session_generate captures on the idle stream
            HIPC(eb);
        }
        const q3_status st = frame_launch(s);
        const hipError_t e = hipStreamEndCapture(s->stream, &s->graph);
        if (st == Q3_OK && e == hipSuccess && s->graph) break;
        if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
        (void)hipGetLastError();
        if (attempt >= 3) { Q3C(st); return set_err(Q3_HIP_ERROR, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
    }
    if (!s->aql && !s->aql_tried) {
        s->aql_tried = true;
        const char* e = getenv("Q3_AQL");
        int mode = e ? atoi(e) : 3;
        // Packets without ANY boundary fence give WRONG codes (state that crosses frames moves with plain accesses):
        // mode 2 and the fence halves are probes and need an explicit opt-in, so that a stray environment variable cannot
        // silently corrupt a server's output.
        const bool unsafe_ok = getenv("Q3_AQL_UNSAFE") && atoi(getenv("Q3_AQL_UNSAFE")) == 1;
        const bool wants_unsafe = mode == 2 || getenv("Q3_AQL_ACQ") || getenv("Q3_AQL_REL");
        if (wants_unsafe && !unsafe_ok) {
            static std::atomic<bool> told{false};
            if (!told.exchange(true)) fprintf(stderr, "[q3] Q3_AQL=2 / Q3_AQL_ACQ / Q3_AQL_REL drop kernel-boundary fences and produce wrong results with the product kernels; "
                                                      "ignored without Q3_AQL_UNSAFE=1 (frames stay on %s)\n", mode == 2 ? "hipGraphLaunch" : "HIP's fence policy");
            if (mode == 2) mode = 0;
        }
        if (mode > 0) {
            q3::AqlPolicy pol; pol.fence = mode == 2 ? 0 : 1;
            pol.acquire = pol.release = pol.fence;
            if (mode == 3) pol.node_policy = frame_fence_policy;      // fence-free boundaries between the kernels that move their data write-through
            if (unsafe_ok) {
                if (const char* a = getenv("Q3_AQL_ACQ")) pol.acquire = atoi(a);        // probes: the two fences of a boundary priced separately
                if (const char* r = getenv("Q3_AQL_REL")) pol.release = atoi(r);
                fprintf(stderr, "[q3] WARNING: Q3_AQL_UNSAFE=1: frames are submitted with acquire=%d release=%d — results are NOT valid with the product kernels\n", pol.acquire, pol.release);
            }
            std::string why;
            s->aql = q3::aql_program_create(s->graph, s->m->device, pol, &why);
            if (s->aql) s->aql_mode = mode == 2 ? 2 : mode == 3 ? 3 : 1;
            else if (getenv("Q3_AQL_VERBOSE")) fprintf(stderr, "[q3] AQL submission unavailable, staying on hipGraphLaunch: %s\n", why.c_str());
        }
    }
    if (!s->aql && !s->graph_exec) HIPC(hipGraphInstantiate(&s->graph_exec, s->graph, nullptr, nullptr, 0));
    return Q3_OK;
}
// `n` replays of the captured frame enqueued WITHOUT waiting for them (streaming read-ahead): on the own queue behind whatever it
// holds, or as graph launches on the session's stream. The caller reserved the K/V pages (kv_reserve_frames) and, on the own queue,
// made sure everything the first frame reads has landed (the queue is not ordered with the HIP stream). sync_frames waits for both.
static q3_status frames_enqueue(q3_session* s, int n) {
    if (n <= 0) return Q3_OK;
    if (s->aql) {
        std::string why; int handed = 0;
        const bool ok = q3::aql_submit(s->aql, n, &why, &handed);
        s->frames_run += handed;
        if (!ok) { s->aql_failed = true; s->codes_host_valid = false; return set_err(Q3_HIP_ERROR, "AQL frame submission: %s", why.c_str()); }
    } else {
        if (!s->graph_exec) return set_err(Q3_INVALID_ARG, "frames_enqueue: no captured frame");
        for (int i = 0; i < n; ++i) HIPC(hipGraphLaunch(s->graph_exec, s->stream));
        s->frames_run += n;
    }
    s->codes_host_valid = false;
    return Q3_OK;
}

extern "C" q3_status q3_session_generate(q3_session* s, int n_frames, int use_graph) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (!s->prefilled) return set_err(Q3_INVALID_ARG, "session not prefilled");
    if (s->aql_failed) return set_err(Q3_HIP_ERROR, "session unusable: an earlier frame submission on the AQL queue failed or timed out");
    HIPC(hipSetDevice(s->m->device));
    if (s->debug || s->profile) use_graph = 0;
    int todo = n_frames;
    { const int left = session_remaining(s); if (todo > left) todo = left; }
    if (todo <= 0) return Q3_OK;
    Q3C(kv_reserve_frames(s, todo));        // paged KV: every page these frames can reach, before the first of them is queued
    if (use_graph) Q3C(frame_capture(s, false));
    bool on_aql = use_graph && s->aql;
    if (on_aql) HIPC(sync_frames(s));     // the queue is not ordered with the HIP stream: prefill / swaps must have landed
    bool eos_on = false;
    for (const auto& q : s->seq) eos_on = eos_on || q.req.opts.eos_token_id >= 0;
    const int check_every = 32;
    while (todo > 0) {
        const int burst = eos_on ? (todo < check_every ? todo : check_every) : todo;
        if (on_aql) {
            // (a failed submission or wait marks the session failed: the ring may still hold — or run — packets that write this
            // session's buffers, so nothing may replay behind them; frames_run stays in step with what the device was given)
            q3_status st = frames_enqueue(s, burst);
            if (st == Q3_OK && sync_frames(s) != hipSuccess) st = Q3_HIP_ERROR;
            if (st != Q3_OK) { s->aql_failed = true; s->codes_host_valid = false; return st; }
        } else
        for (int i = 0; i < burst; ++i) {
            if (use_graph) HIPC(hipGraphLaunch(s->graph_exec, s->stream));
            else Q3C(frame_launch(s));
            s->frames_run += 1;
        }
        todo -= burst;
        s->codes_host_valid = false;
        if (eos_on) { Q3C(refresh_codes(s)); if (all_done(s)) break; }
    }
    HIPC(sync_frames(s));
    if (s->profile && !s->prof_events.empty()) {
        for (size_t i = 0; i < s->prof_events.size(); ++i) {
            float ms = 0; hipEventElapsedTime(&ms, s->prof_events[i].first, s->prof_events[i].second);
            s->prof_linear.ms += ms; s->prof_linear.bytes += s->prof_event_bytes[i]; s->prof_linear.launches += 1;
        }
        s->prof_events.clear(); s->prof_event_bytes.clear(); s->prof_pool_next = 0;
    }
    return Q3_OK;
}

static q3_status decode_range_on(q3_session* s, int b, int f0, int f1, hipStream_t st, float* pcm_host, size_t cap, size_t* n_samples);

// Continuous batching: swap a finished row of a running session for a new request (include/q3tts.h). The reference keeps all
// per-utterance state per call (KV caches, sampling context, penalty mask, trailing text: lib.rs:743-756, 1484-1541); here
// that state is the row's slice of the session's device arrays, so a swap = prefill the request in a one-row side session
// (the unchanged prefill path) and copy its slice in: K/V extents of the prompt positions, last hidden state, first sampled
// token, penalty mask, counters, the pre-drawn PCG stream, projected text rows. The captured frame graph is untouched — it
// only ever reads these arrays — and the other rows do not notice: their state, and therefore their bits, are unchanged.
// Row j of a prefilled side session becomes row b of the host session: the per-row state the captured frame graph reads is
// copied in, the prompt's K/V pages are relinked (contiguous extents: copied). Both streams are idle (the caller drained them).
static q3_status transplant_row(q3_session* s, int b, q3_session* side, int j, int limit) {
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    const SeqInfo& sq = side->seq[(size_t)j];
    const int H = c.hidden, S = sq.prefill_len, nkv = c.n_kv_heads;
    const size_t row_bytes = (size_t)HEAD_DIM * 4;
    if (s->paged) {
        // the prompt's K/V is not copied: the side session's pages become the row's (its old ones go back to the pool), and the
        // row's table entries are rewritten
        kv_release_row(s, b);
        std::vector<float*>& row = s->kv_rows[(size_t)b];
        row.assign(side->kv_rows[(size_t)j].begin(), side->kv_rows[(size_t)j].end());
        side->kv_rows[(size_t)j].clear();
        HIPC(hipMemcpyAsync(s->kv_table + (size_t)b * KV_MAX_PAGES, row.data(), row.size() * 8, hipMemcpyHostToDevice, s->stream));
    } else
    for (int l = 0; l < c.n_layers; ++l) {
        const size_t so = (size_t)l * side->kv_layer_stride + (size_t)j * nkv * side->max_seq * HEAD_DIM, dof = (size_t)l * s->kv_layer_stride + (size_t)b * nkv * s->max_seq * HEAD_DIM;
        HIPC(hipMemcpy2DAsync(s->kcache + dof, s->max_seq * row_bytes, side->kcache + so, side->max_seq * row_bytes, S * row_bytes, nkv, hipMemcpyDeviceToDevice, s->stream));
        HIPC(hipMemcpy2DAsync(s->vcache + dof, s->max_seq * row_bytes, side->vcache + so, side->max_seq * row_bytes, S * row_bytes, nkv, hipMemcpyDeviceToDevice, s->stream));
    }
    auto d2d = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s->stream); };
    HIPC(d2d(s->LASTH + (size_t)b * H, side->LASTH + (size_t)j * H, (size_t)H * 4));
    HIPC(d2d(s->tok + b, side->tok + j, 4));
    HIPC(d2d(s->seen + (size_t)b * c.codec_vocab, side->seen + (size_t)j * c.codec_vocab, (size_t)c.codec_vocab));
    HIPC(d2d(s->token_count + b, side->token_count + j, 4));
    HIPC(d2d(s->pos + b, side->pos + j, 4));
    HIPC(d2d(s->frame_idx + b, side->frame_idx + j, 4));
    // the row's pre-drawn PCG stream: the side session drew max_frames(side) + 1 >= limit + 1 of them (an ICL cap may make it the shorter one)
    HIPC(d2d(s->U + (size_t)b * (s->max_frames + 2), side->U + (size_t)j * (side->max_frames + 2),
             (size_t)((side->max_frames < s->max_frames ? side->max_frames : s->max_frames) + 2) * 4));
    const int row0 = s->repl_base + b * s->row_cap;
    HIPC(d2d(s->rows + (size_t)row0 * H, side->rows + (size_t)sq.row_base * H, (size_t)sq.n_rows * H * 4));
    const int hv[4] = {row0 + (sq.trail_base - sq.row_base), sq.trailing_len, row0 + (sq.pad_row - sq.row_base), limit};
    HIPC(hipMemcpyAsync(s->trail_base + b, &hv[0], 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->trail_len + b, &hv[1], 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->pad_row + b, &hv[2], 4, hipMemcpyHostToDevice, s->stream));
    HIPC(hipMemcpyAsync(s->limit + b, &hv[3], 4, hipMemcpyHostToDevice, s->stream));
    const SampleRow srow = sample_row(sq.req.opts);
    HIPC(hipMemcpyAsync(s->sample_rows + b, &srow, sizeof srow, hipMemcpyHostToDevice, s->stream));
    HIPC(sync_frames(s));
    SeqInfo nq = sq;
    nq.row_base = row0; nq.trail_base = hv[0]; nq.pad_row = hv[2];
    nq.start_run = s->frames_run; nq.limit = limit; nq.n_frames = 0; nq.done = false; nq.stream_pos = 0; nq.req.opts.max_length = limit; nq.idle = false;
    s->seq[(size_t)b] = nq;
    {   // the request's arrays live in the row's own vectors (the caller's pointers need not outlive the call)
        SeqInfo& q = s->seq[(size_t)b];
        q.req.text_ids = q.text.data(); q.req.instruct_ids = q.instruct.data(); q.req.ref_codes = q.ref_codes.data();
        q.req.ref_text_ids = q.ref_text.data(); q.req.xvector = q.xvec.empty() ? nullptr : q.xvec.data();
    }
    if (b == 0) s->stream_pos = 0;       // q3_session_next_chunk (the row-0 streaming call) starts over with the new utterance too
    s->codes_host_valid = false;
    return Q3_OK;
}
// what a side session must satisfy before its rows may enter the host session
static q3_status transplant_check(q3_session* s, q3_session* side, int j, int limit_req, int* limit_out) {
    const SeqInfo& sq = side->seq[(size_t)j];
    if (side->opts.chunk_frames != s->opts.chunk_frames) return set_err(Q3_UNSUPPORTED, "q3_session_replace: chunk_frames is a property of the session");
    const int limit = limit_req < sq.limit ? limit_req : sq.limit;
    if (sq.n_rows > s->row_cap) return set_err(Q3_UNSUPPORTED, "q3_session_replace: the request's %d text rows exceed the session's slot (%d rows: 1024, or the longest text of the original batch)", sq.n_rows, s->row_cap);
    if (s->paged != side->paged) return set_err(Q3_UNSUPPORTED, "q3_session_replace: the sessions disagree on KV paging");
    if (s->kv_bf16 != s->kv_in_bf16) return set_err(Q3_UNSUPPORTED, "q3_session_replace: the session's K/V conversion has not happened yet");
    // max_seq bounds a row in both layouts: the captured frame was specialised for it (key splits, the page-table form of the
    // attention kernel); with pages it reserves nothing — only the pages a row really reaches are taken from the pool
    if (sq.prefill_len + limit + 1 > s->max_seq) return set_err(Q3_KV_OVERFLOW, "q3_session_replace: prompt of %d positions + %d frames exceeds the row's KV extent (%d)", sq.prefill_len, limit, s->max_seq);
    *limit_out = limit;
    return Q3_OK;
}

extern "C" q3_status q3_session_replace(q3_session* s, int b, const q3_request* req) {
    if (!s || !req || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "q3_session_replace: bad argument");
    if (!s->prefilled) return set_err(Q3_INVALID_ARG, "q3_session_replace: session not prefilled");
    if (s->debug || s->profile) return set_err(Q3_UNSUPPORTED, "q3_session_replace: not on debug / profiling sessions");
    const q3_model* m = s->m;
    HIPC(hipSetDevice(m->device));
    q3_request r = *req;
    const int limit_req = r.opts.max_length;
    if (limit_req < 1 || limit_req > s->max_frames) return set_err(Q3_UNSUPPORTED, "q3_session_replace: max_length %d outside 1..%d (the session's frame budget)", limit_req, s->max_frames);
    r.opts.max_length = s->max_frames;                 // the side session draws the row's PCG stream with the host session's stride
    static const bool timing = getenv("Q3_REPLACE_TIMING") != nullptr;      // development aid: where a swap's milliseconds go
    const auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) fprintf(stderr, "[q3 replace] %-8s %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
    };
    q3_session* side_raw = nullptr;
    Q3C(session_create(s->m, &r, 1, 0, 0, &side_raw, s->stream));      // on the host's stream: its frames and this prefill are serial anyway
    std::unique_ptr<q3_session> side(side_raw);
    side->kv_bf16 = s->kv_bf16;                        // the side session prefills in f32 and converts, as the host session did
    lap("create");
    // (sampling options are per row — SampleRow —, resolved by the side session: an ICL request's repetition-penalty floor and
    // length cap, lib.rs:913-929, come along)
    int limit = 0;
    Q3C(transplant_check(s, side.get(), 0, limit_req, &limit));
    Q3C(q3_session_prefill(side.get()));               // ends with a synchronisation of the side stream
    lap("prefill");
    HIPC(sync_frames(s));             // no frame of the host session in flight while its row changes
    Q3C(transplant_row(s, b, side.get(), 0, limit));
    lap("copies");
    side.reset();
    lap("free");
    return Q3_OK;
}

// Streaming with several sequences in one session: the next chunk of row b (StreamingSession::next_chunk, lib.rs:1650-1759,
// one per row). Rows advance in lockstep, so asking row after row costs the frames once: the first call generates them for
// every row, the others find theirs buffered and only run their vocoder. One row: q3_session_next_chunk (with read-ahead).
// ------------------------------------------------------------------------------------------------
// Continuous batcher: a queue of requests through the rows of ONE session (the native form of what a serving loop does
// with q3_session_replace). No thread of its own: the host calls q3_batcher_step from its loop — submit / step / poll /
// fetch may interleave freely (one thread at a time). A step fills free rows from the queue (one-row side prefill + state
// copy, rows of any prompt kind), runs up to n_frames frames of the shared frame graph, and collects the rows that ended
// (codes, and the PCM if the request asked for it). Every request gets the bits of its own batch-1 run.
// ------------------------------------------------------------------------------------------------
struct BatTicket {
    BatReq req; int state = Q3_TICKET_QUEUED; int row = -1; bool want_pcm = false;
    std::vector<uint32_t> codes; std::vector<float> pcm; int n_frames = 0;
    q3_status st = Q3_OK; std::string err;
};
struct q3_batcher {
    q3_model* m = nullptr; int slots = 0, frame_budget = 0, prompt_budget = 0, chunk_frames = 0;
    q3_session* s = nullptr;
    std::vector<int64_t> owner;                       // ticket running in each row, -1 = free
    std::vector<long> commit;                         // KvBudget units row r may still come to hold (worst case of its request), 0 = free row
    std::vector<int64_t> queue;                       // FIFO of waiting tickets
    std::unordered_map<int64_t, std::unique_ptr<BatTicket>> t;
    int64_t next_id = 1;
    // Round 6: the head of the queue is prefilled AHEAD of the row it will enter. A worker thread opens its one-row side session
    // and runs the (unchanged) prefill on a stream of its own while the captured frame keeps replaying for the live rows — the
    // frame leaves most of the chip idle —; when a row ends, the swap is only the state copy (transplant_row) at that frame
    // boundary. The other rows used to stand still for the whole side prefill (1.9 ms for a short prompt, 45 ms for a 4k-token
    // one). Bits are unchanged: the same kernels on the same inputs, only on another stream. Off under a page limit
    // (q3_model_kv_pool_limit: admission must see a row's pages when it decides) and with Q3_BAT_NO_STAGE=1 (A/B aid).
    struct Stage {
        int64_t id = -1; std::thread thr; q3_session* side = nullptr; q3_status st = Q3_OK; std::string err; int limit = 0;
    } stage;
};
static void stage_join(q3_batcher* b) { if (b->stage.thr.joinable()) b->stage.thr.join(); }
static void stage_drop(q3_batcher* b) {
    stage_join(b);
    if (b->stage.side) { q3_session_free(b->stage.side); b->stage.side = nullptr; }
    b->stage.id = -1; b->stage.st = Q3_OK; b->stage.err.clear();
}

// a row that has nothing to do: frozen on the device from its current frame on (rows of a freshly opened session that no
// request occupies yet)
static q3_status session_idle_row(q3_session* s, int b) {
    SeqInfo& q = s->seq[b];
    int ran = s->frames_run - q.start_run; if (ran < 0) ran = 0; if (ran > q.limit) ran = q.limit;
    q.limit = ran;
    HIPC(sync_frames(s));          // no frame in flight while the row's limit and pages change
    HIPC(hipMemcpy(s->limit + b, &q.limit, sizeof(int), hipMemcpyHostToDevice));
    // A frozen row still runs through every frame (its results are dropped): it reads its keys and rewrites the K/V of its
    // frozen position, prefill_len + ran. It keeps the ONE page that position lies in and every table entry it can reach points
    // there (stale keys are as good as any for a row nobody reads); the other pages go back to the pool, and the row takes no
    // more (kv_reserve_frames skips it) — a finished row must not sit on pages the queue is waiting for.
    if (s->paged && !q.idle && !s->kv_rows[(size_t)b].empty()) {
        std::vector<float*>& row = s->kv_rows[(size_t)b];
        size_t keep = (size_t)(q.prefill_len + ran) / KV_PAGE_POS; if (keep >= row.size()) keep = row.size() - 1;
        float* kept = row[keep];
        std::vector<float*> back;
        for (size_t i = 0; i < row.size(); ++i) if (i != keep) back.push_back(row[i]);
        if (!back.empty()) (s->kv_in_bf16 ? s->m->kv_pool16 : s->m->kv_pool).give(back);
        row.assign(1, kept);
        std::vector<unsigned long long> ent(keep + 1, (unsigned long long)kept);
        HIPC(hipMemcpy(s->kv_table + (size_t)b * KV_MAX_PAGES, ent.data(), ent.size() * 8, hipMemcpyHostToDevice));
    }
    q.idle = true;
    s->codes_host_valid = false;
    return Q3_OK;
}

extern "C" q3_status q3_batcher_create(q3_model* m, int slots, int frame_budget, int prompt_budget, q3_batcher** out) {
    if (!m || !out) return set_err(Q3_INVALID_ARG, "q3_batcher_create: null argument");
    if (!m->finalized) return set_err(Q3_INVALID_ARG, "model not finalized");
    if (slots < 1 || slots > Q3_MAX_BATCH) return set_err(Q3_UNSUPPORTED, "q3_batcher_create: %d rows unsupported (1..%d)", slots, Q3_MAX_BATCH);
    if (frame_budget < 1 || prompt_budget < 0) return set_err(Q3_INVALID_ARG, "q3_batcher_create: frame_budget must be >= 1, prompt_budget >= 0");
    {   // the session the first step opens: max_seq = max(prompt_budget, 16) + frame_budget + 1 positions of the RoPE table
        const long need = (long)(prompt_budget > 16 ? prompt_budget : 16) + frame_budget + 1;
        if (need > m->rope_len) return set_err(Q3_KV_OVERFLOW, "q3_batcher_create: prompt_budget + frame_budget = %ld positions exceed the RoPE table (%d)", need, m->rope_len);
    }
    std::unique_ptr<q3_batcher> b(new q3_batcher());
    b->m = m; b->slots = slots; b->frame_budget = frame_budget; b->prompt_budget = prompt_budget;
    b->owner.assign(slots, -1); b->commit.assign(slots, 0);
    *out = b.release();
    return Q3_OK;
}
extern "C" void q3_batcher_free(q3_batcher* b) {
    if (!b) return;
    stage_drop(b);
    if (b->s) q3_session_free(b->s);
    delete b;
}
extern "C" q3_status q3_batcher_submit(q3_batcher* b, const q3_request* req, int want_pcm, int64_t* ticket) {
    if (!b || !req || !ticket) return set_err(Q3_INVALID_ARG, "q3_batcher_submit: null argument");
    if (req->opts.max_length < 1 || req->opts.max_length > b->frame_budget)
        return set_err(Q3_UNSUPPORTED, "q3_batcher_submit: max_length %d outside 1..%d (the batcher's frame budget)", req->opts.max_length, b->frame_budget);
    if (req->n_text < 0 || req->n_instruct < 0 || req->n_ref < 0 || req->n_ref_text < 0) return set_err(Q3_INVALID_ARG, "q3_batcher_submit: negative length");
    std::unique_ptr<BatTicket> t(new BatTicket());
    t->req.own(*req, b->m->cfg.hidden); t->want_pcm = want_pcm != 0;
    const int64_t id = b->next_id++;
    b->t[id] = std::move(t);
    b->queue.push_back(id);
    *ticket = id;
    return Q3_OK;
}

static void bat_fail(BatTicket& t, q3_status st) { t.state = Q3_TICKET_FAILED; t.st = st; t.err = q3_last_error(); t.row = -1; }

// the row's sequence has ended: keep its codes (and PCM), free the row
static q3_status bat_collect(q3_batcher* b, int row) {
    BatTicket& t = *b->t[b->owner[row]];
    int n = 0;
    Q3C(q3_session_codes(b->s, row, nullptr, 0, &n));
    t.codes.resize((size_t)n * 16); t.n_frames = n;
    if (n > 0) Q3C(q3_session_codes(b->s, row, t.codes.data(), n, &n));
    if (t.want_pcm && n > 0) {
        size_t ns = 0;
        t.pcm.resize((size_t)n * samples_per_frame(b->m->cfg));
        Q3C(q3_session_decode(b->s, row, 0, n, t.pcm.data(), t.pcm.size(), &ns));
        t.pcm.resize(ns);
    }
    t.state = Q3_TICKET_DONE; t.row = -1;
    b->owner[row] = -1; b->commit[row] = 0;
    // the device freezes a row at its frame limit, not at EOS: idle it now so that it stops advancing — and taking pages — while
    // the queue is empty or waits for room; its pages but one go back to the pool
    return session_idle_row(b->s, row);
}

extern "C" q3_status q3_batcher_step(q3_batcher* b, int n_frames, int use_graph, int* n_running, int* n_queued, int* n_finished) {
    if (!b) return set_err(Q3_INVALID_ARG, "q3_batcher_step: null batcher");
    if (n_frames < 1) return set_err(Q3_INVALID_ARG, "q3_batcher_step: n_frames must be >= 1");
    int finished = 0;
    // Open the session on `slots` idle rows: copies of a one-token CustomVoice prompt with a one-frame limit (ten prefill
    // positions per row — opening on the first request itself would prefill, and size every row's KV extent for, `slots`
    // copies of what may be a 4k-token prompt), frozen before the first frame. Every request, the first included, then enters
    // through q3_session_replace, so prompt kinds mix freely.
    if (!b->s && !b->queue.empty()) {
        // The idle rows are built from fixed, known-valid values — never from a queued request: a malformed first request must
        // fail alone, at its own q3_session_replace below, not wedge the queue by failing the session every step.
        const q3_request& first = b->t[b->queue.front()]->req.r;
        b->chunk_frames = first.opts.chunk_frames >= 1 ? first.opts.chunk_frames : 10;       // the one option a session shares
        const q3_request d = idle_request(b->chunk_frames);
        std::vector<q3_request> reqs((size_t)b->slots, d);
        q3_session* s = nullptr;
        q3_status st = session_create(b->m, reqs.data(), b->slots, b->frame_budget, b->prompt_budget > 16 ? b->prompt_budget : 16, &s);
        if (st == Q3_OK) st = q3_session_prefill(s);
        if (st != Q3_OK) {
            // nothing a request could have caused (the budgets were checked at q3_batcher_create): a device failure. The head
            // ticket takes the error so that a serving loop sees it on a ticket and the queue moves on.
            if (s) q3_session_free(s);
            const int64_t id = b->queue.front(); b->queue.erase(b->queue.begin());
            bat_fail(*b->t[id], st);
            if (n_running) *n_running = 0; if (n_queued) *n_queued = (int)b->queue.size(); if (n_finished) *n_finished = 1;
            return st;
        }
        b->s = s;
        for (int r = 0; r < b->slots; ++r) Q3C(session_idle_row(s, r));
    }
    if (!b->s) { if (n_running) *n_running = 0; if (n_queued) *n_queued = 0; if (n_finished) *n_finished = finished; return Q3_OK; }
    // Admission under a page limit (q3_model_kv_pool_limit): a request enters a row only if its WORST CASE (prompt + max_length
    // positions; row_worst_units) fits beside what the running rows may still come to hold and what everything else on the model
    // holds now — so a page shortage shows up here, as a request that waits in the queue (rows are running: room will come) or
    // fails on its ticket (it cannot fit even alone), never in the middle of a generation where it would stop every row.
    auto held_units = [&](int r) -> long { return (long)b->s->kv_rows[(size_t)r].size() * (b->s->kv_in_bf16 ? 1 : 2); };
    auto admit = [&](const q3_request& rq, long* units_out, bool* wait) -> bool {
        *wait = false; *units_out = 0;
        if (!b->s->paged) return true;
        int S = 0, lim = 0; request_shape(rq, &S, &lim);
        const long units = row_worst_units(S, lim, b->s->kv_bf16);
        *units_out = units;
        long mine = 0, claimed = 0; int running = 0;
        for (int r = 0; r < b->slots; ++r) {
            const long h = held_units(r);
            mine += h;
            if (b->owner[r] >= 0) { claimed += std::max(b->commit[r], h); running++; } else claimed += h;
        }
        std::lock_guard<std::mutex> g(b->m->kv_budget.mu);
        if (b->m->kv_budget.limit <= 0) return true;
        const long others = b->m->kv_budget.used - mine;
        if (others + claimed + units <= b->m->kv_budget.limit) return true;
        *wait = running > 0;
        return false;
    };
    auto fill = [&]() -> q3_status {               // free rows <- waiting requests
        for (int r = 0; r < b->slots && !b->queue.empty(); ++r) {
            if (b->owner[r] >= 0) continue;
            while (!b->queue.empty()) {
                const int64_t id = b->queue.front();
                BatTicket& t = *b->t[id];
                long units = 0; bool wait = false;
                if (!admit(t.req.r, &units, &wait)) {
                    if (wait) return Q3_OK;          // FIFO: the head of the queue waits for running rows to end
                    b->queue.erase(b->queue.begin());
                    int S = 0, lim = 0; request_shape(t.req.r, &S, &lim);
                    bat_fail(t, set_err(Q3_KV_OVERFLOW, "KV page pool exhausted: the request's %d prompt positions + %d frames need %ld page(s) (f32 equivalents), more than the pool's limit leaves",
                                        S, lim, (units + 1) / 2));
                    finished++; continue;
                }
                b->queue.erase(b->queue.begin());
                t.req.r.opts.chunk_frames = b->chunk_frames;        // the one option a session shares
                q3_status st;
                if (b->stage.id == id) {
                    // prefilled ahead on the worker's stream: wait for it (normally long done), then only the state copy stands
                    // between two frames of the live rows
                    stage_join(b);
                    st = b->stage.st;
                    if (st != Q3_OK) set_err(st, "%s", b->stage.err.c_str());
                    else {
                        st = sync_frames(b->s) == hipSuccess ? Q3_OK : Q3_HIP_ERROR;      // no frame of the host session in flight while its row changes
                        if (st == Q3_OK) st = transplant_row(b->s, r, b->stage.side, 0, b->stage.limit);
                    }
                    stage_drop(b);
                } else
                    st = q3_session_replace(b->s, r, &t.req.r);
                if (st != Q3_OK) { bat_fail(t, st); finished++; continue; }     // does not fit: the ticket carries the reason; try the next one
                t.state = Q3_TICKET_RUNNING; t.row = r; b->owner[r] = id; b->commit[r] = units;
                break;
            }
        }
        return Q3_OK;
    };
    // the head of the queue starts its prefill on the worker (see q3_batcher::Stage); called with frames about to be queued
    auto stage_begin = [&]() {
        static const bool off = getenv("Q3_BAT_NO_STAGE") != nullptr;
        if (off || b->stage.id >= 0 || b->queue.empty() || b->s->debug || b->s->profile) return;
        // not while this thread may still CAPTURE the host session's frame (the first graph step): the worker's allocations and
        // null-stream zero-fills invalidate a capture in progress on this HIP runtime, thread-local capture mode or not
        if (use_graph ? b->s->graph == nullptr : false) return;
        { std::lock_guard<std::mutex> g(b->m->kv_budget.mu); if (b->m->kv_budget.limit > 0) return; }
        bool any_free = false;
        for (int r = 0; r < b->slots; ++r) any_free = any_free || b->owner[r] < 0;
        if (any_free) return;                        // a free row takes the head at once (fill): nothing to run ahead of
        const int64_t id = b->queue.front();
        BatTicket& t = *b->t[id];
        q3_request rq = t.req.r;                     // (arrays owned by the ticket, which lives until it is fetched)
        rq.opts.chunk_frames = b->chunk_frames;
        const int limit_req = rq.opts.max_length;
        if (limit_req < 1 || limit_req > b->s->max_frames) return;      // the synchronous path reports it on the ticket
        rq.opts.max_length = b->s->max_frames;       // the side session draws the row's PCG stream with the host session's stride
        b->stage.id = id; b->stage.st = Q3_OK; b->stage.err.clear(); b->stage.side = nullptr; b->stage.limit = 0;
        q3_batcher* bp = b;
        b->stage.thr = std::thread([bp, rq, limit_req]() {
            q3_batcher::Stage& g = bp->stage;
            q3_session* side = nullptr;
            q3_status st = hipSetDevice(bp->m->device) == hipSuccess ? Q3_OK : set_err(Q3_HIP_ERROR, "hipSetDevice");
            if (st == Q3_OK) st = session_create(bp->m, &rq, 1, 0, 0, &side);          // a stream of its own: runs beside the frames
            if (st == Q3_OK) { side->kv_bf16 = bp->s->kv_bf16; st = transplant_check(bp->s, side, 0, limit_req, &g.limit); }
            if (st == Q3_OK) st = q3_session_prefill(side);                            // ends with a synchronisation of that stream
            g.side = side; g.st = st;
            if (st != Q3_OK) g.err = q3_last_error();
        });
    };
    // Run in pieces that end where the next row reaches its frame limit: that row is collected and refilled at once instead of
    // idling to the end of the step (a row that ends on EOS is noticed at q3_session_generate's 32-frame check or at the
    // end of the piece)
    for (int left = n_frames; left > 0;) {
        Q3C(fill());
        int piece = left, busy = 0;
        for (int r = 0; r < b->slots; ++r) {
            if (b->owner[r] < 0) continue;
            const SeqInfo& q = b->s->seq[r];
            const int rem = q.limit - (b->s->frames_run - q.start_run);
            if (rem > 0) { busy++; if (rem < piece) piece = rem; }
        }
        if (busy > 0) {
            stage_begin();
            const q3_status gst = q3_session_generate(b->s, piece, use_graph);
            if (gst == Q3_KV_OVERFLOW && b->s->kv_overflow_row >= 0 && b->owner[b->s->kv_overflow_row] >= 0) {
                // (only reachable when something outside this batcher took the pages its admission counted on) nothing ran: the
                // row that needs the page fails alone and is frozen; the others go on
                const int row = b->s->kv_overflow_row;
                bat_fail(*b->t[b->owner[row]], gst);
                b->owner[row] = -1; b->commit[row] = 0;
                Q3C(session_idle_row(b->s, row));
                finished++;
                continue;
            }
            Q3C(gst);
            left -= piece;
        }
        int collected = 0;
        for (int r = 0; r < b->slots; ++r) {
            if (b->owner[r] < 0) continue;
            {   // a row without a live EOS id ends exactly at its frame limit, which the host knows: no device read-back (a
                // synchronisation and nine blocking copies per step) while no row can have ended
                const SeqInfo& q = b->s->seq[r];
                if (q.req.opts.eos_token_id < 0 && b->s->frames_run - q.start_run < q.limit) continue;
            }
            int n = 0, done = 0;
            Q3C(q3_session_frames(b->s, r, &n, &done));
            if (done) { Q3C(bat_collect(b, r)); finished++; collected++; }
        }
        if (busy == 0 && collected == 0) break;          // nothing runs and nothing is waiting for a row
    }
    Q3C(fill());                                   // the next step starts with full rows
    int running = 0;
    for (int r = 0; r < b->slots; ++r) running += b->owner[r] >= 0 ? 1 : 0;
    if (n_running) *n_running = running;
    if (n_queued) *n_queued = (int)b->queue.size();
    if (n_finished) *n_finished = finished;
    return Q3_OK;
}

extern "C" q3_status q3_batcher_poll(q3_batcher* b, int64_t ticket, int* state, int* n_frames, size_t* n_samples) {
    if (!b) return set_err(Q3_INVALID_ARG, "q3_batcher_poll: null batcher");
    auto it = b->t.find(ticket);
    if (it == b->t.end()) return set_err(Q3_INVALID_ARG, "q3_batcher_poll: unknown ticket %lld", (long long)ticket);
    const BatTicket& t = *it->second;
    if (state) *state = t.state;
    int nf = t.n_frames;
    if (t.state == Q3_TICKET_RUNNING && b->s && t.row >= 0) {       // frames run so far (an EOS inside them is only looked at when the row is collected)
        const SeqInfo& q = b->s->seq[t.row];
        nf = b->s->frames_run - q.start_run; if (nf > q.limit) nf = q.limit; if (nf < 0) nf = 0;
    }
    if (n_frames) *n_frames = nf;
    if (n_samples) *n_samples = t.pcm.size();
    return Q3_OK;
}

extern "C" q3_status q3_batcher_fetch(q3_batcher* b, int64_t ticket, uint32_t* codes_host, int cap_frames, float* pcm_host, size_t cap_samples) {
    if (!b) return set_err(Q3_INVALID_ARG, "q3_batcher_fetch: null batcher");
    auto it = b->t.find(ticket);
    if (it == b->t.end()) return set_err(Q3_INVALID_ARG, "q3_batcher_fetch: unknown ticket %lld", (long long)ticket);
    BatTicket& t = *it->second;
    if (t.state == Q3_TICKET_FAILED) {
        const q3_status st = t.st; const std::string err = t.err;
        b->t.erase(it);
        return set_err(st, "%s", err.c_str());
    }
    if (t.state != Q3_TICKET_DONE) return set_err(Q3_INVALID_ARG, "q3_batcher_fetch: ticket %lld has not finished", (long long)ticket);
    if (codes_host) {
        if (cap_frames < t.n_frames) return set_err(Q3_INVALID_ARG, "codes buffer too small (%d < %d frames)", cap_frames, t.n_frames);
        memcpy(codes_host, t.codes.data(), t.codes.size() * 4);
    }
    if (pcm_host) {
        if (cap_samples < t.pcm.size()) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        memcpy(pcm_host, t.pcm.data(), t.pcm.size() * 4);
    }
    b->t.erase(it);
    return Q3_OK;
}

extern "C" q3_status q3_session_next_chunk_row(q3_session* s, int b, float* pcm_host, size_t cap, size_t* n_samples, int* done) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    if (!s->prefilled) Q3C(q3_session_prefill(s));
    Q3C(refresh_codes(s));
    SeqInfo& q = s->seq[b];
    const int chunk = s->opts.chunk_frames > 0 ? s->opts.chunk_frames : 10;
    while (!q.done && q.n_frames - q.stream_pos < chunk && session_remaining(s) > 0) {
        Q3C(q3_session_generate(s, chunk - (q.n_frames - q.stream_pos), 1));
        Q3C(refresh_codes(s));
    }
    int avail = q.n_frames - q.stream_pos;
    if (avail > chunk) avail = chunk;
    if (avail <= 0) { if (n_samples) *n_samples = 0; if (done) *done = 1; return Q3_OK; }
    const int spf = samples_per_frame(s->m->cfg);
    if (s->stream_mode == 1 && !q.icl) {
        // continuous mode (q3_session_set_stream_mode): left context re-run, sample-exact with the whole-utterance decode
        const int a0 = q.stream_pos, e = q.stream_pos + avail, c0 = a0 > CODEC_CTX_FRAMES ? a0 - CODEC_CTX_FRAMES : 0;
        Q3C(codec_reserve(s->m, s->cws, chunk + CODEC_CTX_FRAMES, s->max_frames));
        HIPC(hipMemcpyAsync(s->cws.frames, s->codes + (size_t)b * s->max_frames * 16, (size_t)e * 16 * 4, hipMemcpyDeviceToDevice, s->stream));
        Q3C(codec_decode_dev(s->m, s->cws, e, s->stream, nullptr, c0));
        HIPC(sync_frames(s));
        if (n_samples) *n_samples = (size_t)avail * spf;
        if (pcm_host) {
            if (cap < (size_t)avail * spf) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            HIPC(hipMemcpy(pcm_host, s->cws.pcm + (size_t)(a0 - c0) * spf, (size_t)avail * spf * 4, hipMemcpyDeviceToHost));
        }
    } else {
        Q3C(decode_range_on(s, b, q.stream_pos, q.stream_pos + avail, s->stream, pcm_host, cap, n_samples));
    }
    q.stream_pos += avail;
    if (done) *done = (q.done && q.stream_pos >= q.n_frames) ? 1 : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_frames(q3_session* s, int b, int* n_frames, int* done) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    Q3C(refresh_codes(s));
    if (n_frames) *n_frames = s->seq[b].n_frames;
    if (done) *done = s->seq[b].done ? 1 : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_codes(q3_session* s, int b, uint32_t* codes_host, int cap_frames, int* n_frames) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    Q3C(refresh_codes(s));
    const int n = s->seq[b].n_frames;
    if (n_frames) *n_frames = n;
    if (codes_host) {
        if (cap_frames < n) return set_err(Q3_INVALID_ARG, "codes buffer too small (%d < %d frames)", cap_frames, n);
        memcpy(codes_host, &s->codes_host[(size_t)b * s->max_frames * 16], (size_t)n * 16 * 4);
    }
    return Q3_OK;
}

// context-free decode of frames [f0, f1) of sequence b on `st` (the reference's per-chunk decode, lib.rs:1755-1758)
static q3_status decode_range_on(q3_session* s, int b, int f0, int f1, hipStream_t st, float* pcm_host, size_t cap, size_t* n_samples) {
    const int T = f1 - f0, spf = samples_per_frame(s->m->cfg);
    if (n_samples) *n_samples = (size_t)T * spf;
    if (T == 0) return Q3_OK;
    Q3C(codec_reserve(s->m, s->cws, T));
    HIPC(hipMemcpyAsync(s->cws.frames, s->codes + ((size_t)b * s->max_frames + f0) * 16, (size_t)T * 16 * 4, hipMemcpyDeviceToDevice, st));
    Q3C(codec_decode_dev(s->m, s->cws, T, st, nullptr));
    HIPC(hipStreamSynchronize(st));
    if (pcm_host) {
        if (cap < (size_t)T * spf) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        HIPC(hipMemcpy(pcm_host, s->cws.pcm, (size_t)T * spf * 4, hipMemcpyDeviceToHost));
    }
    return Q3_OK;
}

extern "C" q3_status q3_session_decode(q3_session* s, int b, int f0, int f1, float* pcm_host, size_t cap, size_t* n_samples) {
    if (!s || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad sequence index");
    HIPC(hipSetDevice(s->m->device));
    Q3C(refresh_codes(s));
    if (f0 < 0 || f1 < f0 || f1 > s->seq[b].n_frames) return set_err(Q3_INVALID_ARG, "bad frame range [%d,%d) of %d", f0, f1, s->seq[b].n_frames);
    const int T = f1 - f0, spf = samples_per_frame(s->m->cfg);
    const SeqInfo& q = s->seq[b];
    if (!q.ref_codes.empty() && s->prefilled && f0 == 0 && f1 == q.n_frames) {
        // ICL full-utterance decode (lib.rs:1022-1041): decode [ref_frames ; generated], then cut the first
        // ref_len * samples / total_frames samples
        const int n_ref = (int)(q.ref_codes.size() / 16), total = n_ref + T;
        const size_t all = (size_t)total * spf, cut = (size_t)n_ref * all / (size_t)(total > 0 ? total : 1);
        if (n_samples) *n_samples = all - cut;
        Q3C(codec_reserve(s->m, s->cws, total));
        // the row's reference frames from the host copy (a row swapped in by q3_session_replace brings its own)
        HIPC(hipMemcpyAsync(s->cws.frames, q.ref_codes.data(), (size_t)n_ref * 16 * 4, hipMemcpyHostToDevice, s->stream));
        if (T > 0) HIPC(hipMemcpyAsync(s->cws.frames + (size_t)n_ref * 16, s->codes + (size_t)b * s->max_frames * 16, (size_t)T * 16 * 4, hipMemcpyDeviceToDevice, s->stream));
        Q3C(codec_decode_dev(s->m, s->cws, total, s->stream, nullptr));
        HIPC(sync_frames(s));
        if (pcm_host) {
            if (cap < all - cut) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            HIPC(hipMemcpy(pcm_host, s->cws.pcm + cut, (all - cut) * 4, hipMemcpyDeviceToHost));
        }
        return Q3_OK;
    }
    return decode_range_on(s, b, f0, f1, s->stream, pcm_host, cap, n_samples);
}

// Enqueue (no host sync) the vocoder for frames [a, e) of sequence b on the decode stream; PCM lands in s->pcm_all.
static q3_status seg_decode_enqueue(q3_session* s, int b, int a, int e) {
    const int spf = samples_per_frame(s->m->cfg);
    const int c0 = a > CODEC_CTX_FRAMES ? a - CODEC_CTX_FRAMES : 0;
    HIPC(hipMemcpyAsync(s->seg_ws.frames, s->codes + (size_t)b * s->max_frames * 16, (size_t)e * 16 * 4, hipMemcpyDeviceToDevice, s->dec_stream));
    Q3C(codec_decode_dev(s->m, s->seg_ws, e, s->dec_stream, nullptr, c0));
    HIPC(hipMemcpyAsync(s->pcm_all + ((size_t)b * s->max_frames + a) * spf, s->seg_ws.pcm + (size_t)(a - c0) * spf,
                        (size_t)(e - a) * spf * 4, hipMemcpyDeviceToDevice, s->dec_stream));
    return Q3_OK;
}

// synthesize_with_timing for the whole batch. With Q3_DECODE_OVERLAP=1 (and no ICL sequence) the vocoder does not
// wait for the last frame: every Q3_DECODE_SEG (default 128) generated frames a helper thread enqueues the segment's
// decode (exact: CODEC_CTX_FRAMES of left context re-run, see codec_decode_dev) on a second stream beside the frame
// loop. OFF by default: measured on MI355X (1.7B, 8 x 640 frames) the frame loop slows from 2785 to 3308 ms while the
// decode tail only shrinks from 639 to 289 ms (3444 -> 3620 ms per step) — the frame loop's workgroups need a whole
// CU's registers, so vocoder waves already resident on a CU block them, and neither stream priorities nor a CU mask
// on the decode stream (32 / 64 / 96 / 128 CUs: 5624 / 3949 / 3757 / 3607 ms) recover it.
extern "C" q3_status q3_session_run(q3_session* s, int use_graph, float** pcm_host, const size_t* cap, size_t* n_samples, q3_timing* timing) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    s->precapture = use_graph != 0;
    Q3C(q3_session_prefill(s));
    const auto t1 = clk::now();
    const int overlap_env = [] { const char* e = getenv("Q3_DECODE_OVERLAP"); return e ? atoi(e) : 0; }();     // read per call: opt-in, tests set it per test
    static const int seg_env = [] { const char* e = getenv("Q3_DECODE_SEG"); const int v = e ? atoi(e) : 128; return v < 16 ? 16 : v; }();
    bool overlap = overlap_env != 0 && !s->debug && !s->profile && s->max_frames > seg_env;
    for (auto& q : s->seq) if (!q.ref_codes.empty()) overlap = false;
    const int spf = samples_per_frame(s->m->cfg);
    if (!overlap) {
        Q3C(q3_session_generate(s, s->max_frames, use_graph));
        Q3C(refresh_codes(s));
        const auto t2 = clk::now();
        int total = 0;
        // Four utterances are vocoded at a time, each on its own stream and workspace: most of a decode saturates the
        // chip, but its front (the pre-transformer's ~90 launches on 10-160 workgroups, ≈4 of 28 ms) is latency-bound
        // and fills in beside the other utterance's convolutions. Q3_DECODE_PAIRS=n: n at a time (1 = serial; A/B aid).
        static const int conc = [] { const char* e = getenv("Q3_DECODE_PAIRS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
        bool any_icl = false;
        for (auto& q : s->seq) any_icl = any_icl || !q.ref_codes.empty();
        if (conc > 1 && s->B > 1 && !any_icl) {
            while ((int)s->par_ws.size() < conc - 1) {
                s->par_ws.emplace_back();
                hipStream_t st = nullptr;
                HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                s->par_streams.push_back(st);
            }
            auto ws_of = [&](int k) -> CodecWS& { return k == 0 ? s->cws : s->par_ws[(size_t)k - 1]; };
            auto st_of = [&](int k) { return k == 0 ? s->stream : s->par_streams[(size_t)k - 1]; };
            if (pcm_host)
                for (int b = 0; b < s->B; ++b)
                    if (pcm_host[b] && s->seq[b].n_frames && (!cap || cap[b] < (size_t)s->seq[b].n_frames * spf)) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            for (int b0 = 0; b0 < s->B; b0 += conc) {
                const int nb = (s->B - b0) < conc ? (s->B - b0) : conc;
                for (int k = 0; k < nb; ++k) {
                    const int b = b0 + k, T = s->seq[b].n_frames;
                    if (n_samples) n_samples[b] = (size_t)T * spf;
                    total += T;
                    if (T == 0) continue;
                    Q3C(codec_reserve(s->m, ws_of(k), T));
                    HIPC(hipMemcpyAsync(ws_of(k).frames, s->codes + (size_t)b * s->max_frames * 16, (size_t)T * 16 * 4, hipMemcpyDeviceToDevice, st_of(k)));
                    Q3C(codec_decode_dev(s->m, ws_of(k), T, st_of(k), nullptr));
                    // the samples leave for the host on the utterance's own stream, beside the other utterances' decodes
                    // (synthesize returns host samples, lib.rs:718-784); pinned caller buffers make this a true async copy
                    if (pcm_host && pcm_host[b])
                        HIPC(hipMemcpyAsync(pcm_host[b], ws_of(k).pcm, (size_t)T * spf * 4, hipMemcpyDeviceToHost, st_of(k)));
                }
                for (int k = 0; k < nb; ++k) HIPC(hipStreamSynchronize(st_of(k)));
            }
        } else
        for (int b = 0; b < s->B; ++b) {
            size_t n = 0;
            Q3C(q3_session_decode(s, b, 0, s->seq[b].n_frames, pcm_host ? pcm_host[b] : nullptr, cap ? cap[b] : 0, &n));
            if (n_samples) n_samples[b] = n;
            total += s->seq[b].n_frames;
        }
        const auto t3 = clk::now();
        if (timing) { timing->prefill_ms = ms(t0, t1); timing->generation_ms = ms(t1, t2); timing->decode_ms = ms(t2, t3); timing->generation_frames = total; }
        return Q3_OK;
    }
    HIPC(hipSetDevice(s->m->device));
    if (!s->dec_stream) {
        static const int cus = [] { const char* e = getenv("Q3_DECODE_CUS"); return e ? atoi(e) : 0; }();   // tuning aid
        int least = 0, greatest = 0;
        HIPC(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (cus > 0) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < cus && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
            HIPC(hipExtStreamCreateWithCUMask(&s->dec_stream, 8, mask));
        } else {
            HIPC(hipStreamCreateWithPriority(&s->dec_stream, hipStreamNonBlocking, least));
        }
    }
    Q3C(codec_reserve(s->m, s->seg_ws, seg_env + CODEC_CTX_FRAMES, s->max_frames));
    if (s->pcm_all_floats < (size_t)s->B * s->max_frames * spf) {
        if (s->pcm_all) dev_free(s->pcm_all);
        s->pcm_all_floats = (size_t)s->B * s->max_frames * spf;
        HIPC(dev_malloc((void**)&s->pcm_all, s->pcm_all_floats * 4));
    }
    std::vector<int> dec_pos((size_t)s->B, 0);
    std::thread worker; q3_status wst = Q3_OK; std::string werr;
    auto join = [&]() -> q3_status {
        if (worker.joinable()) worker.join();
        if (wst != Q3_OK) return set_err(wst, "%s", werr.c_str());
        return Q3_OK;
    };
    struct Job { int b, a, e; };
    auto dispatch = [&](bool final_pass) -> q3_status {
        std::vector<Job> jobs;
        for (int b = 0; b < s->B; ++b) {
            const SeqInfo& q = s->seq[b];
            const int e = (q.done || final_pass) ? q.n_frames : (q.n_frames < s->frames_run ? q.n_frames : s->frames_run);
            if (e > dec_pos[(size_t)b] && (final_pass || q.done || e - dec_pos[(size_t)b] >= 16)) {
                // keep every call within the workspace: at most seg_env new frames per job
                for (int a = dec_pos[(size_t)b]; a < e; a += seg_env) jobs.push_back({b, a, a + seg_env < e ? a + seg_env : e});
                dec_pos[(size_t)b] = e;
            }
        }
        if (jobs.empty()) return Q3_OK;
        Q3C(join());
        worker = std::thread([s, jobs, &wst, &werr]() {
            if (hipSetDevice(s->m->device) != hipSuccess) { wst = Q3_HIP_ERROR; werr = "hipSetDevice failed in the decode thread"; return; }
            for (const Job& j : jobs) {
                const q3_status st = seg_decode_enqueue(s, j.b, j.a, j.e);
                if (st != Q3_OK) { wst = st; werr = q3_last_error(); return; }
            }
        });
        return Q3_OK;
    };
    q3_status st = Q3_OK;
    while (st == Q3_OK && session_remaining(s) > 0 && !all_done(s)) {
        st = q3_session_generate(s, seg_env, use_graph);
        if (st == Q3_OK) st = refresh_codes(s);
        if (st == Q3_OK && session_remaining(s) > 0 && !all_done(s)) st = dispatch(false);
    }
    const auto t2 = clk::now();
    if (st == Q3_OK) st = dispatch(true);
    { const q3_status js = join(); if (st == Q3_OK) st = js; }
    if (st != Q3_OK) { hipStreamSynchronize(s->dec_stream); return st; }
    HIPC(hipStreamSynchronize(s->dec_stream));
    int total = 0;
    for (int b = 0; b < s->B; ++b) {
        const size_t n = (size_t)s->seq[b].n_frames * spf;
        if (n_samples) n_samples[b] = n;
        if (pcm_host && pcm_host[b] && n) {
            if (!cap || cap[b] < n) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
            HIPC(hipMemcpy(pcm_host[b], s->pcm_all + (size_t)b * s->max_frames * spf, n * 4, hipMemcpyDeviceToHost));
        }
        total += s->seq[b].n_frames;
    }
    const auto t3 = clk::now();
    if (timing) { timing->prefill_ms = ms(t0, t1); timing->generation_ms = ms(t1, t2); timing->decode_ms = ms(t2, t3); timing->generation_frames = total; }
    return Q3_OK;
}

extern "C" q3_status q3_session_set_stream_mode(q3_session* s, int mode) {
    if (!s || (mode != 0 && mode != 1)) return set_err(Q3_INVALID_ARG, "q3_session_set_stream_mode: mode must be 0 (context-free) or 1 (continuous)");
    s->stream_mode = mode;
    return Q3_OK;
}

extern "C" q3_status q3_session_next_chunk(q3_session* s, float* pcm_host, size_t cap, size_t* n_samples, int* done) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (s->B != 1) return set_err(Q3_UNSUPPORTED, "streaming sessions are batch 1 (StreamingSession, lib.rs:1484)");
    if (!s->prefilled) { s->precapture = true; Q3C(q3_session_prefill(s)); }
    Q3C(refresh_codes(s));
    SeqInfo& q = s->seq[0];
    const int chunk = s->opts.chunk_frames > 0 ? s->opts.chunk_frames : 10;
    // generate until chunk_frames frames are buffered or the sequence ends (lib.rs:1663-1748)
    while (!q.done && q.n_frames - s->stream_pos < chunk && session_remaining(s) > 0) {
        int need = chunk - (q.n_frames - s->stream_pos);
        Q3C(q3_session_generate(s, need, 1));
        Q3C(refresh_codes(s));
    }
    int avail = q.n_frames - s->stream_pos;
    if (avail > chunk) avail = chunk;
    if (avail <= 0) { if (n_samples) *n_samples = 0; if (done) *done = 1; return Q3_OK; }
    const bool chunk_done = q.done && s->stream_pos + avail >= q.n_frames;
    // Read-ahead: the frames of the NEXT chunk are enqueued (graph replays, no host wait) while this chunk is vocoded
    // and handed over, so the frame loop keeps going while the host copies out, returns and plays the chunk. First
    // chunk: its vocoder is enqueued first and the replays behind it on the same stream (time-to-first-audio is what it
    // was; the host waits on an event, not on the stream). Later chunks: the replays go first and the chunk's vocoder
    // runs beside them on its own stream — a chunk then costs max(generation, decode) instead of their sum (streaming
    // RTF 0.046 -> 0.042 on the 1.7B model). After EOS the device-side done flag turns extra replays into no-ops.
    // Q3_STREAM_NO_AHEAD=1 restores the serial schedule (A/B aid).
    static const bool no_ahead = getenv("Q3_STREAM_NO_AHEAD") != nullptr;
    int ahead = 0;
    if (!no_ahead && !q.done && (s->aql || s->graph_exec) && !s->aql_failed && !s->debug && !s->profile) {
        ahead = chunk - (q.n_frames - (s->stream_pos + avail));
        if (ahead > session_remaining(s)) ahead = session_remaining(s);
        if (ahead < 0) ahead = 0;
    }
    const bool first = s->stream_pos == 0;
    auto launch_ahead = [&]() -> q3_status {
        Q3C(kv_reserve_frames(s, ahead));                        // (its table updates ride s->stream)
        if (s->aql) HIPC(hipStreamSynchronize(s->stream));       // the own queue is not ordered with the stream
        return frames_enqueue(s, ahead);                         // the next call re-reads codes / EOS state after sync_frames
    };
    hipStream_t dst = s->stream;
    if (ahead > 0 && !first) {
        if (!s->dec_stream) HIPC(hipStreamCreateWithFlags(&s->dec_stream, hipStreamNonBlocking));
        Q3C(launch_ahead());
        dst = s->dec_stream;
    }
    // enqueue this chunk's vocoder on dst
    const int spf = samples_per_frame(s->m->cfg);
    const float* src = nullptr;
    if (s->stream_mode == 1 && !q.icl) {
        // continuous mode: the front runs over frames [0, end), the convolutional stack over [pos - CTX, end); the chunk's
        // samples are identical to the same frames of a whole-utterance decode (codec_decode_dev)
        const int a0 = s->stream_pos, e = s->stream_pos + avail;
        const int c0 = a0 > CODEC_CTX_FRAMES ? a0 - CODEC_CTX_FRAMES : 0;
        Q3C(codec_reserve(s->m, s->cws, chunk + CODEC_CTX_FRAMES, s->max_frames));
        HIPC(hipMemcpyAsync(s->cws.frames, s->codes, (size_t)e * 16 * 4, hipMemcpyDeviceToDevice, dst));
        Q3C(codec_decode_dev(s->m, s->cws, e, dst, nullptr, c0));
        src = s->cws.pcm + (size_t)(a0 - c0) * spf;
    } else {
        // the reference's schedule: the chunk decoded as an independent utterance (lib.rs:1755-1758)
        Q3C(codec_reserve(s->m, s->cws, avail));
        HIPC(hipMemcpyAsync(s->cws.frames, s->codes + (size_t)s->stream_pos * 16, (size_t)avail * 16 * 4, hipMemcpyDeviceToDevice, dst));
        Q3C(codec_decode_dev(s->m, s->cws, avail, dst, nullptr));
        src = s->cws.pcm;
    }
    if (ahead > 0 && first) {
        if (!s->dec_ev) HIPC(hipEventCreateWithFlags(&s->dec_ev, hipEventDisableTiming));
        HIPC(hipEventRecord(s->dec_ev, s->stream));
        Q3C(launch_ahead());
        HIPC(hipEventSynchronize(s->dec_ev));
    } else {
        HIPC(hipStreamSynchronize(dst));
    }
    if (n_samples) *n_samples = (size_t)avail * spf;
    if (pcm_host) {
        if (cap < (size_t)avail * spf) return set_err(Q3_INVALID_ARG, "pcm buffer too small");
        HIPC(hipMemcpy(pcm_host, src, (size_t)avail * spf * 4, hipMemcpyDeviceToHost));
    }
    s->stream_pos += avail;
    if (done) *done = chunk_done ? 1 : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_get(q3_session* s, int what, int b, void* out, size_t bytes) {
    if (!s || !out || b < 0 || b >= s->B) return set_err(Q3_INVALID_ARG, "bad argument");
    const q3_config& c = s->m->cfg;
    HIPC(hipSetDevice(s->m->device));
    HIPC(sync_frames(s));
    const void* src = nullptr; size_t need = 0;
    switch (what) {
        case Q3_GET_PREFILL_EMBEDS: src = s->embeds + (size_t)b * s->prefill_len * c.hidden; need = (size_t)s->prefill_len * c.hidden * 4; break;
        case Q3_GET_LAST_HIDDEN: src = s->LASTH + (size_t)b * c.hidden; need = (size_t)c.hidden * 4; break;
        case Q3_GET_LOGITS: src = s->LOGITS + (size_t)b * c.codec_vocab; need = (size_t)c.codec_vocab * 4; break;
        case Q3_GET_TRAILING: src = s->rows + (size_t)s->seq[b].trail_base * c.hidden; need = (size_t)s->seq[b].trailing_len * c.hidden * 4; break;
        case Q3_GET_PAD_EMBED: src = s->rows + (size_t)s->seq[b].pad_row * c.hidden; need = (size_t)c.hidden * 4; break;
        case Q3_GET_LOGITS_HIST:
            if (!s->logits_hist) return set_err(Q3_INVALID_ARG, "session has no debug capture");
            src = s->logits_hist + (size_t)b * (s->max_frames + 1) * c.codec_vocab; need = (size_t)(s->frames_run + 1) * c.codec_vocab * 4; break;
        case Q3_GET_TOKEN: src = s->tok + b; need = 4; break;
        case Q3_GET_CP_LOGITS: {
            // [15][B][V] on device → [15][V] for sequence b
            need = (size_t)15 * c.cp_vocab * 4;
            if (bytes < need) return set_err(Q3_INVALID_ARG, "buffer too small");
            for (int g = 0; g < 15; ++g)
                HIPC(hipMemcpy((char*)out + (size_t)g * c.cp_vocab * 4, s->CP_LOGITS + ((size_t)g * s->B + b) * c.cp_vocab, (size_t)c.cp_vocab * 4, hipMemcpyDeviceToHost));
            return Q3_OK;
        }
        case Q3_GET_CP_LOGITS_HIST: {
            if (!s->cp_logits_hist) return set_err(Q3_INVALID_ARG, "session has no debug capture");
            need = (size_t)s->frames_run * 15 * c.cp_vocab * 4;
            if (bytes < need) return set_err(Q3_INVALID_ARG, "buffer too small");
            for (int f = 0; f < s->frames_run; ++f)
                for (int g = 0; g < 15; ++g)
                    HIPC(hipMemcpy((char*)out + ((size_t)f * 15 + g) * c.cp_vocab * 4,
                                   s->cp_logits_hist + (((size_t)f * 15 + g) * s->B + b) * c.cp_vocab, (size_t)c.cp_vocab * 4, hipMemcpyDeviceToHost));
            return Q3_OK;
        }
        default: return set_err(Q3_INVALID_ARG, "unknown item %d", what);
    }
    if (bytes < need) return set_err(Q3_INVALID_ARG, "buffer too small (%zu < %zu)", bytes, need);
    HIPC(hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_talker_step(q3_session* s, const float* embeds_host, float* hidden_host, float* logits_host) {
    if (!s || !embeds_host) return set_err(Q3_INVALID_ARG, "null argument");
    if (!s->prefilled) return set_err(Q3_INVALID_ARG, "session not prefilled");
    const q3_config& c = s->m->cfg;
    HIPC(hipSetDevice(s->m->device));
    // the step writes K/V at `pos`: refuse BEFORE running when that slot does not exist (a step at pos == max_seq - 1 is valid)
    std::vector<int> posv(s->B);
    HIPC(sync_frames(s));
    HIPC(hipMemcpy(posv.data(), s->pos, s->B * 4, hipMemcpyDeviceToHost));
    for (int p : posv) if (p >= s->max_seq) return set_err(Q3_KV_OVERFLOW, "KV cache full (%d)", s->max_seq);
    for (int b = 0; b < s->B; ++b) Q3C(kv_reserve_row(s, b, posv[(size_t)b] + 1));       // paged KV: the slot this step writes
    HIPC(hipMemcpyAsync(s->tb.X, embeds_host, (size_t)s->B * c.hidden * 4, hipMemcpyHostToDevice, s->stream));
    Q3C(talker_step(s, s->pos, 0, true));
    // advance positions by one (host-driven teacher forcing)
    HIPC(sync_frames(s));
    for (int& p : posv) p += 1;
    HIPC(hipMemcpy(s->pos, posv.data(), s->B * 4, hipMemcpyHostToDevice));
    if (hidden_host) HIPC(hipMemcpy(hidden_host, s->LASTH, (size_t)s->B * c.hidden * 4, hipMemcpyDeviceToHost));
    if (logits_host) HIPC(hipMemcpy(logits_host, s->LOGITS, (size_t)s->B * c.codec_vocab * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_cp_generate(q3_session* s, const float* last_hidden_host, const float* sem_embed_host,
                                    uint32_t* codes15_host, float* cp_logits_host) {
    if (!s || !last_hidden_host || !codes15_host) return set_err(Q3_INVALID_ARG, "null argument");
    const q3_model* m = s->m; const q3_config& c = m->cfg;
    HIPC(hipSetDevice(m->device));
    // the semantic embedding is looked up from tok on device; teacher forcing passes it as an embedding
    // row, so run pass 1 from an explicit buffer: temporarily stage it through CP_IN/cb.X.
    const int B = s->B, H = c.hidden, CH = c.cp_hidden, V = c.cp_vocab;
    HIPC(hipMemcpyAsync(s->LASTH, last_hidden_host, (size_t)B * H * 4, hipMemcpyHostToDevice, s->stream));
    const LmDims d = cp_dims(c);
    float* sem_dev = nullptr;
    if (sem_embed_host) { HIPC(hipMalloc((void**)&sem_dev, (size_t)B * H * 4)); HIPC(hipMemcpy(sem_dev, sem_embed_host, (size_t)B * H * 4, hipMemcpyHostToDevice)); }
    q3_status st = Q3_OK;
    auto run = [&]() -> q3_status {
        for (int p = 0; p < c.n_groups; ++p) {
            CpGatherArgs g{};
            g.pass = p; g.last_hidden = s->LASTH; g.H = H; g.codec_emb = m->codec_emb; g.tok = s->tok;
            g.cp_emb = p >= 2 ? m->cp_emb[p - 2] : nullptr;
            g.cp_logits = p >= 2 ? s->CP_LOGITS + (size_t)(p - 2) * B * V : nullptr;
            g.cp_vocab = V; g.codes = s->codes; g.frame_idx = s->frame_idx; g.max_frames = s->max_frames; g.B = B;
            float* dst = m->mtp_w.t1 ? s->CP_IN : s->cb.X; const int ld = m->mtp_w.t1 ? H : CH;
            g.out = dst; g.ld_out = ld;
            if (p == 1 && sem_dev) HIPC(launch_copy_rows(sem_dev, H, dst, ld, B, H, s->stream));
            else HIPC(launch_cp_gather(g, s->stream));
            if (m->mtp_w.t1) {
                LinArgs a;
                a.N = CH; a.K = H; set_w(a, m->mtp_w, B, CH, H); a.x = s->CP_IN; a.ldx = H; a.bias = m->mtp_b; a.y = s->cb.X; a.ldy = CH; a.M = B; a.epi = EPI_NONE;
                HIPC(launch_linear(a, s->stream));
            }
            for (int i = 0; i < c.cp_layers; ++i)
                Q3C(lm_layer(s, d, m->cl[i], s->cb, s->ckcache + (size_t)i * s->ckv_layer_stride, s->cvcache + (size_t)i * s->ckv_layer_stride,
                             c.n_groups + 1, nullptr, p, 1));
            if (p >= 1) {
                LinArgs h;
                h.N = V; h.K = CH; set_w(h, m->cp_head[p - 1], B, V, CH); h.x = s->cb.X; h.ldx = CH; h.norm_w = m->cp_norm; h.eps = c.rms_eps;
                h.y = s->CP_LOGITS + (size_t)(p - 1) * B * V; h.ldy = V; h.M = B; h.epi = EPI_NONE;
                HIPC(launch_linear(h, s->stream));
            }
        }
        HIPC(sync_frames(s));
        return Q3_OK;
    };
    st = run();
    if (sem_dev) hipFree(sem_dev);
    Q3C(st);
    std::vector<float> lg((size_t)15 * B * V);
    HIPC(hipMemcpy(lg.data(), s->CP_LOGITS, lg.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < 15; ++g) {
            const float* row = &lg[((size_t)g * B + b) * V];
            int best = 0; for (int i = 1; i < V; ++i) if (row[i] > row[best]) best = i;
            codes15_host[(size_t)b * 15 + g] = (uint32_t)best;
            if (cp_logits_host) memcpy(cp_logits_host + ((size_t)b * 15 + g) * V, row, (size_t)V * 4);
        }
    return Q3_OK;
}

extern "C" q3_status q3_frame_embed(q3_model* m, uint32_t sem_token, const uint32_t* codes15, const float* text_add_host, float* out_host) {
    if (!m || !m->finalized || !codes15 || !text_add_host || !out_host) return set_err(Q3_INVALID_ARG, "bad argument");
    const q3_config& c = m->cfg;
    HIPC(hipSetDevice(m->device));
    const int H = c.hidden, V = c.cp_vocab;
    if (sem_token >= (uint32_t)c.codec_vocab) return set_err(Q3_INVALID_ARG, "semantic token out of range");
    DevPool pool;
    float *rows, *logits, *out; uint32_t *tok, *codes; int *zero, *one;
    HIPC(pool.alloc(&rows, (size_t)H)); HIPC(pool.alloc(&logits, (size_t)V)); HIPC(pool.alloc(&out, (size_t)H));
    HIPC(pool.alloc(&tok, 1)); HIPC(pool.alloc(&codes, 16)); HIPC(pool.alloc(&zero, 1)); HIPC(pool.alloc(&one, 1));
    // codes 0..13 pre-written; code 14 enters through a one-hot logits row
    uint32_t frame[16] = {0};
    for (int g = 0; g < 14; ++g) { if (codes15[g] >= (uint32_t)V) return set_err(Q3_INVALID_ARG, "code out of range"); frame[1 + g] = codes15[g]; }
    if (codes15[14] >= (uint32_t)V) return set_err(Q3_INVALID_ARG, "code out of range");
    std::vector<float> lg((size_t)V, 0.0f); lg[codes15[14]] = 1.0f;
    const int h_one = 1;
    HIPC(hipMemcpy(rows, text_add_host, (size_t)H * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(logits, lg.data(), (size_t)V * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(tok, &sem_token, 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(codes, frame, 64, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(one, &h_one, 4, hipMemcpyHostToDevice));
    FrameEmbedArgs f{};
    f.codec_emb = m->codec_emb; f.tok = tok; f.cp_logits_last = logits; f.cp_vocab = V;
    for (int g = 0; g < 15; ++g) f.cp_embs[g] = g < (int)m->cp_emb.size() ? m->cp_emb[(size_t)g] : nullptr;
    f.codes = codes; f.frame_idx = zero; f.max_frames = 1; f.text_rows = rows; f.trail_base = zero; f.trail_len = one; f.pad_row = zero;
    f.out = out; f.H = H; f.B = 1; f.n_acoustic = 15;
    HIPC(launch_frame_embed(f, 0));
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(out_host, out, (size_t)H * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_sample(int device, const float* logits_host, const uint8_t* seen_host, const float* u_host, int rows, int vocab,
                               const q3_options* o, int token_count, uint32_t* tokens_host) {
    if (!logits_host || !u_host || !o || !tokens_host || rows < 1) return set_err(Q3_INVALID_ARG, "bad argument");
    if (vocab < 2 || vocab > 4096) return set_err(Q3_UNSUPPORTED, "vocab %d unsupported by the device sampler (2..4096)", vocab);
    HIPC(hipSetDevice(device));
    DevPool pool;
    float *lg, *u; uint8_t* seen = nullptr; uint32_t* tok;
    HIPC(pool.alloc(&lg, (size_t)rows * vocab)); HIPC(pool.alloc(&u, (size_t)rows)); HIPC(pool.alloc(&tok, (size_t)rows));
    HIPC(hipMemcpy(lg, logits_host, (size_t)rows * vocab * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(u, u_host, (size_t)rows * 4, hipMemcpyHostToDevice));
    if (seen_host) { HIPC(pool.alloc(&seen, (size_t)rows * vocab)); HIPC(hipMemcpy(seen, seen_host, (size_t)rows * vocab, hipMemcpyHostToDevice)); }
    SampleArgs a; memset(&a, 0, sizeof a);
    a.logits = lg; a.ld = vocab; a.seen = seen; a.u = u; a.u_stride = 1; a.tok = tok; a.token_count_static = token_count < 0 ? 0 : token_count;
    a.vocab = vocab; a.B = rows;
    a.apply_temp = (o->temperature != 1.0 && o->temperature > 0.0) ? 1 : 0;
    a.inv_temp = (float)(1.0 / o->temperature);
    a.greedy = o->temperature < 0.01 ? 1 : 0;
    a.top_k = o->top_k; a.use_top_p = (o->top_p < 1.0 && o->top_p > 0.0) ? 1 : 0; a.top_p = (float)o->top_p;
    const bool pen = token_count >= 0;     // token_count < 0: plain `sample` without the penalty pipeline
    a.use_rep = (pen && seen && o->repetition_penalty != 1.0 && !(fabs(o->repetition_penalty - 1.0) < 1e-9)) ? 1 : 0;
    a.rep_pen = (float)o->repetition_penalty; a.rep_inv = 1.0f / (float)o->repetition_penalty;
    a.eos_id = pen ? o->eos_token_id : -1; a.min_new_tokens = pen ? o->min_new_tokens : 0; a.codec_eos = CODEC_EOS; a.use_suppress = pen ? 1 : 0;
    HIPC(launch_sample(a, 0));
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(tokens_host, tok, (size_t)rows * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_fused_residual_rmsnorm(int device, int dtype, const void* x_host, const void* res_host, const void* w_host,
                                               int rows, int cols, float eps, void* normed_host, void* sum_host) {
    if (!x_host || !res_host || !w_host || !normed_host || !sum_host || rows < 1 || cols < 1) return set_err(Q3_INVALID_ARG, "bad argument");
    if (dtype != Q3_DTYPE_F32 && dtype != Q3_DTYPE_BF16) return set_err(Q3_UNSUPPORTED, "dtype %d unsupported", dtype);
    HIPC(hipSetDevice(device));
    const size_t es = dtype == Q3_DTYPE_F32 ? 4 : 2, n = (size_t)rows * cols;
    DevPool pool;
    char *x, *r, *w, *nm, *sm;
    HIPC(pool.alloc(&x, n * es)); HIPC(pool.alloc(&r, n * es)); HIPC(pool.alloc(&w, (size_t)cols * es)); HIPC(pool.alloc(&nm, n * es)); HIPC(pool.alloc(&sm, n * es));
    HIPC(hipMemcpy(x, x_host, n * es, hipMemcpyHostToDevice)); HIPC(hipMemcpy(r, res_host, n * es, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(w, w_host, (size_t)cols * es, hipMemcpyHostToDevice));
    if (dtype == Q3_DTYPE_F32) HIPC(launch_fused_residual_rmsnorm_f32((float*)x, (float*)r, (float*)w, (float*)nm, (float*)sm, rows, cols, eps, 0));
    else HIPC(launch_fused_residual_rmsnorm_bf16((uint16_t*)x, (uint16_t*)r, (uint16_t*)w, (uint16_t*)nm, (uint16_t*)sm, rows, cols, eps, 0));
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(normed_host, nm, n * es, hipMemcpyDeviceToHost)); HIPC(hipMemcpy(sum_host, sm, n * es, hipMemcpyDeviceToHost));
    return Q3_OK;
}

extern "C" q3_status q3_linear(int device, const float* x_host, const uint16_t* w_host, const float* bias_host, int M, int N, int K, float* y_host) {
    if (!x_host || !w_host || !y_host || M < 1 || N < 1 || K < 8 || K % 8) return set_err(Q3_INVALID_ARG, "bad argument (K must be a multiple of 8)");
    HIPC(hipSetDevice(device));
    DevPool pool;
    float *x, *y, *b = nullptr; uint16_t* w;
    // the engine's own choice (pick_mode): 4-row tiles for narrow projections at small M — not for short-K wide ones, not beyond 16 rows
    const int mode = (M <= 16 && N < 4096 && !short_k_wide(N, K) && (M <= 2 || (N <= 1024 && M <= 8))) ? 2 : 1;
    const size_t wt_elems = tiled_elems(mode, N, K);
    std::vector<uint16_t> wt(wt_elems);
    retile_bf16(w_host, N, K, wt.data(), mode);
    HIPC(pool.alloc(&x, (size_t)M * K)); HIPC(pool.alloc(&y, (size_t)M * N)); HIPC(pool.alloc(&w, wt_elems));
    HIPC(hipMemcpy(x, x_host, (size_t)M * K * 4, hipMemcpyHostToDevice)); HIPC(hipMemcpy(w, wt.data(), wt_elems * 2, hipMemcpyHostToDevice));
    if (bias_host) { HIPC(pool.alloc(&b, (size_t)N)); HIPC(hipMemcpy(b, bias_host, (size_t)N * 4, hipMemcpyHostToDevice)); }
    const int step = mode == 1 ? Q3_MAX_BATCH : 16;          // up to 64 rows per launch on the 16-row tiles (wide-session kernels beyond 16)
    float* ws = nullptr; size_t ws_bytes = 0;
    if (M > 16 && N % 128 == 0 && K % 128 == 0) { ws_bytes = gemm_wide_ws_bytes(M < step ? M : step, N, K, EPI_NONE); HIPC(pool.alloc(&ws, ws_bytes / 4)); }
    for (int m0 = 0; m0 < M; m0 += step) {
        LinArgs a;
        a.W = w; a.N = N; a.K = K; a.x = x + (size_t)m0 * K; a.ldx = K; a.bias = b; a.y = y + (size_t)m0 * N; a.ldy = N; a.M = (M - m0) < step ? (M - m0) : step; a.epi = EPI_NONE;
        a.tiled = mode; a.Kpad = kpad_for(mode, K); a.ws = ws; a.ws_bytes = ws_bytes;
        HIPC(launch_linear(a, 0));
    }
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(y_host, y, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return Q3_OK;
}

#ifdef Q3_TRACE
// development builds only (not declared in include/q3tts.h): arm the per-node stamp buffer BEFORE the first
// q3_session_generate (the captured graph keeps the slice pointers), then read the stamps of the last replayed frame.
extern "C" q3_status q3_debug_trace_enable(q3_session* s, int max_nodes) {
    if (!s || max_nodes < 1) return set_err(Q3_INVALID_ARG, "q3_debug_trace_enable");
    const size_t bytes = (size_t)max_nodes * TRACE_NODE * 8;
    HIPC(hipMalloc((void**)&s->trace_buf, bytes));
    HIPC(hipMemset(s->trace_buf, 0, bytes));
    s->trace_cap = max_nodes;
    return Q3_OK;
}
extern "C" q3_status q3_debug_trace_read(q3_session* s, unsigned long long* stamps_host, int* meta_host, int cap_nodes, int* n_nodes) {
    if (!s || !s->trace_buf || !n_nodes) return set_err(Q3_INVALID_ARG, "q3_debug_trace_read");
    HIPC(sync_frames(s));
    const int n = (int)s->trace_meta.size();
    *n_nodes = n;
    if (stamps_host && meta_host) {
        if (cap_nodes < n) return set_err(Q3_INVALID_ARG, "trace buffer too small");
        HIPC(hipMemcpy(stamps_host, s->trace_buf, (size_t)n * TRACE_NODE * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const auto& t = s->trace_meta[i];
            const int v[7] = {t.kind, t.a, t.b, t.c, t.d, t.e, t.f};
            memcpy(meta_host + (size_t)i * 7, v, sizeof v);
        }
    }
    return Q3_OK;
}
#endif

extern "C" q3_status q3_session_submit_info(q3_session* s, int* path, int* nodes) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (path) *path = s->aql ? 1 + s->aql_mode : s->graph ? 1 : 0;
    if (nodes) *nodes = s->aql ? q3::aql_program_nodes(s->aql) : 0;
    return Q3_OK;
}

extern "C" q3_status q3_session_frame_bytes(q3_session* s, int kv_len, double* weight_bytes, double* kv_bytes) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    const q3_config& c = s->m->cfg;
    auto layer_params = [](int H, int I, int nh, int nkv) { return (double)H * (nh + 2 * nkv) * HEAD_DIM + (double)H * nh * HEAD_DIM + 3.0 * (double)H * I; };
    const double talker = c.n_layers * layer_params(c.hidden, c.inter, c.n_heads, c.n_kv_heads) + (double)c.codec_vocab * c.hidden;
    const double cp_pass = c.cp_layers * layer_params(c.cp_hidden, c.cp_inter, c.cp_heads, c.cp_kv_heads) + (double)c.cp_vocab * c.cp_hidden +
                           (c.hidden != c.cp_hidden ? (double)c.cp_hidden * c.hidden : 0.0);
    if (weight_bytes) *weight_bytes = 2.0 * (talker + 15.0 * cp_pass);          // bf16; SURVEY §8(d)
    if (kv_bytes) {
        const double kv_tok = 2.0 * c.n_kv_heads * HEAD_DIM * (s->kv_bf16 ? 2.0 : 4.0) * c.n_layers;   // f32 K/V (default) or a bf16 session's
        const double cp_tok = 2.0 * c.cp_kv_heads * HEAD_DIM * 4.0 * c.cp_layers;
        *kv_bytes = s->B * (kv_tok * kv_len + cp_tok * 135.0);
    }
    return Q3_OK;
}

// profiling read-out: accumulated GPU milliseconds / algorithmic bytes / launches of the bf16 GEMV family
extern "C" q3_status q3_session_profile_read(q3_session* s, double* ms, double* bytes, long* launches, int reset) {
    if (!s) return set_err(Q3_INVALID_ARG, "null session");
    if (ms) *ms = s->prof_linear.ms; if (bytes) *bytes = s->prof_linear.bytes; if (launches) *launches = s->prof_linear.launches;
    if (reset) s->prof_linear = ProfAcc();
    return Q3_OK;
}

// the distinct GEMV launches (and how often each ran) since profiling was enabled / last reset: rows of 8 ints
// {M, N, K, epilogue, fused input RMSNorm (0 / 1), reserved (0), tiling, count} — what q3_bench_linear can replay
extern "C" q3_status q3_session_profile_shapes(q3_session* s, int* rows, int cap_rows, int* n_rows, int reset) {
    if (!s || !n_rows) return set_err(Q3_INVALID_ARG, "null argument");
    *n_rows = (int)s->prof_shapes.size();
    if (rows) {
        if (cap_rows < *n_rows) return set_err(Q3_INVALID_ARG, "shape buffer too small (%d < %d rows)", cap_rows, *n_rows);
        for (int i = 0; i < *n_rows; ++i) {
            const ProfShape& q = s->prof_shapes[i];
            const int v[8] = {q.M, q.N, q.K, q.epi, q.rms, q.produce, q.tiled, q.count};
            memcpy(rows + (size_t)i * 8, v, sizeof v);
        }
    }
    if (reset) s->prof_shapes.clear();
    return Q3_OK;
}

// ------------------------------------------------------------------------------------------------
// micro-benchmark of one GEMV shape (kernel development aid, used by tools/bench_kernels.py):
// `iters` back-to-back launches cycling over `n_copies` distinct weight buffers (so the stream comes
// from HBM, not the 256 MiB Infinity Cache), captured in one hipGraph and timed with HIP events.
// epi: LinEpi; rms: fused input RMSNorm; tiled: 1 = 16-row MFMA tiles, 2 = 4-row tiles, 3 = 16-row tiles with split-K in two,
// 0 = first-generation VALU kernel, < 0 = the engine's choice for an unsplit launch.
// ------------------------------------------------------------------------------------------------
extern "C" q3_status q3_bench_linear(int device, int M, int N, int K, int epi, int rms, int tiled, int iters, int n_copies,
                                     double* avg_us) {
    if (M < 1 || M > Q3_MAX_BATCH || N < 16 || K < 32 || iters < 1 || n_copies < 1 || !avg_us || epi < EPI_NONE || epi > EPI_SWIGLU || rms < 0 || rms > 1)
        return set_err(Q3_INVALID_ARG, "q3_bench_linear: bad argument (epi 0..3, rms 0/1)");
    if (tiled < 0) tiled = (M <= 16 && N < 4096 && !short_k_wide(N, K) && (M <= 2 || (N <= 1024 && M <= 8))) ? 2 : 1;    // the engine's choice (pick_mode)
    const bool sk2 = tiled == 3;          // 3 = 16-row tiles, split-K in two (LinArgs::ksplit): y alternates between two buffers,
    if (sk2) tiled = 1;                   // each launch clearing the other one as its side job, as the frame loop's neighbours do
    HIPC(hipSetDevice(device));
    DevPool pool;
    const size_t welems = tiled == 2 ? tiled_elems(2, N, K) : tiled_elems(1, N, K);
    const int nmat = epi == EPI_SWIGLU ? 2 : 1;
    uint16_t* w; float *x, *y, *nw, *res;
    HIPC(pool.alloc(&w, welems * nmat * n_copies));
    HIPC(pool.alloc(&x, (size_t)Q3_MAX_BATCH * K)); HIPC(pool.alloc(&y, (size_t)2 * Q3_MAX_BATCH * N)); HIPC(pool.alloc(&nw, (size_t)K)); HIPC(pool.alloc(&res, (size_t)Q3_MAX_BATCH * N));
    HIPC(hipMemset(y, 0, (size_t)2 * Q3_MAX_BATCH * N * 4));
    {   // random-ish bf16 weights / f32 activations (never zeros: DVFS, guide §5.4 rule 25)
        std::vector<uint16_t> hw(welems);
        q3_synth_fill(1, "bench.w", Q3_DTYPE_BF16, 0.02f, 0.0f, (int64_t)welems, hw.data());
        for (int c = 0; c < nmat * n_copies; ++c) HIPC(hipMemcpy(w + (size_t)c * welems, hw.data(), welems * 2, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)Q3_MAX_BATCH * K), hn((size_t)K, 1.0f);
        q3_synth_fill(2, "bench.x", Q3_DTYPE_F32, 1.0f, 0.0f, (int64_t)hx.size(), hx.data());
        HIPC(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(nw, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st; HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* ws = nullptr; size_t ws_bytes = 0;
    if (M > 16 && tiled == 1 && N % 128 == 0 && K % 128 == 0) { ws_bytes = gemm_wide_ws_bytes(M, N, K, epi); HIPC(pool.alloc(&ws, ws_bytes / 4)); }
    auto one = [&](int i) -> hipError_t {
        LinArgs a;
        a.ws = ws; a.ws_bytes = ws_bytes;
        const int c = i % n_copies;
        a.W = w + (size_t)c * nmat * welems; a.W2 = nmat == 2 ? a.W + welems : nullptr;
        a.N = N; a.K = K; a.Kpad = tiled == 2 ? up128(K) : up32(K); a.tiled = tiled; a.x = x; a.ldx = K; a.y = y; a.ldy = N; a.M = M; a.epi = epi;
        if (rms) { a.norm_w = nw; a.eps = 1e-6f; }
        if (epi == EPI_RESID) { a.resid = res; a.ldr = N; }
        if (sk2) { a.ksplit = 2; a.y = y + (size_t)(i & 1) * Q3_MAX_BATCH * N; a.zero = y + (size_t)((i + 1) & 1) * Q3_MAX_BATCH * N; a.zero_n = M * N; }
        return launch_linear(a, st);
    };
    for (int i = 0; i < 4; ++i) HIPC(one(i));
    HIPC(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    HIPC(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipError_t e = hipSuccess;
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = one(i);
    hipError_t e2 = hipStreamEndCapture(st, &g);
    if (e != hipSuccess || e2 != hipSuccess) return set_err(Q3_HIP_ERROR, "bench capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIPC(hipGraphLaunch(ge, st)); HIPC(hipStreamSynchronize(st));     // warm
    hipEvent_t ev0, ev1; HIPC(hipEventCreate(&ev0)); HIPC(hipEventCreate(&ev1));
    double sum = 0.0;          // mean over 5 replays (not the best one)
    for (int rep = 0; rep < 5; ++rep) {
        HIPC(hipEventRecord(ev0, st));
        HIPC(hipGraphLaunch(ge, st));
        HIPC(hipEventRecord(ev1, st));
        HIPC(hipStreamSynchronize(st));
        float ms = 0; HIPC(hipEventElapsedTime(&ms, ev0, ev1));
        sum += ms;
    }
    *avg_us = sum / 5.0 * 1000.0 / iters;
    hipEventDestroy(ev0); hipEventDestroy(ev1); hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    return Q3_OK;
}

