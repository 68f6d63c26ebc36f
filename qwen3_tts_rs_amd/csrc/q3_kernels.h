// q3_kernels.h — host-side launchers of the gfx950 kernels (q3_kernels_lm.hip,
// q3_kernels_codec.hip). Everything takes an explicit hipStream_t; no allocation, no sync — all
// launchers are hipGraph-capturable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q3 {

constexpr int HEAD_DIM = 128;        // talker / code-predictor head dim (kernels are specialised)
constexpr int MAX_SPLITS = 64;       // KV splits of the decode attention (16 unless the context is long, see q3_session_create)
constexpr int PART_STRIDE = HEAD_DIM + 2;   // partial record: acc[128], m, l
// Paged talker KV (replaces the reference's preallocated per-call cache, kv_cache.rs:234-310): a PAGE holds KV_PAGE_POS
// consecutive positions of ONE sequence for every layer and KV head, K and V (29.4 MB at 28 layers x 8 KV heads; a
// (layer, head) run is 64 KB). Pages are slots of layer-major slabs (KvPool, q3_engine.h); a page is named by the
// address of its layer-0 K run. A sequence owns a row of KV_MAX_PAGES page pointers in device memory (64 x 128 = 8192
// positions = the RoPE table); the K row of position p of (layer l, head h) lives at
//   pages[p / 128] + l * layer_stride + h * 16384 + (p % 128) * 128 floats, its V row kv_vdelta floats further.
constexpr int KV_PAGE_POS = 128, KV_PAGE_SHIFT = 7, KV_MAX_PAGES = 64;

// ---- Q3_TRACE (development builds only: tools/trace_build.sh -> libq3tts_trace.so; never defined in the product) ----
// Per-node timeline of the captured frame graph at 10 ns resolution: every instrumented kernel gets, per launch, its own
// slice of a device buffer through its argument struct (a captured node keeps its kernargs, so each replay overwrites
// the same slice), and lane 0 of every workgroup stores s_memrealtime stamps taken at fixed points: slot 0 = entry,
// 1 = inputs landed (vmcnt(0)), 2 = main work done, 3 = results stored (issued), 4 = stores acknowledged (vmcnt(0)).
// tools/trace_frame.py turns them into the Gantt table of one frame (launch gaps, entry skew, phase lengths).
#ifdef Q3_TRACE
constexpr int TRACE_SLOTS = 8, TRACE_WGS = 512, TRACE_DWGS = 16, TRACE_WAVES = 16;     // + every wave of the first TRACE_DWGS workgroups
constexpr int TRACE_NODE = TRACE_SLOTS * (TRACE_WGS + TRACE_DWGS * TRACE_WAVES);          // u64 per node
#define Q3_TRACE_FIELD unsigned long long* trace = nullptr;
#if defined(__HIPCC__)
#define Q3T_DECL unsigned long long q3t_[q3::TRACE_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0};
#define Q3T(i) do { q3t_[i] = (unsigned long long)wall_clock64(); } while (0)
#define Q3T_W(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); q3t_[i] = (unsigned long long)wall_clock64(); } while (0)
// stamp taken once the kernel argument `val` has arrived in its SGPR (slot 7: the kernarg round trip of a node)
#define Q3T_K(i, val) do { asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(val) : "memory"); q3t_[i] = (unsigned long long)wall_clock64(); } while (0)
#define Q3T_FLUSH(a, wg) do { if ((a).trace && threadIdx.x == 0 && (int)(wg) < q3::TRACE_WGS) { \
        _Pragma("unroll") for (int q3i_ = 0; q3i_ < q3::TRACE_SLOTS; ++q3i_) (a).trace[(size_t)(wg) * q3::TRACE_SLOTS + q3i_] = q3t_[q3i_]; } \
    if ((a).trace && (threadIdx.x & 63) == 0 && (int)(wg) < q3::TRACE_DWGS && (int)(threadIdx.x >> 6) < q3::TRACE_WAVES) { \
        _Pragma("unroll") for (int q3i_ = 0; q3i_ < q3::TRACE_SLOTS; ++q3i_) \
            (a).trace[(size_t)q3::TRACE_SLOTS * (q3::TRACE_WGS + (wg) * q3::TRACE_WAVES + (threadIdx.x >> 6)) + q3i_] = q3t_[q3i_]; } } while (0)
#endif
#else
#define Q3_TRACE_FIELD
#define Q3T_DECL
#define Q3T(i) do { } while (0)
#define Q3T_W(i) do { } while (0)
#define Q3T_K(i, val) do { } while (0)
#define Q3T_FLUSH(a, wg) do { } while (0)
#endif

// ---- bf16-weight skinny GEMM ("GEMV family"): y[m][n] = sum_k x[m][k] * W[n][k] ----
enum LinEpi { EPI_NONE = 0, EPI_RESID = 1, EPI_SILU = 2, EPI_SWIGLU = 3 };
struct LinArgs {
    const uint16_t* W = nullptr;    // [N][K] bf16
    const uint16_t* W2 = nullptr;   // second matrix (SwiGLU "up")
    const float* x = nullptr; int ldx = 0;        // [M][ldx]
    const float* norm_w = nullptr; float eps = 0; // fused input RMSNorm if norm_w != nullptr
    const float* bias = nullptr;                  // [N] or nullptr
    const float* resid = nullptr; int ldr = 0;    // EPI_RESID: y = resid + acc
    float* y = nullptr; int ldy = 0;
    int M = 1, N = 0, K = 0;
    int epi = EPI_NONE;
    int tiled = 0;                  // 1: 16-row MFMA tiles [N/16][Kpad/32][64 lanes][8 bf16]; 2: 4-row tiles [N/4][Kpad/128][64][8]
    int Kpad = 0;                   // K rounded up to 32 (tiled == 1) or 128 (tiled == 2)
    // Split-K in two (k_gemv_sk2: tiled == 1, no fused norm, epilogue none / resid): the K halves of a 16-row tile run as
    // two workgroups that ADD their results into y with f32 atomics. y MUST hold zeros when the launch starts; with exactly
    // two addends onto zero the sum does not depend on their order (f32 addition is commutative: 0 + a + b == 0 + b + a bit
    // for bit), so the result is deterministic. The half that owns k = 0 carries bias and residual.
    int ksplit = 1;
    // side job of any GEMV launch: store zeros to zero[0 .. zero_n) (zero_n % 4 == 0) — how the target of the NEXT
    // split-K launch is cleared without a launch of its own (the buffer must be dead for this launch's consumers)
    float* zero = nullptr; int zero_n = 0;
    // workspace of the wide-session GEMM (q3_kernels_wide.hip; 17 <= M <= 64): slice sums [S][matrices][M][N] + sum(x^2) [S][M].
    // nullptr / too small: the launch falls back to k_gemv_wide
    float* ws = nullptr; size_t ws_bytes = 0;
    Q3_TRACE_FIELD
};
hipError_t launch_linear(const LinArgs& a, hipStream_t st);       // dispatches on a.tiled
// wide sessions: 128 weight rows x one K slice per workgroup + a slice-sum / epilogue launch; hipErrorNotSupported = shape outside the family
hipError_t launch_gemm_wide(const LinArgs& a, hipStream_t st);
size_t gemm_wide_ws_bytes(int M, int N, int K, int epi);
hipError_t launch_gemv_tiled(const LinArgs& a, hipStream_t st);   // MFMA bf16x3 kernel (16-row tiles, tiled == 1)
hipError_t launch_gemv_tiled4(const LinArgs& a, hipStream_t st);  // 4-row tiles on the 4x4x4 16-block MFMA (tiled == 2)
hipError_t launch_linear_rowmajor(const LinArgs& a, hipStream_t st);   // first-generation VALU kernel

// ---- long-prompt prefill (q3_kernels_prefill.hip): real GEMM over the same tiled weight image + query-blocked attention ----
struct GemmArgs {
    const uint16_t* W = nullptr; const uint16_t* W2 = nullptr;   // mode-1 tiled bf16 images ([N/16][Kpad/32][64 lanes][8])
    const float* x = nullptr; int ldx = 0;                       // [M][ldx] f32
    const float* norm_w = nullptr; const float* den = nullptr;   // fused input RMSNorm: x*norm_w, result / den[m] (launch_row_den)
    const float* bias = nullptr;
    const float* resid = nullptr; int ldr = 0;
    float* y = nullptr; int ldy = 0;
    int M = 0, N = 0, K = 0, Kpad = 0;
    int epi = EPI_NONE;
    // optional: x (times norm_w when set) already split into its three exact bf16 terms by launch_split_rows —
    // [3][plane_elems] bf16, row pitch Kpad. The GEMM then stages plain copies instead of redoing the split in every
    // one of its N/64 workgroup columns.
    // The buffer must be readable for whole 128-row tiles: rows M .. ceil(M/128)*128 of every plane are fetched (any
    // content) by the LDS-DMA geometry and never stored.
    const uint16_t* xp = nullptr; size_t xp_plane = 0;
    int n_split = 1;                                             // set by launch_lm_gemm: N range cut into this many XCD work units per M tile
    // third geometry: optional workspace for split-K partial sums (8 ranges x [gate, up] x M x N floats); set by the caller
    float* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;
    int k_ranges = 1, k_split = 1, wg_per_split = 0;             // set by launch_lm_gemm
    int sup_m = 4, sup_n = 8;                                    // set by launch_lm_gemm (third geometry): M x N tiles of the super-tile one XCD runs at a time
};
hipError_t launch_lm_gemm(const GemmArgs& a, hipStream_t st);
// planes[pl][row][k] (pitch Kpad, zero beyond K) = pl-th bf16 term of x[row][k] * (norm_w ? norm_w[k] : 1); K % 8 == 0
hipError_t launch_split_rows(const float* x, int ldx, const float* norm_w, uint16_t* planes, size_t plane_elems, int rows, int K,
                             int Kpad, hipStream_t st);
hipError_t launch_row_den(const float* x, int ldx, float* den, int rows, int cols, float eps, hipStream_t st);

// standalone analogue of kernels/fused_residual_rmsnorm.cu: (normed, sum) for [rows][cols]
hipError_t launch_fused_residual_rmsnorm_f32(const float* x, const float* res, const float* w, float* normed,
                                             float* sum, int rows, int cols, float eps, hipStream_t st);
hipError_t launch_fused_residual_rmsnorm_bf16(const uint16_t* x, const uint16_t* res, const uint16_t* w,
                                              uint16_t* normed, uint16_t* sum, int rows, int cols, float eps,
                                              hipStream_t st);
// y = rms_norm(x) * w   (rows = batch)
hipError_t launch_rmsnorm(const float* x, int ldx, const float* w, float* y, int ldy, int rows, int cols, float eps,
                          hipStream_t st);

// ---- attention decode ----
struct AttnArgs {
    const float* qkv; int ld_qkv;           // [B][nh*128 + 2*nkv*128] raw projections
    const float* q_norm_w; const float* k_norm_w; float eps;
    const float* rope_cos; const float* rope_sin;   // [max_pos][64]
    const int* pos_dev; int pos_static;     // position of the new token: pos_dev[b] if non-null
    float* kcache; float* vcache;           // [B][nkv][max_seq][128]
    int max_seq;
    // paged form (the talker): kv_pages != nullptr -> kcache / vcache / max_seq are unused; kv_pages[seq * KV_MAX_PAGES + p / 128]
    // is the page of position p, kv_layer_off = floats from the page's layer-0 K run to this layer's, kv_vdelta = floats from
    // a K row to its V row (both constants of the pool's slab layout)
    const unsigned long long* kv_pages = nullptr; size_t kv_layer_off = 0, kv_vdelta = 0;
    int kv_row_pages = 0;                   // most pages a row of this session can ever hold (0 = unknown: up to KV_MAX_PAGES)
    int kv_bf16 = 0;                        // launch_attn_fused only: the pages hold bf16 (same geometry, 2-byte elements; q3_session_set_kv_dtype)
    float* qbuf;                            // [B][nh][128] normed+roped q
    float* part;                            // [B][nh][n_splits][PART_STRIDE]
    float* out; int ld_out;                 // [B][nh*128]
    int B, nh, nkv, n_splits;
    // multi-row steps (chunked prefill, 2-token code-predictor pass): B counts ROWS; row r belongs to sequence
    // r / rows_per_seq and sits at position base_pos(sequence) + r % rows_per_seq. 0/1 = one row per sequence.
    int rows_per_seq;
    // prefill only: K/V of every cached position as bf16x3 planes in 32-key tiles (launch_kv_planes), kvp_tiles tiles
    // allocated per (sequence, kv head); selects the bf16-matrix-core attention when set
    const unsigned char* kvp = nullptr; int kvp_tiles = 0;
    // launch_attn_fused only — code-predictor layer 0 of a pass whose input is a table row (the folded k_cp_gather):
    // every workgroup re-derives row = argmax(g_logits[b]) (2048 logits: cheaper than a launch) and takes q|k|v from
    // g_qkv_tab[row] instead of qkv[b]; the (kv head 0, split 0) workgroup also records the code and copies the
    // residual-stream row g_proj_tab[row] to g_x[b].
    // launch_attn_first2: row 2b+1 of sequence b is table row g_tok[b] (no argmax, no code to record).
    const float* g_logits = nullptr; int g_vocab = 0; const uint32_t* g_tok = nullptr;
    const float* g_qkv_tab = nullptr;
    const float* g_proj_tab = nullptr; int g_proj_dim = 0; float* g_x = nullptr; int g_ldx = 0;    // g_x + b * g_ldx = the sequence's residual row
    uint32_t* g_codes = nullptr; const int* g_frame_idx = nullptr; int g_max_frames = 0, g_code_slot = 0;
    // side job (launch_attn_fused / launch_attn_cp): store zeros to zero[0 .. zero_n) — clears the target of the split-K
    // o-projection that follows (LinArgs::ksplit)
    float* zero = nullptr; int zero_n = 0;
    // wide sessions (launch_attn_fused / launch_attn_cp): q|k|v arrives as the K-slice sums of the split-K GEMM
    // (launch_gemm_wide_partial: qkv_part[s][row][ld_qkv], sum(x^2) per slice qkv_ssq[s][row]) and the kernel does what
    // k_wide_epilogue would have — slices added in order, then / sqrt(mean(x^2) + eps) — on the 3 x 128 values it needs:
    // one launch less per layer. qkv is unused then. qkv_S <= 8.
    const float* qkv_part = nullptr; const float* qkv_ssq = nullptr; int qkv_S = 0, qkv_K = 0; float qkv_eps = 0.0f;
    Q3_TRACE_FIELD
};
#if defined(__HIPCC__)
// K row of (sequence, kv head, position) — contiguous extent or page (one table fetch: the cold paths; k_attn_fused keeps the
// sequence's table row in registers instead). The V row is `+ kv_vd(a)` floats further.
// Row rotation inside a (page, head) run: position p sits in row (p + kv_rot) % 128. Every run starts on a 64 KB boundary, and
// the workgroups of a decode-attention launch all read the same position range of their sequences at the same time — without
// the rotation the 64 (sequence, head) streams of a B = 8 launch walk addresses that are equal modulo 64 KB (the contiguous
// layout staggers them by max_seq * 512 B) and crowd the same memory channels.
__device__ __forceinline__ int kv_rot(unsigned long long page, int kvh) {
    const unsigned h = (unsigned)(page >> 19);            // slot number of the page within its slab region (512 KB apart at 8 KV heads)
    return (int)((kvh * 16 + (h & 15) * 8 + ((h >> 4) & 7)) & (KV_PAGE_POS - 1));
}
__device__ __forceinline__ float* kv_krow(const AttnArgs& a, int seq, int kvh, int p) {
    if (a.kv_pages) {
        const unsigned long long page = a.kv_pages[(size_t)seq * KV_MAX_PAGES + (p >> KV_PAGE_SHIFT)];
        // the page address as an OFFSET from a pointer the compiler knows to be global (a kernel argument): a pointer made from
        // an integer is generic, and its loads become flat_load — which also count on lgkmcnt, so every LDS / scalar wait
        // then waits for the K/V stream (k_attn_fused: +1 us per launch until its page pointers were formed this way)
        float* pg = reinterpret_cast<float*>(reinterpret_cast<char*>(const_cast<unsigned long long*>(a.kv_pages)) +
                                             (ptrdiff_t)(page - reinterpret_cast<unsigned long long>(a.kv_pages)));
        return pg + a.kv_layer_off + ((size_t)kvh * KV_PAGE_POS + ((p + kv_rot(page, kvh)) & (KV_PAGE_POS - 1))) * HEAD_DIM;
    }
    return a.kcache + (((size_t)seq * a.nkv + kvh) * a.max_seq + p) * HEAD_DIM;
}
__device__ __forceinline__ ptrdiff_t kv_vd(const AttnArgs& a) { return a.kv_pages ? (ptrdiff_t)a.kv_vdelta : a.vcache - a.kcache; }
#endif
// the GEMM of launch_gemm_wide WITHOUT its slice-sum launch (EPI_NONE with a fused input norm only): the consumer adds
// the slices. hipErrorNotSupported: shape outside the family.
struct WidePartial { const float* part; const float* ssq; int S; };
int gemm_wide_min_rows();      // rows from which launch_gemm_wide takes a projection (17; Q3_WIDE_GEMM_MIN = 17 .. 64: A/B aid)
hipError_t launch_gemm_wide_partial(const LinArgs& a, hipStream_t st, WidePartial* out);
constexpr size_t KVP_TILE_BYTES = 6 * 32 * HEAD_DIM * 2;
// planes of positions [0, n_pos) of every (sequence, kv head) pair, from the f32 cache launch_qknorm_rope_kv filled
hipError_t launch_kv_planes(const AttnArgs& kv, int n_pairs, int n_pos, int tiles_alloc,
                            unsigned char* kvp, hipStream_t st);
// pages of f32 K/V -> pages of bf16 (RNE), the first n_pos positions of every (layer, kv head) run: src / dst = device arrays
// of page addresses [n_pages]; layer_stride / v_delta in ELEMENTS (the same counts in both pools)
hipError_t launch_kv_pages_to_bf16(const unsigned long long* src_pages, const unsigned long long* dst_pages, int n_pages, int n_layers, int nkv,
                                   size_t layer_stride, size_t v_delta, hipStream_t st);
hipError_t launch_qknorm_rope_kv(const AttnArgs& a, hipStream_t st);
hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t st);
hipError_t launch_attn_merge(const AttnArgs& a, hipStream_t st);
// the code predictor's 2-token first pass (rows 2b / 2b+1 at positions 0 / 1 of an empty cache), one launch
hipError_t launch_attn_first2(const AttnArgs& a, hipStream_t st);
// prefill: causal attention of rows_per_seq consecutive positions per sequence (B = total rows), after launch_qknorm_rope_kv
hipError_t launch_attn_prefill(const AttnArgs& a, hipStream_t st);
// fused q/k-norm + RoPE + KV append + attention (+ final normalisation when n_splits == 1)
hipError_t launch_attn_fused(const AttnArgs& a, hipStream_t st);
// the same for caches that never exceed 16 positions with the position known at launch (the code predictor's single-row
// passes): one wave per (sequence, q head), one memory round trip; attn_cp_ok says whether the arguments qualify
bool attn_cp_ok(const AttnArgs& a);
hipError_t launch_attn_cp(const AttnArgs& a, hipStream_t st);

// ---- frame glue ----
// gather rows: out[r][0..dim) = f32(table_bf16[ids[r]][0..dim))
hipError_t launch_gather_rows_bf16(const uint16_t* table, const uint32_t* ids, float* out, int n_rows, int dim,
                                   hipStream_t st);
// prefill assembly: out[i] = (text_row[i] >= 0 ? rows[text_row[i]] : 0) + (codec_id[i] >= 0 ? codec_emb[codec_id[i]]
//                   : codec_id[i] == -2 ? xvec : 0)
hipError_t launch_assemble_rows(const float* rows, const int* text_row, const uint16_t* codec_emb, const int* codec_id,
                                const float* xvec, float* out, int n, int H, hipStream_t st,
                                const uint32_t* ref_codes = nullptr, const uint16_t* const* cp_embs = nullptr);   // codec_id <= -3: ICL ref frame -3-id
hipError_t launch_copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t st);

struct CpGatherArgs {
    int pass;                        // 0..15
    const float* last_hidden; int H; // [B][H] (pass 0)
    const uint16_t* codec_emb;       // talker codec embedding [codec_vocab][H] (pass 1)
    const uint32_t* tok;             // [B] current semantic token
    const uint16_t* cp_emb;          // table of group pass-2 (pass >= 2): [cp_vocab][H]
    const float* cp_logits;          // [B][cp_vocab] logits of the previous pass (pass >= 2)
    int cp_vocab;
    uint32_t* codes;                 // [B][max_frames][16]
    const int* frame_idx;            // [B]
    int max_frames;
    float* out; int ld_out;          // [B][H]
    // 1.7B: pass >= 1 rows taken from a PRE-PROJECTED f32 table [vocab][proj_dim] (small_to_mtp_projection applied to
    // every embedding row once at model finalize) instead of the bf16 embedding table; nullptr = embedding table
    const float* proj_tab = nullptr; int proj_dim = 0;
    // pass >= 1: the row's layer-0 q|k|v projection (input RMSNorm included) from a table built at finalize, copied to
    // qkv_out[b][0 .. qkv_dim) so that the layer-0 qkv GEMV of this pass can be skipped
    const float* qkv_tab = nullptr; int qkv_dim = 0; float* qkv_out = nullptr; int ld_qkv_out = 0;
    int B;
};
hipError_t launch_cp_gather(const CpGatherArgs& a, hipStream_t st);

struct FrameEmbedArgs {
    const uint16_t* codec_emb; const uint16_t* cp_embs[15];      // the 15 acoustic tables, IN the kernel arguments (round 5: was a device array — one more dependent round trip in front of every gather)
    const uint32_t* tok; const float* cp_logits_last; int cp_vocab;  // final pass logits → code 14
    uint32_t* codes; const int* frame_idx; int max_frames;
    const float* text_rows; const int* trail_base; const int* trail_len; const int* pad_row;   // per-seq
    float* out; int H; int B; int n_acoustic;
};
hipError_t launch_frame_embed(const FrameEmbedArgs& a, hipStream_t st);

// per-sequence sampling options (SynthesisOptions is per call in the reference, lib.rs:1786-1836): when SampleArgs::rows is
// set, row b's values replace the scalar fields below, so the sequences of one session — and a row swapped in later — may
// differ in temperature, top-k / top-p, repetition penalty, EOS id and min_new_tokens while sharing one captured sampler
struct SampleRow {
    float inv_temp; int apply_temp; int greedy;
    int top_k; float top_p; int use_top_p;
    float rep_pen, rep_inv; int use_rep;
    int eos_id; int min_new_tokens; int pad;
};
struct SampleArgs {
    const float* logits; int ld;     // [B][vocab]
    uint8_t* seen;                   // [B][vocab] or nullptr
    const float* u; int u_stride;    // u[b*u_stride + (draw_idx ? draw_idx[b] : 0)]
    const int* draw_idx;
    uint32_t* tok;                   // [B] out
    int* token_count;                // [B] device counters (incremented) or nullptr
    int token_count_static;
    int* frame_idx; int* pos;        // advanced by one after sampling when advance != 0
    int advance;
    const SampleRow* rows;           // per-sequence options (nullable), see SampleRow
    // per-sequence frame limits (nullable): a sequence whose frame_idx has reached limit[b] is FROZEN — its counters
    // (frame_idx, pos, token_count) stop, so it keeps re-running its last position in bounds while the other rows of the
    // session go on (rows end at different frames: own max_length, a row swapped in later — q3_session_replace)
    const int* limit;
    float* logits_hist; int hist_stride_b; int hist_cap;  // optional capture [B][cap][vocab] at index token_count
    int vocab, B;
    float inv_temp; int apply_temp; int greedy;
    int top_k; float top_p; int use_top_p;
    float rep_pen, rep_inv; int use_rep;
    int eos_id; int min_new_tokens; int codec_eos; int use_suppress;
};
hipError_t launch_sample(const SampleArgs& a, hipStream_t st);

// ---- codec decoder (f32, [C][L] layout) ----
struct ConvArgs {
    const float* x; const float* w; const float* b; float* y;
    int cin, cout, L, k, dil;
    const float* snake_a = nullptr; const float* snake_b = nullptr;   // SnakeBeta applied to x on load
    const float* resid = nullptr;    // y += resid
    const float* scale = nullptr;    // y = resid + scale[c] * conv   (layer scale / gamma), needs resid
    int act = 0;                     // 1 = GELU(erf), 2 = clamp(-1,1), 3 = ReLU, 4 = tanh(ReLU), 5 = sigmoid, 6 = ELU
    const float* post_a = nullptr; const float* post_ib = nullptr;   // SnakeBeta of the CONSUMER applied to the output
    float* y2 = nullptr;             // if set: y = raw output, y2 = activated output; else y = activated output
    const void* wpk = nullptr;       // bf16x3-packed copy of w (launch_pack_conv_w) → bf16 matrix-core kernel; else f32 MFMA
    int planes = 3;                  // bf16 planes per operand used by the matrix-core kernel: 3 = f32-exact products, 2 = ~2^-17
};
// one-time split of f32 conv weights [cout][cin][K] into bf16 hi/mid/lo MFMA A-operand tiles (cout % 32 == 0, cin % 16 == 0)
hipError_t launch_pack_conv_w(const float* w, void* out, int cout, int cin, int K, hipStream_t st);
size_t packed_conv_w_bytes(int cout, int cin, int K);
hipError_t launch_conv1d(const ConvArgs& a, hipStream_t st);
// a whole residual unit (decoder_block.rs:81-92) in one launch: y += conv1x1(snake_mid(conv7_dil(xa) + b1)) + b2, ya = snake_post(y).
// hipErrorNotSupported (nothing launched) outside 96 / 192 channels with packed weights: the caller runs the two convs.
struct ResUnitArgs {
    const float* xa; float* y; float* ya;
    const void* w1pk; const void* w2pk; const float* b1; const float* b2;
    const float* mid_a; const float* mid_ib; const float* post_a; const float* post_ib;
    int C, L, dil;
    int planes = 3;                  // as ConvArgs::planes
};
hipError_t launch_resunit(const ResUnitArgs& r, hipStream_t st);
// polyphase transposed conv: wp = per-phase causal-conv weights [stride][cout][cin][taps]
hipError_t launch_transconv1d_taps(const float* x, const float* wp, const float* b, float* y, int cin, int cout, int L,
                                   int stride, int taps, const float* snake_a, const float* snake_ib, hipStream_t st,
                                   const float* post_a = nullptr, const float* post_ib = nullptr, float* y2 = nullptr,
                                   const void* wpk = nullptr,    // wpk: per-phase packed weights, phase-major
                                   int planes = 3);
hipError_t launch_snake_tables(const float* alpha, const float* beta, float* a, float* ib, int C, hipStream_t st);
hipError_t launch_dwconv7(const float* x, const float* w, const float* b, float* y, int C, int L, hipStream_t st);
hipError_t launch_layernorm_c(const float* x, const float* w, const float* b, float* y, int C, int L, float eps,
                              hipStream_t st);
hipError_t launch_rmsnorm_c(const float* x, const float* w, float* y, int C, int L, float eps, hipStream_t st);
hipError_t launch_rope_c(float* q, float* k, const float* cs, const float* sn, int nh, int hd, int L, hipStream_t st);
hipError_t launch_attn_c(const float* q, const float* k, const float* v, float* o, int nh, int hd, int L, float scale,
                         hipStream_t st);
hipError_t launch_silu_mul(const float* g, const float* u, float* y, int64_t n, hipStream_t st);
hipError_t launch_rvq_embed(const uint32_t* frames, int n_frames, const float* first_cb, const float* const* rest_cbs,
                            float* first_out, float* rest_out, int cb_dim, int cb_size, hipStream_t st);
hipError_t launch_norm_codebook(const float* esum, const float* usage, float* out, int rows, int dim, hipStream_t st);

#if defined(__HIPCC__)
// Separately rounded products / sums / differences (the reference's and the oracle's operation order: every product rounded before it is
// added). ROCm's __fmul_rn / __fadd_rn / __fsub_rn are plain `x * y` / `x + y` / `x - y`, which hipcc is free to contract into an FMA with a
// neighbouring operation — and did or did not from one template instance of the same source line to the next (the RoPE of k_attn_cp:
// v_pk_fma_f32 in one instance, two roundings in the other). These carry no contract flag, so nothing fuses with them.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
// ---- activation transport of the frame (round 6) --------------------------------------------------------------
// Everything one kernel of a frame hands to a LATER kernel of the same frame (activations, the code predictor's K/V rows,
// logits, partial records, zeroed split-K targets) is stored WRITE-THROUGH (sc1: the bytes leave the XCD's L2 for the fabric,
// where the other XCDs' L2s are kept coherent) and drained (act_drain: s_waitcnt vmcnt(0) before the wave ends), and is read
// with sc1 loads (never served by a CU's vector L1, the one cache no other CU's store ever refreshes). Between two kernels
// that both keep to this, a packet boundary needs neither the release fence (L2 write-back) nor the acquire fence (L1 / scalar
// cache invalidate) HIP puts on every packet — the library's own queue (q3_aql.cpp) then submits those nodes without them,
// which is worth ~0.3 us per dependent node. The accesses are also correct WITH the fences: hipGraphLaunch replays the
// very same kernels. A 16-byte access is a raw buffer load / store with the sc1 cache policy (aux = 16) through a descriptor of
// the array's base pointer (a relaxed agent-scope __hip_atomic_load / store lowers to an sc1 access only up to 8 bytes).
// What must NOT go through these: anything the scalar unit fetches (s_load is served by the scalar cache, which only the
// acquire fence invalidates) — per-frame counters are written by the frame's last kernel and the first packet of every frame
// keeps its acquire fence (q3_session.hip: frame_fence_policy).
typedef __attribute__((ext_vector_type(4))) float act_f4_t;
typedef __attribute__((ext_vector_type(2))) float act_f2_t;
constexpr int ACT_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t act_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 act_ld4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const act_f4_t v = __builtin_bit_cast(act_f4_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, ACT_SC1));
    return float4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ float2 act_ld2(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const act_f2_t v = __builtin_bit_cast(act_f2_t, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, ACT_SC1));
    return float2{v[0], v[1]};
}
__device__ __forceinline__ float act_ld1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, ACT_SC1));
}
__device__ __forceinline__ void act_st4(__amdgpu_buffer_rsrc_t r, int byte_off, const float4& v) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const act_f4_t f = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, f), r, byte_off, 0, ACT_SC1);
}
__device__ __forceinline__ void act_st2(__amdgpu_buffer_rsrc_t r, int byte_off, const float2& v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const act_f2_t f = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, f), r, byte_off, 0, ACT_SC1);
}
__device__ __forceinline__ void act_st1(__amdgpu_buffer_rsrc_t r, int byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, ACT_SC1);
}
// every wave that stored: its write-through stores (and no-return atomics) have reached the fabric before it ends
__device__ __forceinline__ void act_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// q|k|v columns col0 / col1 (.. +VEC-1) of activation row b from the slice sums (AttnArgs::qkv_part): the arithmetic of
// k_wide_epilogue<EPI_NONE, RMS> — v = 0; v += slice s (ascending); v / sqrt(sum_s ssq / K + eps) — with every load of
// the eight possible slices in flight at once
template <typename V, int NC>
__device__ __forceinline__ void qkv_from_slices(const AttnArgs& a, int b, const int (&col)[NC], V (&out)[NC]) {
    // (the slices come from the GEMM launch in front of the caller: L1-bypassing loads, see "activation transport" below)
    const int plane = a.B * a.ld_qkv;
    const __amdgpu_buffer_rsrc_t p0 = act_rsrc(a.qkv_part + (size_t)b * a.ld_qkv), qs = act_rsrc(a.qkv_ssq);
    V x[NC][8]; float q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = i < a.qkv_S ? i : a.qkv_S - 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if constexpr (sizeof(V) == 8) x[c][i] = act_ld2(p0, (s * plane + col[c]) * 4);
            else x[c][i] = act_ld1(p0, (s * plane + col[c]) * 4);
        }
        q[i] = act_ld1(qs, (s * a.B + b) * 4);
    }
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < a.qkv_S) tot += q[i];
    const float den = sqrtf(tot / (float)a.qkv_K + a.qkv_eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if constexpr (sizeof(V) == 8) {
            float2 v = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < a.qkv_S) { v.x += x[c][i].x; v.y += x[c][i].y; }
            out[c].x = v.x / den; out[c].y = v.y / den;
        } else {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < a.qkv_S) v += x[c][i];
            out[c] = v / den;
        }
    }
}
// zero side job (LinArgs::zero / AttnArgs::zero): workgroup `wg` of `nwg` clears its share with 16-byte write-through stores
__device__ __forceinline__ void zero_job(float* z, int n, int wg, int nwg, int tid, int nthreads) {
    if (!z) return;
    const int per = (((n + nwg - 1) / nwg) + 3) & ~3;
    const int beg = wg * per, end = (beg + per) < n ? (beg + per) : n;
    const __amdgpu_buffer_rsrc_t zr = act_rsrc(z);
    for (int i = beg + tid * 4; i < end; i += nthreads * 4) act_st4(zr, i * 4, float4{0.f, 0.f, 0.f, 0.f});
}
#endif

}  // namespace q3
