// q3_kernels_codec.hip — gfx950 kernels of the 12 Hz codec decoder / vocoder (always F32, like the
// reference: lib.rs:344-353). All activations are [C][L] row-major (time contiguous), so every conv
// and every "linear" (a K=1 conv) shares one LDS-tiled kernel with the input tile + left halo
// dil*(k-1) staged once per input-channel chunk; SnakeBeta is fused into the tile load, bias /
// GELU / layer-scale / residual / clamp into the store.
//
// Reference anchors: decoder_12hz.rs:411-505 (decode), 536-691 (pre-transformer), causal_conv.rs:94-103,
// causal_trans_conv.rs:88-100, convnext_block.rs:110-141, decoder_block.rs:81-92/240-247,
// snake_beta.rs:58-77.
#include "q3_kernels.h"

#include <math.h>
#include <stdlib.h>

namespace q3 {

// sin(x)^2 for SnakeBeta. libm-grade sinf costs ~200 lane-operations here (the vocoder evaluates it 1.8e9 times per
// 640 frames: 10 of 50 ms); only the square is needed, so the quadrant sign drops out: k = rint(x/pi), r = x - k*pi by a
// three-term Cody-Waite reduction (exact FMAs; k*PI_A exact for |k| < 2^13), r in [-pi/2, pi/2], odd Taylor polynomial
// to r^11 (truncation 6e-8 at the end points). |x| >= 8192 falls back to sinf.
__device__ __forceinline__ float sin_sq(float x) {
    if (fabsf(x) >= 8192.0f) { const float s = sinf(x); return s * s; }
    const float k = rintf(x * 0.31830988618379067154f);
    float r = fmaf(-k, 3.140625f, x);
    r = fmaf(-k, 9.67502593994140625e-4f, r);
    r = fmaf(-k, 1.509957990978376432e-7f, r);
    const float z = r * r;
    float p = fmaf(z, -2.50521083854417188e-8f, 2.75573192239858907e-6f);
    p = fmaf(z, p, -1.98412698412698413e-4f);
    p = fmaf(z, p, 8.33333333333333333e-3f);
    p = fmaf(z, p, -1.66666666666666667e-1f);
    const float s = fmaf(r * z, p, r);
    return s * s;
}
__device__ __forceinline__ float snake_f(float x, float a, float ib) { return add_rn(x, mul_rn(sin_sq(x * a), ib)); }   // unfused, as the reference

// ------------------------------------------------------------------------------------------------
// Generic causal conv1d:  y[co][t*ostride + ooff] = epi( b[co] + Σ_ci Σ_kk w[co][ci][kk] · f(x[ci][t-(k-1-kk)·dil]) )
// Block tile: 32 output channels × 128 time steps, 256 threads, 4×4 outputs per thread;
// input channels in chunks of 16 through LDS.
// ------------------------------------------------------------------------------------------------
constexpr int CV_CO = 32, CV_T = 128, CV_CI = 16, CV_MAXK = 7, CV_MAXHALO = 54;

// epilogue activations: 1 GELU(erf) (ConvNeXt), 2 clamp (applied after the residual, see the kernels), and the speaker
// encoder's (speaker.rs:53-61, 136-138, 327-329): 3 ReLU, 4 ReLU then tanh, 5 sigmoid = 1 / (exp(-x) + 1); the speech
// encoder's (Mimi SEANet): 6 ELU
__device__ __forceinline__ float conv_act(float v, int act) {
    if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 3) return fmaxf(v, 0.0f);
    if (act == 4) return tanhf(fmaxf(v, 0.0f));
    if (act == 5) return 1.0f / (expf(-v) + 1.0f);
    if (act == 6) return v > 0.0f ? v : expf(v) - 1.0f;
    return v;
}

struct ConvDev {
    const float* x; const float* w; const float* b; float* y;
    int cin, cout, L, k, dil;
    const float* snake_a; const float* snake_ib;
    const float* resid; const float* scale;
    int act;
    int ostride, ooff, oL;        // output indexing (transposed conv phases write strided)
    size_t w_phase_stride;        // weight offset per blockIdx.z
    int ooff_phase;               // output offset added per blockIdx.z
    const float* post_a; const float* post_ib;   // SnakeBeta applied to the OUTPUT (the consumer's activation)
    float* y2;                    // if set: y gets the raw value, y2 the activated one; else y gets the activated value
    const void* wpk;              // bf16x3-packed weights (launch_pack_conv_w) or nullptr
    size_t wpk_phase_stride;      // 16-byte units per phase
    int phases;                   // k_conv_bf16x3: phases folded into its 1-D grid (set by the launcher)
    int planes;                   // bf16 planes per operand in the matrix-core kernels: 3 (f32-exact products) or 2
};

__global__ __launch_bounds__(256) void k_conv1d(ConvDev a) {
    __shared__ float xs[CV_CI][CV_T + CV_MAXHALO + 2];
    __shared__ __attribute__((aligned(16))) float ws[CV_CI][CV_MAXK][CV_CO];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int t0 = blockIdx.x * CV_T, co0 = blockIdx.y * CV_CO;
    const float* w = a.w + (size_t)blockIdx.z * a.w_phase_stride;
    const int ooff = a.ooff + (int)blockIdx.z * a.ooff_phase;
    const int halo = (a.k - 1) * a.dil;
    const int W = CV_T + halo;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

    for (int ci0 = 0; ci0 < a.cin; ci0 += CV_CI) {
        __syncthreads();
        // stage x tile (with SnakeBeta applied) — coalesced along t
        for (int e = tid; e < CV_CI * W; e += 256) {
            const int ci = e / W, tt = e - ci * W;
            const int c = ci0 + ci, t = t0 - halo + tt;
            float v = 0.0f;
            if (c < a.cin && t >= 0 && t < a.L) {
                v = a.x[(size_t)c * a.L + t];
                if (a.snake_a) v = snake_f(v, a.snake_a[c], a.snake_ib[c]);
            }
            xs[ci][tt] = v;
        }
        // stage weights [ci][kk][co]
        for (int e = tid; e < CV_CI * a.k * CV_CO; e += 256) {
            const int co = e % CV_CO, r = e / CV_CO, kk = r % a.k, ci = r / a.k;
            const int c = ci0 + ci, o = co0 + co;
            ws[ci][kk][co] = (c < a.cin && o < a.cout) ? w[((size_t)o * a.cin + c) * a.k + kk] : 0.0f;
        }
        __syncthreads();
        const int nci = (a.cin - ci0) < CV_CI ? (a.cin - ci0) : CV_CI;
        for (int ci = 0; ci < nci; ++ci) {
            for (int kk = 0; kk < a.k; ++kk) {
                const float4 wv = *reinterpret_cast<const float4*>(&ws[ci][kk][ty * 4]);
                const int xo = tx + kk * a.dil;
                float xv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = xs[ci][xo + 32 * j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][j] = fmaf(wv.x, xv[j], acc[0][j]);
                    acc[1][j] = fmaf(wv.y, xv[j], acc[1][j]);
                    acc[2][j] = fmaf(wv.z, xv[j], acc[2][j]);
                    acc[3][j] = fmaf(wv.w, xv[j], acc[3][j]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = co0 + ty * 4 + i;
        if (o >= a.cout) continue;
        const float bias = a.b ? a.b[o] : 0.0f;
        const float sc = a.scale ? a.scale[o] : 1.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + tx + 32 * j;
            if (t >= a.L) continue;
            float v = acc[i][j] + bias;
            v = conv_act(v, a.act);
            if (a.scale) v = v * sc;
            const size_t oi = (size_t)o * a.oL + (size_t)t * a.ostride + ooff;
            if (a.resid) v = a.resid[oi] + v;
            if (a.act == 2) v = fminf(fmaxf(v, -1.0f), 1.0f);
            if (a.post_a) {
                const float va = snake_f(v, a.post_a[o], a.post_ib[o]);
                if (a.y2) { a.y[oi] = v; a.y2[oi] = va; } else a.y[oi] = va;
            } else {
                a.y[oi] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Same conv as an implicit GEMM on the f32 matrix cores: v_mfma_f32_32x32x2_f32 is an exact f32 fmaf chain
// at the f32 vector peak (157 TF) that leaves the VALU free for the SnakeBeta staging and the epilogue.
// Reduction index r = ci*K + kk; one MFMA consumes two consecutive r: A[i][k] = W[co0+i][r0+k] (lane i = l&31,
// k = l>>5), B[k][j] = x'[ci(r0+k)][t0 + j - (K-1-kk)·dil]. Block = 4 waves; CO_W = 2: 64 co × 128 t (each wave
// 32 co × 64 t = two 32×32 accumulators sharing one A fragment); CO_W = 1: 32 co × 256 t (cout = 96).
// LDS: x tile [16 ci][T_T + halo] (SnakeBeta applied once at staging), w tile [16·K][32·CO_W] (co fastest).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int K, int CO_W, int CI_T>
__global__ __launch_bounds__(256) void k_conv1d_mfma(ConvDev a) {
    constexpr int CO_T = 32 * CO_W, T_W = 4 / CO_W, T_T = 64 * T_W;
    constexpr int R_T = CI_T * K;
    constexpr int XW = T_T + (K > 1 ? CV_MAXHALO : 0) + 2;
    __shared__ float xs[CI_T][XW];
    __shared__ float ws[R_T][CO_T + 1];                 // +1: conflict-free transposing stores
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wco = wave % CO_W, wt = wave / CO_W;
    const int t0 = blockIdx.x * T_T, co0 = blockIdx.y * CO_T;
    const float* w = a.w + (size_t)blockIdx.z * a.w_phase_stride;
    const int ooff = a.ooff + (int)blockIdx.z * a.ooff_phase;
    const int halo = (K - 1) * a.dil;
    const int W = T_T + halo;
    f32x16_t acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    const int li = lane & 31, lk = lane >> 5;

    for (int ci0 = 0; ci0 < a.cin; ci0 += CI_T) {
        __syncthreads();
        // x tile: one wave per input-channel row, lanes along time (coalesced, no index division)
        for (int ci = wave; ci < CI_T; ci += 4) {
            const int c = ci0 + ci;
            const float* xrow = a.x + (size_t)c * a.L;
            const bool cok = c < a.cin;
            float sa = 0.0f, sib = 0.0f;
            if (a.snake_a && cok) { sa = a.snake_a[c]; sib = a.snake_ib[c]; }
            for (int tt = lane; tt < W; tt += 64) {
                const int t = t0 - halo + tt;
                float v = 0.0f;
                if (cok && t >= 0 && t < a.L) {
                    v = xrow[t];
                    if (a.snake_a) v = snake_f(v, sa, sib);
                }
                xs[ci][tt] = v;
            }
        }
        // w tile: for a fixed output channel the chunk's (ci, kk) run is contiguous in memory → coalesced reads
        const int r_valid = (a.cin - ci0) * K;
        for (int e = tid; e < R_T * CO_T; e += 256) {
            const int co = e / R_T, r = e - co * R_T;
            const int o = co0 + co;
            ws[r][co] = (o < a.cout && r < r_valid) ? w[((size_t)o * a.cin + ci0) * K + r] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int r0 = 0; r0 < R_T; r0 += 2) {
            const int r = r0 + lk;
            const int ci = r / K, kk = r - ci * K;
            const float av = ws[r][wco * 32 + li];
            const float* xrow = &xs[ci][wt * 64 + li + kk * a.dil];
            const float b0 = xrow[0], b1 = xrow[32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int t = t0 + wt * 64 + half * 32 + li;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = co0 + wco * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            if (o >= a.cout || t >= a.L) continue;
            float v = (half == 0 ? acc0[reg] : acc1[reg]) + (a.b ? a.b[o] : 0.0f);
            v = conv_act(v, a.act);
            if (a.scale) v = v * a.scale[o];
            const size_t oi = (size_t)o * a.oL + (size_t)t * a.ostride + ooff;
            if (a.resid) v = a.resid[oi] + v;
            if (a.act == 2) v = fminf(fmaxf(v, -1.0f), 1.0f);
            if (a.post_a) {
                const float va = snake_f(v, a.post_a[o], a.post_ib[o]);
                if (a.y2) { a.y[oi] = v; a.y2[oi] = va; } else a.y[oi] = va;
            } else {
                a.y[oi] = v;
            }
        }
    }
}

template <int K>
static hipError_t launch_conv_mfma_k(const ConvDev& a, int phases, hipStream_t st) {
    if (a.cout % 64 == 0) {
        dim3 grid((a.L + 127) / 128, a.cout / 64, phases);
        hipLaunchKernelGGL((k_conv1d_mfma<K, 2, 16>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid((a.L + 255) / 256, (a.cout + 31) / 32, phases);
        hipLaunchKernelGGL((k_conv1d_mfma<K, 1, 16>), grid, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}
// returns hipErrorNotSupported when the shape should use the VALU kernel
static hipError_t launch_conv_mfma(const ConvDev& a, int phases, hipStream_t st) {
    static const bool off = getenv("Q3_CONV_VALU") != nullptr;       // A/B aid
    if (off || a.cout < 32) return hipErrorNotSupported;
    switch (a.k) {
        case 1: return launch_conv_mfma_k<1>(a, phases, st);
        case 2: return launch_conv_mfma_k<2>(a, phases, st);
        case 3: return launch_conv_mfma_k<3>(a, phases, st);
        case 7: return launch_conv_mfma_k<7>(a, phases, st);
        default: return hipErrorNotSupported;
    }
}

// ------------------------------------------------------------------------------------------------
// Third generation: implicit GEMM on the bf16 matrix cores at f32-equivalent accuracy ("bf16x3").
// Weights AND activations are split exactly into three bf16 terms (w = wh + wm + wl, x = xh + xm + xl, 24 mantissa
// bits each); a product w·x is formed by six v_mfma_f32_32x32x16_bf16 — hh, hm, mh, hl, lh, mm; the dropped
// ml / lm / ll terms are below 2^-24 relative — with bf16·bf16 products exact in the f32 accumulator. Six bf16 MFMAs
// cost 6/16 of one f32 MFMA per FLOP: 417 TF/s f32-equivalent peak against 157 TF/s of v_mfma_f32_32x32x2_f32.
//   * weights: split once at model finalize (launch_pack_conv_w) into MFMA A-operand tiles, order
//     [co/32][kk][ci/16][plane][lane]: slot `lane` (row i = lane%32, k-group = lane/32) = 8 bf16 of
//     W[co0+i][ci0 + 8·(lane/32) .. +8][kk] — one coalesced 16-B load per lane per plane;
//   * activations: split once per workgroup at LDS staging (SnakeBeta applied there too) into three planes laid out
//     [t][32 ci] bf16 with an 80-byte row pitch, so the B operand of column t (8 consecutive ci) is one 16-byte
//     aligned ds_read_b128 and a tap is just a row offset kk·dil;
//   * reduction order per output: ci chunks of 16 ascending, taps ascending inside a 32-channel stage — fixed and
//     independent of the output position (the segment-exact decode of the engine relies on that).
// Workgroup = WCO x WT waves; wave tile = (32·CO_M) co x (32·T_M) t.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 cbf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int cu32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 cbf16x2_t;
typedef __attribute__((ext_vector_type(2))) float cf32x2_t;
constexpr int XP = 80;               // LDS bytes per time row per plane: 32 ci x 2 B + 16 pad

__device__ __forceinline__ uint32_t cpk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32, RNE (builtin: see gemv file)
    const cf32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, cbf16x2_t));
}
// exact 3-way split of a pair: returns packed (a | b << 16) per plane
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cpk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = cpk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = cpk_bf16(sa, sb);
}

// pack f32 conv weights [cout][cin][K] (one phase) into bf16x3 A-operand tiles; one thread per (tile, lane)
__global__ __launch_bounds__(256) void k_pack_conv_w(const float* __restrict__ w, cu32x4_t* __restrict__ out, int cout, int cin, int K) {
    const int nc16 = cin >> 4;
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;        // ((co32*K + kk)*nc16 + c16)*64 + lane
    const size_t n_slots = (size_t)(cout >> 5) * K * nc16 * 64;
    if (slot >= n_slots) return;
    const int lane = (int)(slot & 63);
    size_t tile = slot >> 6;
    const int c16 = (int)(tile % nc16); tile /= nc16;
    const int kk = (int)(tile % K); const int co32 = (int)(tile / K);
    const int co = co32 * 32 + (lane & 31), ci = c16 * 16 + (lane >> 5) * 8;
    cu32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = w[((size_t)co * cin + ci + 2 * e) * K + kk], b = w[((size_t)co * cin + ci + 2 * e + 1) * K + kk];
        uint32_t hh, mm, ll; split3_pair(a, b, hh, mm, ll);
        h[e] = hh; m[e] = mm; l[e] = ll;
    }
    cu32x4_t* o = out + ((slot >> 6) * 3) * 64 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

hipError_t launch_pack_conv_w(const float* w, void* out, int cout, int cin, int K, hipStream_t st) {
    if (cout % 32 || cin % 16 || K < 1) return hipErrorInvalidValue;
    const size_t n_slots = (size_t)(cout / 32) * K * (cin / 16) * 64;
    hipLaunchKernelGGL(k_pack_conv_w, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, st, w, (cu32x4_t*)out, cout, cin, K);
    return hipGetLastError();
}
size_t packed_conv_w_bytes(int cout, int cin, int K) { return (size_t)(cout / 32) * K * (cin / 16) * 3 * 1024; }

// NP = 2: the h and m planes only (products hm + mh + hh): operands carry 16-17 mantissa bits, a product is good to
// ~2^-17 — half the MFMAs, two thirds of the fragment and LDS traffic (the opt-in "x2" vocoder mode)
template <int NP>
__device__ __forceinline__ f32x16_t mfmaP(const cu32x4_t (&A)[NP], const cu32x4_t (&B)[NP], f32x16_t acc) {
#define Q3_MF(ap, bp) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8_t, A[ap]), __builtin_bit_cast(cbf16x8_t, B[bp]), acc, 0, 0, 0)
    if constexpr (NP == 3) { Q3_MF(1, 1); Q3_MF(2, 0); Q3_MF(0, 2); }
    Q3_MF(1, 0); Q3_MF(0, 1); Q3_MF(0, 0);
#undef Q3_MF
    return acc;
}

// Summation order of the 1x1 convs whose cin is a multiple of 512 (the pre-transformer / ConvNeXt linears): 128-channel
// segments are accumulated from zero and added to the total in ascending order — the order k_lin_small_bf16x3 (short
// sequences, one wave per segment) can reproduce, so both kernels give a position the same bits.
__host__ __device__ __forceinline__ bool conv_segmented(int k, int cin) { return k == 1 && (cin & 511) == 0; }

// CIS = input channels per LDS stage: 32, or 128 for the 1x1 convs on 64-column tiles (with a single tap a 32-channel
// stage is two 16-channel MFMA steps between two barriers and a global round trip).
// LDS row pitch = CIS x 2 B + 16: an odd number of 16-byte slots, so the 16 rows of a ds_read_b128 group spread over all.
template <int K, int CO_M, int T_M, int WCO, int WT, bool PRE = false, bool SEG = false, int CIS = 32, int NP = 3>
__global__ __launch_bounds__(64 * WCO * WT, (CO_M == 2 || T_M >= 8) ? 2 : 3) void k_conv_bf16x3(ConvDev a) {     // waves per SIMD the register budget must allow
    constexpr int NT = 64 * WCO * WT;
    constexpr int CO_WG = 32 * CO_M * WCO, T_WG = 32 * T_M * WT;
    constexpr int XPB = CIS * 2 + 16, NOCT = CIS / 8;           // LDS bytes per time row per plane; channel octets per stage
    // ring of weight-fragment register sets. Two (one step of prefetch) for the tapped three-plane convs — a step is 24 MFMAs
    // and three sets cost more registers than they hide (k = 7: 918 -> 934 us; k = 2: 853 -> 887). Three where a step is
    // short: single-tap convs (416 -> 384, 307 -> 286, 199 -> 186 us) and everything in two-plane mode (12 MFMAs per step:
    // 640-frame decode 14.6 -> 14.3 ms). Prefetch depth only: same bits.
    constexpr int NA = (NP == 2 || K == 1) ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [3 planes][W rows][XP bytes]
    // per-output-row epilogue operands (bias, layer scale, the consumer's SnakeBeta pair), fetched once per workgroup:
    // read from global inside the store loop they cannot be hoisted over the stores and cost a round trip per row
    __shared__ float s_prm[4][CO_WG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: keeps tile / weight addresses scalar
    const int li = lane & 31, lk = lane >> 5;
    const int wco = wave % WCO, wt = wave / WCO;
    // XCD-aware tile order (1-D grid of nt x nco tiles): workgroup b runs on XCD b % 8, each XCD has its own L2, and the
    // nco channel tiles of one time tile all stage the same x. They are therefore made consecutive *on one XCD*
    // (ids 8 apart) instead of nt dispatches apart: the x tile comes from HBM once, not nco times
    // (profiles/r1_pmc_vocoder_hbm_T640.txt).
    // The phases of a polyphase transposed conv (a.phases > 1) join the same grouping: phase p writes every stride-th
    // output sample, so the phases of one tile fill the same cache lines — issued from one XCD back to back the lines
    // are completed in that L2; as separate grid planes on whichever XCD they reached HBM as partial-line writes
    // (PMC: 2.8 GB written per launch for a 0.47 GB output).
    int bx, by, bz;
    {
        const int P = a.phases, nco = a.cout / CO_WG, per = nco * P, nt = (int)gridDim.x / per, nt8 = nt & ~7, lin = (int)blockIdx.x;
        int r, tt;
        if (lin < nt8 * per) { r = lin >> 3; tt = (r / per) * 8 + (lin & 7); r = r % per; }
        else { const int rem = lin - nt8 * per; tt = nt8 + rem / per; r = rem % per; }
        bx = tt; bz = r % P; by = r / P;
    }
    const int t0 = bx * T_WG, co0 = by * CO_WG + wco * (32 * CO_M);
    for (int i = tid; i < CO_WG; i += NT) {
        const int o = by * CO_WG + i;
        const bool ok = o < a.cout;
        s_prm[0][i] = (a.b && ok) ? a.b[o] : 0.0f;
        s_prm[1][i] = (a.scale && ok) ? a.scale[o] : 1.0f;
        s_prm[2][i] = (a.post_a && ok) ? a.post_a[o] : 0.0f;
        s_prm[3][i] = (a.post_a && ok) ? a.post_ib[o] : 0.0f;
    }
    const int halo = (K - 1) * a.dil, W = T_WG + halo;
    const size_t plane = (size_t)W * XPB;
    const unsigned plane32 = (unsigned)W * XPB, bfrag = (unsigned)((wt * (32 * T_M) + li) * XPB + lk * 16);
    const cu32x4_t* __restrict__ wpk = reinterpret_cast<const cu32x4_t*>(a.wpk) + (size_t)bz * a.wpk_phase_stride;
    const int ooff = a.ooff + bz * a.ooff_phase;
    const int nc16 = a.cin >> 4;

    f32x16_t acc[CO_M][T_M];
#pragma unroll
    for (int i = 0; i < CO_M; ++i)
#pragma unroll
        for (int j = 0; j < T_M; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    constexpr int SM = SEG ? CO_M : 1, SN = SEG ? T_M : 1;
    f32x16_t tot[SM][SN];
    constexpr bool seg = SEG;                                     // the launcher instantiates SEG = conv_segmented(k, cin)
    if (SEG) {
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int j = 0; j < SN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.0f;
    }

    // weight fragments through a buffer descriptor: lane offset in a VGPR, tile offset in an SGPR — no address VALU per
    // load (as 64-bit pointer arithmetic the six loads of a step cost 13 VALU instructions, six of them 64-bit multiplies,
    // issued with the matrix pipe empty)
    const uint64_t wpa = reinterpret_cast<uint64_t>(wpk);          // wave-uniform, but derived through VALU (tile order): tell the compiler
    const uint64_t wpu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wpa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)wpa);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(wpu), 0, 0x7fffffff, 0x00020000);
    const int co32 = __builtin_amdgcn_readfirstlane(co0 >> 5);
    auto load_A = [&](cu32x4_t (&A)[CO_M][NP], int ci0, int kk, int c16l) {
#pragma unroll
        for (int cm = 0; cm < CO_M; ++cm) {
            const int tile = ((co32 + cm) * K + kk) * nc16 + (ci0 >> 4) + c16l;            // < 2^21 tiles of 3 KB
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                A[cm][pl] = __builtin_bit_cast(cu32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, (tile * 3 + pl) * 1024, 0));
        }
    };
    // (Requesting the NEXT stage's x into registers right after the publishing barrier, so that its round trip runs under
    // the MFMAs, was tried and lost: 24 more live VGPRs cost more than the hidden latency — 640-frame decode 30.4 -> 33 ms.)
    for (int ci0 = 0; ci0 < a.cin; ci0 += CIS) {
        const int n16 = (a.cin - ci0) >= CIS ? CIS / 16 : (a.cin - ci0) >> 4;
        // the first weight fragments of the stage do not depend on the staging: request them before the barrier
        const int n_steps = K * n16;
        cu32x4_t A[NA][CO_M][NP];
        load_A(A[0], ci0, 0, 0);
        if constexpr (NA == 3) load_A(A[1], ci0, (n_steps > 1 ? 1 : 0) / n16, (n_steps > 1 ? 1 : 0) % n16);
        __syncthreads();
        // stage [W rows][32 ci] of x. Work item = (time row, channel octet): consecutive threads take consecutive rows
        // (coalesced global reads along t), 8 loads in flight, split, one 16-byte LDS store per plane.
        // Work items = (64-row chunk, channel octet), dealt round-robin to the waves: wave-uniform, so no division by the
        // runtime tile width and scalar channel-row addresses. A wave's items are requested in batches of NB, ALL
        // their loads unconditional (clamped addresses, zeroed by a select afterwards): a load guarded by a run-time
        // condition makes hipcc branch around it and wait for it, one memory round trip per item.
        constexpr int NCHK = (T_WG + (K - 1) * 9 + 63) / 64, NBF = (NOCT * NCHK + WCO * WT - 1) / (WCO * WT);
        constexpr int NB = NBF < 1 ? 1 : (NBF > 4 ? 4 : NBF);        // a wave's whole share in one batch up to dilation 9 (3 waves: 4 items)
        const int n_items = NOCT * ((W + 63) >> 6);
        for (int c0 = wave; c0 < n_items; c0 += NB * WCO * WT) {
            float v[NB][8];
            bool ok[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = c0 + i * WCO * WT;
                const int q = c % NOCT, tt = (c / NOCT) * 64 + lane;
                const int ca = ci0 + q * 8, t = t0 - halo + tt;
                ok[i] = c < n_items && tt < W && ca < a.cin && t >= 0 && t < a.L;     // cin % 8 == 0: the octet is all in or all out
                const int cc = ca < a.cin ? ca : a.cin - 8, tc = t < 0 ? 0 : (t < a.L ? t : a.L - 1);
                const float* xr = a.x + (size_t)cc * a.L + tc;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = xr[(size_t)e * a.L];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = c0 + i * WCO * WT;
                if (c >= n_items) break;                             // wave-uniform
                const int q = c % NOCT, tt = (c / NOCT) * 64 + lane;
                const int ca = ci0 + q * 8;
                if (a.snake_a && ca < a.cin) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[i][e] = snake_f(v[i][e], a.snake_a[ca + e], a.snake_ib[ca + e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = ok[i] ? v[i][e] : 0.0f;
                if (tt < W) {
                    cu32x4_t h, m, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { uint32_t hh, mm, ll; split3_pair(v[i][2 * e], v[i][2 * e + 1], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
                    unsigned char* row = smem + (unsigned)tt * XPB + q * 16;
                    *reinterpret_cast<cu32x4_t*>(row) = h;
                    *reinterpret_cast<cu32x4_t*>(row + plane) = m;
                    if constexpr (NP == 3) *reinterpret_cast<cu32x4_t*>(row + 2 * plane) = l;
                }
            }
        }
        __syncthreads();
        // steps (kk, c16l) flattened; the weight fragments of step s + NA - 1 are requested before the MFMAs of step s
        auto do_step = [&](const cu32x4_t (&A)[CO_M][NP], int s) {
            const int kk = s / n16, c16l = s - kk * n16;
            cu32x4_t B[T_M][NP];
            // one VGPR add per plane and step: lane part fixed for the kernel (bfrag), step part scalar, column tiles as
            // immediate offsets
            const unsigned so = (unsigned)(kk * a.dil * XPB + c16l * 32);
            const unsigned char* bp0 = smem + (bfrag + so);
            const unsigned char* bp1 = smem + (bfrag + so + plane32);
            const unsigned char* bp2 = smem + (bfrag + so + 2 * plane32);
#pragma unroll
            for (int tm = 0; tm < T_M; ++tm) {
                B[tm][0] = *reinterpret_cast<const cu32x4_t*>(bp0 + tm * (32 * XPB));
                B[tm][1] = *reinterpret_cast<const cu32x4_t*>(bp1 + tm * (32 * XPB));
                if constexpr (NP == 3) B[tm][2] = *reinterpret_cast<const cu32x4_t*>(bp2 + tm * (32 * XPB));
            }
#pragma unroll
            for (int cm = 0; cm < CO_M; ++cm)
#pragma unroll
                for (int tm = 0; tm < T_M; ++tm) acc[cm][tm] = mfmaP<NP>(A[cm], B[tm], acc[cm][tm]);
        };
        // Steps in pairs of straight-line code, the prefetch UNCONDITIONAL (index clamped: the last pair re-requests the
        // last step). `s_waitcnt vmcnt` counts in issue order: behind a prefetch that sits in a conditional the compiler
        // cannot know how many younger loads are in flight and waits for all of them — vmcnt(0) right after issuing the
        // next step's fragments, i.e. no prefetch at all (skipping the in-loop fragment loads: 1030 -> 671 us).
        const int last = n_steps - 1;
        // (pinned: left to itself hipcc sinks the requests into the second half of the current step, where the previous
        // fragments' registers come free — half a step of cover for an L2 round trip)
        auto fetch = [&](cu32x4_t (&Ad)[CO_M][NP], int sp) {
            sp = sp < last ? sp : last;
            __builtin_amdgcn_sched_barrier(0); load_A(Ad, ci0, sp / n16, sp % n16); __builtin_amdgcn_sched_barrier(0);
        };
        int st = 0;
        if constexpr (NA == 3) {                                     // two steps of prefetch
            for (; st + 2 < n_steps; st += 3) {
                fetch(A[2], st + 2); do_step(A[0], st);
                fetch(A[0], st + 3); do_step(A[1], st + 1);
                fetch(A[1], st + 4); do_step(A[2], st + 2);
            }
            if (st < n_steps) do_step(A[0], st);
            if (st + 1 < n_steps) do_step(A[1], st + 1);
        } else {
            for (; st + 1 < n_steps; st += 2) {
                fetch(A[1], st + 1); do_step(A[0], st);
                fetch(A[0], st + 2); do_step(A[1], st + 1);
            }
            if (st < n_steps) do_step(A[0], st);
        }
        if (SEG && ((ci0 + CIS) & 127) == 0) {                    // segment boundary: total += segment sum
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int j = 0; j < SN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { tot[i][j][r] = tot[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.0f; }
        }
    }
    if (SEG) {
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
            for (int j = 0; j < SN; ++j) acc[i][j] = tot[i][j];
    }
    // epilogue: lane (li, lk) holds column t = li of rows (reg&3) + 8*(reg>>2) + 4*lk. PRE (residual epilogues of the
    // single-co-tile geometries): the residual values of a 32-row tile are requested in one batch before the first use —
    // inside the element loop they serialise into one memory round trip per element (96 channels: 1670 -> 950 us); on
    // the two-co-tile geometries the extra registers spill and it loses, so those keep the in-loop load.
#pragma unroll
    for (int cm = 0; cm < CO_M; ++cm) {
        float rs[PRE ? T_M : 1][16];
        if constexpr (PRE) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = co0 + cm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
#pragma unroll
                for (int tm = 0; tm < T_M; ++tm) {
                    const int t = t0 + wt * (32 * T_M) + tm * 32 + li;
                    rs[tm][reg] = (a.resid && o < a.cout && t < a.L) ? a.resid[(size_t)o * a.oL + (size_t)t * a.ostride + ooff] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = co0 + cm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            const bool ook = o < a.cout;
            const int pi = o - by * CO_WG;                      // row inside the workgroup's channel range
            const float bias = s_prm[0][pi], sc = s_prm[1][pi], pa = s_prm[2][pi], pib = s_prm[3][pi];
#pragma unroll
            for (int tm = 0; tm < T_M; ++tm) {
                const int t = t0 + wt * (32 * T_M) + tm * 32 + li;
                if (ook && t < a.L) {
                    float v = acc[cm][tm][reg] + bias;
                    v = conv_act(v, a.act);
                    if (a.scale) v = v * sc;
                    const size_t oi = (size_t)o * a.oL + (size_t)t * a.ostride + ooff;
                    if (a.resid) v = (PRE ? rs[PRE ? tm : 0][reg] : a.resid[oi]) + v;
                    if (a.act == 2) v = fminf(fmaxf(v, -1.0f), 1.0f);
                    if (a.post_a) {
                        const float va = snake_f(v, pa, pib);
                        if (a.y2) { a.y[oi] = v; a.y2[oi] = va; } else a.y[oi] = va;
                    } else {
                        a.y[oi] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A whole residual unit in one launch (decoder_block.rs:81-92: x + conv1(snake2(conv7_dil(snake1(x))))) for the layers
// whose channel range fits one workgroup (96 and 192 channels: 74 % of the decoder's activation bytes).
// As two launches the unit moves six C x L tensors through HBM (conv7: read act, write mid; 1x1: read mid, read
// residual, write raw, write act) and the 1x1 runs at 9-21 % MFMA occupancy, bound by exactly that traffic
// (profiles/r2_pmc_vocoder_mfma_T640.txt). Here the workgroup (C/32 waves, wave = 32 channels x 128 time columns) keeps
// the conv7 result on chip: bias + SnakeBeta in the accumulators, split into the three bf16 planes, parked in LDS in
// B-operand order (the x staging area is reused, 64 columns at a time, so the LDS footprint — and the occupancy — stay
// those of the conv7 kernel), multiplied by the 1x1 weights, residual added from the raw tensor IN PLACE (a column is
// read and written by the one workgroup that owns it), activated copy for the next consumer written beside it.
// Four tensors instead of six, one launch instead of two, the x tile staged once for all channels.
// Arithmetic per output element: exactly the two-launch sequence (same MFMA order, same split, same epilogue
// expressions) — bit-identical, tested.
// ------------------------------------------------------------------------------------------------
struct ResUnitDev {
    const float* xa;                 // snake1(x) [C][L]: the conv7 operand (written by the producer's epilogue)
    float* y;                        // raw x [C][L]: residual in, unit output out (in place)
    float* ya;                       // snake_next(output) [C][L] (must not alias xa: neighbours still read its halo), or nullptr
    const void* w1pk; const void* w2pk;
    const float* b1; const float* b2;
    const float* mid_a; const float* mid_ib;       // SnakeBeta between the two convs
    const float* post_a; const float* post_ib;     // the next consumer's SnakeBeta
    int C, L, dil;
};

template <int NW, int NP = 3>
__global__ __launch_bounds__(64 * NW, 3) void k_resunit_bf16x3(ResUnitDev a) {
    constexpr int K = 7, T_M = 4, NT = 64 * NW, C = 32 * NW, T_WG = 128;
    constexpr int XPB = 80, NOCT = 4;                             // x staging: 32 channels per stage (as k_conv_bf16x3)
    constexpr int XP2 = C * 2 + 16;                               // mid rows: all C channels of a column, odd number of 16-byte slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // max(x staging [3][W][80], mid [3][32][XP2])
    __shared__ float s_prm[6][C];                                 // b1, mid_a, mid_ib, b2, post_a, post_ib
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int t0 = (int)blockIdx.x * T_WG;
    for (int i = tid; i < C; i += NT) {
        s_prm[0][i] = a.b1 ? a.b1[i] : 0.0f; s_prm[1][i] = a.mid_a[i]; s_prm[2][i] = a.mid_ib[i];
        s_prm[3][i] = a.b2 ? a.b2[i] : 0.0f;
        s_prm[4][i] = a.post_a ? a.post_a[i] : 0.0f; s_prm[5][i] = a.post_a ? a.post_ib[i] : 0.0f;
    }
    const int halo = (K - 1) * a.dil, W = T_WG + halo;
    const unsigned plane32 = (unsigned)W * XPB, bfrag = (unsigned)(li * XPB + lk * 16);
    constexpr int nc16 = C / 16;

    f32x16_t acc[T_M];
#pragma unroll
    for (int j = 0; j < T_M; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    auto rsrc = [&](const void* p) {
        const uint64_t pa = reinterpret_cast<uint64_t>(p);
        const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)pa);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(pu), 0, 0x7fffffff, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t w1rs = rsrc(a.w1pk), w2rs = rsrc(a.w2pk);
    auto load_A = [&](cu32x4_t (&A)[NP], int ci0, int kk, int c16l) {
        const int tile = (wave * K + kk) * nc16 + (ci0 >> 4) + c16l;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            A[pl] = __builtin_bit_cast(cu32x4_t, __builtin_amdgcn_raw_buffer_load_b128(w1rs, lane * 16, (tile * 3 + pl) * 1024, 0));
    };
    // ---- conv7: the main loop of k_conv_bf16x3<7, 1, 4, NW, 1> ----
    for (int ci0 = 0; ci0 < C; ci0 += 32) {
        constexpr int n16 = 2, n_steps = K * n16;
        cu32x4_t A[2][NP];
        load_A(A[0], ci0, 0, 0);
        __syncthreads();
        constexpr int NCHK = (T_WG + (K - 1) * 9 + 63) / 64, NBF = (NOCT * NCHK + NW - 1) / NW;
        constexpr int NB = NBF < 1 ? 1 : (NBF > 4 ? 4 : NBF);
        const int n_items = NOCT * ((W + 63) >> 6);
        for (int c0 = wave; c0 < n_items; c0 += NB * NW) {
            float v[NB][8];
            bool ok[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = c0 + i * NW;
                const int q = c % NOCT, tt = (c / NOCT) * 64 + lane;
                const int ca = ci0 + q * 8, t = t0 - halo + tt;
                ok[i] = c < n_items && tt < W && t >= 0 && t < a.L;
                const int tc = t < 0 ? 0 : (t < a.L ? t : a.L - 1);
                const float* xr = a.xa + (size_t)ca * a.L + tc;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = xr[(size_t)e * a.L];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int c = c0 + i * NW;
                if (c >= n_items) break;                             // wave-uniform
                const int q = c % NOCT, tt = (c / NOCT) * 64 + lane;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = ok[i] ? v[i][e] : 0.0f;
                if (tt < W) {
                    cu32x4_t h, m, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { uint32_t hh, mm, ll; split3_pair(v[i][2 * e], v[i][2 * e + 1], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
                    unsigned char* row = smem + (unsigned)tt * XPB + q * 16;
                    *reinterpret_cast<cu32x4_t*>(row) = h;
                    *reinterpret_cast<cu32x4_t*>(row + plane32) = m;
                    if constexpr (NP == 3) *reinterpret_cast<cu32x4_t*>(row + 2 * plane32) = l;
                }
            }
        }
        __syncthreads();
        auto do_step = [&](const cu32x4_t (&Af)[NP], int s) {
            const int kk = s >> 1, c16l = s & 1;
            cu32x4_t B[T_M][NP];
            const unsigned so = (unsigned)(kk * a.dil * XPB + c16l * 32);
            const unsigned char* bp0 = smem + (bfrag + so);
            const unsigned char* bp1 = smem + (bfrag + so + plane32);
            const unsigned char* bp2 = smem + (bfrag + so + 2 * plane32);
#pragma unroll
            for (int tm = 0; tm < T_M; ++tm) {
                B[tm][0] = *reinterpret_cast<const cu32x4_t*>(bp0 + tm * (32 * XPB));
                B[tm][1] = *reinterpret_cast<const cu32x4_t*>(bp1 + tm * (32 * XPB));
                if constexpr (NP == 3) B[tm][2] = *reinterpret_cast<const cu32x4_t*>(bp2 + tm * (32 * XPB));
            }
#pragma unroll
            for (int tm = 0; tm < T_M; ++tm) acc[tm] = mfmaP<NP>(Af, B[tm], acc[tm]);
        };
        auto fetch = [&](cu32x4_t (&Ad)[NP], int sp) {
            sp = sp < n_steps - 1 ? sp : n_steps - 1;
            __builtin_amdgcn_sched_barrier(0); load_A(Ad, ci0, sp >> 1, sp & 1); __builtin_amdgcn_sched_barrier(0);
        };
        for (int st = 0; st < n_steps; st += 2) {                  // 14 steps: pairs of straight-line code, prefetch unconditional
            fetch(A[1], st + 1); do_step(A[0], st);
            fetch(A[0], st + 2); do_step(A[1], st + 1);
        }
    }
    // ---- the 1x1 conv over the activated conv7 result, 64 columns (two column tiles) at a time ----
    auto load_A2 = [&](cu32x4_t (&A)[NP], int s) {
        const int tile = wave * nc16 + s;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            A[pl] = __builtin_bit_cast(cu32x4_t, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane * 16, (tile * 3 + pl) * 1024, 0));
    };
    // One column tile (32 columns) at a time (round 5). With both tiles of a 64-column half in flight — two accumulator tiles,
    // their 32 residual values, six B fragments and the ring of weight fragments beside the other half's conv7 accumulators —
    // the kernel wanted 228 VGPRs and ran at three waves per SIMD only by spilling 36 of them in this phase; a tile at a time
    // needs 142, no spill. The parked tile is then 20 KB instead of 40, so the workgroup's LDS is the x staging's (32-44 KB)
    // and FOUR workgroups fit a CU for the dilation-1 and -3 units: twelve waves, three on every SIMD, where three workgroups
    // of three waves left the SIMDs at 3 / 2 / 2 / 2. The 1x1 weight fragments are fetched once per tile instead of once per
    // pair (18 KB per wave, L2-resident). Per output element the sequence of products and sums is unchanged: same bits.
    constexpr unsigned plane2 = 32u * XP2;
    const unsigned bfrag2 = (unsigned)(li * XP2 + lk * 16);
#pragma unroll
    for (int tq = 0; tq < T_M; ++tq) {
        cu32x4_t A2[3][NP];
        load_A2(A2[0], 0); load_A2(A2[1], 1);                       // in flight across the barriers and the activation below
        float rs[16];                                               // the residual of this tile, requested before anything waits
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            const int t = t0 + tq * 32 + li;
            rs[reg] = t < a.L ? a.y[(size_t)o * a.L + t] : 0.0f;
        }
        __syncthreads();                                            // every wave is done reading the area (x tiles / previous tile)
        {
            const f32x16_t& ac = acc[tq];
            unsigned char* row = smem + (unsigned)li * XP2 + wave * 64 + lk * 8;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pi = wave * 32 + 8 * g + 4 * lk + j;
                    v[j] = snake_f(ac[4 * g + j] + s_prm[0][pi], s_prm[1][pi], s_prm[2][pi]);
                }
                uint32_t h0, m0, l0, h1, m1, l1;
                split3_pair(v[0], v[1], h0, m0, l0); split3_pair(v[2], v[3], h1, m1, l1);
                *reinterpret_cast<uint2*>(row + g * 16) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(row + g * 16 + plane2) = make_uint2(m0, m1);
                if constexpr (NP == 3) *reinterpret_cast<uint2*>(row + g * 16 + 2 * plane2) = make_uint2(l0, l1);
            }
        }
        __syncthreads();
        f32x16_t acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
        auto step2 = [&](const cu32x4_t (&Af)[NP], int s) {
            cu32x4_t B[NP];
            const unsigned char* bp = smem + (bfrag2 + (unsigned)s * 32);
            B[0] = *reinterpret_cast<const cu32x4_t*>(bp);
            B[1] = *reinterpret_cast<const cu32x4_t*>(bp + plane2);
            if constexpr (NP == 3) B[2] = *reinterpret_cast<const cu32x4_t*>(bp + 2 * plane2);
            acc2 = mfmaP<NP>(Af, B, acc2);
        };
        auto fetch2 = [&](cu32x4_t (&Ad)[NP], int sp) {
            sp = sp < nc16 - 1 ? sp : nc16 - 1;
            __builtin_amdgcn_sched_barrier(0); load_A2(Ad, sp); __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(nc16 % 3 == 0, "ring of three");
        for (int s = 0; s < nc16; s += 3) {                          // two steps of prefetch: a step is only 6 MFMAs
            fetch2(A2[2], s + 2); step2(A2[0], s);
            fetch2(A2[0], s + 3); step2(A2[1], s + 1);
            fetch2(A2[1], s + 4); step2(A2[2], s + 2);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int pi = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
            const float bias = s_prm[3][pi], pa = s_prm[4][pi], pib = s_prm[5][pi];
            const int t = t0 + tq * 32 + li;
            if (t < a.L) {
                float v = acc2[reg] + bias;
                v = rs[reg] + v;
                const size_t oi = (size_t)pi * a.L + t;
                a.y[oi] = v;
                if (a.ya) a.ya[oi] = a.post_a ? snake_f(v, pa, pib) : v;
            }
        }
    }
}

static int conv_planes(int asked) { return asked == 2 ? 2 : 3; }      // ConvArgs::planes: anything but 2 means the exact products

hipError_t launch_resunit(const ResUnitArgs& r, hipStream_t st) {
    static const bool off = getenv("Q3_CONV_F32") != nullptr || getenv("Q3_CODEC_NO_UNIT_FUSE") != nullptr;    // A/B aids
    // 192 channels = six waves per workgroup: measured 1803 us per unit against 1050 + 414 for the two launches (the conv7
    // already prefers two 96-channel workgroups over one of 192: 1.32 vs 1.50 ms) — off unless Q3_CODEC_UNIT_FUSE_192=1
    static const bool fuse192 = getenv("Q3_CODEC_UNIT_FUSE_192") != nullptr;
    if (off || !r.w1pk || !r.w2pk || (r.C != 96 && !(r.C == 192 && fuse192)) || r.L < 4096 || (r.dil != 1 && r.dil != 3 && r.dil != 9) ||
        !r.mid_a || !r.xa || !r.y || r.ya == r.xa)
        return hipErrorNotSupported;
    ResUnitDev a{};
    a.xa = r.xa; a.y = r.y; a.ya = r.ya; a.w1pk = r.w1pk; a.w2pk = r.w2pk; a.b1 = r.b1; a.b2 = r.b2;
    a.mid_a = r.mid_a; a.mid_ib = r.mid_ib; a.post_a = r.post_a; a.post_ib = r.post_ib; a.C = r.C; a.L = r.L; a.dil = r.dil;
    const int W = 128 + 6 * r.dil;
    const int np = conv_planes(r.planes);
    const size_t lds_x = (size_t)np * W * 80, lds_mid = (size_t)np * 32 * (r.C * 2 + 16);       // x staging | one parked 32-column tile
    const size_t lds = lds_x > lds_mid ? lds_x : lds_mid;
    const dim3 grid((r.L + 127) / 128);
    if (np == 2) {
        if (r.C == 96) hipLaunchKernelGGL((k_resunit_bf16x3<3, 2>), grid, dim3(192), lds, st, a);
        else hipLaunchKernelGGL((k_resunit_bf16x3<6, 2>), grid, dim3(384), lds, st, a);
    } else {
        if (r.C == 96) hipLaunchKernelGGL((k_resunit_bf16x3<3, 3>), grid, dim3(192), lds, st, a);
        else hipLaunchKernelGGL((k_resunit_bf16x3<6, 3>), grid, dim3(384), lds, st, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Short sequences (streaming chunks: the pre-transformer and ConvNeXt linears at L <= 32). The tiled kernel above walks
// cin in 32 serial stages of (weight round trip, two barriers) with 8-16 workgroups on the chip: 29-52 us for a
// 10-column product whose 3-6 MB of weights stream in 2 us. Here a workgroup owns one 32-co tile and the single column
// tile; wave w takes the 128-channel segment w of cin, requests ALL its operands up front (24 KB of weight fragments
// + the x columns straight from global, split in registers), and the segment sums are added in ascending order through
// LDS. That is the segmented summation order conv_segmented() defines, which the tiled kernel follows for the same
// shapes — a frame position gets bit-identical values from either kernel (continuous streaming mode depends on it).
// ------------------------------------------------------------------------------------------------
template <int W, int NP = 3>
__global__ __launch_bounds__(64 * W) void k_lin_small_bf16x3(ConvDev a) {
    __shared__ float part[W][16][64];
    constexpr int NR = 16 / W;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    const int co32 = blockIdx.x, co0 = co32 * 32;
    const int tc = (int)blockIdx.y * 32 + li;                       // column tile blockIdx.y (one tile for the streaming chunks)
    const cu32x4_t* __restrict__ wpk = reinterpret_cast<const cu32x4_t*>(a.wpk);
    const float* __restrict__ x = a.x;
    const int nc16 = a.cin >> 4, nseg = a.cin >> 7;
    const bool tok = tc < a.L;
    float tot[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) tot[j] = 0.0f;
    for (int s0 = 0; s0 < nseg; s0 += W) {
        const int seg = s0 + wave;
        if (seg < nseg) {                                         // wave-uniform
            cu32x4_t A[8][NP];
            float xv[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const size_t tile = (size_t)co32 * nc16 + seg * 8 + i;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) A[i][pl] = wpk[(tile * 3 + pl) * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ci = seg * 128 + i * 16 + lk * 8 + e;
                    xv[i][e] = tok ? x[(size_t)ci * a.L + tc] : 0.0f;
                }
            if (a.snake_a) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ci = seg * 128 + i * 16 + lk * 8 + e;
                        if (tok) xv[i][e] = snake_f(xv[i][e], a.snake_a[ci], a.snake_ib[ci]);
                    }
            }
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cu32x4_t B[NP];
#pragma unroll
                for (int e = 0; e < 4; ++e) { uint32_t h, m, l; split3_pair(xv[i][2 * e], xv[i][2 * e + 1], h, m, l); B[0][e] = h; B[1][e] = m; if constexpr (NP == 3) B[2][e] = l; }
                acc = mfmaP<NP>(A[i], B, acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
        }
        __syncthreads();
        const int n = (nseg - s0) < W ? (nseg - s0) : W;
#pragma unroll
        for (int j = 0; j < NR; ++j)
            for (int q = 0; q < n; ++q) tot[j] = tot[j] + part[q][wave + j * W][lane];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int reg = wave + j * W;
        const int o = co0 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
        if (o < a.cout && tok) {
            float v = tot[j] + (a.b ? a.b[o] : 0.0f);
            v = conv_act(v, a.act);
            if (a.scale) v = v * a.scale[o];
            const size_t oi = (size_t)o * a.oL + (size_t)tc * a.ostride + a.ooff;
            if (a.resid) v = a.resid[oi] + v;
            if (a.act == 2) v = fminf(fmaxf(v, -1.0f), 1.0f);
            if (a.post_a) {
                const float va = snake_f(v, a.post_a[o], a.post_ib[o]);
                if (a.y2) { a.y[oi] = v; a.y2[oi] = va; } else a.y[oi] = va;
            } else {
                a.y[oi] = v;
            }
        }
    }
}

template <int K, int CO_M, int T_M, int WCO, int WT, int NP>
static hipError_t launch_bf16x3_vp(const ConvDev& a, int phases, hipStream_t st) {
    constexpr int CO_WG = 32 * CO_M * WCO, T_WG = 32 * T_M * WT;
    constexpr bool K1 = K == 1;
    // 1x1 convs on 64-column tiles (the pre-transformer / ConvNeXt linears): 128 channels per stage, 33 -> 29 and
    // 75 -> 68 us per launch; on 128-column tiles 64 channels per stage LOST (326 -> 476 us: the residual convs are
    // bound by their epilogue traffic, and the bigger stage only delays it). Q3_CONV_CIS32=1: 32 everywhere (A/B aid)
    static const bool cis32 = getenv("Q3_CONV_CIS32") != nullptr;
    constexpr int CIS1 = T_WG <= 64 ? 128 : 32;
    const bool wide = K1 && CIS1 > 32 && !cis32 && a.cin >= 2 * CIS1;
    const size_t lds = (size_t)NP * (T_WG + (K - 1) * a.dil) * ((wide ? CIS1 : 32) * 2 + 16);
    dim3 grid(((a.L + T_WG - 1) / T_WG) * (a.cout / CO_WG) * phases);         // tile order: see the kernel
    ConvDev ap = a; ap.phases = phases;
    const bool segm = conv_segmented(a.k, a.cin);
    const dim3 blk(64 * WCO * WT);
    if constexpr (K1) {
        const bool pre = CO_M == 1 && a.resid;
#define Q3_CV(P, S, C) hipLaunchKernelGGL((k_conv_bf16x3<K, CO_M, T_M, WCO, WT, P, S, C, NP>), grid, blk, lds, st, ap)
        if (wide) {
            if (segm) { if (pre) Q3_CV(CO_M == 1, true, CIS1); else Q3_CV(false, true, CIS1); }
            else { if (pre) Q3_CV(CO_M == 1, false, CIS1); else Q3_CV(false, false, CIS1); }
        } else {
            if (segm) { if (pre) Q3_CV(CO_M == 1, true, 32); else Q3_CV(false, true, 32); }
            else { if (pre) Q3_CV(CO_M == 1, false, 32); else Q3_CV(false, false, 32); }
        }
#undef Q3_CV
    // (64 channels per stage for the two-tap phases of the transposed convs — a 32-channel stage is only four MFMA steps
    // between two barriers — was measured and lost at every width: 614 -> 645, 791 -> 903, 880 -> 1209, 841 -> 1260 us;
    // the larger x tile costs the 96-channel geometry its third workgroup per CU)
    } else {
        hipLaunchKernelGGL((k_conv_bf16x3<K, CO_M, T_M, WCO, WT, false, false, 32, NP>), grid, blk, lds, st, ap);
    }
    return hipGetLastError();
}
template <int K, int CO_M, int T_M, int WCO, int WT>
static hipError_t launch_bf16x3_v(const ConvDev& a, int phases, hipStream_t st) {
    // two-plane instances only for the tap counts the vocoder's heavy layers use (1, 2, 7); its one k = 3 conv (pre_conv,
    // 0.06 ms) and the encoders' k = 3 / 5 convs always run three planes — in every decode mode alike, so chunked and
    // whole-utterance decodes still agree
    if constexpr (K == 1 || K == 2 || K == 7)
        if (a.planes == 2) return launch_bf16x3_vp<K, CO_M, T_M, WCO, WT, 2>(a, phases, st);
    return launch_bf16x3_vp<K, CO_M, T_M, WCO, WT, 3>(a, phases, st);
}
template <int K>
static hipError_t launch_bf16x3_k(const ConvDev& a, int phases, hipStream_t st) {
    const long big_tiles = (long)((a.L + 127) / 128) * (a.cout / 128) * phases;
    // 96- and 192-channel layers with taps: wave tiles of 128 time columns (T_M = 4) — a weight fragment serves 4 column
    // tiles instead of 2, half the fragment traffic through the texture path per MFMA (k = 7: 1182 -> 1034 us; the
    // 128-co geometry needs 256 VGPRs for it and loses). Q3_CONV_TM4=0: the T_M = 2 geometry (A/B aid)
    // ... and since round 3 also the 128-multiples (four waves x 32 co x 128 t instead of 2 x 2 waves x 64 x 64): k = 7
    // 808 -> 788 and 1023 -> 989 us, the 768 -> 384 transposed conv 791 -> 708 us. Q3_CONV_TM4=1: the round-2 choice
    static const int geo = [] { const char* e = getenv("Q3_CONV_TM4"); return e ? atoi(e) : 2; }();
    // (192 co x 64 t in ONE 3-wave workgroup — x staged once instead of twice, but twice the weight-fragment traffic per MFMA —
    // measured 1125-1137 us against 1064-1074 for two workgroups of 96 co x 128 t: rejected)
    if (geo && K >= 2 && a.L >= 4096 && (a.cout == 192 || a.cout == 96)) return launch_bf16x3_v<K, 1, 4, 3, 1>(a, phases, st);
    if (geo >= 2 && K >= 2 && a.L >= 4096 && a.cout % 128 == 0 && big_tiles >= 192) {      // experiment: 4 waves x (32 co x 128 / 256 t)
        if (geo == 2) return launch_bf16x3_v<K, 1, 4, 4, 1>(a, phases, st);
        if (geo == 3) return launch_bf16x3_v<K, 1, 8, 4, 1>(a, phases, st);
    }
    // 1x1 convs with a residual (second conv of every residual unit) are bound by their epilogue traffic, not by x
    // staging: the single-co-tile 64 co x 128 t geometry with batched residual loads wins at every width
    if (K == 1 && a.resid && a.cout % 64 == 0 && a.cout != 96 && big_tiles >= 192) return launch_bf16x3_v<K, 1, 2, 2, 2>(a, phases, st);
    // wide 1x1 convs without a residual on SMALL grids (the ConvNeXt pw1: 4096 output channels at 1280 / 2560 columns = 320 / 640
    // workgroups of 128 x 128 — barely more than one per CU, each walking all of K): 64 co x 64 t tiles fill the chip four times
    // over instead: 136 -> < 100 us and 243 -> 174 us (round 5; Q3_CONV_PW1_GEO=0 restores the 128 x 128 tiles: A/B aid)
    {
        static const int pw = [] { const char* e = getenv("Q3_CONV_PW1_GEO"); return e ? atoi(e) : 2; }();
        if (pw && K == 1 && !a.resid && a.cout % 128 == 0 && big_tiles >= 192 && big_tiles < 1024) {
            if (pw == 1) return launch_bf16x3_v<K, 1, 2, 2, 2>(a, phases, st);      // 64 co x 128 t
            return launch_bf16x3_v<K, 1, 1, 2, 2>(a, phases, st);                   // 64 co x 64 t
        }
    }
    if (a.cout % 128 == 0 && big_tiles >= 192) return launch_bf16x3_v<K, 2, 2, 2, 2>(a, phases, st);   // 128 co x 128 t (64 x 128: 5-13 % slower at k = 7)
    if (a.cout == 192) return launch_bf16x3_v<K, 1, 2, 3, 2>(a, phases, st);   // 2 x (96 co x 128 t): 1.32 ms vs 1.50 for 192 x 128 (k = 7)
    if (a.cout == 96) return launch_bf16x3_v<K, 1, 2, 3, 2>(a, phases, st);                             //  96 co x 128 t
    if (a.cout % 64 == 0) return launch_bf16x3_v<K, 1, 1, 2, 2>(a, phases, st);                         //  64 co x  64 t (small grids)
    return hipErrorNotSupported;
}
// returns hipErrorNotSupported when the shape has no packed weights / unsupported geometry
static hipError_t launch_conv_bf16x3(const ConvDev& a, int phases, hipStream_t st) {
    static const bool off = getenv("Q3_CONV_F32") != nullptr;        // A/B aid: force the f32-MFMA generation
    if (off || !a.wpk || a.cin % 16 || a.cout % 32) return hipErrorNotSupported;
    static const bool no_small = getenv("Q3_CONV_NO_SMALL") != nullptr;   // A/B aid
    // ... and, since it is the same bits, for the narrow deep linears of a whole utterance too (the pre-transformer's o / down
    // projections, 1024 -> 512 at 640 frames: 80 workgroups of the tiled kernel walk 8 serial stages, 38.8 us; as 320
    // one-shot workgroups of 32 x 32 outputs the chip is full once). Only while the grid stays within two rounds —
    // wider outputs (q|k|v, gate|up, the ConvNeXt linears) are already full grids and gain nothing.
    static const bool no_small_tiles = getenv("Q3_CONV_NO_SMALL_TILES") != nullptr;   // A/B aid
    const long small_wgs = (long)(a.cout / 32) * ((a.L + 31) / 32);
    const bool small_tiles = a.L > 32 && a.cin >= 1024 && small_wgs <= 512 && !no_small_tiles;
    if (conv_segmented(a.k, a.cin) && (a.L <= 32 || small_tiles) && phases == 1 && a.cout % 64 == 0 && !no_small) {
        const dim3 grid(a.cout / 32, (a.L + 31) / 32);
        if (a.planes == 2) {                                          // the mode's summation is the tiled kernel's in either form
            if (a.cin >= 1024) hipLaunchKernelGGL((k_lin_small_bf16x3<8, 2>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_lin_small_bf16x3<4, 2>), grid, dim3(256), 0, st, a);
        } else {
            if (a.cin >= 1024) hipLaunchKernelGGL((k_lin_small_bf16x3<8, 3>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_lin_small_bf16x3<4, 3>), grid, dim3(256), 0, st, a);
        }
        return hipGetLastError();
    }
    switch (a.k) {
        case 1: return launch_bf16x3_k<1>(a, phases, st);
        case 2: return launch_bf16x3_k<2>(a, phases, st);
        case 3: return launch_bf16x3_k<3>(a, phases, st);
        case 5: return launch_bf16x3_k<5>(a, phases, st);     // speaker encoder front TDNN
        case 7: return launch_bf16x3_k<7>(a, phases, st);
        default: return hipErrorNotSupported;
    }
}

// single-output-channel conv (final 96→1, k=7): one thread per time step
__global__ __launch_bounds__(256) void k_conv_out1(ConvDev a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.L) return;
    float acc = 0.0f;
    for (int c = 0; c < a.cin; ++c) {
        const float sa = a.snake_a ? a.snake_a[c] : 0.0f, sib = a.snake_a ? a.snake_ib[c] : 0.0f;
        for (int kk = 0; kk < a.k; ++kk) {
            const int ts = t - (a.k - 1 - kk) * a.dil;
            if (ts < 0) continue;
            float v = a.x[(size_t)c * a.L + ts];
            if (a.snake_a) v = snake_f(v, sa, sib);
            acc = fmaf(a.w[(size_t)c * a.k + kk], v, acc);
        }
    }
    float v = acc + (a.b ? a.b[0] : 0.0f);
    if (a.act == 2) v = fminf(fmaxf(v, -1.0f), 1.0f);
    a.y[t] = v;
}

// the same for k = 7, dilation 1, L % 4 == 0: four consecutive outputs per thread from three aligned float4 loads per
// channel (x[t-8 .. t+3]) instead of 28 scalar ones — the scalar kernel is bound by its 670 load instructions per output
// (0.62 ms per 640-frame decode for a 472 MB read). Per output the same fmaf chain (channels, then taps, ascending).
__global__ __launch_bounds__(256) void k_conv_out1_v4(ConvDev a) {
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= a.L) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = 0; c < a.cin; ++c) {
        const float* xr = a.x + (size_t)c * a.L + t;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 p0 = t >= 8 ? *reinterpret_cast<const float4*>(xr - 8) : z;
        const float4 p1 = t >= 4 ? *reinterpret_cast<const float4*>(xr - 4) : z;
        const float4 p2 = *reinterpret_cast<const float4*>(xr);
        float v[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
        if (a.snake_a) {
            const float sa = a.snake_a[c], sib = a.snake_ib[c];
#pragma unroll
            for (int i = 2; i < 12; ++i) v[i] = snake_f(v[i], sa, sib);      // snake(0) = 0: positions before the start stay 0
        }
        float w[7];
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) w[kk] = a.w[(size_t)c * 7 + kk];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) acc[j] = fmaf(w[kk], v[j + kk + 2], acc[j]);
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = acc[j] + (a.b ? a.b[0] : 0.0f);
        if (a.act == 2) v = fminf(fmaxf(v, -1.0f), 1.0f);
        o[j] = v;
    }
    *reinterpret_cast<float4*>(a.y + t) = make_float4(o[0], o[1], o[2], o[3]);
}

hipError_t launch_conv1d(const ConvArgs& c, hipStream_t st) {
    if (c.k > CV_MAXK || (c.k - 1) * c.dil > CV_MAXHALO || c.L <= 0) return hipErrorInvalidValue;
    ConvDev a{};
    a.x = c.x; a.w = c.w; a.b = c.b; a.y = c.y; a.cin = c.cin; a.cout = c.cout; a.L = c.L; a.k = c.k; a.dil = c.dil;
    a.snake_a = c.snake_a; a.snake_ib = c.snake_b; a.resid = c.resid; a.scale = c.scale; a.act = c.act;
    a.ostride = 1; a.ooff = 0; a.oL = c.L; a.w_phase_stride = 0; a.ooff_phase = 0;
    a.post_a = c.post_a; a.post_ib = c.post_ib; a.y2 = c.y2;
    a.wpk = c.wpk; a.wpk_phase_stride = 0; a.planes = conv_planes(c.planes);
    if (c.cout == 1) {
        if (c.k == 7 && c.dil == 1 && c.L % 4 == 0 && (((uintptr_t)c.x | (uintptr_t)c.y) & 15) == 0)
            hipLaunchKernelGGL(k_conv_out1_v4, dim3((c.L / 4 + 255) / 256), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_conv_out1, dim3((c.L + 255) / 256), dim3(256), 0, st, a);
    } else if (hipError_t e = launch_conv_bf16x3(a, 1, st); e != hipErrorNotSupported) {
        return e;
    } else if (hipError_t e = launch_conv_mfma(a, 1, st); e != hipErrorNotSupported) {
        return e;
    } else {
        dim3 grid((c.L + CV_T - 1) / CV_T, (c.cout + CV_CO - 1) / CV_CO, 1);
        hipLaunchKernelGGL(k_conv1d, grid, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

// Transposed conv, polyphase form. Host code pre-arranges the weight [cin][cout][k] into per-phase
// causal-conv weights wp[ph][cout][cin][taps] (taps = k/stride: tap 0 ↔ x[j-1] (w[.][.][ph+s]),
// last tap ↔ x[j] (w[.][.][ph])), so phase ph is a k=taps causal conv whose outputs land at
// t = j*stride + ph; the right-trim k - s of causal_trans_conv.rs:79 is implicit (length L*stride).
hipError_t launch_transconv1d_taps(const float* x, const float* wp, const float* b, float* y, int cin, int cout, int L,
                                   int stride, int taps, const float* snake_a, const float* snake_ib, hipStream_t st,
                                   const float* post_a, const float* post_ib, float* y2, const void* wpk, int planes) {
    ConvDev a{};
    a.planes = conv_planes(planes);
    a.wpk = wpk; a.wpk_phase_stride = packed_conv_w_bytes(cout, cin, taps) / 16;
    a.post_a = post_a; a.post_ib = post_ib; a.y2 = y2;
    a.x = x; a.w = wp; a.b = b; a.y = y; a.cin = cin; a.cout = cout; a.L = L; a.k = taps; a.dil = 1;
    a.snake_a = snake_a; a.snake_ib = snake_ib; a.resid = nullptr; a.scale = nullptr; a.act = 0;
    a.ostride = stride; a.ooff = 0; a.oL = L * stride;
    a.w_phase_stride = (size_t)cout * cin * taps; a.ooff_phase = 1;
    if (hipError_t e = launch_conv_bf16x3(a, stride, st); e != hipErrorNotSupported) return e;
    if (hipError_t e = launch_conv_mfma(a, stride, st); e != hipErrorNotSupported) return e;
    dim3 grid((L + CV_T - 1) / CV_T, (cout + CV_CO - 1) / CV_CO, stride);
    hipLaunchKernelGGL(k_conv1d, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// depthwise causal conv k=7 (ConvNeXt, convnext_block.rs:116)
__global__ __launch_bounds__(256) void k_dwconv7(const float* x, const float* w, const float* b, float* y, int L) {
    const int c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    const float* xr = x + (size_t)c * L;
    float acc = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
        const int ts = t - 6 + kk;
        if (ts >= 0) acc = fmaf(w[c * 7 + kk], xr[ts], acc);
    }
    y[(size_t)c * L + t] = acc + b[c];
}
hipError_t launch_dwconv7(const float* x, const float* w, const float* b, float* y, int C, int L, hipStream_t st) {
    hipLaunchKernelGGL(k_dwconv7, dim3((L + 255) / 256, C), dim3(256), 0, st, x, w, b, y, L);
    return hipGetLastError();
}

// channel-wise norms for [C][L] tensors: block = 32 time steps × 8 channel groups
template <bool LAYERNORM>
__global__ __launch_bounds__(256) void k_norm_c(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                float* __restrict__ y, int C, int L, float eps) {      // y never aliases x
    __shared__ float s1[8][32], s2[8][32];
    const int tx = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tx;
    float a = 0.0f, q = 0.0f;
    if (t < L) {
        // eight loads in flight per thread, summed in channel order (the order — and with it the result — is the same
        // for every L: the segment-exact decode relies on position-independent arithmetic)
        int c = cg;
        for (; c + 56 < C; c += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(c + 8 * u) * L + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += v[u]; q += v[u] * v[u]; }
        }
        for (; c < C; c += 8) { const float v = x[(size_t)c * L + t]; a += v; q += v * v; }
    }
    s1[cg][tx] = a; s2[cg][tx] = q;
    __syncthreads();
    float sum = 0.0f, sq = 0.0f;
#pragma unroll
    for (int g = 0; g < 8; ++g) { sum += s1[g][tx]; sq += s2[g][tx]; }
    if (t >= L) return;
    if (LAYERNORM) {
        const float mean = sum / (float)C, var = sq / (float)C - mean * mean;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll 8
        for (int c = cg; c < C; c += 8) y[(size_t)c * L + t] = (x[(size_t)c * L + t] - mean) * inv * w[c] + b[c];
    } else {
        const float den = sqrtf(sq / (float)C + eps);
#pragma unroll 8
        for (int c = cg; c < C; c += 8) y[(size_t)c * L + t] = x[(size_t)c * L + t] / den * w[c];
    }
}
// One memory round trip instead of sixteen (round 5): 17.5 us -> a few for the pre-transformer's 17 RMSNorms and the two ConvNeXt
// LayerNorms of a decode. 16 columns x 16 channel groups per workgroup (twice the workgroups), and a thread requests ALL of its
// CPT = C / 16 channels at once and keeps them in registers: the statistics, then the normalised values, come from the same
// loads (k_norm_c read x twice, eight loads at a time). Sums run in channel order inside a thread and in group order across the
// 16 groups — position-independent like before (the segment-exact decode relies on that), but a different order than k_norm_c's,
// so the two kernels do not produce the same last bits: every caller goes through the launchers below, which pick by C only.
template <bool LAYERNORM, int CPT>
__global__ __launch_bounds__(256) void k_norm_c2(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                 float* __restrict__ y, int C, int L, float eps) {      // y never aliases x
    __shared__ float s1[16][16], s2[16][16];
    const int tx = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tx;
    const bool in = t < L;
    float v[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) v[u] = in ? x[(size_t)(cg + 16 * u) * L + t] : 0.0f;
    float wv[CPT], bv[LAYERNORM ? CPT : 1];
#pragma unroll
    for (int u = 0; u < CPT; ++u) { wv[u] = w[cg + 16 * u]; if (LAYERNORM) bv[u] = b[cg + 16 * u]; }
    float a = 0.0f, q = 0.0f;
#pragma unroll
    for (int u = 0; u < CPT; ++u) { a += v[u]; q += v[u] * v[u]; }
    s1[cg][tx] = a; s2[cg][tx] = q;
    __syncthreads();
    float sum = 0.0f, sq = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) { sum += s1[g][tx]; sq += s2[g][tx]; }
    if (!in) return;
    if (LAYERNORM) {
        const float mean = sum / (float)C, var = sq / (float)C - mean * mean;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int u = 0; u < CPT; ++u) y[(size_t)(cg + 16 * u) * L + t] = (v[u] - mean) * inv * wv[u] + bv[u];
    } else {
        const float den = sqrtf(sq / (float)C + eps);
#pragma unroll
        for (int u = 0; u < CPT; ++u) y[(size_t)(cg + 16 * u) * L + t] = v[u] / den * wv[u];
    }
}
hipError_t launch_layernorm_c(const float* x, const float* w, const float* b, float* y, int C, int L, float eps,
                              hipStream_t st) {
    static const bool old = getenv("Q3_NORM_C_OLD") != nullptr;        // A/B aid
    if (!old && C == 1024) hipLaunchKernelGGL((k_norm_c2<true, 64>), dim3((L + 15) / 16), dim3(256), 0, st, x, w, b, y, C, L, eps);
    else if (!old && C == 512) hipLaunchKernelGGL((k_norm_c2<true, 32>), dim3((L + 15) / 16), dim3(256), 0, st, x, w, b, y, C, L, eps);
    else hipLaunchKernelGGL(k_norm_c<true>, dim3((L + 31) / 32), dim3(256), 0, st, x, w, b, y, C, L, eps);
    return hipGetLastError();
}
hipError_t launch_rmsnorm_c(const float* x, const float* w, float* y, int C, int L, float eps, hipStream_t st) {
    static const bool old = getenv("Q3_NORM_C_OLD") != nullptr;        // A/B aid
    if (!old && C == 1024) hipLaunchKernelGGL((k_norm_c2<false, 64>), dim3((L + 15) / 16), dim3(256), 0, st, x, w, (const float*)nullptr, y, C, L, eps);
    else if (!old && C == 512) hipLaunchKernelGGL((k_norm_c2<false, 32>), dim3((L + 15) / 16), dim3(256), 0, st, x, w, (const float*)nullptr, y, C, L, eps);
    else hipLaunchKernelGGL(k_norm_c<false>, dim3((L + 31) / 32), dim3(256), 0, st, x, w, (const float*)nullptr, y, C, L, eps);
    return hipGetLastError();
}

// rotate-half RoPE on q and k in [nh*hd][L] layout (decoder_12hz.rs:682-691)
__global__ __launch_bounds__(256) void k_rope_c(float* q, float* k, const float* cs, const float* sn, int hd, int L) {
    const int half = hd / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int hi = blockIdx.y;                 // h*half + i
    if (t >= L) return;
    const int h = hi / half, i = hi % half;
    const float c = cs[(size_t)t * half + i], s = sn[(size_t)t * half + i];
    float* p = blockIdx.z == 0 ? q : k;
    const size_t i1 = ((size_t)h * hd + i) * L + t, i2 = ((size_t)h * hd + i + half) * L + t;
    const float x1 = p[i1], x2 = p[i2];
    p[i1] = add_rn(mul_rn(x1, c), mul_rn(-x2, s));
    p[i2] = add_rn(mul_rn(x2, c), mul_rn(x1, s));
}
hipError_t launch_rope_c(float* q, float* k, const float* cs, const float* sn, int nh, int hd, int L, hipStream_t st) {
    hipLaunchKernelGGL(k_rope_c, dim3((L + 255) / 256, nh * hd / 2, 2), dim3(256), 0, st, q, k, cs, sn, hd, L);
    return hipGetLastError();
}

// causal MHA over [nh*hd][L] tensors, hd == 64: one wave per (query i, head h); lane = key index
// inside a 64-key chunk for QKᵀ/softmax, lane = output dim for the accumulator.
__global__ __launch_bounds__(64) void k_attn_c(const float* q, const float* k, const float* v, float* o, int L,
                                               float scale) {
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const size_t hb = (size_t)h * 64 * L;
    const float qd = q[hb + (size_t)lane * L + i];      // lane d holds q[d]
    float m = -INFINITY, l = 0.0f, acc = 0.0f;          // acc: lane d holds out[d]
    for (int j0 = 0; j0 <= i; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = j <= i;
        float s = 0.0f;
        for (int d = 0; d < 64; ++d) {
            const float qv = __shfl(qd, d);
            const float kv = ok ? k[hb + (size_t)d * L + j] : 0.0f;
            s = fmaf(qv, kv, s);
        }
        s = ok ? s * scale : -INFINITY;
        float cm = s;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cm = fmaxf(cm, __shfl_xor(cm, off));
        const float mn = fmaxf(m, cm);
        const float corr = expf(m - mn);
        const float p = ok ? expf(s - mn) : 0.0f;
        float ps = p;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ps += __shfl_xor(ps, off);
        l = l * corr + ps;
        acc *= corr;
        for (int d = 0; d < 64; ++d) {
            float c = ok ? p * v[hb + (size_t)d * L + j] : 0.0f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
            if (lane == d) acc += c;
        }
        m = mn;
    }
    o[hb + (size_t)lane * L + i] = acc / l;
}
// The same attention on the matrix cores (v_mfma_f32_32x32x2f32: true f32 products). One wave per (32-query tile,
// head), key tiles of 32 in ascending order with an online softmax. The [C][L] layout makes every operand a coalesced
// global read: Sᵀ = K·Qᵀ (A = K: lane (key, d-pair) reads k[d][j0 + key]; B = Q, the tile's 32 x 64 values held in 32
// registers) puts a QUERY in each lane's column, so the softmax statistics are per-lane reductions over the 16
// accumulator registers plus one cross-half shuffle, and the probabilities are already in B-operand position for
// Oᵀ = Vᵀ·Pᵀ: MFMA step r pairs the key of register r in the lower lane half with the key of register r in the upper
// half, which only fixes which V column each half reads. V goes through LDS (coalesced load, transposed read).
// A query's arithmetic depends only on its own index (tiles are aligned to absolute positions), so prefix decodes
// reproduce the whole-utterance values bit for bit, as the streaming modes require.
typedef __attribute__((ext_vector_type(16))) float f32x16a_t;
// Round 2: four waves per (query tile, head), wave w walking key tiles w, w + 4, ... with its own V patch in LDS and no
// workgroup barrier inside the loop (LDS operations of one wave execute in order); the four partial (max, sum, O) are
// merged in wave order at the end — a fixed order per query, so prefix decodes still reproduce the whole-utterance bits.
// One wave per tile walked up to 20 key tiles alone: 95 us per layer at 640 frames.
constexpr int ATC_NW = 4;
__global__ __launch_bounds__(64 * ATC_NW) void k_attn_c_mfma(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                             float* __restrict__ o, int L, float scale) {
    constexpr int VP = 33;
    __shared__ float vs_all[ATC_NW][64 * VP];                                  // per wave: V patch, then its partial O
    __shared__ float s_m[ATC_NW][32], s_l[ATC_NW][32];
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* vs = vs_all[wave];
    const int qt = (int)gridDim.x - 1 - (int)blockIdx.x, h = blockIdx.y;      // longest (last) query tiles first
    const int i0 = qt * 32, i = i0 + li;
    const size_t hb = (size_t)h * 64 * L;
    float qb[32];                                                              // B operand of step s: Q[i][d = 2s + lk]
#pragma unroll
    for (int s = 0; s < 32; ++s) qb[s] = i < L ? q[hb + (size_t)(2 * s + lk) * L + i] : 0.0f;
    f32x16a_t oacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;                                          // lsum: this half's share of the denominator
    for (int j0 = 32 * wave; j0 <= i0; j0 += 32 * ATC_NW) {
        const int j = j0 + li;
        const bool jok = j < L;
        // V tile -> LDS (rows d, columns key); issued first so that it overlaps the score MFMAs
        float vr[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) vr[s] = jok ? v[hb + (size_t)(2 * s + lk) * L + j] : 0.0f;
        f32x16a_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
        float ka[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) ka[s] = jok ? k[hb + (size_t)(2 * s + lk) * L + j] : 0.0f;
#pragma unroll
        for (int s = 0; s < 32; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], qb[s], sacc, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // the previous tile's patch reads are done
#pragma unroll
        for (int s = 0; s < 32; ++s) vs[(2 * s + lk) * VP + li] = vr[s];
        // scores: register r of this lane = (key j0 + (r&3) + 8*(r>>2) + 4*lk, query i)
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float sv = (key <= i && key < L) ? sacc[r] * scale : -INFINITY;
            sacc[r] = sv; cm = fmaxf(cm, sv);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        // a key tile past the first can be all masked for the early queries of this tile only when j0 > i: not here
        // (j0 <= i0 <= i), so cm is finite
        const float mn = fmaxf(m, cm);
        const float corr = expf(m - mn);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = expf(sacc[r] - mn); sacc[r] = p; ps += p; }
        lsum = lsum * corr + ps;
        m = mn;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= corr;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // the patch is written
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lk;
#pragma unroll
            for (int t = 0; t < 2; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(t * 32 + li) * VP + key], sacc[r], oacc[t], 0, 0, 0);
        }
    }
    // partials to LDS: O[d][query] (d = t*32 + (r&3) + 8*(r>>2) + 4*lk) in this wave's patch, (m, l) per query
    const float lw = lsum + __shfl_xor(lsum, 32);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) vs[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * VP + li] = oacc[t][r];
    if (lk == 0) { s_m[wave][li] = m; s_l[wave][li] = lw; }
    __syncthreads();
    // merge in wave order: thread (w', lane) handles d = w' * 16 + (lane >> 5) * 8 .. +8 of query li
    if (i < L) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATC_NW; ++w) M = fmaxf(M, s_m[w][li]);
        float wg[ATC_NW], den = 0.0f;
#pragma unroll
        for (int w = 0; w < ATC_NW; ++w) { wg[w] = s_m[w][li] == -INFINITY ? 0.0f : expf(s_m[w][li] - M); den += s_l[w][li] * wg[w]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = wave * 16 + lk * 8 + e;
            float acc = 0.0f;
#pragma unroll
            for (int w = 0; w < ATC_NW; ++w) acc += vs_all[w][d * VP + li] * wg[w];
            o[hb + (size_t)d * L + i] = acc / den;
        }
    }
}
hipError_t launch_attn_c(const float* q, const float* k, const float* v, float* o, int nh, int hd, int L, float scale,
                         hipStream_t st) {
    if (hd != 64) return hipErrorInvalidValue;
    static const bool valu = getenv("Q3_ATTN_C_VALU") != nullptr;       // A/B aid: the first-generation VALU kernel
    if (valu) hipLaunchKernelGGL(k_attn_c, dim3(L, nh), dim3(64), 0, st, q, k, v, o, L, scale);
    else hipLaunchKernelGGL(k_attn_c_mfma, dim3((L + 31) / 32, nh), dim3(64 * ATC_NW), 0, st, q, k, v, o, L, scale);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_silu_mul(const float* g, const float* u, float* y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = g[i]; y[i] = (v / (1.0f + expf(-v))) * u[i]; }
}
hipError_t launch_silu_mul(const float* g, const float* u, float* y, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_silu_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, u, y, n);
    return hipGetLastError();
}

// split-RVQ lookup (decoder_12hz.rs:420-446): first_out[d][t] = first_cb[code0 % cb_size][d],
// rest_out[d][t] = ((0 + cb_0[c1][d]) + cb_1[c2][d]) + … (15 tables, in order)
__global__ __launch_bounds__(256) void k_rvq_embed(const uint32_t* frames, int n_frames, const float* first_cb,
                                                   const float* const* rest_cbs, float* first_out, float* rest_out,
                                                   int cb_dim, int cb_size) {
    const int t = blockIdx.x, d = threadIdx.x;
    if (d >= cb_dim) return;
    const uint32_t* f = frames + (size_t)t * 16;
    first_out[(size_t)d * n_frames + t] = first_cb[(size_t)(f[0] % (uint32_t)cb_size) * cb_dim + d];
    float acc = 0.0f;
    for (int i = 0; i < 15; ++i) {
        uint32_t c = f[1 + i];
        if (c >= (uint32_t)cb_size) c = cb_size - 1;      // host validates; clamp keeps the read in-bounds
        acc = add_rn(acc, rest_cbs[i][(size_t)c * cb_dim + d]);
    }
    rest_out[(size_t)d * n_frames + t] = acc;
}
hipError_t launch_rvq_embed(const uint32_t* frames, int n_frames, const float* first_cb, const float* const* rest_cbs,
                            float* first_out, float* rest_out, int cb_dim, int cb_size, hipStream_t st) {
    if (cb_dim > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rvq_embed, dim3(n_frames), dim3(256), 0, st, frames, n_frames, first_cb, rest_cbs, first_out,
                       rest_out, cb_dim, cb_size);
    return hipGetLastError();
}

// codebook = embedding_sum / clamp(cluster_usage, 1e-7) (decoder_12hz.rs:199-225)
__global__ __launch_bounds__(256) void k_norm_codebook(const float* esum, const float* usage, float* out, int dim) {
    const int r = blockIdx.x;
    const float u = fmaxf(usage[r], 1e-7f);
    for (int d = threadIdx.x; d < dim; d += 256) out[(size_t)r * dim + d] = esum[(size_t)r * dim + d] / u;
}
hipError_t launch_norm_codebook(const float* esum, const float* usage, float* out, int rows, int dim, hipStream_t st) {
    hipLaunchKernelGGL(k_norm_codebook, dim3(rows), dim3(256), 0, st, esum, usage, out, dim);
    return hipGetLastError();
}

// snake tables: a = exp(alpha), ib = 1/(exp(beta)+1e-9) (snake_beta.rs:63-73), computed once at load
__global__ __launch_bounds__(256) void k_snake_tables(const float* alpha, const float* beta, float* a, float* ib, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) { a[c] = expf(alpha[c]); ib[c] = 1.0f / (expf(beta[c]) + 1e-9f); }
}
hipError_t launch_snake_tables(const float* alpha, const float* beta, float* a, float* ib, int C, hipStream_t st) {
    hipLaunchKernelGGL(k_snake_tables, dim3((C + 255) / 256), dim3(256), 0, st, alpha, beta, a, ib, C);
    return hipGetLastError();
}

}  // namespace q3
