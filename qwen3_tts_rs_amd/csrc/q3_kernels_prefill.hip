// q3_kernels_prefill.hip — long-prompt prefill of the talker (run_prefill_layers,
// talker.rs:823-841) as real GEMMs plus a query-blocked causal attention, for prompts of hundreds to thousands of
// positions (VoiceDesign instruct text, ICL reference transcripts + frames: BASELINE config[4]).
//
// The decode path streams every weight matrix once per 16 activation rows; over a 4k-position prompt that is 256
// weight passes and ~43k launches per sequence. Here a chunk of up to 128 positions per sequence goes through each
// layer at once:
//   * k_lm_gemm     y[m][n] = Σ_k x[m][k]·W[n][k] on v_mfma_f32_16x16x32_bf16 with the SAME pre-tiled bf16 weight image
//                   the decode GEMV uses (no second copy of the model) and the same exact bf16x3 split of the f32
//                   activations, done once per workgroup while staging x into LDS; workgroup tile 64 weight rows x
//                   128 activation rows, so a weight tile is reused by 128 rows and a staged x tile by 64 weight rows;
//   * k_attn_prefill  GQA causal attention for 8 consecutive query positions per workgroup: one K/V load serves
//                   8 positions x 2 query heads (the decode kernel's per-position online softmax, unchanged
//                   arithmetic), reading the K/V the preceding k_qknorm_rope_kv launch appended to the cache.
// Same arithmetic contract as the decode path (bf16 weights, f32 activations / accumulation / KV); only summation
// orders differ, so prefill logits agree with the oracle to the same ~1e-5 as the decode kernels.
#include "q3_kernels.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

namespace q3 {

typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8_t;
typedef __attribute__((ext_vector_type(4))) float pf32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int pu32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 pbf16x2_t;
typedef __attribute__((ext_vector_type(2))) float pf32x2_t;

__device__ __forceinline__ uint32_t ppk_bf16(float lo, float hi) {
    const pf32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pbf16x2_t));
}
__device__ __forceinline__ void psplit3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = ppk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = ppk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = ppk_bf16(sa, sb);
}
__device__ __forceinline__ float pwave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float phalf_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// den[m] = sqrt(mean(x[m][:]^2) + eps): the RMSNorm denominator of each activation row (one wave per row)
__global__ __launch_bounds__(256) void k_row_den(const float* __restrict__ x, int ldx, float* __restrict__ den, int rows, int cols, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float ss = 0.0f;
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
    }
    ss = pwave_sum(ss);
    if (lane == 0) den[row] = sqrtf(ss / (float)cols + eps);
}
hipError_t launch_row_den(const float* x, int ldx, float* den, int rows, int cols, float eps, hipStream_t st) {
    if (cols % 4 || ldx % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_row_den, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, den, rows, cols, eps);
    return hipGetLastError();
}

// The exact bf16x3 split of a whole activation matrix, once (one wave per row, 8 k per lane per step): the GEMM below
// would otherwise redo it in each of its N/64 workgroup columns — 64-192 times — and be VALU-bound on it.
template <bool RMS>
__global__ __launch_bounds__(256) void k_split_rows(const float* __restrict__ x, int ldx, const float* __restrict__ norm_w,
                                                    uint16_t* __restrict__ planes, size_t plane_elems, int rows, int K, int Kpad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    uint16_t* pr = planes + (size_t)row * Kpad;
    for (int k = lane * 8; k < Kpad; k += 512) {
        float v[8];
        if (k < K) {                                            // K % 8 == 0: the octet is all in or all out
            const float4 a = *reinterpret_cast<const float4*>(xr + k), b = *reinterpret_cast<const float4*>(xr + k + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if constexpr (RMS) {
                const float4 n0 = *reinterpret_cast<const float4*>(norm_w + k), n1 = *reinterpret_cast<const float4*>(norm_w + k + 4);
                v[0] *= n0.x; v[1] *= n0.y; v[2] *= n0.z; v[3] *= n0.w; v[4] *= n1.x; v[5] *= n1.y; v[6] *= n1.z; v[7] *= n1.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        }
        pu32x4_t h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { uint32_t hh, mm, ll; psplit3_pair(v[2 * e], v[2 * e + 1], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
        *reinterpret_cast<pu32x4_t*>(pr + k) = h;
        *reinterpret_cast<pu32x4_t*>(pr + plane_elems + k) = m;
        *reinterpret_cast<pu32x4_t*>(pr + 2 * plane_elems + k) = l;
    }
}
hipError_t launch_split_rows(const float* x, int ldx, const float* norm_w, uint16_t* planes, size_t plane_elems, int rows, int K,
                             int Kpad, hipStream_t st) {
    if (K % 8 || Kpad % 8 || ldx % 4 || K > Kpad || (plane_elems % 8)) return hipErrorInvalidValue;
    if (norm_w) hipLaunchKernelGGL((k_split_rows<true>), dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, norm_w, planes, plane_elems, rows, K, Kpad);
    else hipLaunchKernelGGL((k_split_rows<false>), dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, norm_w, planes, plane_elems, rows, K, Kpad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GEMM. Workgroup = 4 waves, tile = 64 weight rows (4 MFMA row tiles) x 128 activation rows (8 column tiles of 16);
// wave w owns column tiles 2w, 2w+1 and all 4 row tiles. Per stage of 64 k: x[128][64] f32 is staged as three bf16
// planes [row][64 k] (144-byte pitch: the 16 rows of a ds_read_b128 phase land on distinct 4-bank groups); weight
// tiles come straight from global memory (the 4 waves read the same 8 tiles: L1 hits).
// ------------------------------------------------------------------------------------------------
constexpr int GP = 144;      // LDS bytes per activation row per plane: 64 k x 2 B + 16 pad

template <int EPI, bool RMS, bool PL = false>      // PL: x comes pre-split (GemmArgs::xp), staging is a copy
__global__ __launch_bounds__(256) void k_lm_gemm(GemmArgs a) {
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 128 * GP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mj = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * 128, rt0 = blockIdx.y * 4;            // first activation row, first 16-row weight tile
    const int S = a.Kpad >> 5;                                         // k-steps of 32 in the tiled image
    const pu32x4_t* __restrict__ w1 = reinterpret_cast<const pu32x4_t*>(a.W);
    const pu32x4_t* __restrict__ w2 = reinterpret_cast<const pu32x4_t*>(NW == 2 ? a.W2 : a.W);
    constexpr size_t plane = (size_t)128 * GP;

    pf32x4_t acc[NW][4][2];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[w][r][c] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

    // staging role: thread t handles activation row t/2, k half (t%2)*32 .. +32 of the 64-k stage
    const int srow = tid >> 1, sk = (tid & 1) * 32;
    const bool srow_ok = (m0 + srow) < a.M;
    const float* __restrict__ xrow = a.x + (size_t)(srow_ok ? m0 + srow : 0) * a.ldx;

    // pre-split x: the next stage's 3 x 64 bytes per thread are requested before this stage's MFMAs and committed to
    // LDS after the next barrier, so their round trip is off the critical path
    pu32x4_t xt[3][4];
    auto fetch_x = [&](int k0) {
        const uint16_t* src = a.xp + (size_t)(srow_ok ? m0 + srow : 0) * a.Kpad + k0 + sk;
        const bool kok = srow_ok && (k0 + sk) < a.Kpad;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < 4; ++q) xt[pl][q] = kok ? *reinterpret_cast<const pu32x4_t*>(src + pl * a.xp_plane + q * 8) : pu32x4_t{0u, 0u, 0u, 0u};
    };
    pu32x4_t A[NW][4][2];
    auto load_A = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int s = (k0 >> 5) + ks;
                const size_t off = ((size_t)(rt0 + r) * S + (s < S ? s : S - 1)) * 64 + lane;
                A[0][r][ks] = __builtin_nontemporal_load(w1 + off);
                if constexpr (NW == 2) A[1][r][ks] = __builtin_nontemporal_load(w2 + off);
            }
    };
    if constexpr (PL) fetch_x(0);
    for (int k0 = 0; k0 < a.Kpad; k0 += 64) {
        // weight tiles of this stage first (independent of the LDS traffic below)
        load_A(k0);
        __syncthreads();
        if constexpr (PL) {
            unsigned char* dst = smem + (size_t)srow * GP + sk * 2;            // 32 k of one row: 64 bytes per plane
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<pu32x4_t*>(dst + pl * plane + q * 16) = xt[pl][q];
        } else {
            float v[32];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + sk + q * 4;
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (srow_ok && k < a.K) {                      // K % 4 == 0
                    f = *reinterpret_cast<const float4*>(xrow + k);
                    if constexpr (RMS) { const float4 n = *reinterpret_cast<const float4*>(a.norm_w + k); f.x *= n.x; f.y *= n.y; f.z *= n.z; f.w *= n.w; }
                }
                v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
            }
            unsigned char* dst = smem + (size_t)srow * GP + sk * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                      // 8 floats -> one 16-byte store per plane
                pu32x4_t h, m, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) { uint32_t hh, mm, ll; psplit3_pair(v[8 * q + 2 * e], v[8 * q + 2 * e + 1], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
                *reinterpret_cast<pu32x4_t*>(dst + q * 16) = h;
                *reinterpret_cast<pu32x4_t*>(dst + plane + q * 16) = m;
                *reinterpret_cast<pu32x4_t*>(dst + 2 * plane + q * 16) = l;
            }
        }
        __syncthreads();
        if constexpr (PL) fetch_x(k0 + 64 < a.Kpad ? k0 + 64 : k0);      // unconditional: see k_lm_gemm2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if ((k0 >> 5) + ks >= S) break;
            pu32x4_t B[2][3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned char* bp = smem + (size_t)(wave * 32 + c * 16 + mj) * GP + ks * 64 + kg * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) B[c][pl] = *reinterpret_cast<const pu32x4_t*>(bp + pl * plane);
            }
            // plane outermost (lo, mid, hi: the same per-accumulator order as before): consecutive MFMAs write different
            // accumulators, so none waits for the previous one's result
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            acc[w][r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8_t, A[w][r][ks]),
                                                                                  __builtin_bit_cast(pbf16x8_t, B[c][pl]), acc[w][r][c], 0, 0, 0);
        }
    }
    // epilogue: lane (mj, kg) holds activation row mj of the column tile and weight rows kg*4 .. +3 of the row tile
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int m = m0 + wave * 32 + c * 16 + mj;
        if (m >= a.M) continue;
        const float den = RMS ? a.den[m] : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = (rt0 + r) * 16 + kg * 4;
            if (n >= a.N) continue;                                // N % 4 == 0
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[0][r][c][e];
                float v2 = NW == 2 ? acc[NW - 1][r][c][e] : 0.0f;
                if constexpr (RMS) { v = v / den; if constexpr (NW == 2) v2 = v2 / den; }
                if (a.bias) v = v + a.bias[n + e];
                if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)m * a.ldr + n + e] + v;
                if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
                if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
                o[e] = v;
            }
            *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Second GEMM geometry (pre-split inputs only): 2 x 2 wave grid, a wave owns RT row tiles x 4 column tiles
// (RT = 4, or 2 + 2 for the gate/up pair), workgroup tile (128 / NW) weight rows x 128 activation rows. Per 64-k stage
// the vector-memory pipe still moves 80 KB (48 KB of planes + the weight fragments, each read by two waves) but now
// under 96 MFMAs per wave instead of 48: the first geometry is bound by that pipe (64 B/clk per CU), this one by the
// matrix cores.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool RMS>
__global__ __launch_bounds__(256) void k_lm_gemm2(GemmArgs a) {
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int RT = 4 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 128 * GP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mj = lane & 15, kg = lane >> 4;
    const int wr = wave & 1, wc = wave >> 1;
    // XCD-aware work order. Workgroup b runs on XCD b % 8 and each XCD has its own 4 MB L2. A work unit = one 128-row
    // M tile x one n_split-th of the N tiles; unit u belongs to XCD u % 8, and the workgroups of a unit are consecutive
    // on that XCD, so the unit's activation planes (1.5 MB at K = 2048) are fetched into that L2 once and every weight
    // tile streams past them. In plain (M tile, N tile) order the planes of all 33 M tiles cycle through the L2 between
    // two uses: each GEMM re-read its input N/128 times from the Infinity Cache (9.6 GB per layer at 4105 positions)
    // and was bound by exactly that.
    int mt, ntile;
    {
        const int nMt = (a.M + 127) / 128, nNt = a.N / (128 / NW), Q = a.n_split, per = nNt / Q;
        const int lin = (int)blockIdx.x, xcd = lin & 7, j = lin >> 3;
        const int u = xcd + 8 * (j / per);                         // this XCD's (j / per)-th unit
        if (u >= nMt * Q) return;                                   // padding workgroups of the last round
        mt = u / Q; ntile = (u - mt * Q) * per + j % per;
    }
    const int m0 = mt * 128, rt0 = ntile * (2 * RT) + wr * RT;     // first activation row, this wave's first 16-row weight tile
    const int S = a.Kpad >> 5;
    const pu32x4_t* __restrict__ w1 = reinterpret_cast<const pu32x4_t*>(a.W);
    const pu32x4_t* __restrict__ w2 = reinterpret_cast<const pu32x4_t*>(NW == 2 ? a.W2 : a.W);
    constexpr size_t plane = (size_t)128 * GP;

    pf32x4_t acc[NW][RT][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[w][r][c] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

    const int srow = tid >> 1, sk = (tid & 1) * 32;            // staging role: activation row, 32-k half of the stage
    const bool srow_ok = (m0 + srow) < a.M;
    pu32x4_t xt[3][4];
    auto fetch_x = [&](int k0) {
        const uint16_t* src = a.xp + (size_t)(srow_ok ? m0 + srow : 0) * a.Kpad + k0 + sk;
        const bool kok = srow_ok && (k0 + sk) < a.Kpad;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int q = 0; q < 4; ++q) xt[pl][q] = kok ? *reinterpret_cast<const pu32x4_t*>(src + pl * a.xp_plane + q * 8) : pu32x4_t{0u, 0u, 0u, 0u};
    };
    fetch_x(0);
    for (int k0 = 0; k0 < a.Kpad; k0 += 64) {
        pu32x4_t A[NW][RT][2];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int s = (k0 >> 5) + ks;
                const size_t off = ((size_t)(rt0 + r) * S + (s < S ? s : S - 1)) * 64 + lane;
                A[0][r][ks] = __builtin_nontemporal_load(w1 + off);
                if constexpr (NW == 2) A[1][r][ks] = __builtin_nontemporal_load(w2 + off);
            }
        __syncthreads();
        {
            unsigned char* dst = smem + (size_t)srow * GP + sk * 2;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<pu32x4_t*>(dst + pl * plane + q * 16) = xt[pl][q];
        }
        __syncthreads();
        fetch_x(k0 + 64 < a.Kpad ? k0 + 64 : k0);                // unconditional (past the end: this stage again): a prefetch
                                                                 // in a uniform branch makes hipcc wait for ALL loads at the join
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if ((k0 >> 5) + ks >= S) break;
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) {                  // lo, mid, hi: the per-accumulator order of the first geometry
                pu32x4_t B[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    B[c] = *reinterpret_cast<const pu32x4_t*>(smem + pl * plane + (size_t)(wc * 64 + c * 16 + mj) * GP + ks * 64 + kg * 16);
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[w][r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8_t, A[w][r][ks]),
                                                                                  __builtin_bit_cast(pbf16x8_t, B[c]), acc[w][r][c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int m = m0 + wc * 64 + c * 16 + mj;
        if (m >= a.M) continue;
        const float den = RMS ? a.den[m] : 1.0f;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int n = (rt0 + r) * 16 + kg * 4;
            if (n >= a.N) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[0][r][c][e];
                float v2 = NW == 2 ? acc[NW - 1][r][c][e] : 0.0f;
                if constexpr (RMS) { v = v / den; if constexpr (NW == 2) v2 = v2 / den; }
                if (a.bias) v = v + a.bias[n + e];
                if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)m * a.ldr + n + e] + v;
                if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
                if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
                o[e] = v;
            }
            *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Third GEMM geometry (pre-split inputs, Kpad % 128 == 0): 8 waves (4 x 2), workgroup tile (256 / NW) weight rows x 128
// activation rows, wave tile 64 x 64 as in the second geometry. What changes is the data movement:
//   * the three activation planes of a 64-k stage go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 6 pieces of
//     1 KB per wave and stage), no staging registers and no ds_write pass; the LDS image is lane-linear, so the
//     bank swizzle (16-byte slot c of row r lives in slot c ^ (r & 7)) is applied to each lane's SOURCE address and
//     again on the fragment reads: the four 16-lane groups of a ds_read_b128 then touch 16 distinct slots;
//   * two LDS buffers (96 KB) and two register sets of weight fragments: stage s+1 is requested before the MFMAs of
//     stage s and waited for (vmcnt(0)) after them, ONE barrier per stage instead of two;
//   * 256 weight rows share one set of planes: per 64-k stage and CU the texture path moves 112 KB under
//     2 x 96 MFMAs per SIMD (3072 clk) instead of 160 KB (two 128 x 128 workgroups).
// Per accumulator the MFMA order is the second geometry's (k-step, then lo, mid, hi); K is summed in ranges (below).
// ------------------------------------------------------------------------------------------------
constexpr int G3_PLANE = 128 * 128;            // bytes per plane per buffer: 128 activation rows x 64 k x 2 B
constexpr int G3_BUF = 3 * G3_PLANE;

template <int EPI, bool RMS>
__global__ __launch_bounds__(512) void k_lm_gemm3(GemmArgs a) {
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int RT = 4 / NW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G3_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction: keeps addresses scalar
    const int mj = lane & 15, kg = lane >> 4;
    const int wr = wave & 3, wc = wave >> 2;          // waves w and w + 4 (one SIMD) share their weight fragments
    int mt, ntile;
    {   // XCD-aware work order. Workgroup b runs on XCD b % 8, one workgroup per CU, so the 32 workgroups an XCD runs at
        // a time are 32 consecutive values of b / 8. They form one SUPER-TILE of sup_m M tiles x sup_n N tiles and walk
        // K together: per 64-k stage the XCD's L2 fetches sup_m plane pieces (48 KB each) and sup_n weight pieces
        // (32 KB) for 32 workgroups - 448 KB at 4 x 8 - where units of one M tile x 8 N tiles (the second geometry's
        // order) fetched 4 x 48 + 32 x 32 = 1216 KB: the four concurrent units had four different N ranges. Super-tiles
        // go round-robin over the XCDs with the M group running fastest, so with 8 M groups an XCD keeps its M group
        // (planes L2/MALL-warm) while it walks the N groups.
        const int nMt = (a.M + 127) / 128, nNt = a.N / (256 / NW);
        const int nMg = (nMt + a.sup_m - 1) / a.sup_m, nNg = (nNt + a.sup_n - 1) / a.sup_n, per = a.sup_m * a.sup_n;
        const int lin = (int)blockIdx.x % a.wg_per_split, xcd = lin & 7, j = lin >> 3;        // blockIdx / wg_per_split = K split (below)
        const int sup = xcd + 8 * (j / per), w = j % per;
        if (sup >= nMg * nNg) return;
        mt = (sup % nMg) * a.sup_m + w / a.sup_n; ntile = (sup / nMg) * a.sup_n + w % a.sup_n;
        if (mt >= nMt || ntile >= nNt) return;
    }
    const int m0 = mt * 128, rt0 = ntile * (4 * RT) + wr * RT;
    const int S = a.Kpad >> 5, nst = a.Kpad >> 6;      // nst is even (launcher: Kpad % 128 == 0)

    pf32x4_t acc[NW][RT][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[w][r][c] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

    // LDS-DMA role: piece q = wave + 8 i (i < 6) is rows (q % 16) * 8 .. +8 of plane q / 16; lane l lands in 16-byte
    // slot l of the piece = row l / 8, slot l % 8, and therefore fetches source slot (l % 8) ^ (row & 7). Rows past M
    // are read like any other (the planes buffer holds whole 128-row tiles: GemmArgs::xp) and never stored.
    const unsigned xoff = (unsigned)(((lane >> 3) * a.Kpad + (((lane & 7) ^ (lane >> 3)) << 3)) * 2);     // bytes, per lane
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(a.xp) + (size_t)m0 * a.Kpad * 2;
    auto stage_x = [&](int st, int buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = wave + 8 * i, pl = q >> 4, rg = q & 15;
            const unsigned char* src = xbase + ((size_t)pl * a.xp_plane + (size_t)rg * 8 * a.Kpad + (size_t)st * 64) * 2;   // scalar
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + xoff),
                                             (__attribute__((address_space(3))) void*)(smem + buf * G3_BUF + pl * G3_PLANE + rg * 1024),
                                             16, 0, 0);
        }
    };
    const unsigned char* const w1 = reinterpret_cast<const unsigned char*>(a.W) + (size_t)rt0 * S * 1024;
    const unsigned char* const w2 = reinterpret_cast<const unsigned char*>(NW == 2 ? a.W2 : a.W) + (size_t)rt0 * S * 1024;
    const unsigned aoff = (unsigned)lane * 16;
    auto load_A = [&](pu32x4_t (&A)[NW][RT][2], int st) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const size_t off = ((size_t)r * S + st * 2 + ks) * 1024;                                 // scalar
                A[0][r][ks] = *reinterpret_cast<const pu32x4_t*>(w1 + off + aoff);
                if constexpr (NW == 2) A[1][r][ks] = *reinterpret_cast<const pu32x4_t*>(w2 + off + aoff);
            }
    };
    // fragment reads: lane (mj, kg) of column tile c wants row wc*64 + c*16 + mj, source slot ks*4 + kg at its swizzled
    // place; (row & 7) = mj & 7 because wc*64 + c*16 is a multiple of 8
    const unsigned fb = (unsigned)((wc * 64 + mj) * 128);
    const unsigned fo0 = fb + (unsigned)(((0 + kg) ^ (mj & 7)) * 16), fo1 = fb + (unsigned)(((4 + kg) ^ (mj & 7)) * 16);
    auto load_B = [&](pu32x4_t (&B)[4], int buf, int g) {          // group g = ks * 3 + (2 - pl): k-step, then lo, mid, hi
        const int ks = g / 3, pl = 2 - g % 3;
        const unsigned char* p = smem + buf * G3_BUF + pl * G3_PLANE + (ks ? fo1 : fo0);
#pragma unroll
        for (int c = 0; c < 4; ++c) B[c] = *reinterpret_cast<const pu32x4_t*>(p + c * (16 * 128));
    };
    auto mfma_group = [&](const pu32x4_t (&A)[NW][RT][2], const pu32x4_t (&B)[4], int g) {
        const int ks = g / 3;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[w][r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8_t, A[w][r][ks]),
                                                                          __builtin_bit_cast(pbf16x8_t, B[c]), acc[w][r][c], 0, 0, 0);
    };
    // K RANGES. K is cut into a.k_ranges (8 when the stage count allows, else 1) contiguous ranges; each range is accumulated
    // from zero and the range sums are added in ascending order — by this workgroup when it walks all of K (tot below),
    // or, for small M, by k_gemm_splitk_reduce when the launcher spreads the ranges over a.k_split workgroups (split-K:
    // a 509-position prompt gives the down projection 32 workgroups of 96 serial stages on 256 CUs). Either way an
    // output is the same sum in the same order: a position prefilled by this GEMM gets the same bits whatever batch it is
    // prefilled in (which positions those are depends on the prompt length only: q3_session_prefill's tail rule).
    const int R = a.k_ranges, spr = nst / R;                         // stages per range: even (launcher)
    const int kz = (int)blockIdx.x / a.wg_per_split;
    const int r_begin = kz * R / a.k_split, r_end = (kz + 1) * R / a.k_split;
    const int st_begin = r_begin * spr, st_end = r_end * spr;
    pf32x4_t tot[NW][RT][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) tot[w][r][c] = pf32x4_t{0.f, 0.f, 0.f, 0.f};
    auto range_done = [&](int range) {
        if (a.k_split == 1) {
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { tot[w][r][c] = tot[w][r][c] + acc[w][r][c]; acc[w][r][c] = pf32x4_t{0.f, 0.f, 0.f, 0.f}; }
        } else {                                                     // partial sums [range][w][m][n] for the reduce kernel
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m = m0 + wc * 64 + c * 16 + mj;
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const int n = (rt0 + r) * 16 + kg * 4;
                        if (m < a.M && n < a.N)
                            *reinterpret_cast<pf32x4_t*>(a.splitk_ws + (((size_t)range * NW + w) * a.M + m) * a.N + n) = acc[w][r][c];
                        acc[w][r][c] = pf32x4_t{0.f, 0.f, 0.f, 0.f};
                    }
            }
        }
    };

    pu32x4_t A0[NW][RT][2], A1[NW][RT][2];
    stage_x(st_begin, 0); load_A(A0, st_begin);
    {
        // Interleaved schedule: the barrier releases all 8 waves at once, so anything issued in a block of its own (the
        // next stage's 6 LDS-DMA pieces and 8 weight-fragment loads, the next group's 4 fragment reads) is time in which
        // no wave of the CU feeds the matrix cores. Here every group of 16 MFMAs is cut into 4 chunks and each chunk is
        // followed by at most two of those instructions, which issue in the shadow of the chunk's MFMAs; all requests
        // of the next stage are out by the middle of the current one.
        auto step = [&](const pu32x4_t (&Ac)[NW][RT][2], pu32x4_t (&An)[NW][RT][2], int buf, int stn) {
            pu32x4_t Ba[4], Bb[4];
            auto read_B = [&](pu32x4_t& d, int g, int c) {
                const int ks = g / 3, pl = 2 - g % 3;
                d = *reinterpret_cast<const pu32x4_t*>(smem + buf * G3_BUF + pl * G3_PLANE + (ks ? fo1 : fo0) + c * (16 * 128));
            };
            auto piece = [&](int i) {
                const int q = wave + 8 * i, pl = q >> 4, rg = q & 15;
                const unsigned char* src = xbase + ((size_t)pl * a.xp_plane + (size_t)rg * 8 * a.Kpad + (size_t)stn * 64) * 2;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + xoff),
                                                 (__attribute__((address_space(3))) void*)(smem + (buf ^ 1) * G3_BUF + pl * G3_PLANE + rg * 1024),
                                                 16, 0, 0);
            };
            auto fetch_A = [&](int idx) {
                const int w = NW == 2 ? idx >> 2 : 0, r = NW == 2 ? (idx >> 1) & 1 : idx >> 1, ks = idx & 1;
                const size_t off = ((size_t)r * S + stn * 2 + ks) * 1024;
                // plain (cached) loads: the sup_m workgroups of a super-tile column share these fragments through the L2
                // (nontemporal: 49.4 instead of 46.1 ms per 4105-position prefill at 4 x 8)
                An[w][r][ks] = *reinterpret_cast<const pu32x4_t*>((w ? w2 : w1) + off + aoff);
            };
            auto chunk = [&](const pu32x4_t (&B)[4], int g, int j) {          // column tile j x the wave's 4 row tiles:
                const int ks = g / 3;                                         // needs B[j] only, read a whole group earlier
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        acc[w][r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8_t, Ac[w][r][ks]),
                                                                              __builtin_bit_cast(pbf16x8_t, B[j]), acc[w][r][j], 0, 0, 0);
            };
#pragma unroll
            for (int c = 0; c < 4; ++c) read_B(Ba[c], 0, c);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gp = 0; gp < 3; ++gp)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int g = 2 * gp + h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (h == 0) chunk(Ba, g, j); else chunk(Bb, g, j);
                        __builtin_amdgcn_sched_barrier(0);
                        if (g < 5) { if (h == 0) read_B(Bb[j], g + 1, j); else read_B(Ba[j], g + 1, j); }
                        if (g < 3 && (j & 1) == 0) piece(2 * g + (j >> 1));
                        if (g < 4 && (j & 1) == 1) fetch_A(2 * g + (j >> 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        };
        int range = r_begin, range_end = st_begin + spr;
        for (int st = st_begin; st < st_end; st += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            step(A0, A1, 0, st + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            step(A1, A0, 1, st + 2 < st_end ? st + 2 : st_end - 1);      // past the end: a spare request of the last stage, never read
            if (st + 2 == range_end) { range_done(range); ++range; range_end += spr; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA may outlive the workgroup
    }
    if (a.k_split > 1) return;                                      // epilogue in k_gemm_splitk_reduce
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int m = m0 + wc * 64 + c * 16 + mj;
        if (m >= a.M) continue;
        const float den = RMS ? a.den[m] : 1.0f;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int n = (rt0 + r) * 16 + kg * 4;
            if (n >= a.N) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = tot[0][r][c][e];
                float v2 = NW == 2 ? tot[NW - 1][r][c][e] : 0.0f;
                if constexpr (RMS) { v = v / den; if constexpr (NW == 2) v2 = v2 / den; }
                if (a.bias) v = v + a.bias[n + e];
                if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)m * a.ldr + n + e] + v;
                if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
                if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
                o[e] = v;
            }
            *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = make_float4(o[0], o[1], o[2], o[3]);
            // (Writing the consumer's bf16x3 planes from this epilogue — SwiGLU output, residual stream times the next norm
            // weight — instead of separate k_split_rows passes was built and measured: 45.3 vs 45.3 ms per 4105-position
            // prefill; the 8-byte plane stores cost the epilogue what the split passes cost on their own.)
        }
    }
}

// split-K second pass: y[m][n] = epilogue(range sums added in ascending order) — the arithmetic of k_lm_gemm3's own epilogue
template <int EPI, bool RMS>
__global__ __launch_bounds__(256) void k_gemm_splitk_reduce(GemmArgs a) {
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int n4 = a.N >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)a.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
    pf32x4_t t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        t[w] = pf32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < a.k_ranges; ++r)
            t[w] = t[w] + *reinterpret_cast<const pf32x4_t*>(a.splitk_ws + (((size_t)r * NW + w) * a.M + m) * a.N + n);
    }
    const float den = RMS ? a.den[m] : 1.0f;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = t[0][e];
        float v2 = NW == 2 ? t[NW - 1][e] : 0.0f;
        if constexpr (RMS) { v = v / den; if constexpr (NW == 2) v2 = v2 / den; }
        if (a.bias) v = v + a.bias[n + e];
        if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)m * a.ldr + n + e] + v;
        if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
        if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
        o[e] = v;
    }
    *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = make_float4(o[0], o[1], o[2], o[3]);
}

static bool lm_gemm_geo3(const GemmArgs& a) {
    if (!a.xp || getenv("Q3_GEMM_GEO1")) return false;
    const char* ge = getenv("Q3_GEMM_GEO");
    const int force = ge ? atoi(ge) : 0;
    const int nt3 = a.epi == EPI_SWIGLU ? 128 : 256;
    const bool ok3 = a.Kpad % 128 == 0 && a.N % nt3 == 0;
    const int nMt = (a.M + 127) / 128;
    (void)nMt;                                                   // any grid size: with split-K small grids fill the chip too (round 2:
    return ok3 && force != 2;                                    //   forced geometry 3 beat geometry 2 at 200 / 500 / 1000 positions already)
}

hipError_t launch_lm_gemm(const GemmArgs& a, hipStream_t st) {
    if (a.Kpad % 32 || a.K % 4 || a.K > a.Kpad || a.ldx % 4 || a.ldy % 4 || a.N % 64 || a.M < 1 || !a.W) return hipErrorInvalidValue;
    const bool rms = a.norm_w != nullptr;
    if (rms && !a.den) return hipErrorInvalidValue;
    static const bool geo1 = getenv("Q3_GEMM_GEO1") != nullptr;          // A/B aid: the 64 x 128 geometry
    // geometry 3 (256-row workgroup tiles, LDS-DMA, one barrier per stage) when it fills the chip: one workgroup per CU,
    // so its grid wants >= 256 workgroups; Q3_GEMM_GEO=2 / 3 forces a geometry (A/B aid, read per call)
    if (a.xp && !geo1) {
        const int nt3 = a.epi == EPI_SWIGLU ? 128 : 256;
        const int nMt = (a.M + 127) / 128;
        if (lm_gemm_geo3(a)) {
            // super-tile shape: 4 x 8 when the grid fills the chip; fewer M tiles per super-tile for small problems so
            // that every XCD still gets work (Q3_GEMM3_SUP="m,n" overrides, A/B aid)
            const int nNt = a.N / nt3, T = nMt * nNt;
            GemmArgs b = a;
            b.sup_n = nNt < 8 ? nNt : 8;
            b.sup_m = (T / 8 + b.sup_n - 1) / b.sup_n; b.sup_m = b.sup_m < 1 ? 1 : (b.sup_m > 4 ? 4 : b.sup_m);
            if (const char* se = getenv("Q3_GEMM3_SUP")) { int m_ = 0, n_ = 0; if (sscanf(se, "%d,%d", &m_, &n_) == 2 && m_ > 0 && n_ > 0) { b.sup_m = m_; b.sup_n = n_; } }
            const int nsup = ((nMt + b.sup_m - 1) / b.sup_m) * ((nNt + b.sup_n - 1) / b.sup_n);
            dim3 g3((unsigned)(8 * ((nsup + 7) / 8) * b.sup_m * b.sup_n));
            // K ranges and split-K (see the kernel): 8 ranges when the stage count is a multiple of 16; the ranges are
            // spread over 2 / 4 / 8 workgroups while that still leaves the grid within one round of the chip and the
            // partial sums fit the workspace (Q3_GEMM3_KSPLIT = forced split, A/B aid)
            const int nst = a.Kpad >> 6;
            b.k_ranges = (nst % 16 == 0) ? 8 : 1;
            b.k_split = 1;
            const int nw = a.epi == EPI_SWIGLU ? 2 : 1;
            if (b.k_ranges == 8 && a.splitk_ws) {
                const char* ke = getenv("Q3_GEMM3_KSPLIT");
                int ks = 1;
                if (ke) ks = atoi(ke);
                else while (ks < 8 && T * ks * 2 <= 256) ks *= 2;
                if (ks != 1 && ks != 2 && ks != 4 && ks != 8) ks = 1;
                if ((size_t)8 * nw * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) ks = 1;
                b.k_split = ks;
            }
            b.wg_per_split = (int)g3.x;
            g3.x *= b.k_split;
#define Q3_GEMM3(E, R) do { hipLaunchKernelGGL((k_lm_gemm3<E, R>), g3, dim3(512), 0, st, b); \
                            if (b.k_split > 1) hipLaunchKernelGGL((k_gemm_splitk_reduce<E, R>), dim3((unsigned)(((size_t)a.M * (a.N / 4) + 255) / 256)), dim3(256), 0, st, b); } while (0)
            switch (a.epi) {
                case EPI_NONE: if (rms) Q3_GEMM3(EPI_NONE, true); else Q3_GEMM3(EPI_NONE, false); return hipGetLastError();
                case EPI_RESID: if (rms) return hipErrorInvalidValue; Q3_GEMM3(EPI_RESID, false); return hipGetLastError();
                case EPI_SILU: if (rms) return hipErrorInvalidValue; Q3_GEMM3(EPI_SILU, false); return hipGetLastError();
                case EPI_SWIGLU: if (!rms || !a.W2) return hipErrorInvalidValue; Q3_GEMM3(EPI_SWIGLU, true); return hipGetLastError();
                default: return hipErrorInvalidValue;
            }
#undef Q3_GEMM3
        }
    }
    if (a.xp && !geo1) {
        const int nt = a.epi == EPI_SWIGLU ? 64 : 128;
        if (a.N % nt == 0) {
            static const int q_env = [] { const char* e = getenv("Q3_GEMM_NSPLIT"); return e ? atoi(e) : 0; }();   // tuning aid
            const int nMt = (a.M + 127) / 128, nNt = a.N / nt;
            int Q = q_env > 0 ? q_env : 2;
            while (Q > 1 && nNt % Q) --Q;
            GemmArgs b = a; b.n_split = Q;
            const int units = nMt * Q, rounds = (units + 7) / 8;
            dim3 g2(8 * rounds * (nNt / Q));
#define Q3_GEMM2(E, R) hipLaunchKernelGGL((k_lm_gemm2<E, R>), g2, dim3(256), 0, st, b)
            switch (a.epi) {
                case EPI_NONE: if (rms) Q3_GEMM2(EPI_NONE, true); else Q3_GEMM2(EPI_NONE, false); return hipGetLastError();
                case EPI_RESID: if (rms) return hipErrorInvalidValue; Q3_GEMM2(EPI_RESID, false); return hipGetLastError();
                case EPI_SILU: if (rms) return hipErrorInvalidValue; Q3_GEMM2(EPI_SILU, false); return hipGetLastError();
                case EPI_SWIGLU: if (!rms || !a.W2) return hipErrorInvalidValue; Q3_GEMM2(EPI_SWIGLU, true); return hipGetLastError();
                default: return hipErrorInvalidValue;
            }
#undef Q3_GEMM2
        }
    }
    dim3 grid((a.M + 127) / 128, a.N / 64);
#define Q3_GEMM(E, R) do { if (a.xp) hipLaunchKernelGGL((k_lm_gemm<E, R, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((k_lm_gemm<E, R, false>), grid, dim3(256), 0, st, a); } while (0)
    switch (a.epi) {
        case EPI_NONE: if (rms) Q3_GEMM(EPI_NONE, true); else Q3_GEMM(EPI_NONE, false); break;
        case EPI_RESID: if (rms) return hipErrorInvalidValue; Q3_GEMM(EPI_RESID, false); break;
        case EPI_SILU: if (rms) return hipErrorInvalidValue; Q3_GEMM(EPI_SILU, false); break;
        case EPI_SWIGLU: if (!rms || !a.W2) return hipErrorInvalidValue; Q3_GEMM(EPI_SWIGLU, true); break;
        default: return hipErrorInvalidValue;
    }
#undef Q3_GEMM
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Causal GQA attention for RQ consecutive query rows of one sequence per workgroup. grid (row blocks, nkv, sequences);
// 256 threads = 8 groups of 32 lanes, a group owns one cached position at a time (float4 K/V loads, 512 B per
// position) and applies it to every (query row, query head) of the block whose position is >= that key: the decode
// kernel's online softmax per group, groups merged through LDS. q comes from qbuf (normed + roped by
// k_qknorm_rope_kv, which also appended this chunk's K/V rows before this launch).
// ------------------------------------------------------------------------------------------------
template <int NREP, int RQ>
__global__ __launch_bounds__(256) void k_attn_prefill(AttnArgs a) {
    constexpr int NQ = NREP * RQ;
    constexpr int NH = NQ > 8 ? 8 : NQ;                       // (row, head) states merged per LDS round (32 KB)
    __shared__ float sm_m[NH][8], sm_l[NH][8];
    __shared__ __attribute__((aligned(16))) float sm_acc[NH][8][HEAD_DIM];
    const int blk = blockIdx.x, kvh = blockIdx.y, seq = blockIdx.z;
    const int tid = threadIdx.x, grp = tid >> 5, li = tid & 31;
    const int rps = a.rows_per_seq;
    const int i0 = blk * RQ;                                   // first row of the block inside the sequence's chunk
    const int base_pos = a.pos_dev ? a.pos_dev[seq] : a.pos_static;
    const int nrows = (rps - i0) < RQ ? (rps - i0) : RQ;
    const int last_pos = base_pos + i0 + nrows - 1;
    const float scale = 0.08838834764831845f;

    float4 q[NQ];
#pragma unroll
    for (int i = 0; i < RQ; ++i)
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            const int row = seq * rps + i0 + (i < nrows ? i : 0);
            q[i * NREP + r] = *reinterpret_cast<const float4*>(a.qbuf + ((size_t)row * a.nh + kvh * NREP + r) * HEAD_DIM + li * 4);
        }
    float m[NQ], l[NQ];
    float4 acc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { m[j] = -INFINITY; l[j] = 0.0f; acc[j] = make_float4(0.f, 0.f, 0.f, 0.f); }

    const ptrdiff_t vd = kv_vd(a);
    for (int p = grp; p <= last_pos; p += 8) {
        const float* kr = kv_krow(a, seq, kvh, p) + li * 4;
        const float4 kk = *reinterpret_cast<const float4*>(kr);
        const float4 vv = *reinterpret_cast<const float4*>(kr + vd);
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
            if (i < nrows && p <= base_pos + i0 + i) {         // causal: key position <= query position
#pragma unroll
                for (int r = 0; r < NREP; ++r) {
                    const int j = i * NREP + r;
                    float s = q[j].x * kk.x + q[j].y * kk.y + q[j].z * kk.z + q[j].w * kk.w;
                    s = phalf_sum(s) * scale;
                    const float mn = fmaxf(m[j], s);
                    const float corr = expf(m[j] - mn);
                    const float pe = expf(s - mn);
                    l[j] = l[j] * corr + pe;
                    acc[j].x = acc[j].x * corr + pe * vv.x; acc[j].y = acc[j].y * corr + pe * vv.y;
                    acc[j].z = acc[j].z * corr + pe * vv.z; acc[j].w = acc[j].w * corr + pe * vv.w;
                    m[j] = mn;
                }
            }
        }
    }
#pragma unroll
    for (int h0 = 0; h0 < NQ; h0 += NH) {
        if (h0) __syncthreads();
#pragma unroll
        for (int jj = 0; jj < NH; ++jj) {
            const int j = h0 + jj;
            if (li == 0) { sm_m[jj][grp] = m[j]; sm_l[jj][grp] = l[j]; }
            *reinterpret_cast<float4*>(&sm_acc[jj][grp][li * 4]) = acc[j];
        }
        __syncthreads();
        for (int t = tid; t < NH * HEAD_DIM; t += 256) {
            const int jj = t / HEAD_DIM, d = t % HEAD_DIM, j = h0 + jj, i = j / NREP, r = j % NREP;
            if (i >= nrows) continue;
            float M = -INFINITY;
#pragma unroll
            for (int g = 0; g < 8; ++g) M = fmaxf(M, sm_m[jj][g]);
            float L = 0.0f, A = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float wgt = sm_m[jj][g] == -INFINITY ? 0.0f : expf(sm_m[jj][g] - M);
                L += sm_l[jj][g] * wgt;
                A += sm_acc[jj][g][d] * wgt;
            }
            const int row = seq * rps + i0 + i;
            a.out[(size_t)row * a.ld_out + (kvh * NREP + r) * HEAD_DIM + d] = A / L;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same attention on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation), flash
// style. Workgroup = 4 waves on one (sequence, kv head, block of 128/NREP query rows); wave w owns 32 query vectors
// (head w % NREP, rows 32·(w / NREP) .. +32) and keeps Q (64 registers per lane), the 32x128 output accumulator and the
// running (max, sum) of its rows; K/V tiles of 32 cached positions are staged once per workgroup in LDS and shared
// by the 4 waves. Per tile: S = Q·Kᵀ (64 MFMAs; the head dimension is walked as d = 64·(lane/32) + step, the same
// permutation on both operands, so every LDS read is a float4), scale, causal mask, online softmax in the accumulator
// layout (row statistics by 5-step lane reductions), P through a wave-private LDS patch into A-operand layout,
// O += P·V (64 MFMAs). The VALU version above needs ~2500 lane-operations per (query, head, key); this one 256 MFMA flops.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float pf32x16_t;
constexpr int KVP = 132;     // LDS pitch (floats) of a staged K / V row: 128 + 4 → float4 reads of 16 rows hit distinct banks
constexpr int PP = 36;       // pitch of the P patch rows

__device__ __forceinline__ float prow_max(float v) {       // over the 32 lanes of a half-wave
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

template <int NREP>
__global__ __launch_bounds__(256) void k_attn_prefill_mfma(AttnArgs a) {
    constexpr int ROWS_WG = 128 / NREP;
    __shared__ __attribute__((aligned(16))) float sK[32 * KVP];
    __shared__ __attribute__((aligned(16))) float sV[32 * KVP];
    __shared__ __attribute__((aligned(16))) float sP[4][32 * PP];
    const int blk = blockIdx.x, kvh = blockIdx.y, seq = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    const int rps = a.rows_per_seq;
    const int base_pos = a.pos_dev ? a.pos_dev[seq] : a.pos_static;
    const int head = kvh * NREP + (wave % NREP);
    const int r0 = blk * ROWS_WG + (wave / NREP) * 32;         // first chunk row of this wave
    const int wg_rows = (rps - blk * ROWS_WG) < ROWS_WG ? (rps - blk * ROWS_WG) : ROWS_WG;
    const int wg_last_pos = base_pos + blk * ROWS_WG + wg_rows - 1;
    const int n_tiles = wg_last_pos / 32 + 1;
    const float scale = 0.08838834764831845f;

    // Q: lane (i = li, lk) holds Q[i][64·lk + s], s = 0..63
    float q[64];
    {
        const int i = r0 + li;
        const bool ok = i < rps;
        const float* src = a.qbuf + ((size_t)(seq * rps + (ok ? i : 0)) * a.nh + head) * HEAD_DIM + lk * 64;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 f = ok ? *reinterpret_cast<const float4*>(src + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
            q[4 * t] = f.x; q[4 * t + 1] = f.y; q[4 * t + 2] = f.z; q[4 * t + 3] = f.w;
        }
    }
    pf32x16_t O[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[b][r] = 0.0f;
    float mrow[16], lrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.0f; }
    const int my_first_pos = base_pos + r0, my_last_pos = base_pos + r0 + 31;   // (rows past the chunk end are discarded)

    const ptrdiff_t vd = kv_vd(a);
    const int skey = tid >> 3, sc = (tid & 7) * 16;            // staging role: key row, 16-float column chunk
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();
        {
            const int p = tile * 32 + skey;
            const bool ok = p <= wg_last_pos;
            const float* kp = kv_krow(a, seq, kvh, ok ? p : 0) + sc;
            const float* vp = kp + vd;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 kf = ok ? *reinterpret_cast<const float4*>(kp + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 vf = ok ? *reinterpret_cast<const float4*>(vp + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&sK[skey * KVP + sc + 4 * t]) = kf;
                *reinterpret_cast<float4*>(&sV[skey * KVP + sc + 4 * t]) = vf;
            }
        }
        __syncthreads();
        if (tile * 32 > my_last_pos) continue;                 // wave-uniform: every key of the tile is in this wave's future
        // S = Q·Kᵀ: B operand lane (j = li, lk) = K[j][64·lk + s]
        pf32x16_t S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
        const float* krow = &sK[li * KVP + lk * 64];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * t);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * t], kf.x, S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * t + 1], kf.y, S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * t + 2], kf.z, S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * t + 3], kf.w, S, 0, 0, 0);
        }
        // lane (key j = li, lk) holds S[row(r, lk)][j]; scale, causal mask, online softmax per row
        const int keypos = tile * 32 + li;
        const bool need_mask = tile * 32 + 31 > my_first_pos;
        float* pw = &sP[wave][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
            float sv = S[r] * scale;
            if (need_mask && keypos > my_first_pos + i) sv = -INFINITY;
            const float mx = prow_max(sv);
            const float mn = fmaxf(mrow[r], mx);               // finite: key 0 is visible to every row
            const float corr = expf(mrow[r] - mn);
            const float pe = expf(sv - mn);
            const float ps = phalf_sum(pe);
            lrow[r] = lrow[r] * corr + ps;
            mrow[r] = mn;
#pragma unroll
            for (int b = 0; b < 4; ++b) O[b][r] *= corr;
            pw[i * PP + li] = pe;
        }
        // O += P·V: A operand lane (i = li, lk) = P[i][16·lk + s]; B operand lane (d = li, lk) = V[16·lk + s][32·b + d]
        // (LDS accesses of one wave are in order: the patch written above is complete when these reads issue)
        float pa[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 f = *reinterpret_cast<const float4*>(pw + li * PP + lk * 16 + 4 * t);
            pa[4 * t] = f.x; pa[4 * t + 1] = f.y; pa[4 * t + 2] = f.z; pa[4 * t + 3] = f.w;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float* vcol = &sV[(lk * 16) * KVP + b * 32 + li];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2)
                O[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s2], vcol[s2 * KVP], O[b], 0, 0, 0);
        }
    }
    // out[row][head·128 + 32·b + li] = O[b][r] / l_row
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int row = r0 + i;
        if (row < rps) {
            float* dst = a.out + (size_t)(seq * rps + row) * a.ld_out + head * HEAD_DIM + li;
            const float inv = lrow[r];
#pragma unroll
            for (int b = 0; b < 4; ++b) dst[b * 32] = O[b][r] / inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Third generation: the transposed product. Sᵀ = K·Qᵀ (A = the staged K tile, B = the wave's Q registers) puts ONE
// QUERY in each lane's accumulator column, so the online-softmax statistics are per-lane reductions over the 16
// registers plus one cross-half shuffle (the S = Q·Kᵀ form above pays 160 lane shuffles per tile for them), and the
// probabilities are already in B-operand position for Oᵀ = Vᵀ·Pᵀ — no LDS patch for P. MFMA step r of the second
// product pairs the key of register r in the lower lane half with the key of register r in the upper half, which only
// selects the V row each half reads. The next K/V tile is requested into registers before the current tile's MFMAs
// and committed to LDS after the barrier, so the global round trip is off the critical path; long (late) query blocks
// are dispatched first. 4105-position prefill of the 1.7B talker: 3.03 -> see DESIGN §4.5 ms of attention per layer.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int rps_blocks(int rps, int rows_wg) { return (rps + rows_wg - 1) / rows_wg; }
template <int NREP>
__global__ __launch_bounds__(256, 2) void k_attn_prefill_t(AttnArgs a) {      // 2 waves per SIMD: <= 256 VGPRs + AGPRs
    constexpr int ROWS_WG = 128 / NREP;
    __shared__ __attribute__((aligned(16))) float sK[32 * KVP];
    __shared__ __attribute__((aligned(16))) float sV[32 * KVP];
    // n_splits == 2: the key range of every query block is halved over two workgroups (partials in a.part, merged by
    // k_attn_merge). All workgroups of a launch are resident at once, so the launch lasts as long as its longest
    // workgroup — the last query block's walk over every key tile; halving that chain is worth more than the merge costs.
    const int halves = a.n_splits == 2 ? 2 : 1;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8; (sequence, kv head) pair p is served by XCD p % 8 only, so the
    // 4.2 MB of K/V a pair's query blocks all walk (4105 positions) stay in that XCD's L2 instead of cycling eight
    // pairs' worth through every L2. Within an XCD: long (late) query blocks first.
    int blk, half, kvh, seq;
    {
        const int nb = (rps_blocks(a.rows_per_seq, ROWS_WG)) * halves, npairs = (a.B / a.rows_per_seq) * a.nkv;
        const int lin = (int)blockIdx.x, xcd = lin & 7, j = lin >> 3;
        const int pair = xcd + 8 * (j / nb);                    // this XCD's (j / nb)-th pair
        if (pair >= npairs) return;                             // padding workgroups of the last round
        const int bid = nb - 1 - j % nb;
        blk = bid / halves; half = bid - blk * halves; kvh = pair % a.nkv; seq = pair / a.nkv;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    const int rps = a.rows_per_seq;
    const int base_pos = a.pos_dev ? a.pos_dev[seq] : a.pos_static;
    const int head = kvh * NREP + (wave % NREP);
    const int r0 = blk * ROWS_WG + (wave / NREP) * 32;         // first chunk row of this wave
    const int wg_rows = (rps - blk * ROWS_WG) < ROWS_WG ? (rps - blk * ROWS_WG) : ROWS_WG;
    const int wg_last_pos = base_pos + blk * ROWS_WG + wg_rows - 1;
    const int n_tiles = wg_last_pos / 32 + 1;
    const float scale = 0.08838834764831845f;

    // Q as the B operand: lane (query i = li, lk) holds Q[i][64·lk + s], s = 0..63 (the head dimension is walked as
    // d = 64·(lane/32) + step on both operands, so every LDS read of K is a float4)
    float q[64];
    {
        const int i = r0 + li;
        const bool ok = i < rps;
        const float* src = a.qbuf + ((size_t)(seq * rps + (ok ? i : 0)) * a.nh + head) * HEAD_DIM + lk * 64;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 f = ok ? *reinterpret_cast<const float4*>(src + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
            q[4 * t] = f.x; q[4 * t + 1] = f.y; q[4 * t + 2] = f.z; q[4 * t + 3] = f.w;
        }
    }
    pf32x16_t O[4];                                            // Oᵀ: tile b holds d = 4·row(r, lk) + b, column = query li
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[b][r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;                          // lsum: this half's share of the row's denominator
    const int qpos = base_pos + r0 + li;                       // this lane's query position
    const int my_first_pos = base_pos + r0, my_last_pos = base_pos + r0 + 31;

    const ptrdiff_t vd = kv_vd(a);
    const int skey = tid >> 3, sc = (tid & 7) * 16;            // staging role: key row, 16-float column chunk
    float4 kst[4], vst[4];
    auto fetch = [&](int tile) {
        const int p = tile * 32 + skey;
        const bool ok = p <= wg_last_pos;
        const float* kp = kv_krow(a, seq, kvh, ok ? p : 0) + sc;
        const float* vp = kp + vd;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kst[t] = ok ? *reinterpret_cast<const float4*>(kp + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
            vst[t] = ok ? *reinterpret_cast<const float4*>(vp + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int h0 = (n_tiles + 1) / 2;
    const int t_begin = (halves == 2 && half == 1) ? h0 : 0, t_end = (halves == 2 && half == 0) ? h0 : n_tiles;
    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                                       // the previous tile's LDS reads are done
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<float4*>(&sK[skey * KVP + sc + 4 * t]) = kst[t];
            *reinterpret_cast<float4*>(&sV[skey * KVP + sc + 4 * t]) = vst[t];
        }
        __syncthreads();
        fetch(tile + 1 < t_end ? tile + 1 : tile);             // lands under this tile's MFMAs (unconditional: see k_lm_gemm2)
        if (tile * 32 > my_last_pos) continue;                 // wave-uniform: every key of the tile is in this wave's future
        pf32x16_t S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
        const float* krow = &sK[li * KVP + lk * 64];           // A operand: lane (key j = li, lk) = K[j][64·lk + s]
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * t);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, q[4 * t], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, q[4 * t + 1], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, q[4 * t + 2], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, q[4 * t + 3], S, 0, 0, 0);
        }
        // register r of this lane = (key tile·32 + (r&3) + 8(r>>2) + 4·lk, query li): scale, causal mask, online softmax
        const bool need_mask = tile * 32 + 31 > my_first_pos;
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int keypos = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            float sv = S[r] * scale;
            if (need_mask && keypos > qpos) sv = -INFINITY;
            S[r] = sv; cm = fmaxf(cm, sv);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);                         // finite from tile 0 on: key 0 is visible to every query
        const float corr = expf(m - mn);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float pe = expf(S[r] - mn); S[r] = pe; ps += pe; }
        lsum = lsum * corr + ps;
        m = mn;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[b][r] *= corr;
        // Oᵀ += Vᵀ·Pᵀ: A operand lane (row li, lk) = V[key_r(lk)][4·li + b] — output row i of tile b is d = 4i + b, so a
        // lane's four tiles read one float4 of V per step (16 ds_read_b128 per tile instead of 64 ds_read_b32) and
        // finish as four consecutive d; B operand = S[r] (this lane's query column)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 vf = *reinterpret_cast<const float4*>(&sV[((r & 3) + 8 * (r >> 2) + 4 * lk) * KVP + 4 * li]);
            O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, S[r], O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, S[r], O[1], 0, 0, 0);
            O[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, S[r], O[2], 0, 0, 0);
            O[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, S[r], O[3], 0, 0, 0);
        }
    }
    const float den = lsum + __shfl_xor(lsum, 32);
    const int row = r0 + li;
    if (halves == 2) {
        if (row < rps) {                                       // partial record: O (relative to m), m, l — k_attn_merge's format
            float* rec = a.part + (((size_t)(seq * rps + row) * a.nh + head) * 2 + half) * PART_STRIDE;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
                *reinterpret_cast<float2*>(rec + 4 * i) = make_float2(O[0][r], O[1][r]);
                *reinterpret_cast<float2*>(rec + 4 * i + 2) = make_float2(O[2][r], O[3][r]);
            }
            if (lk == 0) { rec[HEAD_DIM] = m; rec[HEAD_DIM + 1] = den; }
        }
        return;
    }
    if (row < rps) {
        float* dst = a.out + (size_t)(seq * rps + row) * a.ld_out + head * HEAD_DIM;
#pragma unroll
        for (int r = 0; r < 16; ++r) {                         // register r of the four tiles = d 4i .. 4i+3, i = row(r, lk)
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
            *reinterpret_cast<float4*>(dst + 4 * i) = make_float4(O[0][r] / den, O[1][r] / den, O[2][r] / den, O[3][r] / den);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fourth generation: the same transposed flash attention on the bf16 matrix cores. The f32-input MFMA above runs at
// the f32 VECTOR rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 is 16x that. Every f32 operand is written as its exact
// three-term bf16 split x = h + m + l and a product a·b is taken as the six MFMA terms
//     ah·bh + ah·bm + am·bh + am·bm + ah·bl + al·bh          (dropped: am·bl, al·bm, al·bl <= 2^-24 |a||b|)
// accumulated in f32 — 6/16 of the f32-MFMA time for the same ~1 ulp-of-f32 products.
//   * K and V of the cached positions are split ONCE per layer by k_kv_planes into 32-key tiles
//     [pair][tile][K h,m,l | Vᵀ h,m,l][8 KB]: K rows [key][128 d], V transposed [d][32 keys] with the keys of a tile in
//     the order the S accumulator hands them to the second product (below), so both are plain 16-byte fragment reads;
//   * workgroup = 8 waves on one (sequence, kv head, 256/NREP query rows): a K/V tile (48 KB) is staged once for 256
//     query vectors by LDS-DMA, double-buffered, one barrier per tile (the structure of k_lm_gemm3); bank swizzles on
//     the source side: K slot ^= key & 15, Vᵀ slot ^= (d >> 2) & 3;
//   * Sᵀ = K·Qᵀ: Q lives in registers as its three planes (96 VGPRs); 48 MFMAs per tile and wave. Register r of the
//     accumulator is key (r&3) + 8(r>>2) + 4·(lane/32) of query lane%32, so after the per-lane online softmax the
//     probabilities of registers 8t .. 8t+7 ARE the B operand of k-step t of Oᵀ = Vᵀ·Pᵀ once split into planes —
//     k_kv_planes stores V's keys in exactly that order (position p of a tile = key (p&3) + 8(2(p>>4) + ((p>>2)&1)) +
//     4((p>>3)&1)); 48 MFMAs for the second product.
// ------------------------------------------------------------------------------------------------
constexpr int X3_PLANE = 32 * HEAD_DIM * 2;          // bytes of one plane of one 32-key tile
constexpr int X3_TILE = 6 * X3_PLANE;                // K h, m, l, Vᵀ h, m, l
typedef __attribute__((ext_vector_type(16))) float pf32x16b_t;

__global__ __launch_bounds__(256) void k_kv_planes(AttnArgs a, int n_pos, int tiles_alloc, unsigned char* __restrict__ kvp) {
    __shared__ float sV[32 * 129];
    const int tile = blockIdx.x, pair = blockIdx.y, tid = threadIdx.x;
    unsigned char* dst = kvp + ((size_t)pair * tiles_alloc + tile) * X3_TILE;
    const int key = tid >> 3, dc = (tid & 7) * 16;
    const int p = tile * 32 + key;
    const bool ok = p < n_pos;                                   // positions past the prompt: zeros (masked by causality anyway)
    const float* krow = kv_krow(a, pair / a.nkv, pair % a.nkv, ok ? p : 0) + dc;      // pair = sequence * nkv + kv head
    const float* vrow = krow + kv_vd(a);
    float kv[16], vv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 kf = ok ? *reinterpret_cast<const float4*>(krow + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vf = ok ? *reinterpret_cast<const float4*>(vrow + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        kv[4 * t] = kf.x; kv[4 * t + 1] = kf.y; kv[4 * t + 2] = kf.z; kv[4 * t + 3] = kf.w;
        vv[4 * t] = vf.x; vv[4 * t + 1] = vf.y; vv[4 * t + 2] = vf.z; vv[4 * t + 3] = vf.w;
    }
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
        pu32x4_t h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { uint32_t hh, mm, ll; psplit3_pair(kv[8 * h8 + 2 * e], kv[8 * h8 + 2 * e + 1], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
        unsigned char* o = dst + key * 256 + (dc + 8 * h8) * 2;
        *reinterpret_cast<pu32x4_t*>(o) = h;
        *reinterpret_cast<pu32x4_t*>(o + X3_PLANE) = m;
        *reinterpret_cast<pu32x4_t*>(o + 2 * X3_PLANE) = l;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) sV[key * 129 + dc + e] = vv[e];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = tid + 256 * i, d = item >> 2, slot = item & 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pp = slot * 8 + e;
            const int kk = (pp & 3) + 8 * (2 * (pp >> 4) + ((pp >> 2) & 1)) + 4 * ((pp >> 3) & 1);
            v[e] = sV[kk * 129 + d];
        }
        pu32x4_t h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { uint32_t hh, mm, ll; psplit3_pair(v[2 * e], v[2 * e + 1], hh, mm, ll); h[e] = hh; m[e] = mm; l[e] = ll; }
        unsigned char* o = dst + 3 * X3_PLANE + d * 64 + slot * 16;
        *reinterpret_cast<pu32x4_t*>(o) = h;
        *reinterpret_cast<pu32x4_t*>(o + X3_PLANE) = m;
        *reinterpret_cast<pu32x4_t*>(o + 2 * X3_PLANE) = l;
    }
}
hipError_t launch_kv_planes(const AttnArgs& kv, int n_pairs, int n_pos, int tiles_alloc,
                            unsigned char* kvp, hipStream_t st) {
    const int tiles = (n_pos + 31) / 32;
    const int cap = kv.kv_pages ? KV_MAX_PAGES * KV_PAGE_POS : kv.max_seq;
    if (tiles < 1 || tiles > tiles_alloc || n_pos > cap || kv.nkv < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_kv_planes, dim3(tiles, n_pairs), dim3(256), 0, st, kv, n_pos, tiles_alloc, kvp);
    return hipGetLastError();
}

template <int NREP>
__global__ __launch_bounds__(512) void k_attn_prefill_x3(AttnArgs a) {
    constexpr int ROWS_WG = 256 / NREP;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * X3_TILE];
    const int halves = a.n_splits == 2 ? 2 : 1;
    int blk, half, kvh, seq, pair;
    {   // XCD-aware 1-D grid of the third generation: a (sequence, kv head) pair is served by one XCD; long blocks first
        const int nb = (rps_blocks(a.rows_per_seq, ROWS_WG)) * halves, npairs = (a.B / a.rows_per_seq) * a.nkv;
        const int lin = (int)blockIdx.x, xcd = lin & 7, j = lin >> 3;
        pair = xcd + 8 * (j / nb);
        if (pair >= npairs) return;
        const int bid = nb - 1 - j % nb;
        blk = bid / halves; half = bid - blk * halves; kvh = pair % a.nkv; seq = pair / a.nkv;
    }
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rps = a.rows_per_seq;
    const int base_pos = a.pos_dev ? a.pos_dev[seq] : a.pos_static;
    const int head = kvh * NREP + (wave % NREP);
    const int r0 = blk * ROWS_WG + (wave / NREP) * 32;         // first chunk row of this wave
    const int wg_rows = (rps - blk * ROWS_WG) < ROWS_WG ? (rps - blk * ROWS_WG) : ROWS_WG;
    const int wg_last_pos = base_pos + blk * ROWS_WG + wg_rows - 1;
    const int n_tiles = wg_last_pos / 32 + 1;
    const float scale = 0.08838834764831845f;

    // Q planes as the B operand: lane (query li, lk), k-step t holds d = 16t + 8·lk .. +8
    pu32x4_t qh[8], qm[8], ql[8];
    {
        const int i = r0 + li;
        const bool ok = i < rps;
        const float* src = a.qbuf + ((size_t)(seq * rps + (ok ? i : 0)) * a.nh + head) * HEAD_DIM + lk * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 f0 = ok ? *reinterpret_cast<const float4*>(src + 16 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 f1 = ok ? *reinterpret_cast<const float4*>(src + 16 * t + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t h, m, l;
            psplit3_pair(f0.x, f0.y, h, m, l); qh[t][0] = h; qm[t][0] = m; ql[t][0] = l;
            psplit3_pair(f0.z, f0.w, h, m, l); qh[t][1] = h; qm[t][1] = m; ql[t][1] = l;
            psplit3_pair(f1.x, f1.y, h, m, l); qh[t][2] = h; qm[t][2] = m; ql[t][2] = l;
            psplit3_pair(f1.z, f1.w, h, m, l); qh[t][3] = h; qm[t][3] = m; ql[t][3] = l;
        }
    }
    pf32x16b_t O[4];                                           // Oᵀ tile b: register r = d 32b + (r&3) + 8(r>>2) + 4·lk of query li
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[b][r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;
    const int qpos = base_pos + r0 + li;
    const int my_first_pos = base_pos + r0, my_last_pos = base_pos + r0 + 31;

    // LDS-DMA role: wave w moves 1-KB block w of each of the six planes (K: keys 4w .. 4w+3, Vᵀ: d 16w .. 16w+15)
    const unsigned char* const tiles = a.kvp + (size_t)pair * a.kvp_tiles * X3_TILE;
    const int skey = wave * 4 + (lane >> 4), sd = wave * 16 + (lane >> 2);
    const unsigned koff = (unsigned)(skey * 256 + (((lane & 15) ^ (skey & 15)) << 4));
    const unsigned voff = (unsigned)(sd * 64 + (((lane & 3) ^ ((sd >> 2) & 3)) << 4));
    auto stage = [&](int tile, int buf) {
        const unsigned char* src = tiles + (size_t)tile * X3_TILE;                    // scalar
        unsigned char* dst = smem + buf * X3_TILE + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < 6; ++pl)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pl * X3_PLANE + (pl < 3 ? koff : voff)),
                                             (__attribute__((address_space(3))) void*)(dst + pl * X3_PLANE), 16, 0, 0);
    };
    // fragment reads. K (A operand of Sᵀ): lane (key li, lk), k-step t: slot 2t + lk of row li, swizzled by li & 15.
    // Vᵀ (A operand of Oᵀ): lane (d = 32b + li, lk), k-step t: slot 2t + lk of row d, swizzled by (li >> 2) & 3.
    const unsigned kfrag = (unsigned)(li * 256), ksw = (unsigned)(li & 15);
    const unsigned vfrag = (unsigned)(3 * X3_PLANE + li * 64), vsw = (unsigned)((li >> 2) & 3);

    const int h0 = (n_tiles + 1) / 2;
    const int t_begin = (halves == 2 && half == 1) ? h0 : 0, t_end = (halves == 2 && half == 0) ? h0 : n_tiles;
    if (t_begin < t_end) stage(t_begin, 0);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int buf = (tile - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // this tile has landed; the other buffer's reads are done
        if (tile + 1 < t_end) stage(tile + 1, buf ^ 1);
        if (tile * 32 > my_last_pos) continue;                 // wave-uniform: every key of the tile is in this wave's future
        const unsigned char* sb = smem + buf * X3_TILE;
        pf32x16b_t S0, S1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S0[r] = 0.0f; S1[r] = 0.0f; }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned so = kfrag + (((unsigned)(2 * t + lk) ^ ksw) << 4);
            const pbf16x8_t kh = __builtin_bit_cast(pbf16x8_t, *reinterpret_cast<const pu32x4_t*>(sb + so));
            const pbf16x8_t km = __builtin_bit_cast(pbf16x8_t, *reinterpret_cast<const pu32x4_t*>(sb + X3_PLANE + so));
            const pbf16x8_t kl = __builtin_bit_cast(pbf16x8_t, *reinterpret_cast<const pu32x4_t*>(sb + 2 * X3_PLANE + so));
            S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, __builtin_bit_cast(pbf16x8_t, qh[t]), S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, __builtin_bit_cast(pbf16x8_t, qm[t]), S1, 0, 0, 0);
            S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, __builtin_bit_cast(pbf16x8_t, qh[t]), S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, __builtin_bit_cast(pbf16x8_t, qm[t]), S1, 0, 0, 0);
            S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, __builtin_bit_cast(pbf16x8_t, ql[t]), S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, __builtin_bit_cast(pbf16x8_t, qh[t]), S1, 0, 0, 0);
        }
        // register r of this lane = (key tile·32 + (r&3) + 8(r>>2) + 4·lk, query li): scale, causal mask, online softmax
        const bool need_mask = tile * 32 + 31 > my_first_pos;
        float S[16];
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int keypos = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            float sv = (S0[r] + S1[r]) * scale;
            if (need_mask && keypos > qpos) sv = -INFINITY;
            S[r] = sv; cm = fmaxf(cm, sv);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);                         // finite from tile 0 on: key 0 is visible to every query
        const float corr = expf(m - mn);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float pe = expf(S[r] - mn); S[r] = pe; ps += pe; }
        lsum = lsum * corr + ps;
        m = mn;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[b][r] *= corr;
        // Pᵀ planes: registers 8t .. 8t+7 are the eight keys of k-step t this lane supplies
        pu32x4_t ph[2], pm[2], pl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h, mm, l;
                psplit3_pair(S[8 * t + 2 * e], S[8 * t + 2 * e + 1], h, mm, l);
                ph[t][e] = h; pm[t][e] = mm; pl[t][e] = l;
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {                   // two d tiles at a time: their MFMAs alternate accumulators
                pbf16x8_t vh[2], vm[2], vl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned vo = vfrag + (unsigned)((2 * bp + u) * 32 * 64) + (((unsigned)(2 * t + lk) ^ vsw) << 4);
                    vh[u] = __builtin_bit_cast(pbf16x8_t, *reinterpret_cast<const pu32x4_t*>(sb + vo));
                    vm[u] = __builtin_bit_cast(pbf16x8_t, *reinterpret_cast<const pu32x4_t*>(sb + X3_PLANE + vo));
                    vl[u] = __builtin_bit_cast(pbf16x8_t, *reinterpret_cast<const pu32x4_t*>(sb + 2 * X3_PLANE + vo));
                }
                const pbf16x8_t Ph = __builtin_bit_cast(pbf16x8_t, ph[t]), Pm = __builtin_bit_cast(pbf16x8_t, pm[t]), Pl = __builtin_bit_cast(pbf16x8_t, pl[t]);
                pf32x16b_t& Oa = O[2 * bp]; pf32x16b_t& Ob = O[2 * bp + 1];
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[0], Ph, Oa, 0, 0, 0);    // small terms first
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[1], Ph, Ob, 0, 0, 0);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], Pl, Oa, 0, 0, 0);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], Pl, Ob, 0, 0, 0);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm[0], Pm, Oa, 0, 0, 0);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm[1], Pm, Ob, 0, 0, 0);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm[0], Ph, Oa, 0, 0, 0);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vm[1], Ph, Ob, 0, 0, 0);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], Pm, Oa, 0, 0, 0);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], Pm, Ob, 0, 0, 0);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[0], Ph, Oa, 0, 0, 0);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh[1], Ph, Ob, 0, 0, 0);
            }
    }
    const float den = lsum + __shfl_xor(lsum, 32);
    const int row = r0 + li;
    if (row >= rps) return;
    if (halves == 2) {                                         // partial record: O (relative to m), m, l — k_attn_merge's format
        float* rec = a.part + (((size_t)(seq * rps + row) * a.nh + head) * 2 + half) * PART_STRIDE;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                float* o = rec + 32 * b + 8 * q4 + 4 * lk;     // PART_STRIDE = 130 floats: 8-byte aligned records only
                *reinterpret_cast<float2*>(o) = make_float2(O[b][4 * q4], O[b][4 * q4 + 1]);
                *reinterpret_cast<float2*>(o + 2) = make_float2(O[b][4 * q4 + 2], O[b][4 * q4 + 3]);
            }
        if (lk == 0) { rec[HEAD_DIM] = m; rec[HEAD_DIM + 1] = den; }
        return;
    }
    float* dst = a.out + (size_t)(seq * rps + row) * a.ld_out + head * HEAD_DIM;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<float4*>(dst + 32 * b + 8 * q4 + 4 * lk) =
                make_float4(O[b][4 * q4] / den, O[b][4 * q4 + 1] / den, O[b][4 * q4 + 2] / den, O[b][4 * q4 + 3] / den);
}

hipError_t launch_attn_prefill(const AttnArgs& a, hipStream_t st) {
    const int nrep = a.nh / a.nkv;
    const int rps = a.rows_per_seq;
    if (rps < 1 || a.B % rps) return hipErrorInvalidValue;
    static const bool valu = getenv("Q3_PREFILL_ATTN_VALU") != nullptr;      // A/B aid: the VALU kernel
    if (!valu && (nrep == 1 || nrep == 2 || nrep == 4)) {
        if (a.kvp) {                                              // bf16x3 generation: 256/nrep query rows per workgroup
            const int rows_x3 = 256 / nrep;
            const int nb = ((rps + rows_x3 - 1) / rows_x3) * ((a.n_splits == 2 && a.part) ? 2 : 1);
            const int npairs = (a.B / rps) * a.nkv, rounds = (npairs + 7) / 8;
            dim3 gx((unsigned)(8 * rounds) * nb);
            if (nrep == 1) hipLaunchKernelGGL((k_attn_prefill_x3<1>), gx, dim3(512), 0, st, a);
            else if (nrep == 2) hipLaunchKernelGGL((k_attn_prefill_x3<2>), gx, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_attn_prefill_x3<4>), gx, dim3(512), 0, st, a);
            return hipGetLastError();
        }
        const int rows_wg = 128 / nrep;
        dim3 grid((rps + rows_wg - 1) / rows_wg, a.nkv, a.B / rps);
        static const bool gen2 = getenv("Q3_PREFILL_ATTN_GEN2") != nullptr;  // A/B aid: the S = Q·Kᵀ generation
        if (!gen2 && a.n_splits == 2 && a.part) grid.x *= 2;
        if (gen2) {
            if (nrep == 1) hipLaunchKernelGGL((k_attn_prefill_mfma<1>), grid, dim3(256), 0, st, a);
            else if (nrep == 2) hipLaunchKernelGGL((k_attn_prefill_mfma<2>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_attn_prefill_mfma<4>), grid, dim3(256), 0, st, a);
        } else {
            const int npairs = (a.B / rps) * a.nkv, rounds = (npairs + 7) / 8;
            dim3 g1((unsigned)(8 * rounds) * grid.x);            // grid.x already counts (query blocks x key halves)
            if (nrep == 1) hipLaunchKernelGGL((k_attn_prefill_t<1>), g1, dim3(256), 0, st, a);
            else if (nrep == 2) hipLaunchKernelGGL((k_attn_prefill_t<2>), g1, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_attn_prefill_t<4>), g1, dim3(256), 0, st, a);
        }
        return hipGetLastError();
    }
    constexpr int RQ = 8;
    dim3 grid((rps + RQ - 1) / RQ, a.nkv, a.B / rps);
    if (nrep == 1) hipLaunchKernelGGL((k_attn_prefill<1, RQ>), grid, dim3(256), 0, st, a);
    else if (nrep == 2) hipLaunchKernelGGL((k_attn_prefill<2, RQ>), grid, dim3(256), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace q3
