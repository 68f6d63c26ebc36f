// q3_kernels_gemv.hip — the weight-streaming kernel of the decode path, second generation:
// skinny GEMM y[m][n] = Σ_k x[m][k]·W[n][k] for the M ≤ 16 live tokens of a batch, on the matrix
// cores without giving up f32 accuracy.
//
//   * Weights are bf16 in HBM, PRE-TILED at upload into MFMA A-operand order: tile (n/16, k/32) is
//     one contiguous 1 KiB block whose 16-byte slot `lane` holds W[n0 + (lane&15)][k0 + (lane>>4)*8 .. +8],
//     so one `global_load_dwordx4` per lane streams a whole tile, fully coalesced, and consecutive
//     k-steps of a row-tile are consecutive in memory.
//   * x stays f32 in global memory (it is a few KB and L2-resident); each lane loads the 8 values of its
//     (m = lane&15, k-group = lane>>4) slot and splits them EXACTLY into three bf16 terms
//     x = hi + mid + lo (24 mantissa bits), so three `v_mfma_f32_16x16x32_bf16` per tile reproduce the
//     f32 product W·x to f32-roundoff — the "bf16x3" trick; bf16·bf16 products are exact in the f32
//     accumulator, only the summation order differs from a VALU fmaf chain.
//   * A workgroup = 8 waves owns ONE 16-row tile (two for SwiGLU: gate and up) and splits K eight ways;
//     partial 16×16 tiles are reduced through LDS in fixed wave order (deterministic), then the epilogue
//     (÷rms, +bias, +residual, SiLU, SwiGLU) runs on 256 threads and stores 64-byte row segments.
//   * Fused input RMSNorm costs nothing extra: Σx² is accumulated from the same x loads, the norm weight
//     is folded into x before the split (z = x·w), and the 1/sqrt(mean+eps) scalar is applied to the
//     finished dot product: y = (Σ W·(x·w)) / den.  No LDS staging, one barrier per kernel.
#include "q3_kernels.h"

#include <math.h>
#include <stdlib.h>

namespace q3 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// one v_cvt_pk_bf16_f32 (RNE); written as a compiler builtin, NOT inline asm, so that hipcc's hazard
// recognizer pads the VALU-write → MFMA-operand-read wait states itself (an asm statement is opaque to it:
// the 4x4x4 kernel read stale operands with the asm form)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

struct Split3 { u32x4_t hi, mid, lo; };

// exact 3-way bf16 split of 8 floats (packed pairs: element 2i in the low half of word i)
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const uint32_t h = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const uint32_t m = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = cvt_pk_bf16(sa, sb);
    }
    return s;
}

__device__ __forceinline__ f32x4_t mfma3(const u32x4_t& w, const Split3& s, f32x4_t acc) {
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, s.hi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, s.mid), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, s.lo), acc, 0, 0, 0);
    return acc;
}

// One k-step for this lane: finish the B operand (mask, Σx², norm weight, bf16x3 split) and run the MFMAs.
template <int NW, bool RMS>
__device__ __forceinline__ void gemv_step(bool valid, const float4& x0, const float4& x1, const float4& n0, const float4& n1,
                                          const u32x4_t& wa, const u32x4_t& wb, f32x4_t& acc0, f32x4_t& acc1, float& ss) {
    float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = valid ? xv[e] : 0.0f;
    if constexpr (RMS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(xv[e], xv[e], ss);
        xv[0] *= n0.x; xv[1] *= n0.y; xv[2] *= n0.z; xv[3] *= n0.w; xv[4] *= n1.x; xv[5] *= n1.y; xv[6] *= n1.z; xv[7] *= n1.w;
    }
    const Split3 sp = split3(xv);
    acc0 = mfma3(wa, sp, acc0);
    if constexpr (NW == 2) acc1 = mfma3(wb, sp, acc1);
}

// NWAVES waves split K; weights for up to G k-steps are requested up front (G KiB per wave in flight per
// matrix) before any of them is consumed, so a wave's whole slice is usually one HBM round trip.
template <int EPI, bool RMS, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_gemv_mfma(LinArgs a) {
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int G = (NW == 2 || RMS) ? 4 : 6;
    __shared__ __attribute__((aligned(16))) float red[NWAVES][NW][256];
    __shared__ float ssq[NWAVES][4][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int S = a.Kpad >> 5;                       // k-steps of 32
    const int s0 = (wave * S) / NWAVES, s1 = ((wave + 1) * S) / NWAVES;
    const size_t tile_base = (size_t)blockIdx.x * S * 64 + lane;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + tile_base;
    const u32x4_t* __restrict__ wp2 = NW == 2 ? reinterpret_cast<const u32x4_t*>(a.W2) + tile_base : wp;
    const bool act = m < a.M;
    const float* __restrict__ xr = a.x + (size_t)(act ? m : 0) * a.ldx + kg * 8;
    const float* __restrict__ nwp = RMS ? a.norm_w + kg * 8 : nullptr;

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.0f;
    for (int sb = s0; sb < s1; sb += G) {
        // every load of the group (weights from HBM, x / norm weight from L2) is issued before any is consumed
        u32x4_t wa[G], wb[G];
        float4 xa[G], xb[G], na[G], nb[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);      // clamp: duplicate load, masked below
            wa[i] = __builtin_nontemporal_load(wp + (size_t)s * 64);
            if constexpr (NW == 2) wb[i] = __builtin_nontemporal_load(wp2 + (size_t)s * 64);
            const int ko = (s * 32 + kg * 8) < a.K ? s * 32 : 0;    // K tail (K % 32 != 0): clamp, masked below
            xa[i] = *reinterpret_cast<const float4*>(xr + ko);
            xb[i] = *reinterpret_cast<const float4*>(xr + ko + 4);
            if constexpr (RMS) {
                na[i] = *reinterpret_cast<const float4*>(nwp + ko);
                nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = sb + i;
            if (s < s1) {
                const bool valid = act && (s * 32 + kg * 8) < a.K;
                gemv_step<NW, RMS>(valid, xa[i], xb[i], RMS ? na[i] : xa[i], RMS ? nb[i] : xb[i], wa[i], NW == 2 ? wb[i] : wa[i],
                                   acc0, acc1, ss);
            }
        }
    }
    // partial tile → LDS, layout [col m][row]: lane (m, kg) owns rows kg*4 .. kg*4+3
    *reinterpret_cast<f32x4_t*>(&red[wave][0][m * 16 + kg * 4]) = acc0;
    if constexpr (NW == 2) *reinterpret_cast<f32x4_t*>(&red[wave][1][m * 16 + kg * 4]) = acc1;
    if constexpr (RMS) ssq[wave][kg][m] = ss;
    __syncthreads();
    if (tid < 256) {
        const int col = tid >> 4, row = tid & 15;
        if (col < a.M) {
            float v = 0.0f, v2 = 0.0f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) {
                v += red[w][0][tid];
                if constexpr (NW == 2) v2 += red[w][1][tid];
            }
            if constexpr (RMS) {
                float tot = 0.0f;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w)
#pragma unroll
                    for (int g = 0; g < 4; ++g) tot += ssq[w][g][col];
                const float den = sqrtf(tot / (float)a.K + a.eps);
                v = v / den;
                if constexpr (NW == 2) v2 = v2 / den;
            }
            const int n = blockIdx.x * 16 + row;
            if (n < a.N) {
                if (a.bias) v = v + a.bias[n];
                if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)col * a.ldr + n] + v;
                if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
                if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
                a.y[(size_t)col * a.ldy + n] = v;
            }
        }
    }
}

template <int EPI, bool RMS>
static hipError_t launch_gemv_t(const LinArgs& a, hipStream_t st) {
    const int tiles = (a.N + 15) / 16;
    const int S = a.Kpad >> 5;
    // 16 waves per tile when the tile count alone cannot fill the chip or the per-wave slice gets long
    bool big = (tiles < 256 && S >= 32) || S >= 128;
    static const int force = [] { const char* e = getenv("Q3_GEMV_WAVES"); return e ? atoi(e) : 0; }();   // tuning aid
    if (force == 8) big = false; else if (force == 16) big = true;
    if (big) hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 16>), dim3(tiles), dim3(16 * 64), 0, st, a);
    else hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 8>), dim3(tiles), dim3(8 * 64), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Small-N variant: 4-row tiles on v_mfma_f32_4x4x4_16b_bf16. A 16-row tile leaves most of the chip idle
// when N <= 2048 (64-128 workgroups); here a workgroup owns 4 output rows, so N = 1024 already gives 256
// workgroups. The instruction's 16 independent 4x4x4 blocks are used as 16 K-slices: lane = 4*b + i loads
// the 16 bytes W[n0+i][k0 + 8b .. +8) (tile = 4 rows x 128 k = 1 KiB, again one coalesced load per lane),
// lane = 4*b + j supplies x[m = 4*mg + j][k0 + 8b .. +8) split into three bf16 terms; block b accumulates
// its own slice and the 16 partial 4x4 results are summed across lanes (xor 4, 8, 16, 32) at the end.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ f32x4_t mfma4(unsigned a0, unsigned a1, unsigned b0, unsigned b1, f32x4_t acc) {
    const u32x2_t a = {a0, a1}, b = {b0, b1};
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma4_tile(const u32x4_t& w, const Split3& s, f32x4_t acc) {
    acc = mfma4(w[0], w[1], s.hi[0], s.hi[1], acc);  acc = mfma4(w[2], w[3], s.hi[2], s.hi[3], acc);
    acc = mfma4(w[0], w[1], s.mid[0], s.mid[1], acc); acc = mfma4(w[2], w[3], s.mid[2], s.mid[3], acc);
    acc = mfma4(w[0], w[1], s.lo[0], s.lo[1], acc);  acc = mfma4(w[2], w[3], s.lo[2], s.lo[3], acc);
    return acc;
}
__device__ __forceinline__ float sum_over_blocks(float v) {   // lanes 4b+j, all b
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}

template <int EPI, bool RMS, int MG>
__global__ __launch_bounds__(512) void k_gemv_mfma4(LinArgs a) {
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int G = 2;
    __shared__ float red[8][NW][MG][4][4];
    __shared__ float ssq[8][MG][4];
    const int nwv = blockDim.x >> 6;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 3, kb = lane >> 2;
    const int S = a.Kpad >> 7;                       // k-steps of 128
    const int s0 = (wave * S) / nwv, s1 = ((wave + 1) * S) / nwv;
    const size_t tile_base = (size_t)blockIdx.x * S * 64 + lane;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + tile_base;
    const u32x4_t* __restrict__ wp2 = NW == 2 ? reinterpret_cast<const u32x4_t*>(a.W2) + tile_base : wp;
    const float* xr[MG]; bool act[MG];
#pragma unroll
    for (int g = 0; g < MG; ++g) {
        const int m = g * 4 + j;
        act[g] = m < a.M;
        xr[g] = a.x + (size_t)(act[g] ? m : 0) * a.ldx + kb * 8;
    }
    const float* __restrict__ nwp = RMS ? a.norm_w + kb * 8 : nullptr;

    f32x4_t acc[NW][MG];
    float ss[MG];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int g = 0; g < MG; ++g) acc[w][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < MG; ++g) ss[g] = 0.0f;

    for (int sb = s0; sb < s1; sb += G) {
        u32x4_t wa[G], wb[G];
        float4 xa[G][MG], xb[G][MG], na[G], nb[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
            wa[i] = __builtin_nontemporal_load(wp + (size_t)s * 64);
            if constexpr (NW == 2) wb[i] = __builtin_nontemporal_load(wp2 + (size_t)s * 64);
            const int ko = (s * 128 + kb * 8) < a.K ? s * 128 : 0;
#pragma unroll
            for (int g = 0; g < MG; ++g) {
                xa[i][g] = *reinterpret_cast<const float4*>(xr[g] + ko);
                xb[i][g] = *reinterpret_cast<const float4*>(xr[g] + ko + 4);
            }
            if constexpr (RMS) {
                na[i] = *reinterpret_cast<const float4*>(nwp + ko);
                nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = sb + i;
            if (s < s1) {
                const bool kok = (s * 128 + kb * 8) < a.K;
#pragma unroll
                for (int g = 0; g < MG; ++g) {
                    const bool valid = act[g] && kok;
                    float xv[8] = {xa[i][g].x, xa[i][g].y, xa[i][g].z, xa[i][g].w, xb[i][g].x, xb[i][g].y, xb[i][g].z, xb[i][g].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[e] = valid ? xv[e] : 0.0f;
                    if constexpr (RMS) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss[g] = fmaf(xv[e], xv[e], ss[g]);
                        xv[0] *= na[i].x; xv[1] *= na[i].y; xv[2] *= na[i].z; xv[3] *= na[i].w;
                        xv[4] *= nb[i].x; xv[5] *= nb[i].y; xv[6] *= nb[i].z; xv[7] *= nb[i].w;
                    }
                    const Split3 sp = split3(xv);
                    acc[0][g] = mfma4_tile(wa[i], sp, acc[0][g]);
                    if constexpr (NW == 2) acc[1][g] = mfma4_tile(wb[i], sp, acc[1][g]);
                }
            }
        }
    }
    // sum the 16 k-blocks across lanes; lanes 0..3 (block 0) then hold D[i][j = lane] in acc[.][.][i]
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = sum_over_blocks(acc[w][g][i]);
                if (lane < 4) red[wave][w][g][i][lane] = v;
            }
    if constexpr (RMS) {
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            const float v = sum_over_blocks(ss[g]);
            if (lane < 4) ssq[wave][g][lane] = v;
        }
    }
    __syncthreads();
    if (tid < 16 * MG) {
        const int m = tid >> 2, i = tid & 3, g = m >> 2, jj = m & 3;
        const int n = blockIdx.x * 4 + i;
        if (m < a.M && n < a.N) {
            float v = 0.0f, v2 = 0.0f;
            for (int w = 0; w < nwv; ++w) {
                v += red[w][0][g][i][jj];
                if constexpr (NW == 2) v2 += red[w][1][g][i][jj];
            }
            if constexpr (RMS) {
                float tot = 0.0f;
                for (int w = 0; w < nwv; ++w) tot += ssq[w][g][jj];
                const float den = sqrtf(tot / (float)a.K + a.eps);
                v = v / den;
                if constexpr (NW == 2) v2 = v2 / den;
            }
            if (a.bias) v = v + a.bias[n];
            if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)m * a.ldr + n] + v;
            if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
            if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
            a.y[(size_t)m * a.ldy + n] = v;
        }
    }
}

template <int EPI, bool RMS>
static hipError_t launch_gemv4_t(const LinArgs& a, hipStream_t st) {
    const int tiles = (a.N + 3) / 4;
    const int S = a.Kpad >> 7;
    const int nwv = S >= 8 ? 8 : (S >= 4 ? 4 : (S >= 2 ? 2 : 1));
    const int mg = (a.M + 3) / 4;
    if (mg <= 1) hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 1>), dim3(tiles), dim3(nwv * 64), 0, st, a);
    else if (mg == 2) hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 2>), dim3(tiles), dim3(nwv * 64), 0, st, a);
    else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 4>), dim3(tiles), dim3(nwv * 64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_gemv_tiled4(const LinArgs& a, hipStream_t st) {
    if (a.Kpad % 128 != 0 || a.K % 8 != 0 || a.K > a.Kpad || a.ldx % 4 != 0 || a.M < 1 || a.M > 16 || a.N < 1)
        return hipErrorInvalidValue;
    const bool rms = a.norm_w != nullptr;
    switch (a.epi) {
        case EPI_NONE: return rms ? launch_gemv4_t<EPI_NONE, true>(a, st) : launch_gemv4_t<EPI_NONE, false>(a, st);
        case EPI_RESID: return rms ? hipErrorInvalidValue : launch_gemv4_t<EPI_RESID, false>(a, st);
        case EPI_SILU: return rms ? hipErrorInvalidValue : launch_gemv4_t<EPI_SILU, false>(a, st);
        case EPI_SWIGLU: return rms ? launch_gemv4_t<EPI_SWIGLU, true>(a, st) : launch_gemv4_t<EPI_SWIGLU, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemv_tiled(const LinArgs& a, hipStream_t st) {
    if (a.Kpad % 32 != 0 || a.K % 8 != 0 || a.K > a.Kpad || a.ldx % 4 != 0 || a.M < 1 || a.M > 16 || a.N < 1)
        return hipErrorInvalidValue;
    const bool rms = a.norm_w != nullptr;
    switch (a.epi) {
        case EPI_NONE: return rms ? launch_gemv_t<EPI_NONE, true>(a, st) : launch_gemv_t<EPI_NONE, false>(a, st);
        case EPI_RESID: return rms ? hipErrorInvalidValue : launch_gemv_t<EPI_RESID, false>(a, st);
        case EPI_SILU: return rms ? hipErrorInvalidValue : launch_gemv_t<EPI_SILU, false>(a, st);
        case EPI_SWIGLU: return rms ? launch_gemv_t<EPI_SWIGLU, true>(a, st) : launch_gemv_t<EPI_SWIGLU, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace q3
