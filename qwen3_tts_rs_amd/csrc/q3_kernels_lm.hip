// q3_kernels_lm.hip — gfx950 (CDNA4, wave64) kernels for the autoregressive half of the hot path:
// talker decode step, 15-step code predictor, frame glue and the on-device sampler.
//
// Numerics contract: weights are bf16 in HBM exactly as the checkpoint stores them,
// every activation / accumulation / KV entry is f32, and the op order of the reference's candle-CPU
// path is kept (x / sqrt(mean+eps) * w, rotate-half RoPE with separately rounded products, softmax
// as exp(x-max)/sum, left-to-right embedding sums). Only the summation ORDER inside dot products
// and block reductions differs from the CPU oracle.
//
// Reference anchors: transformer.rs:247-467 (DecoderLayer/Attention/MLP), fused_ops.rs:49-96 +
// kernels/fused_residual_rmsnorm.cu:39-90, kv_cache.rs:290-347, code_predictor.rs:320-416,
// lib.rs:612-622, 1271-1322, generation/sampling.rs:140-319.
#include "q3_kernels.h"

#include <math.h>

namespace q3 {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {   // over aligned groups of 32 lanes
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide sum for 256-thread blocks; `red` needs 4 floats
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// argmax with first-max tie rule (candle argmax returns the first extremum)
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ int block_argmax_first(const float* vals, int n, float* red_v, int* red_i) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        float v = vals[j];
        if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }   // NaN never wins
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(bv, off); int oi = __shfl_xor(bi, off);
        argmax_combine(bv, bi, ov, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    bv = red_v[0]; bi = red_i[0];
    for (int w = 1; w < nw; ++w) argmax_combine(bv, bi, red_v[w], red_i[w]);
    if (bi == 0x7fffffff) bi = 0;
    return bi;
}

// ------------------------------------------------------------------------------------------------
// bf16-weight skinny GEMM. One wave produces R output rows for all M activations rows; the weight
// row is streamed once from HBM with 16-byte loads (8 bf16 per lane, 1 KiB per wave-instruction),
// x lives in LDS as f32 in a two-plane layout so that both ds_read_b128 of a lane are
// conflict-free: for the 512-wide k-block j, lane l reads plane A at j*512 + 4l (k = 8l..8l+3) and
// plane B at j*512 + 256 + 4l (k = 8l+4..8l+7).
// Optional fused input RMSNorm (x / sqrt(mean(x²)+eps) * w — the candle CPU form) and epilogues:
// +bias, residual add, SiLU, SwiGLU (two weight streams, one x).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xs_index(int k) {   // k multiple of 4
    return (k & ~511) + ((k & 4) ? 256 : 0) + ((k & 511) >> 3) * 4;
}

__device__ __forceinline__ float dot8(const uint4& w, const float4& a, const float4& b, float acc) {
    acc = fmaf(bf16_lo(w.x), a.x, acc); acc = fmaf(bf16_hi(w.x), a.y, acc);
    acc = fmaf(bf16_lo(w.y), a.z, acc); acc = fmaf(bf16_hi(w.y), a.w, acc);
    acc = fmaf(bf16_lo(w.z), b.x, acc); acc = fmaf(bf16_hi(w.z), b.y, acc);
    acc = fmaf(bf16_lo(w.w), b.z, acc); acc = fmaf(bf16_hi(w.w), b.w, acc);
    return acc;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

template <int M> struct LinKC { static constexpr int value = (M >= 8) ? 1024 : 2048; };

template <int M, int R, bool RMS, int EPI>
__global__ __launch_bounds__(256) void k_linear(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KC = LinKC<M>::value;
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, N = a.N;
    const int kc_cap = ((K < KC ? K : KC) + 511) & ~511;   // LDS floats per m-row
    float* xs = smem;                                      // [M][kc_cap]
    float* red = smem + M * kc_cap;                        // [M][4]

    float den[M];
#pragma unroll
    for (int m = 0; m < M; ++m) den[m] = 1.0f;
    if constexpr (RMS) {
        float ss[M];
#pragma unroll
        for (int m = 0; m < M; ++m) ss[m] = 0.0f;
        for (int k = tid * 4; k < K; k += 1024) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float4 v = *reinterpret_cast<const float4*>(a.x + (size_t)(m < a.M ? m : 0) * a.ldx + k);
                ss[m] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) ss[m] = wave_sum(ss[m]);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) red[m * 4 + wave] = ss[m];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float tot = (red[m * 4 + 0] + red[m * 4 + 1]) + (red[m * 4 + 2] + red[m * 4 + 3]);
            den[m] = sqrtf(tot / (float)K + a.eps);
        }
    }

    const int row0 = (blockIdx.x * 4 + wave) * R;
    const uint16_t* wrow[NW][R];
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        valid[r] = (row0 + r) < N;
        const size_t off = (size_t)(valid[r] ? row0 + r : 0) * K;
        wrow[0][r] = a.W + off;
        if constexpr (NW == 2) wrow[1][r] = a.W2 + off;
    }

    float acc[NW][M][R];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[w][m][r] = 0.0f;

    for (int kc0 = 0; kc0 < K; kc0 += KC) {
        const int kc = (K - kc0) < KC ? (K - kc0) : KC;
        __syncthreads();   // previous chunk fully consumed (and `red` reads done)
#pragma unroll
        for (int m = 0; m < M; ++m) {
            for (int k = tid * 4; k < kc; k += 1024) {
                float4 v = *reinterpret_cast<const float4*>(a.x + (size_t)(m < a.M ? m : 0) * a.ldx + kc0 + k);
                if constexpr (RMS) {
                    const float4 nw = *reinterpret_cast<const float4*>(a.norm_w + kc0 + k);
                    v.x = v.x / den[m] * nw.x; v.y = v.y / den[m] * nw.y;
                    v.z = v.z / den[m] * nw.z; v.w = v.w / den[m] * nw.w;
                }
                *reinterpret_cast<float4*>(xs + m * kc_cap + xs_index(k)) = v;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int k0 = lane * 8; k0 < kc; k0 += 512) {
            uint4 wv[NW][R];
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    wv[w][r] = *reinterpret_cast<const uint4*>(wrow[w][r] + kc0 + k0);
            const int xo = (k0 & ~511) + lane * 4;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float4 xa = *reinterpret_cast<const float4*>(xs + m * kc_cap + xo);
                const float4 xb = *reinterpret_cast<const float4*>(xs + m * kc_cap + xo + 256);
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[w][m][r] = dot8(wv[w][r], xa, xb, acc[w][m][r]);
            }
        }
    }

#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[w][m][r] = wave_sum(acc[w][m][r]);

    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!valid[r]) continue;
            const int n = row0 + r;
            const float bias = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (m >= a.M) continue;
                float v = acc[0][m][r];
                if (a.bias) v = v + bias;
                if constexpr (EPI == EPI_RESID) v = a.resid[(size_t)m * a.ldr + n] + v;
                if constexpr (EPI == EPI_SILU) v = silu_f(v);
                if constexpr (EPI == EPI_SWIGLU) v = silu_f(v) * acc[NW - 1][m][r];
                a.y[(size_t)m * a.ldy + n] = v;
            }
        }
    }
}

template <int M, int R, bool RMS, int EPI>
static hipError_t launch_linear_t(const LinArgs& a, hipStream_t st) {
    constexpr int KC = LinKC<M>::value;
    const int kc_cap = ((a.K < KC ? a.K : KC) + 511) & ~511;
    const size_t lds = (size_t)(M * kc_cap + M * 4) * sizeof(float);
    const int blocks = (a.N + 4 * R - 1) / (4 * R);
    hipLaunchKernelGGL((k_linear<M, R, RMS, EPI>), dim3(blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int M, int R>
static hipError_t launch_linear_mr(const LinArgs& a, hipStream_t st) {
    const bool rms = a.norm_w != nullptr;
    switch (a.epi) {
        case EPI_NONE:
            return rms ? launch_linear_t<M, R, true, EPI_NONE>(a, st) : launch_linear_t<M, R, false, EPI_NONE>(a, st);
        case EPI_RESID:
            if (rms) return hipErrorInvalidValue;
            return launch_linear_t<M, R, false, EPI_RESID>(a, st);
        case EPI_SILU:
            if (rms) return hipErrorInvalidValue;
            return launch_linear_t<M, R, false, EPI_SILU>(a, st);
        case EPI_SWIGLU:
            return rms ? launch_linear_t<M, R, true, EPI_SWIGLU>(a, st) : launch_linear_t<M, R, false, EPI_SWIGLU>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_linear(const LinArgs& a, hipStream_t st) {
    return a.tiled == 2 ? launch_gemv_tiled4(a, st) : a.tiled == 1 ? launch_gemv_tiled(a, st) : launch_linear_rowmajor(a, st);
}

hipError_t launch_linear_rowmajor(const LinArgs& a, hipStream_t st) {
    if (a.K % 8 != 0 || a.ldx % 4 != 0 || a.M < 1 || a.M > 8 || a.N < 1) return hipErrorInvalidValue;
    // rows per wave: 2 when that still leaves >= 512 workgroups (2 per CU), else 1
    const bool r2 = (a.N / 8) >= 512;
    const int Mp = a.M <= 1 ? 1 : a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
    switch (Mp) {
        case 1: return r2 ? launch_linear_mr<1, 2>(a, st) : launch_linear_mr<1, 1>(a, st);
        case 2: return r2 ? launch_linear_mr<2, 2>(a, st) : launch_linear_mr<2, 1>(a, st);
        case 4: return r2 ? launch_linear_mr<4, 2>(a, st) : launch_linear_mr<4, 1>(a, st);
        default: return r2 ? launch_linear_mr<8, 2>(a, st) : launch_linear_mr<8, 1>(a, st);
    }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm family (one 256-thread block per row)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rmsnorm(const float* x, int ldx, const float* w, float* y, int ldy, int cols,
                                                 float eps) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float* yr = y + (size_t)blockIdx.x * ldy;
    float ss = 0.0f;
    for (int c = threadIdx.x; c < cols; c += 256) { const float v = xr[c]; ss += v * v; }
    ss = block_sum_256(ss, red);
    const float den = sqrtf(ss / (float)cols + eps);
    for (int c = threadIdx.x; c < cols; c += 256) yr[c] = xr[c] / den * w[c];
}

hipError_t launch_rmsnorm(const float* x, int ldx, const float* w, float* y, int ldy, int rows, int cols, float eps,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_rmsnorm, dim3(rows), dim3(256), 0, st, x, ldx, w, y, ldy, cols, eps);
    return hipGetLastError();
}

// The reference's only hand-written kernel, re-done for wave64: pass 1 s = x + r (stored, rounded to
// T), Σ s² accumulated from the unrounded f32; wave shuffle + LDS cross-wave reduce; pass 2 re-reads
// the rounded s. Output = (normed, sum). f32 math: s / sqrt(mean+eps) * w (CPU form, fused_ops.rs:59-67).
template <typename T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ float ld(const float* p, size_t i) { return p[i]; }
    static __device__ __forceinline__ void st(float* p, size_t i, float v) { p[i] = v; }
};
template <> struct IO<uint16_t> {
    static __device__ __forceinline__ float ld(const uint16_t* p, size_t i) { return bf16_to_f32(p[i]); }
    static __device__ __forceinline__ void st(uint16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }
};

template <typename T>
__global__ __launch_bounds__(256) void k_fused_residual_rmsnorm(const T* x, const T* res, const T* w, T* normed, T* sum,
                                                                int cols, float eps) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * cols;
    float ss = 0.0f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float s = IO<T>::ld(x, base + c) + IO<T>::ld(res, base + c);
        IO<T>::st(sum, base + c, s);
        ss += s * s;
    }
    ss = block_sum_256(ss, red);   // also orders the `sum` stores of this block before pass 2 re-reads
    const float den = sqrtf(ss / (float)cols + eps);
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float s = IO<T>::ld(sum, base + c);
        IO<T>::st(normed, base + c, s / den * IO<T>::ld(w, c));
    }
}

hipError_t launch_fused_residual_rmsnorm_f32(const float* x, const float* res, const float* w, float* normed,
                                             float* sum, int rows, int cols, float eps, hipStream_t st) {
    hipLaunchKernelGGL(k_fused_residual_rmsnorm<float>, dim3(rows), dim3(256), 0, st, x, res, w, normed, sum, cols, eps);
    return hipGetLastError();
}
hipError_t launch_fused_residual_rmsnorm_bf16(const uint16_t* x, const uint16_t* res, const uint16_t* w,
                                              uint16_t* normed, uint16_t* sum, int rows, int cols, float eps,
                                              hipStream_t st) {
    hipLaunchKernelGGL(k_fused_residual_rmsnorm<uint16_t>, dim3(rows), dim3(256), 0, st, x, res, w, normed, sum, cols,
                       eps);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// q/k per-head RMSNorm + rotate-half RoPE + in-place KV append (transformer.rs:263-284,
// kv_cache.rs:290-347). grid (nh + nkv, B), one wave per head; lane i owns the pair (i, i+64).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_qknorm_rope_kv(AttnArgs a) {
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int QD = a.nh * HEAD_DIM, KD = a.nkv * HEAD_DIM;
    const int rps = a.rows_per_seq > 1 ? a.rows_per_seq : 1;
    const int seq = b / rps;
    const int pos = (a.pos_dev ? a.pos_dev[seq] : a.pos_static) + (b - seq * rps);
    const bool is_q = h < a.nh;
    const float* src = a.qkv + (size_t)b * a.ld_qkv + (is_q ? h * HEAD_DIM : QD + (h - a.nh) * HEAD_DIM);
    float x1 = src[lane], x2 = src[lane + 64];
    const float ss = wave_sum(x1 * x1 + x2 * x2);
    const float den = sqrtf(ss / (float)HEAD_DIM + a.eps);
    const float* nw = is_q ? a.q_norm_w : a.k_norm_w;
    x1 = x1 / den * nw[lane];
    x2 = x2 / den * nw[lane + 64];
    const float c = a.rope_cos[(size_t)pos * 64 + lane], s = a.rope_sin[(size_t)pos * 64 + lane];
    const float o1 = sub_rn(mul_rn(x1, c), mul_rn(x2, s));
    const float o2 = add_rn(mul_rn(x2, c), mul_rn(x1, s));
    if (is_q) {
        float* q = a.qbuf + ((size_t)b * a.nh + h) * HEAD_DIM;
        q[lane] = o1; q[lane + 64] = o2;
    } else {
        const int kvh = h - a.nh;
        float* kr = kv_krow(a, seq, kvh, pos); float* vr = kr + kv_vd(a);
        kr[lane] = o1; kr[lane + 64] = o2;
        const float* vs = a.qkv + (size_t)b * a.ld_qkv + QD + KD + kvh * HEAD_DIM;
        vr[lane] = vs[lane]; vr[lane + 64] = vs[lane + 64];
    }
}

hipError_t launch_qknorm_rope_kv(const AttnArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_qknorm_rope_kv, dim3(a.nh + a.nkv, a.B), dim3(64), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GQA decode attention, split over the KV sequence. grid (n_splits, nkv, B), 256 threads = 8 groups
// of 32 lanes; a group owns one cached position at a time (lane = 4 dims, float4 K/V loads: 512 B
// per position, coalesced) and serves all NREP query heads of its KV head from one K/V load.
// Online softmax per group, groups merged through LDS, one partial record per (head, split).
// ------------------------------------------------------------------------------------------------
template <int NREP>
__global__ __launch_bounds__(256) void k_attn_decode(AttnArgs a) {
    __shared__ float sm_m[NREP][8], sm_l[NREP][8];
    __shared__ __attribute__((aligned(16))) float sm_acc[NREP][8][HEAD_DIM];
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, grp = tid >> 5, li = tid & 31;
    const int rps = a.rows_per_seq > 1 ? a.rows_per_seq : 1;
    const int seq = b / rps;
    const int pos = (a.pos_dev ? a.pos_dev[seq] : a.pos_static) + (b - seq * rps);
    const int len = pos + 1;
    const int chunk = (len + a.n_splits - 1) / a.n_splits;
    const int start = split * chunk;
    const int end = (start + chunk) < len ? (start + chunk) : len;
    const float scale = 0.08838834764831845f;   // (1/sqrt(128)) as f32 (affine(scale,0) in candle)

    float4 q[NREP];
#pragma unroll
    for (int r = 0; r < NREP; ++r)
        q[r] = *reinterpret_cast<const float4*>(a.qbuf + ((size_t)b * a.nh + kvh * NREP + r) * HEAD_DIM + li * 4);
    float m[NREP], l[NREP];
    float4 acc[NREP];
#pragma unroll
    for (int r = 0; r < NREP; ++r) { m[r] = -INFINITY; l[r] = 0.0f; acc[r] = make_float4(0.f, 0.f, 0.f, 0.f); }

    const ptrdiff_t vd = kv_vd(a);
    for (int p = start + grp; p < end; p += 8) {
        const float* kr = kv_krow(a, seq, kvh, p) + li * 4;
        const float4 kk = *reinterpret_cast<const float4*>(kr);
        const float4 vv = *reinterpret_cast<const float4*>(kr + vd);
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            float s = q[r].x * kk.x + q[r].y * kk.y + q[r].z * kk.z + q[r].w * kk.w;
            s = half_wave_sum(s) * scale;
            const float mn = fmaxf(m[r], s);
            const float corr = expf(m[r] - mn);
            const float pe = expf(s - mn);
            l[r] = l[r] * corr + pe;
            acc[r].x = acc[r].x * corr + pe * vv.x; acc[r].y = acc[r].y * corr + pe * vv.y;
            acc[r].z = acc[r].z * corr + pe * vv.z; acc[r].w = acc[r].w * corr + pe * vv.w;
            m[r] = mn;
        }
    }
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        if (li == 0) { sm_m[r][grp] = m[r]; sm_l[r][grp] = l[r]; }
        *reinterpret_cast<float4*>(&sm_acc[r][grp][li * 4]) = acc[r];
    }
    __syncthreads();
    for (int t = tid; t < NREP * HEAD_DIM; t += 256) {
        const int r = t / HEAD_DIM, d = t % HEAD_DIM;
        float M = -INFINITY;
#pragma unroll
        for (int g = 0; g < 8; ++g) M = fmaxf(M, sm_m[r][g]);
        float L = 0.0f, A = 0.0f;
        if (M != -INFINITY) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float wgt = sm_m[r][g] == -INFINITY ? 0.0f : expf(sm_m[r][g] - M);
                L += sm_l[r][g] * wgt;
                A += sm_acc[r][g][d] * wgt;
            }
        }
        float* rec = a.part + (((size_t)b * a.nh + kvh * NREP + r) * a.n_splits + split) * PART_STRIDE;
        rec[d] = A;
        if (d == 0) { rec[HEAD_DIM] = M; rec[HEAD_DIM + 1] = L; }
    }
}

hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t st) {
    const int nrep = a.nh / a.nkv;
    dim3 grid(a.n_splits, a.nkv, a.B);
    if (nrep == 1) hipLaunchKernelGGL(k_attn_decode<1>, grid, dim3(256), 0, st, a);
    else if (nrep == 2) hipLaunchKernelGGL(k_attn_decode<2>, grid, dim3(256), 0, st, a);
    else if (nrep == 4) hipLaunchKernelGGL(k_attn_decode<4>, grid, dim3(256), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Fused decode attention: per-head q/k RMSNorm + rotate-half RoPE + in-place KV append + GQA attention in
// ONE launch (replaces k_qknorm_rope_kv + k_attn_decode, and k_attn_merge too when n_splits == 1, which is the
// code predictor's case: 16 positions at most). Every split block re-derives the (tiny) normed/roped q heads
// and the new K/V row in LDS; the split that owns position `pos` appends them to the cache, and every block
// takes position `pos` from LDS, never from the just-written global memory.
// Leading scalars: the 14 dwords the first requests depend on, preloaded into SGPRs (see k_attn_cp): position array, caches,
// q|k|v rows, norm weights, max_seq and n_splits | nkv << 8 | nh << 16.
// PAGED (the talker's cache, KV_PAGE_POS positions per page): the preloaded scalars carry the page table (p_kc), the
// layer's offset inside a page in floats (p_vc, an integer in a pointer's clothes) and the K -> V distance (p_max_seq).
// Every wave asks for the sequence's table row first — one entry per lane, beside the position load — and a row address
// then costs two ds_bpermute (page pointer of p / 128) instead of a dependent trip to memory: the first K/V requests
// leave as early as with one contiguous extent per sequence.
// PAGED = 1: rows of at most 8 pages (1024 positions: every bench session) — the table row is ONE scalar load (the address
// is uniform: 64 bytes through the scalar cache, which the workgroups of a CU share) spread over lanes 0..7; PAGED = 2: up to
// KV_MAX_PAGES pages, one entry per lane from a vector load (2048 waves asking the same 4 KB of table cost ~1 us per launch
// at B = 8 — measured: the frame 1.0 % slower than with contiguous extents — hence the scalar form where it fits).
// KV16 (paged only; an opt-in session mode, q3_session_set_kv_dtype): the cache holds bf16 — the reference GPU path's cache
// dtype (kv_cache.rs:234-310) — in the same page geometry with 2-byte elements. A lane's K / V request is 8 bytes; the new
// position's K / V are rounded to bf16 (RNE) before they are used or stored, as an append-then-attend over a bf16 cache does.
template <int NREP, int PAGED, bool KV16 = false>
__global__ __launch_bounds__(256) void k_attn_fused(const int* p_pos, const float* p_kc, const float* p_vc, const float* p_qkv, const float* p_qw,
                                                    const float* p_kw, int p_max_seq, int p_pk, AttnArgs a_in) {
    AttnArgs a = a_in;
    a.pos_dev = p_pos; a.qkv = p_qkv; a.q_norm_w = p_qw; a.k_norm_w = p_kw;
    a.n_splits = p_pk & 255; a.nkv = (p_pk >> 8) & 255; a.nh = (p_pk >> 16) & 255;
    uint32_t tab_lo = 0, tab_hi = 0;
    size_t pg_off = 0, pg_vd = 0;
    if constexpr (PAGED != 0) {
        const unsigned long long* trow = reinterpret_cast<const unsigned long long*>(p_kc) + (size_t)blockIdx.z * KV_MAX_PAGES;
        if constexpr (PAGED == 1) {
            unsigned long long e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = trow[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) if ((int)(threadIdx.x & 63) == i) { tab_lo = (uint32_t)e[i]; tab_hi = (uint32_t)(e[i] >> 32); }
        } else {
            const unsigned long long e = trow[threadIdx.x & 63];
            tab_lo = (uint32_t)e; tab_hi = (uint32_t)(e >> 32);
        }
        pg_off = reinterpret_cast<size_t>(p_vc) + (size_t)blockIdx.y * KV_PAGE_POS * HEAD_DIM;
        pg_vd = (size_t)p_max_seq;
    } else {
        a.kcache = const_cast<float*>(p_kc); a.vcache = const_cast<float*>(p_vc); a.max_seq = p_max_seq;
    }
    __shared__ __attribute__((aligned(16))) float s_q[NREP][HEAD_DIM];
    __shared__ __attribute__((aligned(16))) float s_k[HEAD_DIM], s_v[HEAD_DIM];
    __shared__ float sm_m[NREP][8], sm_l[NREP][8];
    __shared__ __attribute__((aligned(16))) float sm_acc[NREP][8][HEAD_DIM];
    Q3T_DECL Q3T(0);
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, grp = tid >> 5, li = tid & 31, wave = tid >> 6, lane = tid & 63;
    const int QD = a.nh * HEAD_DIM, KD = a.nkv * HEAD_DIM;
    const int pos = a.pos_dev ? a.pos_dev[b] : a.pos_static;
    const int len = pos + 1;
    const int chunk = (len + a.n_splits - 1) / a.n_splits;
    const int start = split * chunk;
    const int end = (start + chunk) < len ? (start + chunk) : len;
    const float scale = 0.08838834764831845f;
    const size_t cache_base = PAGED != 0 ? 0 : ((size_t)b * a.nkv + kvh) * a.max_seq * HEAD_DIM;
    // K row of position p <= pos. The two 32-lane groups of a wave ask for neighbouring positions (pe = the even group's, pe + 1),
    // so a wave needs at most TWO pages per request: page i = min(pe, pos) / 128 — wave-uniform, read from the table lanes with
    // v_readlane, no LDS and nothing to wait for — and page i + 1; a lane picks by its own position. A lane with nothing to
    // fetch (ok = false) reads row 0 of page 0, as the contiguous form does: ONE row per sequence and head, always cached.
    // (Measured on the B = 8 frame against contiguous extents: ds_bpermute per lane +0.9 %; readlane with min(p, pos) as the
    // dummy row +1.4 % — every dummy became a distinct row of another split's range, 40 % more K/V traffic.)
    const int pos_page = pos >> KV_PAGE_SHIFT;
    unsigned long long pg_zero = 0;
    if constexpr (PAGED != 0)
        pg_zero = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)tab_hi, 0) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)tab_lo, 0);
    auto krow_paged = [&](int p, int pe, bool ok) -> float* {
        int ib = pe >> KV_PAGE_SHIFT; ib = ib < pos_page ? ib : pos_page;
        ib = __builtin_amdgcn_readfirstlane(ib);
        const int ib1 = ib + 1 < KV_MAX_PAGES ? ib + 1 : ib;
        const unsigned long long p0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)tab_hi, ib) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)tab_lo, ib);
        const unsigned long long p1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)tab_hi, ib1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)tab_lo, ib1);
        const unsigned long long pg = ok ? ((p >> KV_PAGE_SHIFT) == ib ? p0 : p1) : pg_zero;
        // (the address is formed as an offset from the table pointer — a kernel argument, known to be global — not cast from
        // the integer: a generic pointer's loads are flat_load, which count on lgkmcnt as well and made every LDS / scalar
        // wait of the loop wait for the K/V stream: +1 us per launch)
        const char* gbase = reinterpret_cast<const char*>(p_kc);
        char* page = const_cast<char*>(gbase) + (ptrdiff_t)(pg - reinterpret_cast<unsigned long long>(gbase));
        const size_t elem = pg_off + (size_t)(((ok ? p : 0) + kv_rot(pg, (int)blockIdx.y)) & (KV_PAGE_POS - 1)) * HEAD_DIM;
        return reinterpret_cast<float*>(page + elem * (KV16 ? 2 : 4));      // KV16: really a uint16_t*, see the callers
    };
    struct KVReg { float4 f; };                      // one K or V fragment in flight: four floats, or four bf16 in .f.x / .f.y (KV16)
    auto kv_unpack = [](const KVReg& r) -> float4 {
        if constexpr (KV16) {
            const uint32_t lo = __float_as_uint(r.f.x), hi = __float_as_uint(r.f.y);
            return make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u));
        } else return r.f;
    };
    auto bf16_rne = [](float v) -> uint32_t { uint32_t u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; };
    // the new position's K / V rows: write-through like everything a fence-free node stores (q3_kernels.h "activation transport") —
    // they are read by later FRAMES only, but nothing may stay dirty in this XCD's L2 behind a packet without a release fence
    auto kv_st = [](auto* p, auto v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    // The first two cached K/V rows of this group are requested NOW, before the q/k-norm + RoPE prologue: they depend
    // on nothing but `pos`, and their round trip then runs under the prologue instead of after its barrier (the new
    // position itself comes from LDS after the barrier, never from the just-written global memory).
    // Every request below is UNCONDITIONAL (a position past the group's share, or the new position, reads row `start`
    // instead and the result is dropped): `s_waitcnt vmcnt` counts in issue order, so behind a load that sits in a
    // conditional hipcc waits for everything in flight — with the prefetch guarded, and the three register sets rotated
    // by copies, every key cost its own memory round trip.
    const size_t base = cache_base + li * 4;
    auto request = [&](int p, KVReg& ko, KVReg& vo) {
        const int ps = (p < end && p != pos) ? p : 0;               // row 0 always exists
        if constexpr (PAGED != 0 && KV16) {
            const uint16_t* kr = reinterpret_cast<const uint16_t*>(krow_paged(p, p - (grp & 1), p < end && p != pos)) + li * 4;
            const float2 k2 = *reinterpret_cast<const float2*>(kr), v2 = *reinterpret_cast<const float2*>(kr + pg_vd);
            ko.f.x = k2.x; ko.f.y = k2.y; vo.f.x = v2.x; vo.f.y = v2.y;
        } else if constexpr (PAGED != 0) {
            const float* kr = krow_paged(p, p - (grp & 1), p < end && p != pos) + li * 4;
            ko.f = *reinterpret_cast<const float4*>(kr);
            vo.f = *reinterpret_cast<const float4*>(kr + pg_vd);
        } else {
            ko.f = *reinterpret_cast<const float4*>(a.kcache + base + (size_t)ps * HEAD_DIM);
            vo.f = *reinterpret_cast<const float4*>(a.vcache + base + (size_t)ps * HEAD_DIM);
        }
    };
    const int p0 = start + grp;
    // folded gather (AttnArgs::g_*): this sequence's q|k|v row comes from the table row of the previous pass's argmax
    const float* qkv_row = a.qkv + (size_t)b * a.ld_qkv;
    if (a.g_logits) {
        __shared__ float g_red_v[4]; __shared__ int g_red_i[4];
        const int row = block_argmax_first(a.g_logits + (size_t)b * a.g_vocab, a.g_vocab, g_red_v, g_red_i);
        qkv_row = a.g_qkv_tab + (size_t)row * a.ld_qkv;
        if (kvh == 0 && split == 0) {
            const float4* ps = reinterpret_cast<const float4*>(a.g_proj_tab + (size_t)row * a.g_proj_dim);
            const __amdgpu_buffer_rsrc_t pd = act_rsrc(a.g_x + (size_t)b * a.g_ldx);
            for (int c = tid; c < a.g_proj_dim / 4; c += 256) act_st4(pd, c * 16, ps[c]);
            if (tid == 0 && a.g_frame_idx[b] < a.g_max_frames)      // write-through: see k_attn_cp
                __hip_atomic_store(&a.g_codes[((size_t)b * a.g_max_frames + a.g_frame_idx[b]) * 16 + a.g_code_slot], (uint32_t)row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // jobs 0..NREP-1: q heads; job NREP: the k head (+ raw v). The prologue's own loads go out FIRST (returns are counted in issue
    // order: behind the K/V requests the prologue would wait for the whole stream), the K/V requests follow, then the arithmetic.
    constexpr int NJ = (NREP + 1 + 3) / 4;
    const bool from_slices = a.qkv_part && !a.g_logits;
    // q | k | v was written by the projection in front of this launch: L1-bypassing loads; partial records / the output row:
    // write-through stores (q3_kernels.h "activation transport"). The K/V pages are cross-frame state: plain accesses.
    const __amdgpu_buffer_rsrc_t qkv_res = act_rsrc(qkv_row);
    float jx1[NJ], jx2[NJ], jv1[NJ], jv2[NJ], jn1[NJ], jn2[NJ];
    const float rc = a.rope_cos[(size_t)pos * 64 + lane], rs = a.rope_sin[(size_t)pos * 64 + lane];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = wave + 4 * jj;
        jx1[jj] = jx2[jj] = jv1[jj] = jv2[jj] = jn1[jj] = jn2[jj] = 0.0f;
        if (j <= NREP) {
            const bool is_q = j < NREP;
            const int scol = is_q ? (kvh * NREP + j) * HEAD_DIM : QD + kvh * HEAD_DIM;
            const int vcol = QD + KD + kvh * HEAD_DIM;
            if (from_slices) {          // the k job brings its v along: one round trip
                const int cols[4] = {scol + lane, scol + lane + 64, is_q ? scol + lane : vcol + lane, is_q ? scol + lane + 64 : vcol + lane + 64};
                float o[4]; qkv_from_slices<float, 4>(a, b, cols, o);
                jx1[jj] = o[0]; jx2[jj] = o[1]; jv1[jj] = o[2]; jv2[jj] = o[3];
            } else {
                // (a table row of the folded gather is static too; one form for both keeps the prologue one code path)
                jx1[jj] = act_ld1(qkv_res, (scol + lane) * 4); jx2[jj] = act_ld1(qkv_res, (scol + lane + 64) * 4);
                if (!is_q) { jv1[jj] = act_ld1(qkv_res, (vcol + lane) * 4); jv2[jj] = act_ld1(qkv_res, (vcol + lane + 64) * 4); }
            }
            const float* nw = is_q ? a.q_norm_w : a.k_norm_w;
            jn1[jj] = nw[lane]; jn2[jj] = nw[lane + 64];
        }
    }
    auto prologue_math = [&]() {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = wave + 4 * jj;
            if (j > NREP) continue;
            const bool is_q = j < NREP;
            float x1 = jx1[jj], x2 = jx2[jj];
            const float ss = wave_sum(x1 * x1 + x2 * x2);
            const float den = sqrtf(ss / (float)HEAD_DIM + a.eps);
            x1 = x1 / den * jn1[jj];
            x2 = x2 / den * jn2[jj];
            const float c = rc, sn = rs;
            const float o1 = sub_rn(mul_rn(x1, c), mul_rn(x2, sn));
            const float o2 = add_rn(mul_rn(x2, c), mul_rn(x1, sn));
            if (is_q) { s_q[j][lane] = o1; s_q[j][lane + 64] = o2; }
            else {
                const float v1 = jv1[jj], v2 = jv2[jj];
                if constexpr (KV16) {                  // the cache holds bf16: this position's K / V are what the cache will hold
                    const uint32_t b1 = bf16_rne(o1), b2 = bf16_rne(o2), c1 = bf16_rne(v1), c2 = bf16_rne(v2);
                    s_k[lane] = __uint_as_float(b1 << 16); s_k[lane + 64] = __uint_as_float(b2 << 16);
                    s_v[lane] = __uint_as_float(c1 << 16); s_v[lane + 64] = __uint_as_float(c2 << 16);
                    if (split == pos / chunk) {
                        uint16_t* kc = reinterpret_cast<uint16_t*>(krow_paged(pos, pos, true)); uint16_t* vc = kc + pg_vd;
                        kv_st(&kc[lane], (uint16_t)b1); kv_st(&kc[lane + 64], (uint16_t)b2); kv_st(&vc[lane], (uint16_t)c1); kv_st(&vc[lane + 64], (uint16_t)c2);
                    }
                } else {
                    s_k[lane] = o1; s_k[lane + 64] = o2; s_v[lane] = v1; s_v[lane + 64] = v2;
                    if (split == pos / chunk) {
                        float* kc; float* vc;
                        if constexpr (PAGED != 0) { kc = krow_paged(pos, pos, true); vc = kc + pg_vd; }
                        else { kc = a.kcache + cache_base + (size_t)pos * HEAD_DIM; vc = a.vcache + cache_base + (size_t)pos * HEAD_DIM; }
                        kv_st(&kc[lane], o1); kv_st(&kc[lane + 64], o2); kv_st(&vc[lane], v1); kv_st(&vc[lane + 64], v2);
                    }
                }
            }
        }
        zero_job(a.zero, a.zero_n, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z, tid, 256);
        __syncthreads();
        Q3T(1);
    };
    float4 q[NREP];
    float m[NREP], l[NREP];
    float4 acc[NREP];
    auto start_keys = [&]() {
#pragma unroll
        for (int r = 0; r < NREP; ++r) q[r] = *reinterpret_cast<const float4*>(&s_q[r][li * 4]);
#pragma unroll
        for (int r = 0; r < NREP; ++r) { m[r] = -INFINITY; l[r] = 0.0f; acc[r] = make_float4(0.f, 0.f, 0.f, 0.f); }
    };
    auto consume = [&](int p, const KVReg& kr_, const KVReg& vr_) {
        float4 kk = kv_unpack(kr_), vv = kv_unpack(vr_);
        if (p == pos) {                                            // the new position: from LDS
            kk = *reinterpret_cast<const float4*>(&s_k[li * 4]);
            vv = *reinterpret_cast<const float4*>(&s_v[li * 4]);
        }
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            float s = q[r].x * kk.x + q[r].y * kk.y + q[r].z * kk.z + q[r].w * kk.w;
            s = half_wave_sum(s) * scale;
            const float mn = fmaxf(m[r], s);
            const float corr = expf(m[r] - mn);
            const float pe = expf(s - mn);
            l[r] = l[r] * corr + pe;
            acc[r].x = acc[r].x * corr + pe * vv.x; acc[r].y = acc[r].y * corr + pe * vv.y;
            acc[r].z = acc[r].z * corr + pe * vv.z; acc[r].w = acc[r].w * corr + pe * vv.w;
            m[r] = mn;
        }
    };
    {
        // The first two cached K/V rows of this group are requested before the q/k-norm + RoPE arithmetic: they depend
        // on nothing but `pos`, and their round trip then runs under the prologue instead of after its barrier (the new
        // position itself comes from LDS after the barrier, never from the just-written global memory).
        // Every request is UNCONDITIONAL (a position past the group's share, or the new position, reads row 0
        // instead and the result is dropped): `s_waitcnt vmcnt` counts in issue order, so behind a load that sits in a
        // conditional hipcc waits for everything in flight — with the prefetch guarded, and the three register sets rotated
        // by copies, every key cost its own memory round trip.
        KVReg kA, vA, kB, vB, kC, vC;
        request(p0, kA, vA);
        request(p0 + 8, kB, vB);
        prologue_math();
        start_keys();
        // three register sets in rotation, the loop unrolled by three so that no set is ever copied: the rows of position
        // p + 16 are requested before position p is consumed. (Batches of 8 keys per group all in flight at once — one round
        // trip per batch — measured slower: 3.667 vs 3.609 ms/frame at B = 8; the 10-16 dummy requests of a short key range
        // cost more than the round trips they save. Round 5, the whole range of a short session in flight at once, in exact batches
        // of four per group and with the prologue's loads ahead of them: 2.675 vs 2.670 ms/frame at 300 frames, 2.80 vs 2.78 at 640.)
        for (int p = p0; p < end; p += 24) {
            request(p + 16, kC, vC); consume(p, kA, vA);
            if (p + 8 >= end) break;
            request(p + 24, kA, vA); consume(p + 8, kB, vB);
            if (p + 16 >= end) break;
            request(p + 32, kB, vB); consume(p + 16, kC, vC);
        }
    }
    Q3T(2);
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        if (li == 0) { sm_m[r][grp] = m[r]; sm_l[r][grp] = l[r]; }
        *reinterpret_cast<float4*>(&sm_acc[r][grp][li * 4]) = acc[r];
    }
    __syncthreads();
    for (int t = tid; t < NREP * HEAD_DIM; t += 256) {
        const int r = t / HEAD_DIM, d = t % HEAD_DIM;
        float M = -INFINITY;
#pragma unroll
        for (int g = 0; g < 8; ++g) M = fmaxf(M, sm_m[r][g]);
        float L = 0.0f, A = 0.0f;
        if (M != -INFINITY) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float wgt = sm_m[r][g] == -INFINITY ? 0.0f : expf(sm_m[r][g] - M);
                L += sm_l[r][g] * wgt;
                A += sm_acc[r][g][d] * wgt;
            }
        }
        const int h = kvh * NREP + r;
        if (a.n_splits == 1) {
            act_st1(act_rsrc(a.out), (b * a.ld_out + h * HEAD_DIM + d) * 4, A / L);
        } else {
            const __amdgpu_buffer_rsrc_t rec = act_rsrc(a.part + (((size_t)b * a.nh + h) * a.n_splits + split) * PART_STRIDE);
            act_st1(rec, d * 4, A);
            if (d == 0) { act_st1(rec, HEAD_DIM * 4, M); act_st1(rec, (HEAD_DIM + 1) * 4, L); }
        }
    }
    Q3T(3); act_drain(); Q3T_W(4); Q3T_FLUSH(a, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

hipError_t launch_attn_fused(const AttnArgs& a, hipStream_t st) {
    const int nrep = a.nh / a.nkv;
    if (a.n_splits > 255 || a.nh > 255 || a.nkv > 255) return hipErrorInvalidValue;
    dim3 grid(a.n_splits, a.nkv, a.B);
    const int pk = a.n_splits | (a.nkv << 8) | (a.nh << 16);
#define Q3_AF(R) hipLaunchKernelGGL((k_attn_fused<R, 0, false>), grid, dim3(256), 0, st, a.pos_dev, (const float*)a.kcache, (const float*)a.vcache, a.qkv, a.q_norm_w, a.k_norm_w, a.max_seq, pk, a)
#define Q3_AFP(R, P) hipLaunchKernelGGL((k_attn_fused<R, P, false>), grid, dim3(256), 0, st, a.pos_dev, reinterpret_cast<const float*>(a.kv_pages), reinterpret_cast<const float*>(a.kv_layer_off), \
                                        a.qkv, a.q_norm_w, a.k_norm_w, (int)a.kv_vdelta, pk, a)
    if (a.kv_pages) {
        if (a.kv_vdelta > 0x7fffffffu) return hipErrorInvalidValue;
        const bool few = a.kv_row_pages > 0 && a.kv_row_pages <= 8;        // the session's rows never hold more than 8 pages
        if (a.kv_bf16) {
#define Q3_AFP16(R, P) hipLaunchKernelGGL((k_attn_fused<R, P, true>), grid, dim3(256), 0, st, a.pos_dev, reinterpret_cast<const float*>(a.kv_pages), reinterpret_cast<const float*>(a.kv_layer_off), \
                                          a.qkv, a.q_norm_w, a.k_norm_w, (int)a.kv_vdelta, pk, a)
            if (nrep == 1) { if (few) Q3_AFP16(1, 1); else Q3_AFP16(1, 2); }
            else if (nrep == 2) { if (few) Q3_AFP16(2, 1); else Q3_AFP16(2, 2); }
            else if (nrep == 4) { if (few) Q3_AFP16(4, 1); else Q3_AFP16(4, 2); }
            else return hipErrorInvalidValue;
#undef Q3_AFP16
        } else
        if (nrep == 1) { if (few) Q3_AFP(1, 1); else Q3_AFP(1, 2); }
        else if (nrep == 2) { if (few) Q3_AFP(2, 1); else Q3_AFP(2, 2); }
        else if (nrep == 4) { if (few) Q3_AFP(4, 1); else Q3_AFP(4, 2); }
        else return hipErrorInvalidValue;
    } else
    if (nrep == 1) Q3_AF(1); else if (nrep == 2) Q3_AF(2); else if (nrep == 4) Q3_AF(4);
    else return hipErrorInvalidValue;
#undef Q3_AF
#undef Q3_AFP
    return hipGetLastError();
}

// f32 pages -> bf16 pages (the prefill always runs on f32 pages; a bf16 session converts once, when its prompt is in):
// grid (pages, layers * nkv * 2): one 128 x 128 run per workgroup, rows keep their rotated places (kv_rot depends on the
// PAGE ADDRESS, so a row moves from src rotation to dst rotation)
__global__ __launch_bounds__(256) void k_kv_pages_to_bf16(const unsigned long long* __restrict__ src_pages, const unsigned long long* __restrict__ dst_pages,
                                                          int nkv, size_t layer_stride, size_t v_delta) {
    const int pgi = blockIdx.x, run = blockIdx.y, tid = threadIdx.x;
    const int is_v = run & 1, kvh = (run >> 1) % nkv, layer = (run >> 1) / nkv;
    const unsigned long long sp = src_pages[pgi], dp = dst_pages[pgi];
    const size_t off = (size_t)layer * layer_stride + (is_v ? v_delta : 0) + (size_t)kvh * KV_PAGE_POS * HEAD_DIM;
    const float* src = reinterpret_cast<const float*>(sp) + off;
    uint16_t* dst = reinterpret_cast<uint16_t*>(dp) + off;
    const int rs = kv_rot(sp, kvh), rd = kv_rot(dp, kvh);
    for (int i = tid; i < KV_PAGE_POS * HEAD_DIM / 4; i += 256) {
        const int p = i / (HEAD_DIM / 4), c = (i % (HEAD_DIM / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)((p + rs) & (KV_PAGE_POS - 1)) * HEAD_DIM + c);
        auto rne = [](float f) -> uint32_t { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; };
        const uint32_t lo = rne(v.x) | (rne(v.y) << 16), hi = rne(v.z) | (rne(v.w) << 16);
        *reinterpret_cast<uint2*>(dst + (size_t)((p + rd) & (KV_PAGE_POS - 1)) * HEAD_DIM + c) = make_uint2(lo, hi);
    }
}
hipError_t launch_kv_pages_to_bf16(const unsigned long long* src_pages, const unsigned long long* dst_pages, int n_pages, int n_layers, int nkv,
                                   size_t layer_stride, size_t v_delta, hipStream_t st) {
    if (n_pages < 1) return hipSuccess;
    hipLaunchKernelGGL(k_kv_pages_to_bf16, dim3(n_pages, n_layers * nkv * 2), dim3(256), 0, st, src_pages, dst_pages, nkv, layer_stride, v_delta);
    return hipGetLastError();
}

// The code predictor's first pass (code_predictor.rs:337-367: talker hidden + semantic embedding as a 2-token causal
// prefill from an empty cache) in ONE launch instead of k_qknorm_rope_kv + k_attn_decode + k_attn_merge: row 2b sits at
// position 0 and sees only itself (softmax over one key: its output is its own V row), row 2b+1 at position 1 sees both.
// grid (nkv, sequences); the per-group / merge arithmetic of the generic kernels is written out for two keys.
template <int NREP>
__global__ __launch_bounds__(256) void k_attn_first2(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float s_q[NREP][HEAD_DIM];          // q heads of row 1 (row 0 needs none)
    __shared__ __attribute__((aligned(16))) float s_k[2][HEAD_DIM], s_v[2][HEAD_DIM];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int QD = a.nh * HEAD_DIM, KD = a.nkv * HEAD_DIM;
    const size_t cache_base = ((size_t)b * a.nkv + kvh) * a.max_seq * HEAD_DIM;
    // jobs: 0..NREP-1 = q heads of row 1; NREP, NREP+1 = k (+ raw v) of rows 0, 1
    for (int j = wave; j < NREP + 2; j += 4) {
        const bool is_q = j < NREP;
        const int row = is_q ? 1 : j - NREP, pos = row;
        const float* base = (row == 1 && a.g_tok) ? a.g_qkv_tab + (size_t)a.g_tok[b] * a.ld_qkv        // folded pass-1 gather
                                                  : a.qkv + (size_t)(2 * b + row) * a.ld_qkv;
        // rows of a.qkv were written by the projection in front of this launch: L1-bypassing loads (harmless on a table row);
        // K/V rows, the output and the residual row are read by later launches of the frame: write-through stores (q3_kernels.h)
        const __amdgpu_buffer_rsrc_t src = act_rsrc(base + (is_q ? (kvh * NREP + j) * HEAD_DIM : QD + kvh * HEAD_DIM));
        float x1 = act_ld1(src, lane * 4), x2 = act_ld1(src, (lane + 64) * 4);
        const float ss = wave_sum(x1 * x1 + x2 * x2);
        const float den = sqrtf(ss / (float)HEAD_DIM + a.eps);
        const float* nw = is_q ? a.q_norm_w : a.k_norm_w;
        x1 = x1 / den * nw[lane];
        x2 = x2 / den * nw[lane + 64];
        const float c = a.rope_cos[(size_t)pos * 64 + lane], sn = a.rope_sin[(size_t)pos * 64 + lane];
        const float o1 = sub_rn(mul_rn(x1, c), mul_rn(x2, sn));
        const float o2 = add_rn(mul_rn(x2, c), mul_rn(x1, sn));
        if (is_q) { s_q[j][lane] = o1; s_q[j][lane + 64] = o2; }
        else {
            const __amdgpu_buffer_rsrc_t vs = act_rsrc(base + QD + KD + kvh * HEAD_DIM);
            const float v1 = act_ld1(vs, lane * 4), v2 = act_ld1(vs, (lane + 64) * 4);
            s_k[row][lane] = o1; s_k[row][lane + 64] = o2; s_v[row][lane] = v1; s_v[row][lane + 64] = v2;
            const __amdgpu_buffer_rsrc_t kc = act_rsrc(a.kcache + cache_base + (size_t)pos * HEAD_DIM);
            const __amdgpu_buffer_rsrc_t vc = act_rsrc(a.vcache + cache_base + (size_t)pos * HEAD_DIM);
            act_st1(kc, lane * 4, o1); act_st1(kc, (lane + 64) * 4, o2); act_st1(vc, lane * 4, v1); act_st1(vc, (lane + 64) * 4, v2);
        }
    }
    if (a.g_tok && kvh == 0) {                                   // the semantic row of the residual stream
        const float4* ps = reinterpret_cast<const float4*>(a.g_proj_tab + (size_t)a.g_tok[b] * a.g_proj_dim);
        const __amdgpu_buffer_rsrc_t pd = act_rsrc(a.g_x + (size_t)b * a.g_ldx);
        for (int c = tid; c < a.g_proj_dim / 4; c += 256) act_st4(pd, c * 16, ps[c]);
    }
    // side job behind the loads: the split-K o-projection that follows adds its halves onto zeros (AttnArgs::zero, as in k_attn_cp)
    zero_job(a.zero, a.zero_n, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, tid, 256);
    __syncthreads();
    const float scale = 0.08838834764831845f;
    for (int r = wave; r < NREP; r += 4) {
        const int h = kvh * NREP + r;
        // row 0: one key, weight exp(0) / 1
        const __amdgpu_buffer_rsrc_t o0 = act_rsrc(a.out + (size_t)(2 * b) * a.ld_out + h * HEAD_DIM);
        act_st1(o0, lane * 4, s_v[0][lane]); act_st1(o0, (lane + 64) * 4, s_v[0][lane + 64]);
        // row 1: two keys
        const float q1 = s_q[r][lane], q2 = s_q[r][lane + 64];
        const float sc0 = wave_sum(q1 * s_k[0][lane] + q2 * s_k[0][lane + 64]) * scale;
        const float sc1 = wave_sum(q1 * s_k[1][lane] + q2 * s_k[1][lane + 64]) * scale;
        const float M = fmaxf(sc0, sc1);
        const float w0 = expf(sc0 - M), w1 = expf(sc1 - M);
        const float L = w0 + w1;
        const __amdgpu_buffer_rsrc_t o1 = act_rsrc(a.out + (size_t)(2 * b + 1) * a.ld_out + h * HEAD_DIM);
        act_st1(o1, lane * 4, (s_v[0][lane] * w0 + s_v[1][lane] * w1) / L);
        act_st1(o1, (lane + 64) * 4, (s_v[0][lane + 64] * w0 + s_v[1][lane + 64] * w1) / L);
    }
    act_drain();
}

hipError_t launch_attn_first2(const AttnArgs& a, hipStream_t st) {
    const int nrep = a.nh / a.nkv;
    if (a.rows_per_seq != 2 || a.B % 2 || a.kv_pages) return hipErrorInvalidValue;      // the code predictor's 17-position cache is never paged
    dim3 grid(a.nkv, a.B / 2);
    if (nrep == 1) hipLaunchKernelGGL(k_attn_first2<1>, grid, dim3(256), 0, st, a);
    else if (nrep == 2) hipLaunchKernelGGL(k_attn_first2<2>, grid, dim3(256), 0, st, a);
    else if (nrep == 4) hipLaunchKernelGGL(k_attn_first2<4>, grid, dim3(256), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Code-predictor decode attention (code_predictor.rs:370-416 through transformer.rs:247-372): the cache of a
// frame's code predictor never holds more than 16 positions, so the generic kernel above — split bookkeeping, eight
// key groups merged through LDS, two barriers — pays a split-capable kernel's latency for a <= 16-key problem
// (6.3 us per launch, 70 launches per frame). Here ONE WAVE owns one (sequence, q head): no LDS, no barrier, and
// every global request — the token's q / k / v, norm weights, RoPE row and all cached K / V rows — is issued before
// the first value is used, so the launch costs one memory round trip (two with the folded gather).
//   * lane l holds head dims 2l, 2l+1 (one 8-byte load per row and lane, 512-byte rows fully coalesced); the
//     rotate-half partner d +- 64 is lane l ^ 32.
//   * the NK partial dot products of a lane are reduced by a TRANSPOSING butterfly: across the two row pairs a lane hands half
//     of its values to lane ^ 32 / ^ 16 (v_permlane32/16_swap) and keeps the other half while more than four are alive, the
//     last four are summed over their 16-lane row by DPP rotations — VALU latency throughout (the first cut chained 17
//     ds_bpermute round trips of ~60 ns each); afterwards a lane of row r holds the complete scores of four keys.
//   * softmax exactly in the reference's form, exp(s - max) / sum with the sum taken in key order (each weight is
//     broadcast with v_readlane from the row that holds it), then sum_p w_p * V[p] in key order.
// Slot p of the NK key slots is the cached row p for p < pos, the new token (registers, never the just-written
// memory) for p == pos, and masked beyond. The q-head-0 wave of each kv group appends K / V.
// ------------------------------------------------------------------------------------------------
// ---- VALU-only cross-lane moves (k_attn_cp): a ds_bpermute (what __shfl_xor compiles to) is an LDS-pipe round trip of
// ~60 ns, and the kernel's first generation chained 17 of them behind its loads (1.65 us of a 2.9 us launch in the frame's
// timeline); DPP row rotations, v_permlane16/32_swap (gfx950) and v_readlane run at VALU latency ----
template <int N> __device__ __forceinline__ float row_ror(float v) {          // lane i of each 16-lane row reads lane (i + N) % 16
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false); }
// sum over the 16 lanes of a row, identical bits in every lane of the row (each step pairs lanes symmetrically)
__device__ __forceinline__ float row_sum16(float v) {
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
    return v;
}
__device__ __forceinline__ float lane_xor32(float v, bool upper) {            // value of lane ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(upper ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor16(float v, bool odd_row) {          // value of lane ^ 16
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(odd_row ? r[0] : r[1]);
}
__device__ __forceinline__ float read_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum_valu(float v) {                      // uniform result
    v = row_sum16(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// Leading scalars = the 14 dwords every first request depends on, preloaded into SGPRs with the wave (see Q3_LIN_PRE in
// q3_kernels_gemv.hip): K cache, q|k|v rows, this position's RoPE rows, the norm weights, V cache as a 32-bit float offset
// from the K cache, and pos | max_seq << 8 | nh << 16 | nkv << 24.
template <int NK, bool GATHER>      // GATHER: the folded gather of a pass's first layer (a.g_logits set)
__global__ __launch_bounds__(64) void k_attn_cp(const float* p_kc, const float* p_qkv, const float* p_rc, const float* p_rs,
                                                const float* p_qw, const float* p_kw, int p_vdelta, int p_pk, AttnArgs a_in) {
    AttnArgs a = a_in;
    a.pos_static = p_pk & 255; a.max_seq = (p_pk >> 8) & 255; a.nh = (p_pk >> 16) & 255; a.nkv = (p_pk >> 24) & 255;
    a.kcache = const_cast<float*>(p_kc); a.vcache = const_cast<float*>(p_kc) + p_vdelta; a.qkv = p_qkv; a.q_norm_w = p_qw; a.k_norm_w = p_kw;
    a.ld_qkv = (a.nh + 2 * a.nkv) * HEAD_DIM;
    Q3T_DECL Q3T(0); Q3T_K(7, a.max_seq);
#ifdef Q3_TRACE
    q3t_[5] = (unsigned long long)clock64();          // shader-clock counter beside the 100 MHz stamps: the clock the frame loop really runs at
#endif
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int nrep = a.nh / a.nkv, kvh = h / nrep;
    const int QD = a.nh * HEAD_DIM, KD = a.nkv * HEAD_DIM;
    const int pos = a.pos_static;
    // the cache rows of a frame's earlier passes, the q | k | v row and the logits come from earlier nodes of the SAME frame: L1-bypassing
    // loads, write-through stores (q3_kernels.h "activation transport")
    const __amdgpu_buffer_rsrc_t kres = act_rsrc(a.kcache), vres = act_rsrc(a.vcache);
    const int cache_base = (((b * a.nkv + kvh) * a.max_seq) * HEAD_DIM + 2 * lane) * 4;       // bytes (a <= 16-position cache: far below 2 GB)
    // The cached rows depend on nothing; slot p >= pos re-reads row 0, which always exists and is finite. GATHER: requested first,
    // under the argmax chain that finds the q | k | v row. Otherwise BEHIND q | k | v: returns are counted in issue order, and ahead
    // of them the norm + RoPE arithmetic waited for all 2 NK cache rows as well (they are not needed before the scores).
    float2 kc[NK], vc[NK];
    auto request_cache = [&]() {
#pragma unroll
        for (int p = 0; p < NK; ++p) {
            const int ro = cache_base + (p < pos ? p : 0) * HEAD_DIM * 4;
            kc[p] = act_ld2(kres, ro);
            vc[p] = act_ld2(vres, ro);
        }
    };
    if constexpr (GATHER) request_cache();
    const float2 qw = *reinterpret_cast<const float2*>(a.q_norm_w + 2 * lane);
    const float2 kw = *reinterpret_cast<const float2*>(a.k_norm_w + 2 * lane);
    const int ri = (2 * lane) & 63;                  // p_rc / p_rs point at this position's rows of the RoPE tables
    const float2 rc = *reinterpret_cast<const float2*>(p_rc + ri), rs = *reinterpret_cast<const float2*>(p_rs + ri);

    const float* qkv_row = a.qkv + (size_t)b * a.ld_qkv;
    if constexpr (GATHER) {      // folded gather (AttnArgs::g_*): the row is the argmax of the previous pass's logits
        const __amdgpu_buffer_rsrc_t lg = act_rsrc(a.g_logits + (size_t)b * a.g_vocab);
        float bv = -INFINITY; int bi = 0x7fffffff;
        auto take4 = [&](const float4& v, int j) {
            if (v.x > bv || (v.x == bv && j < bi)) { bv = v.x; bi = j; }
            if (v.y > bv || (v.y == bv && j + 1 < bi)) { bv = v.y; bi = j + 1; }
            if (v.z > bv || (v.z == bv && j + 2 < bi)) { bv = v.z; bi = j + 2; }
            if (v.w > bv || (v.w == bv && j + 3 < bi)) { bv = v.w; bi = j + 3; }
        };
        if (a.g_vocab == 2048) {          // the production vocabulary: all eight 16-byte requests of a lane in flight at once
            float4 v[8];                  // (a run-time trip count kept them serial: eight round trips, 4.9 us before the first use)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = act_ld4(lg, (lane * 4 + i * 256) * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) take4(v[i], lane * 4 + i * 256);
        } else if ((a.g_vocab & 3) == 0) {
            for (int j = lane * 4; j < a.g_vocab; j += 256) take4(act_ld4(lg, j * 4), j);
        } else {
            for (int j = lane; j < a.g_vocab; j += 64) { const float v = act_ld1(lg, j * 4); if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; } }
        }
        // first-max over the wave: DPP rotations inside the rows (the combine is symmetric, so every lane of a row ends with the
        // row's winner), then the four row winners by v_readlane
        argmax_combine(bv, bi, row_ror<8>(bv), row_ror_i<8>(bi)); argmax_combine(bv, bi, row_ror<4>(bv), row_ror_i<4>(bi));
        argmax_combine(bv, bi, row_ror<2>(bv), row_ror_i<2>(bi)); argmax_combine(bv, bi, row_ror<1>(bv), row_ror_i<1>(bi));
        {
            float wv = read_lane(bv, 0); int wi = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
            for (int r = 1; r < 4; ++r) argmax_combine(wv, wi, read_lane(bv, 16 * r), __builtin_amdgcn_readlane(bi, 16 * r));
            bv = wv; bi = wi;
        }
        const int row = bi == 0x7fffffff ? 0 : bi;
        qkv_row = a.g_qkv_tab + (size_t)row * a.ld_qkv;
        {   // the residual-stream row: every head's wave copies its share (one wave copying all of it finished 2.4 us after the others)
            const int per = (((a.g_proj_dim / 4) + a.nh - 1) / a.nh), c0 = h * per, c1 = (c0 + per) < a.g_proj_dim / 4 ? (c0 + per) : a.g_proj_dim / 4;
            const float4* ps = reinterpret_cast<const float4*>(a.g_proj_tab + (size_t)row * a.g_proj_dim);
            const __amdgpu_buffer_rsrc_t pd = act_rsrc(a.g_x + (size_t)b * a.g_ldx);
            for (int c = c0 + lane; c < c1; c += 64) act_st4(pd, c * 16, ps[c]);
            // write-through as well: the frame's 64-byte code record is completed by k_frame_embed from another XCD, and a dword left
            // dirty in this XCD's L2 until the frame's release would meet that workgroup's copy of the line (round 6: rows 1.. of a
            // release-free code predictor came back with clobbered records, row 0 — same XCD — never)
            if (h == 0 && lane == 0 && a.g_frame_idx[b] < a.g_max_frames)
                __hip_atomic_store(&a.g_codes[((size_t)b * a.g_max_frames + a.g_frame_idx[b]) * 16 + a.g_code_slot], (uint32_t)row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float2 q, k, v;
    if (!GATHER && a.qkv_part) {      // wide sessions: the slice sums of the split-K GEMM (see AttnArgs::qkv_part)
        const int cols[3] = {h * HEAD_DIM + 2 * lane, QD + kvh * HEAD_DIM + 2 * lane, QD + KD + kvh * HEAD_DIM + 2 * lane};
        float2 o[3]; qkv_from_slices<float2, 3>(a, b, cols, o);        // all 24 slice loads of the lane in flight at once
        q = o[0]; k = o[1]; v = o[2];
    } else {
        if constexpr (GATHER) {      // a table row: written once at model finalize
            q = *reinterpret_cast<const float2*>(qkv_row + h * HEAD_DIM + 2 * lane);
            k = *reinterpret_cast<const float2*>(qkv_row + QD + kvh * HEAD_DIM + 2 * lane);
            v = *reinterpret_cast<const float2*>(qkv_row + QD + KD + kvh * HEAD_DIM + 2 * lane);
        } else {
            const __amdgpu_buffer_rsrc_t qr = act_rsrc(qkv_row);
            q = act_ld2(qr, (h * HEAD_DIM + 2 * lane) * 4);
            k = act_ld2(qr, (QD + kvh * HEAD_DIM + 2 * lane) * 4);
            v = act_ld2(qr, (QD + KD + kvh * HEAD_DIM + 2 * lane) * 4);
        }
    }
    if constexpr (!GATHER) request_cache();

    Q3T_W(1);
    // side job behind the last load (vmcnt retires in issue order: a store ahead of the loads would sit in front of every wait for them)
    zero_job(a.zero, a.zero_n, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, lane, 64);
    // per-head RMSNorm (x / sqrt(mean + eps) * w) and rotate-half RoPE with separately rounded products
    const bool upper = lane >= 32, odd_row = (lane & 16) != 0;
    auto norm_rope = [&](float2 x, const float2& w) {
        // (the fused forms are written out here and below: left to the compiler, the GATHER instance took v_pk_mul + add — both products
        // rounded — where the other took mul + fma, and the two instances of one source line disagreed in the last place)
        const float den = sqrtf(wave_sum_valu(fmaf(x.x, x.x, x.y * x.y)) / (float)HEAD_DIM + a.eps);
        x.x = x.x / den * w.x; x.y = x.y / den * w.y;
        const float px = lane_xor32(x.x, upper), py = lane_xor32(x.y, upper);     // the d +- 64 partner
        float2 o;
        if (upper) { o.x = add_rn(mul_rn(x.x, rc.x), mul_rn(px, rs.x)); o.y = add_rn(mul_rn(x.y, rc.y), mul_rn(py, rs.y)); }
        else       { o.x = sub_rn(mul_rn(x.x, rc.x), mul_rn(px, rs.x)); o.y = sub_rn(mul_rn(x.y, rc.y), mul_rn(py, rs.y)); }
        return o;
    };
    q = norm_rope(q, qw);
    k = norm_rope(k, kw);
    if (h == kvh * nrep) {
        const int ro = cache_base + pos * HEAD_DIM * 4;
        act_st2(kres, ro, k);
        act_st2(vres, ro, v);
    }

    float s[NK];
#pragma unroll
    for (int p = 0; p < NK; ++p) {
        const float2 kk = p == pos ? k : kc[p];
        s[p] = fmaf(q.x, kk.x, q.y * kk.y);
    }
    // Reduce the NK per-lane partials over the 64 lanes. Across the two row pairs the butterfly TRANSPOSES while more than four
    // values are alive (a lane hands half of its values to lane ^ 32 / ^ 16 and keeps the other half), the last four are summed
    // over their row with DPP rotations: afterwards a lane of row r holds the complete scores of keys kbase .. kbase + 3.
    int kbase = 0;
    if constexpr (NK == 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = (upper ? s[i + 8] : s[i]) + lane_xor32(upper ? s[i] : s[i + 8], upper);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (odd_row ? s[i + 4] : s[i]) + lane_xor16(odd_row ? s[i] : s[i + 4], odd_row);
        kbase = (upper ? 8 : 0) + (odd_row ? 4 : 0);
    } else if constexpr (NK == 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (upper ? s[i + 4] : s[i]) + lane_xor32(upper ? s[i] : s[i + 4], upper);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += lane_xor16(s[i], odd_row);
        kbase = upper ? 4 : 0;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[i] += lane_xor32(s[i], upper); s[i] += lane_xor16(s[i], odd_row); }
    }
    float sc[4], M = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = (kbase + i) <= pos ? row_sum16(s[i]) * 0.08838834764831845f : -INFINITY;
        M = fmaxf(M, sc[i]);
    }
    M = fmaxf(fmaxf(read_lane(M, 0), read_lane(M, 16)), fmaxf(read_lane(M, 32), read_lane(M, 48)));
    float w4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = (kbase + i) <= pos ? expf(sc[i] - M) : 0.0f;
    float L = 0.0f;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int p = 0; p < NK; ++p) {
        const float wp = read_lane(w4[p & 3], NK == 16 ? 16 * (p >> 2) : NK == 8 ? 32 * (p >> 2) : 0);     // the row that holds key p
        const float2 vv = p == pos ? v : vc[p];
        L += wp;
        acc.x = fmaf(wp, vv.x, acc.x); acc.y = fmaf(wp, vv.y, acc.y);
    }
    Q3T(2);
    act_st2(act_rsrc(a.out), (b * a.ld_out + h * HEAD_DIM + 2 * lane) * 4, make_float2(acc.x / L, acc.y / L));
    Q3T(3); act_drain(); Q3T_W(4);
#ifdef Q3_TRACE
    q3t_[6] = (unsigned long long)clock64();
#endif
    Q3T_FLUSH(a, blockIdx.y * gridDim.x + blockIdx.x);
}

// single-row passes of a cache that never exceeds 16 positions, position known at launch (the code predictor)
bool attn_cp_ok(const AttnArgs& a) {
    const ptrdiff_t vd = a.vcache - a.kcache;
    return !a.kv_pages && !a.pos_dev && a.rows_per_seq <= 1 && a.n_splits == 1 && a.pos_static >= 0 && a.pos_static < 16 && a.pos_static < a.max_seq &&
           a.max_seq < 256 && a.nkv > 0 && a.nh < 256 && a.nh % a.nkv == 0 && a.ld_qkv == (a.nh + 2 * a.nkv) * HEAD_DIM && a.ld_out % 2 == 0 &&
           vd > -(ptrdiff_t)0x7fffffff && vd < (ptrdiff_t)0x7fffffff && (!a.g_logits || (a.g_proj_dim % 4 == 0 && a.g_ldx % 4 == 0));
}
hipError_t launch_attn_cp(const AttnArgs& a, hipStream_t st) {
    if (!attn_cp_ok(a)) return hipErrorInvalidValue;
    dim3 grid(a.nh, a.B);
    const float* rc = a.rope_cos + (size_t)a.pos_static * 64; const float* rs = a.rope_sin + (size_t)a.pos_static * 64;
    const int vdelta = (int)(a.vcache - a.kcache), pk = a.pos_static | (a.max_seq << 8) | (a.nh << 16) | (a.nkv << 24);
#define Q3_ACP(NK) if (a.g_logits) hipLaunchKernelGGL((k_attn_cp<NK, true>), grid, dim3(64), 0, st, (const float*)a.kcache, a.qkv, rc, rs, a.q_norm_w, a.k_norm_w, vdelta, pk, a); \
                   else hipLaunchKernelGGL((k_attn_cp<NK, false>), grid, dim3(64), 0, st, (const float*)a.kcache, a.qkv, rc, rs, a.q_norm_w, a.k_norm_w, vdelta, pk, a)
    if (a.pos_static < 4) Q3_ACP(4); else if (a.pos_static < 8) Q3_ACP(8); else Q3_ACP(16);
#undef Q3_ACP
    return hipGetLastError();
}

// merge the split partials: out[b][h*128+d] = Σ_s A_s[d] e^{m_s-M} / Σ_s l_s e^{m_s-M}
// A separate launch on purpose. Folding it into k_attn_fused as a "last split block to arrive merges" epilogue (release
// fence + ticket atomic + acquire fence + agent-scope loads) was built and measured on MI355X: bit-identical output,
// but the frame got 17 % SLOWER (1.7B, B = 8: 4.34 -> 5.07 ms) — device-scope fences write back / invalidate the XCD's
// L2 under every later launch, while this kernel boundary costs 1.6 us.
template <int NS>      // capacity of the fixed unroll: 16, or 64 for long-context sessions
__global__ __launch_bounds__(128) void k_attn_merge(const float* p_part, float* p_out, int p_splits, int p_nh, int p_ld_out, AttnArgs a_in) {
    AttnArgs a = a_in;                                  // leading scalars: preloaded kernel arguments (see k_attn_cp)
    a.part = const_cast<float*>(p_part); a.out = p_out; a.n_splits = p_splits; a.nh = p_nh; a.ld_out = p_ld_out;
    Q3T_DECL Q3T(0);
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    // the records were written by the attention launch in front of this one: L1-bypassing loads, write-through store (q3_kernels.h)
    const __amdgpu_buffer_rsrc_t rec = act_rsrc(a.part + ((size_t)b * a.nh + h) * a.n_splits * PART_STRIDE);
    // all loads first (fixed unroll, predicated): one memory round trip instead of 2*n_splits dependent ones
    float ms[NS], ls[NS], as[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool ok = s < a.n_splits;
        const int r = (ok ? s : 0) * PART_STRIDE * 4;
        ms[s] = ok ? act_ld1(rec, r + HEAD_DIM * 4) : -INFINITY;
        ls[s] = ok ? act_ld1(rec, r + (HEAD_DIM + 1) * 4) : 0.0f;
        as[s] = ok ? act_ld1(rec, r + d * 4) : 0.0f;
    }
    Q3T_W(1);
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s) M = fmaxf(M, ms[s]);
    float L = 0.0f, A = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float wgt = ms[s] == -INFINITY ? 0.0f : expf(ms[s] - M);
        L += ls[s] * wgt;
        A += as[s] * wgt;
    }
    Q3T(2);
    act_st1(act_rsrc(a.out), (b * a.ld_out + h * HEAD_DIM + d) * 4, A / L);
    Q3T(3); act_drain(); Q3T_W(4); Q3T_FLUSH(a, blockIdx.y * gridDim.x + blockIdx.x);
}

// two partials per (row, head) — the key halves of the long-prompt prefill attention, tens of thousands of records per
// launch: one wave per record pair (float2 per lane, both records in flight at once), four records per workgroup; the
// generic kernel above spends 66 us per layer on 4105 x 16 records, this one the time the 100 MB take
__global__ __launch_bounds__(256) void k_attn_merge2(AttnArgs a) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // idx = row * nh + head
    if (idx >= a.B * a.nh) return;
    const float* r0 = a.part + (size_t)idx * 2 * PART_STRIDE;
    const float* r1 = r0 + PART_STRIDE;
    const float2 a0 = *reinterpret_cast<const float2*>(r0 + 2 * lane), a1 = *reinterpret_cast<const float2*>(r1 + 2 * lane);
    const float m0 = r0[HEAD_DIM], l0 = r0[HEAD_DIM + 1], m1 = r1[HEAD_DIM], l1 = r1[HEAD_DIM + 1];
    const float M = fmaxf(m0, m1);
    const float w0 = m0 == -INFINITY ? 0.0f : expf(m0 - M), w1 = m1 == -INFINITY ? 0.0f : expf(m1 - M);
    float L = 0.0f; L += l0 * w0; L += l1 * w1;                                         // the generic kernel's order
    float Ax = 0.0f, Ay = 0.0f;
    Ax += a0.x * w0; Ax += a1.x * w1; Ay += a0.y * w0; Ay += a1.y * w1;
    const int b = idx / a.nh, h = idx - b * a.nh;
    *reinterpret_cast<float2*>(a.out + (size_t)b * a.ld_out + h * HEAD_DIM + 2 * lane) = make_float2(Ax / L, Ay / L);
}

hipError_t launch_attn_merge(const AttnArgs& a, hipStream_t st) {
    if (a.n_splits == 2 && a.ld_out % 2 == 0) {       // every two-record merge, whatever its size: a prompt prefilled in one pass or in several gets the same bits
        hipLaunchKernelGGL(k_attn_merge2, dim3((unsigned)(((long)a.B * a.nh + 3) / 4)), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (a.n_splits <= 16) hipLaunchKernelGGL(k_attn_merge<16>, dim3(a.nh, a.B), dim3(128), 0, st, (const float*)a.part, a.out, a.n_splits, a.nh, a.ld_out, a);
    else hipLaunchKernelGGL(k_attn_merge<MAX_SPLITS>, dim3(a.nh, a.B), dim3(128), 0, st, (const float*)a.part, a.out, a.n_splits, a.nh, a.ld_out, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// gathers / frame glue
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rows_bf16(const uint16_t* table, const uint32_t* ids, float* out, int dim) {
    const size_t r = blockIdx.x;
    const uint16_t* src = table + (size_t)ids[r] * dim;
    for (int c = threadIdx.x; c < dim; c += 256) out[r * dim + c] = bf16_to_f32(src[c]);
}
hipError_t launch_gather_rows_bf16(const uint16_t* table, const uint32_t* ids, float* out, int n_rows, int dim,
                                   hipStream_t st) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_rows_bf16, dim3(n_rows), dim3(256), 0, st, table, ids, out, dim);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_assemble_rows(const float* rows, const int* text_row, const uint16_t* codec_emb,
                                                       const int* codec_id, const float* xvec, float* out, int H,
                                                       const uint32_t* ref_codes, const uint16_t* const* cp_embs) {
    const size_t i = blockIdx.x;
    const int tr = text_row[i], cid = codec_id[i];
    for (int c = threadIdx.x; c < H; c += 256) {
        float v, cv;
        if (cid >= 0) cv = bf16_to_f32(codec_emb[(size_t)cid * H + c]);
        else if (cid == -2) cv = xvec[c];
        else if (cid <= -3) {
            // ICL reference frame: codec_emb[c0] + e0[c1] + … + e14[c15], left to right (lib.rs:1239-1257)
            const uint32_t* fr = ref_codes + (size_t)(-3 - cid) * 16;
            cv = bf16_to_f32(codec_emb[(size_t)fr[0] * H + c]);
            for (int g = 1; g < 16; ++g) cv = add_rn(cv, bf16_to_f32(cp_embs[g - 1][(size_t)fr[g] * H + c]));
        } else cv = 0.0f;
        if (tr >= 0 && cid != -1) v = rows[(size_t)tr * H + c] + cv;       // text.add(codec)  (talker.rs:480, 694, 706, 782)
        else if (tr >= 0) v = rows[(size_t)tr * H + c];
        else v = cv;
        out[i * H + c] = v;
    }
}
hipError_t launch_assemble_rows(const float* rows, const int* text_row, const uint16_t* codec_emb, const int* codec_id,
                                const float* xvec, float* out, int n, int H, hipStream_t st, const uint32_t* ref_codes,
                                const uint16_t* const* cp_embs) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_assemble_rows, dim3(n), dim3(256), 0, st, rows, text_row, codec_emb, codec_id, xvec, out, H, ref_codes, cp_embs);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_copy_rows(const float* src, int lds, float* dst, int ldd, int cols) {
    const size_t r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += 256) dst[r * ldd + c] = src[r * lds + c];
}
hipError_t launch_copy_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_rows, dim3(rows), dim3(256), 0, st, src, lds, dst, ldd, cols);
    return hipGetLastError();
}

// Code-predictor input of pass p (code_predictor.rs:337-345, 386-396): pass 0 = talker hidden,
// pass 1 = semantic embedding, pass p>=2 = embedding (table p-2) of argmax(previous pass logits);
// the argmax'd code is also recorded as codes[b][frame][p-1].
// grid (B, CPG_PARTS): every part re-derives the row id (an argmax over 2048 logits is cheaper than a second launch)
// and copies its share of the row(s) — one workgroup per sequence spent 8.6 us on the 20 KB of table rows.
constexpr int CPG_PARTS = 4;
__global__ __launch_bounds__(256) void k_cp_gather(CpGatherArgs a) {
    __shared__ float red_v[4]; __shared__ int red_i[4];
    const int b = blockIdx.x, part = blockIdx.y, tid = part * 256 + threadIdx.x, NT = CPG_PARTS * 256;
    float* out = a.out + (size_t)b * a.ld_out;
    if (a.pass == 0) {
        const float* src = a.last_hidden + (size_t)b * a.H;
        for (int c = tid; c < a.H; c += NT) out[c] = src[c];
        return;
    }
    int row;
    if (a.pass == 1) {
        row = (int)a.tok[b];
    } else {
        row = block_argmax_first(a.cp_logits + (size_t)b * a.cp_vocab, a.cp_vocab, red_v, red_i);
        if (tid == 0 && a.frame_idx[b] < a.max_frames) a.codes[((size_t)b * a.max_frames + a.frame_idx[b]) * 16 + (a.pass - 1)] = (uint32_t)row;
    }
    if (a.qkv_tab) {
        const float4* src = reinterpret_cast<const float4*>(a.qkv_tab + (size_t)row * a.qkv_dim);
        float4* dst = reinterpret_cast<float4*>(a.qkv_out + (size_t)b * a.ld_qkv_out);
        for (int c = tid; c < a.qkv_dim / 4; c += NT) dst[c] = src[c];
    }
    if (a.proj_tab) {
        const float* src = a.proj_tab + (size_t)row * a.proj_dim;
        for (int c = tid; c < a.proj_dim; c += NT) out[c] = src[c];
        return;
    }
    const uint16_t* src = (a.pass == 1 ? a.codec_emb : a.cp_emb) + (size_t)row * a.H;
    for (int c = tid; c < a.H; c += NT) out[c] = bf16_to_f32(src[c]);
}
hipError_t launch_cp_gather(const CpGatherArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_cp_gather, dim3(a.B, CPG_PARTS), dim3(256), 0, st, a);
    return hipGetLastError();
}

// lib.rs:605-622 + code_predictor.rs:497-519: record the frame's 16 codes and build the next talker
// input = semantic_embed + ((e0+e1)+…+e14) + (trailing_text[frame] | tts_pad).
// Three dependent memory round trips instead of five (round 5; the node was 14 us of a 2.8 ms frame): everything that does not
// depend on anything else — the sequence's frame index, its sampled token, its text-row bookkeeping and this thread's share of
// the last pass's logits — is requested at once; the frame's earlier codes (address needs the frame index) are requested before
// the argmax runs, so they land under it; the table pointers come with the kernel arguments; then the 17 row gathers.
__global__ __launch_bounds__(256) void k_frame_embed(FrameEmbedArgs a) {
    __shared__ float red_v[4]; __shared__ int red_i[4];
    __shared__ uint32_t codes_s[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    // trip 1: independent requests
    const int f = a.frame_idx[b];
    const uint32_t tok = a.tok[b];
    const int tlen = a.trail_len[b], tbase = a.trail_base[b], prow = a.pad_row[b];
    constexpr int LPT = 16;                                   // logits per thread kept in registers (vocab <= 4096)
    float lv[LPT];
    const float* lg = a.cp_logits_last + (size_t)b * a.cp_vocab;
    const bool in_regs = a.cp_vocab <= LPT * 256;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) { const int j = i * 256 + tid; lv[i] = j < a.cp_vocab ? lg[j] : -INFINITY; }
    }
    // trip 2 (needs f): the frame's earlier codes. A frozen sequence (SampleArgs::limit) may sit at frame_idx == max_frames: it
    // records nothing and reads its last slot
    const bool live = f < a.max_frames;
    uint32_t* frame = a.codes + ((size_t)b * a.max_frames + (live ? f : a.max_frames - 1)) * 16;
    uint32_t my_code = 0;
    if (tid >= 1 && tid < 16 && tid != a.n_acoustic) my_code = frame[tid];
    int last;
    if (in_regs) {                                            // block_argmax_first over the registers: first maximum, NaN never wins
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < LPT; ++i) { const int j = i * 256 + tid; const float v = lv[i]; if (j < a.cp_vocab && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; } }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { float ov = __shfl_xor(bv, off); int oi = __shfl_xor(bi, off); argmax_combine(bv, bi, ov, oi); }
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
        __syncthreads();
        bv = red_v[0]; bi = red_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, red_v[w], red_i[w]);
        last = bi == 0x7fffffff ? 0 : bi;
    } else {
        last = block_argmax_first(lg, a.cp_vocab, red_v, red_i);
    }
    if (tid == 0 && blockIdx.y == 0 && live) {
        frame[0] = tok;
        frame[a.n_acoustic] = (uint32_t)last;
    }
    if (tid < 16) codes_s[tid] = tid == 0 ? tok : (tid == a.n_acoustic ? (uint32_t)last : my_code);
    __syncthreads();
    // trip 3: the gathers
    const int H = a.H;
    const int row = f < tlen ? tbase + f : prow;
    const float* text = a.text_rows + (size_t)row * H;
    const uint16_t* sem = a.codec_emb + (size_t)codes_s[0] * H;
    const int c = blockIdx.y * 256 + tid;
    if (c < H) {
        float e[15];
#pragma unroll
        for (int g = 0; g < 15; ++g) e[g] = g < a.n_acoustic ? bf16_to_f32(a.cp_embs[g][(size_t)codes_s[1 + g] * H + c]) : 0.0f;
        const float sv = bf16_to_f32(sem[c]), tv = text[c];
        float acc = e[0];
#pragma unroll
        for (int g = 1; g < 15; ++g) if (g < a.n_acoustic) acc = add_rn(acc, e[g]);
        const float summed = add_rn(sv, acc);
        a.out[(size_t)b * H + c] = add_rn(summed, tv);
    }
}
hipError_t launch_frame_embed(const FrameEmbedArgs& a, hipStream_t st) {
    if (a.n_acoustic != 15) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_frame_embed, dim3(a.B, (a.H + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// On-device sampler: penalties (lib.rs:1271-1322) → temperature → top-k → top-p → softmax →
// inverse-CDF multinomial (sampling.rs:140-319), one 1024-thread workgroup per sequence.
// Exactness plan: a full bitonic sort of (value desc, index asc) in LDS gives the oracle's sorted
// order; every floating-point SUM the reference does sequentially (top-p softmax/cumsum over the
// sorted row, final softmax/cdf in index order) is done sequentially by one lane over the kept
// entries only (the dropped ones contribute exact zeros), so ids are bit-exact up to expf ulps.
// ------------------------------------------------------------------------------------------------
constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_MAX = 4096;

__device__ __forceinline__ bool sort_before(float av, int ai, float bv, int bi) {
    return av > bv || (av == bv && ai < bi);
}

__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample(SampleArgs a) {
    __shared__ float s_val[SAMPLE_MAX];
    __shared__ float s_oval[SAMPLE_MAX];
    __shared__ uint16_t s_idx[SAMPLE_MAX];
    __shared__ uint16_t s_oidx[SAMPLE_MAX];
    __shared__ float red_v[SAMPLE_THREADS / 64];
    __shared__ int red_i[SAMPLE_THREADS / 64];
    __shared__ int s_keep, s_cut, s_pick;

    const int b = blockIdx.x, tid = threadIdx.x, V = a.vocab;
    if (a.rows) {            // this sequence's own options (SampleRow): a uniform load, the scalar fields of the copy are overwritten
        const SampleRow r = a.rows[b];
        a.inv_temp = r.inv_temp; a.apply_temp = r.apply_temp; a.greedy = r.greedy; a.top_k = r.top_k; a.top_p = r.top_p; a.use_top_p = r.use_top_p;
        a.rep_pen = r.rep_pen; a.rep_inv = r.rep_inv; a.use_rep = r.use_rep; a.eos_id = r.eos_id; a.min_new_tokens = r.min_new_tokens;
    }
    const float* lg = a.logits + (size_t)b * a.ld;
    uint8_t* seen = a.seen ? a.seen + (size_t)b * V : nullptr;
    const int tc = a.token_count ? a.token_count[b] : a.token_count_static;
    // (round 5) what only the last lines of the kernel need — this step's uniform draw (two dependent loads), the counters it
    // advances — is requested NOW by the one thread that uses it: at the end of the kernel each of these was a memory round
    // trip of its own on the frame's critical path (the read-modify-write of pos / frame_idx included)
    float u_pre = 0.0f; int fi_pre = 0, lim_pre = 0x7fffffff, pos_pre = 0;
    if (tid == 0) {
        if (a.u) u_pre = a.u[(size_t)b * a.u_stride + (a.draw_idx ? a.draw_idx[b] : 0)];      // (also in greedy mode: the row options that say so are still in flight)
        if (a.advance) { fi_pre = a.frame_idx[b]; pos_pre = a.pos[b]; if (a.limit) lim_pre = a.limit[b]; }
    }
    int n_sort = 2; while (n_sort < V) n_sort <<= 1;

    if (a.logits_hist && tc < a.hist_cap) {
        float* dst = a.logits_hist + (size_t)b * a.hist_stride_b + (size_t)tc * V;
        for (int i = tid; i < V; i += SAMPLE_THREADS) dst[i] = lg[i];
    }
    // penalties + temperature
    for (int i = tid; i < n_sort; i += SAMPLE_THREADS) {
        float v = -INFINITY;
        if (i < V) {
            v = lg[i];
            if (a.use_rep && seen && seen[i]) v = mul_rn(v, v > 0.0f ? a.rep_inv : a.rep_pen);
            if (a.use_suppress && i >= V - 1024 && i != a.codec_eos) v = -INFINITY;
            if (tc < a.min_new_tokens && i == a.eos_id) v = -INFINITY;
            if (a.apply_temp) v = add_rn(mul_rn(v, a.inv_temp), 0.0f);
        }
        s_val[i] = v; s_idx[i] = (uint16_t)i;
    }
    __syncthreads();

    int pick = 0;
    if (a.greedy) {
        pick = block_argmax_first(s_val, V, red_v, red_i);
    } else {
        // ---- top-k fast path: exact k-th largest value by an 8-bit radix select over order-preserving keys (4 histogram
        // passes in LDS), survivors = every v >= threshold (ties kept, sampling.rs:206-214), ordered by counting ranks
        // under the same (value desc, index asc) order the full sort produces. 78 block-wide bitonic stages (~45 us)
        // become ~14 barriers. Falls back to the full sort when top-k is off or the tie set is larger than SEL_MAX.
        constexpr int SEL_MAX = 1024;
        __shared__ unsigned s_hist[256];
        __shared__ unsigned s_prefix, s_mask, s_krem;
        __shared__ int s_nsel;
        __shared__ float s_thr;
        bool sorted = false;
        if (a.top_k > 0 && a.top_k < V) {
            auto key_of = [](float v) -> unsigned {
                const unsigned u = __float_as_uint(v + 0.0f);            // -0 -> +0: float compare treats them equal
                return (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // ascending float order = ascending key
            };
            if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_krem = (unsigned)a.top_k; }
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (tid < 256) s_hist[tid] = 0u;
                __syncthreads();
                const unsigned prefix = s_prefix, mask = s_mask;
                for (int i = tid; i < V; i += SAMPLE_THREADS) {
                    const unsigned k = key_of(s_val[i]);
                    if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (tid < 64) {
                    // lane l owns bins 4l .. 4l+3; suffix sums from the top bin down locate the bin holding the k-th largest
                    const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
                    const unsigned mine = h0 + h1 + h2 + h3;
                    unsigned above = 0;                                   // elements in bins of higher lanes
                    unsigned run = mine;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const unsigned o = __shfl_down(run, off);
                        if (tid + off < 64) run += o;
                    }
                    above = run - mine;                                   // suffix sum excluding this lane
                    const unsigned krem = s_krem;
                    if (above < krem && krem <= above + mine) {
                        unsigned acc = above; int bin = 4 * tid + 3;
                        const unsigned hs[4] = {h0, h1, h2, h3};
#pragma unroll
                        for (int q = 3; q >= 0; --q) {
                            if (acc + hs[q] >= krem) { bin = 4 * tid + q; break; }
                            acc += hs[q];
                        }
                        s_prefix = prefix | ((unsigned)bin << shift);
                        s_mask = mask | (255u << shift);
                        s_krem = krem - acc;
                    }
                }
                __syncthreads();
            }
            if (tid == 0) {
                const unsigned k = s_prefix;
                const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
                s_thr = __uint_as_float(u); s_nsel = 0;
            }
            __syncthreads();
            const float thr = s_thr;
            // survivors, unordered, into s_oval / s_oidx
            for (int i = tid; i < V; i += SAMPLE_THREADS) {
                const float v = s_val[i];
                if (v >= thr) {
                    const int slot = atomicAdd(&s_nsel, 1);
                    if (slot < SEL_MAX) { s_oval[slot] = v; s_oidx[slot] = (uint16_t)i; }
                }
            }
            __syncthreads();
            const int nsel = s_nsel;
            if (nsel <= SEL_MAX) {
                __syncthreads();
                // rank by counting → s_val / s_idx hold the survivors in sorted order (entries past nsel are never read)
                for (int i = tid; i < nsel; i += SAMPLE_THREADS) {
                    const float v = s_oval[i]; const int id = s_oidx[i];
                    int rank = 0;
                    for (int j = 0; j < nsel; ++j) rank += sort_before(s_oval[j], s_oidx[j], v, id) ? 1 : 0;
                    s_val[rank] = v; s_idx[rank] = (uint16_t)id;
                }
                sorted = true;
                __syncthreads();
                if (tid == 0) s_keep = nsel;
                __syncthreads();
            }
        }
        if (!sorted) {
        // bitonic sort, descending by (value, then ascending index)
        for (int k = 2; k <= n_sort; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (n_sort >> 1); t += SAMPLE_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int p = i + j;
                    const float av = s_val[i], bv = s_val[p];
                    const int ai = s_idx[i], bi = s_idx[p];
                    const bool desc = (i & k) == 0;
                    const bool swap = desc ? sort_before(bv, bi, av, ai) : sort_before(av, ai, bv, bi);
                    if (swap) { s_val[i] = bv; s_val[p] = av; s_idx[i] = (uint16_t)bi; s_idx[p] = (uint16_t)ai; }
                }
                __syncthreads();
            }
        }
        if (tid == 0) { s_keep = 0; }
        __syncthreads();
        if (a.top_k > 0) {
            const int k = a.top_k < V ? a.top_k : V;
            const float thr = s_val[k - 1];
            int cnt = 0;
            for (int i = tid; i < V; i += SAMPLE_THREADS) cnt += (s_val[i] >= thr) ? 1 : 0;
            if (cnt) atomicAdd(&s_keep, cnt);
        } else if (tid == 0) {
            s_keep = V;
        }
        __syncthreads();
        }   // !sorted
        const int n_keep = s_keep;
        // The reference's sequential sums (softmax denominator and running probability over the sorted row, sampling.rs:229-262),
        // in the same order. Up to 64 kept entries (top-k 50: always, ties aside) one WAVE does them: lane l computes its own
        // exp / quotient, and the running sum walks the lanes with v_readlane — an add per entry instead of an LDS round trip + expf
        // (+ a division) per entry behind one thread: 4 + 2.5 us of a 22 us launch. More entries: the one-thread loops.
        if (n_keep <= 64) {
            if (tid < 64) {
                int cut = n_keep;
                if (a.use_top_p) {
                    const float mx = s_val[0];
                    const float e = tid < n_keep ? expf(s_val[tid] - mx) : 0.0f;
                    float sum = 0.0f;
                    for (int i = 0; i < n_keep; ++i) sum += read_lane(e, i);
                    const float pr = e / sum;
                    float cum = 0.0f;
                    for (int i = 0; i < n_keep; ++i) {
                        cum += read_lane(pr, i);
                        if (cum > a.top_p) { cut = i + 1; break; }
                    }
                }
                if (tid == 0) s_cut = cut;
            }
        } else if (tid == 0) {
            int cut = n_keep;
            if (a.use_top_p) {
                const float mx = s_val[0];
                float sum = 0.0f;
                for (int i = 0; i < n_keep; ++i) { const float e = expf(s_val[i] - mx); s_oval[i] = e; sum += e; }
                float cum = 0.0f;
                for (int i = 0; i < n_keep; ++i) {
                    cum += s_oval[i] / sum;
                    if (cum > a.top_p) { cut = i + 1; break; }
                }
            }
            s_cut = cut;
        }
        __syncthreads();
        const int cut = s_cut;
        // order the kept entries by vocabulary index
        for (int i = tid; i < cut; i += SAMPLE_THREADS) {
            const int me = s_idx[i];
            int rank = 0;
            for (int j = 0; j < cut; ++j) rank += (s_idx[j] < me) ? 1 : 0;
            s_oval[rank] = s_val[i];     // safe: s_oval[] temporaries of the top-p pass are dead
            s_oidx[rank] = (uint16_t)me;
        }
        __syncthreads();
        if (cut <= 64) {      // final softmax + inverse CDF in vocabulary order (sampling.rs:264-319), the same way
            if (tid < 64) {
                const float mx = s_val[0];
                const float e = tid < cut ? expf(s_oval[tid] - mx) : 0.0f;
                float sum = 0.0f;
                for (int r = 0; r < cut; ++r) sum += read_lane(e, r);
                const float pr = e / sum;
                const float u = read_lane(u_pre, 0);
                float cdf = 0.0f; int hit = -1;
                if (u > 0.0f) {
                    for (int r = 0; r < cut; ++r) {
                        cdf += read_lane(pr, r);
                        if (cdf >= u) { hit = r; break; }
                    }
                }
                if (tid == 0) s_pick = hit >= 0 ? (int)s_oidx[hit] : 0;
            }
        } else if (tid == 0) {
            const float mx = s_val[0];
            float sum = 0.0f;
            for (int r = 0; r < cut; ++r) sum += expf(s_oval[r] - mx);
            const float u = u_pre;
            float cdf = 0.0f; int pk = 0;
            // `first i with cdf[i] >= u` over the whole vocabulary: for u > 0 that index always has
            // non-zero probability (so scanning the kept entries is exact); for u == 0 it is index 0.
            if (u > 0.0f) {
                for (int r = 0; r < cut; ++r) {
                    cdf += expf(s_oval[r] - mx) / sum;
                    if (cdf >= u) { pk = s_oidx[r]; break; }
                }
            }
            s_pick = pk;
        }
        __syncthreads();
        pick = s_pick;
    }
    if (tid == 0) {
        a.tok[b] = (uint32_t)pick;
        if (seen && pick < V) seen[pick] = 1;
        const bool frozen = a.advance && a.limit && fi_pre >= lim_pre;      // SampleArgs::limit
        if (a.token_count && !frozen) a.token_count[b] = tc + 1;
        if (a.advance && !frozen) { a.frame_idx[b] = fi_pre + 1; a.pos[b] = pos_pre + 1; }
    }
}

hipError_t launch_sample(const SampleArgs& a, hipStream_t st) {
    if (a.vocab > SAMPLE_MAX || a.vocab < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sample, dim3(a.B), dim3(SAMPLE_THREADS), 0, st, a);
    return hipGetLastError();
}

}  // namespace q3
