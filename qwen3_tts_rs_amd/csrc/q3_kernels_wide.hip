// q3_kernels_wide.hip — the projections of WIDE sessions (33 .. 64 sequences advancing in one session) as a real GEMM.
//
// Why a second family (round 3). With M <= 16 rows a projection is a weight stream: one 16-row weight tile per workgroup,
// every workgroup reading all of x (q3_kernels_gemv.hip). At M = 64 that shape is upside down — x is M*K*4 = 512 KB
// (K = 2048) and all 256 workgroups of k_gemv_wide pull it through the L2s (134 MB of L2 -> CU traffic against 16.8 MB of
// weights): 20 us for the talker's q|k|v, 840 GB/s, neither roof (profiles/r2_gemv_wide_batches.txt). Here a workgroup owns
// 128 weight rows x ONE K slice: it reads 128 x Ks weights and only the Ks columns of x, splits that slice of x into its
// three exact bf16 terms ONCE (all 512 threads, into LDS in MFMA B-operand order), and its eight waves — one 16-row weight
// tile each — run the bf16x3 products against the shared LDS image, no cross-wave reduction at all. The K slices of a row
// group are summed by a second, tiny launch (k_wide_epilogue) in fixed slice order — deterministic — which also applies
// 1/rms, bias, residual, SiLU / SwiGLU. Two launches per projection; both are captured in the frame graph. For q|k|v the
// second launch is skipped (launch_gemm_wide_partial): the decode-attention kernels add the slices of the 3 x 128 values
// they need themselves (AttnArgs::qkv_part); the o / down projections of a wide session run on k_gemv_sk2 row blocks.
//   y[m][n] = epi( sum_s sum_{k in slice s} (x[m][k] * norm_w[k]) * W[n][k]  /  sqrt(mean_k x[m][k]^2 + eps) )
// Numerics: the same exact bf16x3 arithmetic as the GEMV family; only the summation order differs.
#include "q3_kernels.h"

#include <math.h>
#include <stdlib.h>

namespace q3 {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
struct Split3 { u32x4_t hi, mid, lo; };
// exact 3-way bf16 split of 8 floats (element 2i in the low half of word i) — the split of q3_kernels_gemv.hip
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
__device__ __forceinline__ f32x4_t mfma_bf16(const u32x4_t& w, const u32x4_t& b, f32x4_t acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ float lane_xor32f(float v, bool upper) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(upper ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor16f(float v, bool odd_row) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(odd_row ? r[0] : r[1]);
}

struct WideArgs {
    const uint16_t* W; const uint16_t* W2;      // mode-1 tiled images ([N/16][Kpad/32][64 lanes][8 bf16]); W2: the SwiGLU "up" matrix
    const float* x; int ldx; const float* norm_w;
    int M, N, K, kst;                            // kst = k-steps of 32 per tile row of the image (Kpad / 32)
    int S, Ks, nmat, rgm;                        // K slices, k per slice (multiple of 128), matrices, 128-row groups per matrix
    unsigned w_bytes, x_bytes;                   // extents of one weight image / of x: the buffer descriptors' num_records
    float* part;                                 // [S][nmat][M][N] slice sums
    float* ssq;                                  // [S][M] sum of x^2 per row and slice (fused RMSNorm)
};

// One workgroup = NWV waves x one 16-row weight tile (128 or 64 weight rows) x one K slice, all M <= 16*MT rows of x.
// The K slice is walked in 128-column chunks through a RING OF FOUR register sets (round 4): the requests of chunk c + 3
// (x, norm weight, weight tiles) are issued before the MFMAs of chunk c. With two sets (round 3) a chunk was requested one
// MFMA phase (~0.4 us) before its staging needed it and every chunk paid the rest of a memory round trip: 8 chunks of the
// talker's gate/up slice took 32 us against ~4 us of matrix time. The workgroup is alone on its CU (512 threads), so the
// 256-VGPR budget of two waves per SIMD is there to be used. Every request is UNCONDITIONAL — `s_waitcnt vmcnt` counts in
// issue order and a load behind a branch makes hipcc wait for everything in flight — and goes through buffer descriptors:
// the requests past the slice's last chunk get an offset beyond num_records, which the hardware answers with zeros
// without touching memory.
template <bool RMS, int MT, int NWV>
__global__ __launch_bounds__(NWV * 64) void k_wide_gemm(WideArgs a) {
    __shared__ __attribute__((aligned(16))) u32x4_t xs[3][MT][4][64];       // [plane][column tile][k-step of the chunk][lane]: 12 KB x MT
    __shared__ float ssq_s[(MT * 4 + NWV - 1) / NWV][NWV][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m16 = lane & 15, kg = lane >> 4;
    const int rg = blockIdx.x, s = blockIdx.y;
    const int mat = rg / a.rgm, rgi = rg - mat * a.rgm;
    const int tile = rgi * NWV + wave;
    const bool tile_ok = tile * 16 < a.N;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(mat ? a.W2 : a.W), 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RMS ? a.norm_w : a.x), 0, RMS ? a.K * 4 : 0, 0x00020000);
    const unsigned w_lane = ((unsigned)(tile_ok ? tile : 0) * (unsigned)a.kst * 64u + (unsigned)lane) * 16u;      // byte offset of this lane in its tile row (image < 4 GB)
    constexpr unsigned OOB = 0x80000000u;            // beyond any num_records: the load returns zeros, no memory access
    const int k_begin = s * a.Ks, k_end = (k_begin + a.Ks) < a.K ? (k_begin + a.Ks) : a.K;
    constexpr int NIT = (MT * 4 + NWV - 1) / NWV;     // staging items (column tile, k-step) per wave and chunk
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float ss_acc[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) ss_acc[j] = 0.0f;

    struct Set { float4 xa[NIT], xb[NIT], na[NIT], nb[NIT]; u32x4_t wa[4]; };
    // requests of a chunk: this wave's share of x (and of the norm weight) first — they come back first and feed the staging —
    // then its four weight tiles
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    };
    auto request = [&](Set& r, int k0) {
        const bool live = k0 < k_end;                  // wave-uniform; a dead chunk's requests all go out of range
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = wave + NWV * j, t = it >> 2, ks = it & 3;
            const int m = 16 * t + m16, kk = k0 + (4 * ks + kg) * 8;
            const bool ok = live && it < MT * 4 && m < a.M;
            const unsigned xo = ok ? ((unsigned)m * (unsigned)a.ldx + (unsigned)kk) * 4u : OOB;
            r.xa[j] = ld4(xrs, xo);
            r.xb[j] = ld4(xrs, xo + 16u);
            if constexpr (RMS) { const unsigned no = live ? (unsigned)kk * 4u : OOB; r.na[j] = ld4(nrs, no); r.nb[j] = ld4(nrs, no + 16u); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r.wa[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(live ? w_lane + (unsigned)((k0 >> 5) + i) * 1024u : OOB), 0, 2));    // nt: streamed once
    };
    // staging: split once, park the B operands of the whole workgroup in LDS (lane-linear fragments: conflict-free b128)
    auto stage = [&](const Set& r) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = wave + NWV * j;
            if (it < MT * 4) {                       // wave-uniform
                float xv[8] = {r.xa[j].x, r.xa[j].y, r.xa[j].z, r.xa[j].w, r.xb[j].x, r.xb[j].y, r.xb[j].z, r.xb[j].w};
                if constexpr (RMS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss_acc[j] = fmaf(xv[e], xv[e], ss_acc[j]);
                    xv[0] *= r.na[j].x; xv[1] *= r.na[j].y; xv[2] *= r.na[j].z; xv[3] *= r.na[j].w;
                    xv[4] *= r.nb[j].x; xv[5] *= r.nb[j].y; xv[6] *= r.nb[j].z; xv[7] *= r.nb[j].w;
                }
                const Split3 sp = split3(xv);
                xs[0][it >> 2][it & 3][lane] = sp.hi; xs[1][it >> 2][it & 3][lane] = sp.mid; xs[2][it >> 2][it & 3][lane] = sp.lo;
            }
        }
    };
    auto multiply = [&](const Set& r) {
        if (tile_ok) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    acc[t] = mfma_bf16(r.wa[ks], xs[0][t][ks][lane], acc[t]);
                    acc[t] = mfma_bf16(r.wa[ks], xs[1][t][ks][lane], acc[t]);
                    acc[t] = mfma_bf16(r.wa[ks], xs[2][t][ks][lane], acc[t]);
                }
        }
    };
    // one chunk: stage it, request the chunk AHEAD chunks later into the set that came free a step ago, multiply
    auto step = [&](Set& cur, Set& nxt, int k0, int ahead) {
        stage(cur);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0); request(nxt, k0 + ahead * 128); __builtin_amdgcn_sched_barrier(0);      // pinned ahead of the MFMAs
        multiply(cur);
        __syncthreads();                                     // the next staging overwrites xs
    };
    if constexpr (NWV == 8) {                                // 512 threads, one workgroup per CU: four sets fit the 256-VGPR budget
        Set R0, R1, R2, R3;
        request(R0, k_begin); request(R1, k_begin + 128); request(R2, k_begin + 256);
        for (int k0 = k_begin; k0 < k_end; k0 += 512) {
            step(R0, R3, k0, 3);
            if (k0 + 128 >= k_end) break;
            step(R1, R0, k0 + 128, 3);
            if (k0 + 256 >= k_end) break;
            step(R2, R1, k0 + 256, 3);
            if (k0 + 384 >= k_end) break;
            step(R3, R2, k0 + 384, 3);
        }
    } else {                                                 // 256-thread geometry: twice the staging items per wave — four sets would leave one wave per
        Set A, B;                                            // SIMD (measured: 9.4 -> 11.9 us); two sets, three workgroups per CU cover for each other
        request(A, k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += 256) {
            step(A, B, k0, 1);
            if (k0 + 128 >= k_end) break;
            step(B, A, k0 + 128, 1);
        }
    }
    // slice sums: lane (column m16 of tile t, row group kg) holds rows kg*4 .. kg*4+3 — 16 bytes contiguous in n
    if (tile_ok) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = 16 * t + m16;
            if (m < a.M)
                *reinterpret_cast<f32x4_t*>(a.part + (((size_t)s * a.nmat + mat) * a.M + m) * a.N + tile * 16 + kg * 4) = acc[t];
        }
    }
    if constexpr (RMS) {
        if (rg == 0) {                               // one workgroup per slice reports sum(x^2); fixed order: k-groups, then k-steps
            const bool upper = lane >= 32, odd = (lane & 16) != 0;
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                float v = ss_acc[j];
                v += lane_xor16f(v, odd); v += lane_xor32f(v, upper);
                if (kg == 0) ssq_s[j][wave][m16] = v;
            }
            __syncthreads();
            if (tid < 16 * MT) {
                // the four k-step items of column tile t are items 4t .. 4t+3: item it was staged by wave it % NWV in pass it / NWV
                const int t = tid >> 4, mm = tid & 15;
                float tot = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { const int it = 4 * t + ks; tot += ssq_s[it / NWV][it % NWV][mm]; }
                if (16 * t + mm < a.M) a.ssq[(size_t)s * a.M + 16 * t + mm] = tot;
            }
        }
    }
}

struct WideEpiArgs {
    const float* part; const float* ssq; int S, nmat, M, N, K; float eps;
    const float* bias; const float* resid; int ldr; float* y; int ldy;
    float* zero; int zero_n;                      // side job (LinArgs::zero): clears the target of a later split-K projection
};
// slice sums -> y: fixed slice order, then 1/rms, +bias, +residual, SiLU, SwiGLU (the epilogues of the GEMV family)
template <int EPI, bool RMS>
__global__ __launch_bounds__(256) void k_wide_epilogue(WideEpiArgs e) {
    const int m = blockIdx.y, n = (blockIdx.x * 256 + threadIdx.x) * 4;
    zero_job(e.zero, e.zero_n, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, threadIdx.x, 256);
    if (n >= e.N) return;
    float4 v = {0.f, 0.f, 0.f, 0.f}, v2 = v;
    const size_t plane = (size_t)e.M * e.N;
    const float* p = e.part + (size_t)m * e.N + n;
    // slice sums in groups of eight requests in flight at once (a plain loop over a run-time S waits one memory round trip per
    // slice: 16 slices = 10 us of a 13 us projection in the first cut); the ADDS keep the slice order
    for (int s0 = 0; s0 < e.S; s0 += 8) {
        float4 a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = (s0 + i) < e.S ? (s0 + i) : (e.S - 1);
            a[i] = *reinterpret_cast<const float4*>(p + (size_t)s * e.nmat * plane);
            if constexpr (EPI == EPI_SWIGLU) b[i] = *reinterpret_cast<const float4*>(p + ((size_t)s * e.nmat + 1) * plane);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (s0 + i < e.S) {
                v.x += a[i].x; v.y += a[i].y; v.z += a[i].z; v.w += a[i].w;
                if constexpr (EPI == EPI_SWIGLU) { v2.x += b[i].x; v2.y += b[i].y; v2.z += b[i].z; v2.w += b[i].w; }
            }
        }
    }
    if constexpr (RMS) {
        float tot = 0.0f;
        for (int s0 = 0; s0 < e.S; s0 += 8) {
            float q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = e.ssq[(size_t)((s0 + i) < e.S ? (s0 + i) : (e.S - 1)) * e.M + m];
#pragma unroll
            for (int i = 0; i < 8; ++i) if (s0 + i < e.S) tot += q[i];
        }
        const float den = sqrtf(tot / (float)e.K + e.eps);
        v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
        if constexpr (EPI == EPI_SWIGLU) { v2.x = v2.x / den; v2.y = v2.y / den; v2.z = v2.z / den; v2.w = v2.w / den; }
    }
    if (e.bias) { const float4 b = *reinterpret_cast<const float4*>(e.bias + n); v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w; }
    if constexpr (EPI == EPI_RESID) {
        const float4 r = *reinterpret_cast<const float4*>(e.resid + (size_t)m * e.ldr + n);
        v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w;
    }
    if constexpr (EPI == EPI_SILU || EPI == EPI_SWIGLU) {
        v.x = v.x / (1.0f + expf(-v.x)); v.y = v.y / (1.0f + expf(-v.y)); v.z = v.z / (1.0f + expf(-v.z)); v.w = v.w / (1.0f + expf(-v.w));
    }
    if constexpr (EPI == EPI_SWIGLU) { v.x *= v2.x; v.y *= v2.y; v.z *= v2.z; v.w *= v2.w; }
    *reinterpret_cast<float4*>(e.y + (size_t)m * e.ldy + n) = v;
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the SwiGLU pair of a wide session WITHOUT split-K (k_wide2_swiglu) — one launch that writes y, fed by a small
// launch that splits x ONCE for the whole chip (k_wide2_split).
//
// Why. k_wide_gemm stages and splits x per workgroup, so a workgroup can only afford a slice of K; the slices are summed by a
// second launch, and for the code predictor's gate/up that second pass moves as many bytes as the weights themselves (6.3 MB
// of slice sums written and re-read for 12.6 MB of weights: 11.8 + 5.2 us per projection at 64 rows, 80 times per frame).
// Here x is split into its three exact bf16 terms ONCE, by 16-32 workgroups, straight into MFMA B-operand order in global
// memory (384 KB at K = 1024: L2-resident) — with the RMSNorm weight folded in and the rows' sum(x^2) reported per K slab —
// and the GEMM workgroup owns 16 output columns of BOTH matrices over ALL of K: its four waves take a quarter of K each,
// read their B operands as plain 16-byte loads (no VALU, no LDS, no barrier in the loop), keep four k-steps of operands in
// flight in a register ring, meet once in LDS (quarters added in order: deterministic) and apply 1/rms, SiLU and the product
// themselves. Same arithmetic as the GEMV family (exact bf16x3 products, f32 accumulation); only the summation order differs.
//   xp[k-step][column tile t][plane][lane] (16 bytes each): lane (kg, m16) = the 8 values x[16 t + m16][32 ks + 8 kg ..] * norm_w
struct Wide2Args {
    const uint16_t* W; const uint16_t* W2;      // mode-1 tiled images (gate, up)
    const float* x; int ldx; const float* norm_w;
    int M, N, K, kst;                            // kst = k-steps per tile row of the image (Kpad / 32)
    unsigned char* xp; float* ssq; int nslab;    // planes; sum(x^2) per (K slab of 256, row): ssq[slab][M]
    float eps; float* y; int ldy;
    float* zero; int zero_n;                     // side job (LinArgs::zero)
    unsigned w_bytes;
};

// grid (column tiles, K slabs of 256): 4 waves x 2 k-steps each
template <bool RMS>
__global__ __launch_bounds__(256) void k_wide2_split(Wide2Args a) {
    __shared__ float ssq_s[4][2][4][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m16 = lane & 15, kg = lane >> 4;
    const int t = blockIdx.x, slab = blockIdx.y, MT = gridDim.x;
    const int m = 16 * t + m16;
    const bool ok = m < a.M;
    float ss[2] = {0.f, 0.f};
    float4 xa[2], xb[2], na[2], nb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ks = slab * 8 + wave * 2 + j, kk = ks * 32 + kg * 8;
        const bool kok = kk < a.K;
        const float* px = a.x + (size_t)(ok ? m : 0) * a.ldx + (kok ? kk : 0);
        xa[j] = *reinterpret_cast<const float4*>(px); xb[j] = *reinterpret_cast<const float4*>(px + 4);
        if constexpr (RMS) { na[j] = *reinterpret_cast<const float4*>(a.norm_w + (kok ? kk : 0)); nb[j] = *reinterpret_cast<const float4*>(a.norm_w + (kok ? kk : 0) + 4); }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ks = slab * 8 + wave * 2 + j, kk = ks * 32 + kg * 8;
        const bool live = ok && kk < a.K;
        float xv[8] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w, xb[j].x, xb[j].y, xb[j].z, xb[j].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = live ? xv[e] : 0.0f;
        if constexpr (RMS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ss[j] = fmaf(xv[e], xv[e], ss[j]);
            xv[0] *= na[j].x; xv[1] *= na[j].y; xv[2] *= na[j].z; xv[3] *= na[j].w;
            xv[4] *= nb[j].x; xv[5] *= nb[j].y; xv[6] *= nb[j].z; xv[7] *= nb[j].w;
        }
        const Split3 sp = split3(xv);
        if (kk < a.K || kg * 8 + (ks * 32) < ((a.K + 31) & ~31)) {      // every k-step of the (32-padded) image is written, zeros past K
            u32x4_t* dst = reinterpret_cast<u32x4_t*>(a.xp) + ((size_t)(ks * MT + t) * 3) * 64 + lane;
            dst[0] = sp.hi; dst[64] = sp.mid; dst[128] = sp.lo;
        }
        if constexpr (RMS) ssq_s[wave][j][kg][m16] = ss[j];
    }
    if constexpr (RMS) {
        __syncthreads();
        if (tid < 16) {                               // fixed order: waves, k-steps, k-groups
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) tot += ssq_s[w][j][g][tid];
            if (16 * t + tid < a.M) a.ssq[(size_t)slab * a.M + 16 * t + tid] = tot;
        }
    }
}

// grid (N / 16): workgroup = 16 output columns of gate AND up over all of K; 4 waves = K quarters.
// (The same loop over two tiles of ONE matrix, for q|k|v, was built and measured: 6.94 vs 6.75 ms per frame at 64 rows — the
// K-slice sums handed straight to the attention kernel stay the better q|k|v.)
template <int MT, int NT>            // NT = 16-column tiles of each matrix per workgroup (2 for the talker's 6144-row pair: one round of 192 workgroups)
__global__ __launch_bounds__(256) void k_wide2_swiglu(Wide2Args a) {
    __shared__ __attribute__((aligned(16))) f32x4_t red[4][2 * NT * MT][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m16 = lane & 15, kg = lane >> 4;
    const int tile0 = blockIdx.x * NT;
    const int KS = a.K >> 5;
    const int q0 = (wave * KS) / 4, q1 = ((wave + 1) * KS) / 4;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W2), 0, (int)a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, KS * MT * 3 * 1024, 0x00020000);
    const unsigned w_lane = ((unsigned)tile0 * (unsigned)a.kst * 64u + (unsigned)lane) * 16u;
    const unsigned w_tile = (unsigned)a.kst * 1024u;                 // bytes from a tile row of the image to the next
    constexpr unsigned OOB = 0x80000000u;
    struct Set { u32x4_t g[NT], u[NT], b[MT][3]; };
    auto request = [&](Set& r, int ks) {             // every request unconditional; a k-step past the quarter goes out of range
        const bool live = ks < q1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const unsigned off = live ? w_lane + (unsigned)n * w_tile + (unsigned)ks * 1024u : OOB;
            r.g[n] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(grs, (int)off, 0, 2));
            r.u[n] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(urs, (int)off, 0, 2));
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                r.b[t][pl] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(brs, (int)(live ? (unsigned)(((ks * MT + t) * 3 + pl) * 64 + lane) * 16u : OOB), 0, 0));
    };
    f32x4_t ag[NT][MT], au[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < MT; ++t) { ag[n][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; au[n][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    auto multiply = [&](const Set& r) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int n = 0; n < NT; ++n) { ag[n][t] = mfma_bf16(r.g[n], r.b[t][pl], ag[n][t]); au[n][t] = mfma_bf16(r.u[n], r.b[t][pl], au[n][t]); }
    };
    auto step = [&](Set& cur, Set& nxt, int ks, int ahead) {
        __builtin_amdgcn_sched_barrier(0); request(nxt, ks + ahead); __builtin_amdgcn_sched_barrier(0);
        multiply(cur);
    };
    if constexpr (NT == 1) {                         // four k-steps of operands in flight
        Set R0, R1, R2, R3;
        request(R0, q0); request(R1, q0 + 1); request(R2, q0 + 2);
        for (int ks = q0; ks < q1; ks += 4) {
            step(R0, R3, ks, 3);
            if (ks + 1 >= q1) break;
            step(R1, R0, ks + 1, 3);
            if (ks + 2 >= q1) break;
            step(R2, R1, ks + 2, 3);
            if (ks + 3 >= q1) break;
            step(R3, R2, ks + 3, 3);
        }
    } else {                                         // two tiles per matrix: 64 VGPRs per k-step and 64 of accumulators — three sets fit without spilling
        Set R0, R1, R2;
        request(R0, q0); request(R1, q0 + 1);
        for (int ks = q0; ks < q1; ks += 3) {
            step(R0, R2, ks, 2);
            if (ks + 1 >= q1) break;
            step(R1, R0, ks + 1, 2);
            if (ks + 2 >= q1) break;
            step(R2, R1, ks + 2, 2);
        }
    }
    zero_job(a.zero, a.zero_n, blockIdx.x, gridDim.x, tid, 256);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < MT; ++t) { red[wave][n * MT + t][lane] = ag[n][t]; red[wave][(NT + n) * MT + t][lane] = au[n][t]; }
    __syncthreads();
    // the (column tile, weight tile) items are dealt to the waves: quarters added in order, 1/rms, SiLU(gate) * up, 16 bytes per lane
    for (int it = wave; it < NT * MT; it += 4) {
        const int n = it / MT, t = it - n * MT;
        f32x4_t g = red[0][it][lane], u = red[0][NT * MT + it][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4_t g2 = red[w][it][lane], u2 = red[w][NT * MT + it][lane];
            g[0] += g2[0]; g[1] += g2[1]; g[2] += g2[2]; g[3] += g2[3];
            u[0] += u2[0]; u[1] += u2[1]; u[2] += u2[2]; u[3] += u2[3];
        }
        const int m = 16 * t + m16;
        if (m < a.M) {
            float tot = 0.0f;
            for (int sl = 0; sl < a.nslab; ++sl) tot += a.ssq[(size_t)sl * a.M + m];
            const float den = sqrtf(tot / (float)a.K + a.eps);
            float4 o;
            { const float gv = g[0] / den, uv = u[0] / den; o.x = (gv / (1.0f + expf(-gv))) * uv; }
            { const float gv = g[1] / den, uv = u[1] / den; o.y = (gv / (1.0f + expf(-gv))) * uv; }
            { const float gv = g[2] / den, uv = u[2] / den; o.z = (gv / (1.0f + expf(-gv))) * uv; }
            { const float gv = g[3] / den, uv = u[3] / den; o.w = (gv / (1.0f + expf(-gv))) * uv; }
            *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + (tile0 + n) * 16 + kg * 4) = o;
        }
    }
}

template <bool RMS, int NWV>
hipError_t launch_gemm_t(const WideArgs& w, hipStream_t st) {
    const dim3 grid(w.rgm * w.nmat, w.S), blk(NWV * 64);
    const int mt = (w.M + 15) / 16;
    if (mt <= 2) hipLaunchKernelGGL((k_wide_gemm<RMS, 2, NWV>), grid, blk, 0, st, w);
    else if (mt == 3) hipLaunchKernelGGL((k_wide_gemm<RMS, 3, NWV>), grid, blk, 0, st, w);
    else hipLaunchKernelGGL((k_wide_gemm<RMS, 4, NWV>), grid, blk, 0, st, w);
    return hipGetLastError();
}
template <int EPI>
hipError_t launch_epi_t(const WideEpiArgs& e, bool rms, hipStream_t st) {
    const dim3 grid((e.N / 4 + 255) / 256, e.M), blk(256);
    if (rms) hipLaunchKernelGGL((k_wide_epilogue<EPI, true>), grid, blk, 0, st, e);
    else hipLaunchKernelGGL((k_wide_epilogue<EPI, false>), grid, blk, 0, st, e);
    return hipGetLastError();
}
}  // namespace

// Geometry: 128 weight rows per workgroup, or 64 when that is what it takes to fill the chip with slices of at least two
// 128-column chunks (the chunk pipeline needs two to overlap anything; fewer, longer slices also mean fewer slice sums).
static void wide_plan(int N, int nmat, int K, int& rows, int& S, int& Ks) {
    const int chunks = K / 128;
    auto plan = [&](int r, int min_chunks, int& s_out, int& ks_out) {
        const int groups = (N / r) * nmat;
        static const int cap = [] { const char* e = getenv("Q3_WIDE_WG_CAP"); const int v = e ? atoi(e) : 256; return v < 64 ? 64 : v; }();      // A/B aid
        int want = (r == 64 ? cap : 256) / groups;     // never more workgroups than CUs: the RMS / four-column-tile variants hold one
                                                       // workgroup per CU (134 VGPRs), so 288 workgroups are two rounds (gate/up: 28.8 -> 38.5 us)
        int max_s = chunks / min_chunks; if (max_s < 1) max_s = 1;
        if (want > max_s) want = max_s;
        if (want < 1) want = 1;
        const int per = (chunks + want - 1) / want;
        ks_out = per * 128; s_out = (chunks + per - 1) / per;
        return groups * s_out;
    };
    static const int minc = [] { const char* e = getenv("Q3_WIDE_MIN_CHUNKS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();      // A/B aid
    int s128, k128, s64, k64;
    const int wg128 = plan(128, minc, s128, k128);
    if (wg128 >= 192 || N % 64 != 0) { rows = 128; S = s128; Ks = k128; return; }
    const int wg64 = plan(64, minc, s64, k64);
    if (wg64 > wg128) { rows = 64; S = s64; Ks = k64; } else { rows = 128; S = s128; Ks = k128; }
}

size_t gemm_wide_ws_bytes(int M, int N, int K, int epi) {
    const int nmat = epi == EPI_SWIGLU ? 2 : 1;
    int rows, S, Ks; wide_plan(N, nmat, K, rows, S, Ks);
    return ((size_t)S * nmat * M * N + (size_t)S * M) * sizeof(float);
}

int gemm_wide_min_rows() {
    static const int v = [] { const char* e = getenv("Q3_WIDE_GEMM_MIN"); const int x = e ? atoi(e) : 17; return x < 17 ? 17 : (x > 64 ? 64 : x); }();
    return v;
}
static hipError_t gemm_wide_impl(const LinArgs& a, hipStream_t st, WidePartial* partial);
hipError_t launch_gemm_wide_partial(const LinArgs& a, hipStream_t st, WidePartial* out) {
    if (!out || a.epi != EPI_NONE || !a.norm_w || a.bias || a.zero) return hipErrorNotSupported;
    return gemm_wide_impl(a, st, out);
}
// hipErrorNotSupported: the shape is outside this family (the caller falls back to k_gemv_wide)
hipError_t launch_gemm_wide(const LinArgs& a, hipStream_t st) { return gemm_wide_impl(a, st, nullptr); }
static hipError_t gemm_wide_impl(const LinArgs& a, hipStream_t st, WidePartial* partial) {
    const bool rms = a.norm_w != nullptr;
    // (until the q|k|v slice sums went to the attention kernels and o / down to the row-block split-K kernel, M <= 32 stayed on
    // k_gemv_wide: two column tiles did not pay for the second launch — B = 32: 6.17 vs 6.79 ms per frame. With those launches
    // gone the GEMM wins from 17 rows: B = 20 4.850 -> 4.644, B = 24 4.881 -> 4.727, B = 32 5.045 -> 4.981. Q3_WIDE_GEMM_MIN=33
    // restores the old split)
    if (a.tiled != 1 || a.M < gemm_wide_min_rows() || a.M > 64 || a.N % 128 != 0 || a.K % 128 != 0 || a.Kpad != a.K || a.ldx % 4 != 0 || a.ldy % 4 != 0 || !a.ws ||
        (a.epi == EPI_RESID && a.ldr % 4 != 0) || ((a.epi == EPI_RESID || a.epi == EPI_SILU) && rms) || a.ksplit != 1)
        return hipErrorNotSupported;
    // SwiGLU pair with a fused input norm: the single-launch form (k_wide2_split + k_wide2_swiglu); Q3_WIDE2=0: the split-K pair (A/B aid)
    static const bool wide2 = [] { const char* e = getenv("Q3_WIDE2"); return !(e && atoi(e) == 0); }();
    static const int wide2_nt2 = [] { const char* e = getenv("Q3_WIDE2_NT2"); return e ? atoi(e) : 4096; }();      // rows from which a workgroup takes two tiles per matrix (A/B aid)
    if (wide2 && !partial && a.epi == EPI_SWIGLU && rms && !a.bias && a.N % 32 == 0 && a.K % 256 == 0) {
        Wide2Args v{};
        v.W = a.W; v.W2 = a.W2; v.x = a.x; v.ldx = a.ldx; v.norm_w = a.norm_w; v.M = a.M; v.N = a.N; v.K = a.K; v.kst = a.Kpad >> 5;
        v.eps = a.eps; v.y = a.y; v.ldy = a.ldy; v.zero = a.zero; v.zero_n = a.zero_n;
        const int MT = (a.M + 15) / 16;
        const size_t plane_bytes = (size_t)(a.K >> 5) * MT * 3 * 1024;
        v.nslab = a.K / 256;
        const size_t wb = (size_t)a.N * a.Kpad * 2;
        if (plane_bytes + (size_t)v.nslab * a.M * sizeof(float) <= a.ws_bytes && wb < 0x7fffffffu && plane_bytes < 0x7fffffffu) {
            v.w_bytes = (unsigned)wb;
            v.xp = reinterpret_cast<unsigned char*>(a.ws); v.ssq = reinterpret_cast<float*>(v.xp + plane_bytes);
            hipLaunchKernelGGL((k_wide2_split<true>), dim3(MT, v.nslab), dim3(256), 0, st, v);
            const bool nt2 = a.N > wide2_nt2;            // one round of workgroups for the talker's 6144 rows
            const dim3 grid(a.N / (nt2 ? 32 : 16)), blk(256);
            if (nt2) {
                if (MT <= 2) hipLaunchKernelGGL((k_wide2_swiglu<2, 2>), grid, blk, 0, st, v);
                else if (MT == 3) hipLaunchKernelGGL((k_wide2_swiglu<3, 2>), grid, blk, 0, st, v);
                else hipLaunchKernelGGL((k_wide2_swiglu<4, 2>), grid, blk, 0, st, v);
            } else {
                if (MT <= 2) hipLaunchKernelGGL((k_wide2_swiglu<2, 1>), grid, blk, 0, st, v);
                else if (MT == 3) hipLaunchKernelGGL((k_wide2_swiglu<3, 1>), grid, blk, 0, st, v);
                else hipLaunchKernelGGL((k_wide2_swiglu<4, 1>), grid, blk, 0, st, v);
            }
            return hipGetLastError();
        }
    }
    WideArgs w{};
    w.W = a.W; w.W2 = a.W2; w.x = a.x; w.ldx = a.ldx; w.norm_w = a.norm_w; w.M = a.M; w.N = a.N; w.K = a.K; w.kst = a.Kpad >> 5;
    w.nmat = a.epi == EPI_SWIGLU ? 2 : 1;
    int rows; wide_plan(a.N, w.nmat, a.K, rows, w.S, w.Ks);
    w.rgm = a.N / rows;
    {
        const size_t wb = (size_t)a.N * a.Kpad * 2, xb = ((size_t)(a.M - 1) * a.ldx + a.K) * 4;
        if (wb >= 0x7fffffffu || xb >= 0x7fffffffu) return hipErrorNotSupported;
        w.w_bytes = (unsigned)wb; w.x_bytes = (unsigned)xb;
    }
    const size_t part_floats = (size_t)w.S * w.nmat * a.M * a.N;
    if ((part_floats + (size_t)w.S * a.M) * sizeof(float) > a.ws_bytes) return hipErrorNotSupported;
    w.part = a.ws; w.ssq = a.ws + part_floats;
    hipError_t e = rows == 128 ? (rms ? launch_gemm_t<true, 8>(w, st) : launch_gemm_t<false, 8>(w, st))
                               : (rms ? launch_gemm_t<true, 4>(w, st) : launch_gemm_t<false, 4>(w, st));
    if (e != hipSuccess) return e;
    if (partial) { partial->part = w.part; partial->ssq = w.ssq; partial->S = w.S; return hipSuccess; }     // the consumer adds the slices
    WideEpiArgs p{};
    p.part = w.part; p.ssq = w.ssq; p.S = w.S; p.nmat = w.nmat; p.M = a.M; p.N = a.N; p.K = a.K; p.eps = a.eps;
    p.bias = a.bias; p.resid = a.resid; p.ldr = a.ldr; p.y = a.y; p.ldy = a.ldy; p.zero = a.zero; p.zero_n = a.zero_n;
    switch (a.epi) {
        case EPI_NONE: return launch_epi_t<EPI_NONE>(p, rms, st);
        case EPI_RESID: return launch_epi_t<EPI_RESID>(p, rms, st);
        case EPI_SILU: return launch_epi_t<EPI_SILU>(p, rms, st);
        case EPI_SWIGLU: return launch_epi_t<EPI_SWIGLU>(p, rms, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace q3
