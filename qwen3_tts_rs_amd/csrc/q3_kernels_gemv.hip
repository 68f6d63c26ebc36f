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

// Kernel-argument preload (hipcc -mllvm -amdgpu-kernarg-preload-count=14, build.sh): the first 14 dwords of a kernel's
// explicit arguments arrive in SGPRs with the wave instead of through an s_load — whose round trip the frame's timeline
// prices at 0.31 us per node, in front of every address a kernel computes (tools/trace_frame.py, `karg` column). A struct
// argument is not preloadable, so the GEMV kernels take the fields their first loads depend on as leading scalars
// (exactly 14 dwords) and the rest as the struct; Q3_LIN_APPLY overwrites the struct copy's fields with the scalars.
#define Q3_LIN_PRE const uint16_t* pW, const uint16_t* pW2, const float* px, const float* pnw, int pM, int pN, int pK, int pKpad, int pldx, int pepi
#define Q3_LIN_APPLY(a, a_in) LinArgs a = a_in; a.W = pW; a.W2 = pW2; a.x = px; a.norm_w = pnw; a.M = pM; a.N = pN; a.K = pK; a.Kpad = pKpad; a.ldx = pldx; a.epi = pepi
#define Q3_LIN_PASS(a) (a).W, (a).W2, (a).x, (a).norm_w, (a).M, (a).N, (a).K, (a).Kpad, (a).ldx, (a).epi, (a)

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

// weight-tile load policy: nontemporal (streamed once per launch) unless -DQ3_NO_NT (experiment: cacheable loads)
#ifdef Q3_NO_NT
#define Q3_WLOAD(p) (*(p))
#else
#define Q3_WLOAD(p) __builtin_nontemporal_load(p)
#endif
struct Split3 { u32x4_t hi, mid, lo; };

// value held by the lane 8 places away inside this lane's row of 16 (DPP row_ror:8 — a VALU move, no LDS, no memory)
__device__ __forceinline__ float4 ror8(const float4& v) {
    float4 r;
    r.x = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v.x), 0x128, 0xf, 0xf, false));
    r.y = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v.y), 0x128, 0xf, 0xf, false));
    r.z = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v.z), 0x128, 0xf, 0xf, false));
    r.w = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v.w), 0x128, 0xf, 0xf, false));
    return r;
}

// Development aid (never defined in product builds): -DQ3_ABLATE=n removes one ingredient of the GEMV kernels to
// price it — 1: bf16x3 split → hi only, 2: + no x loads, 3: no cross-wave reduction, 4: no weight loads. Results
// are wrong by construction.
#ifndef Q3_ABLATE
#define Q3_ABLATE 0
#endif

// exact 3-way bf16 split of 8 floats (packed pairs: element 2i in the low half of word i)
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 s;
#if Q3_ABLATE == 1 || Q3_ABLATE == 2
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.hi[i] = cvt_pk_bf16(x[2 * i], x[2 * i + 1]); s.mid[i] = 0; s.lo[i] = 0; }
    return s;
#endif
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

// B operand of one k-step for this lane: mask, Σx², norm weight, exact bf16x3 split. Needs only x / norm weight
// (L2-resident), so it runs while the weight tiles of the group are still in flight from HBM.
template <bool RMS>
__device__ __forceinline__ Split3 gemv_prep(bool valid, const float4& x0, const float4& x1, const float4& n0, const float4& n1, float& ss) {
#if Q3_ABLATE == 2
    float xv[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    { const Split3 r = split3(xv); ss += valid ? 1.f : 0.f; return r; }
#else
    float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#endif
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = valid ? xv[e] : 0.0f;
    if constexpr (RMS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(xv[e], xv[e], ss);
        xv[0] *= n0.x; xv[1] *= n0.y; xv[2] *= n0.z; xv[3] *= n0.w; xv[4] *= n1.x; xv[5] *= n1.y; xv[6] *= n1.z; xv[7] *= n1.w;
    }
    return split3(xv);
}

// HALF form of the above (M <= 8): lane (m, kg) holds only ITS four floats of the k-group — elements 0..3 in lanes m < 8,
// elements 4..7 in lanes m + 8 — and splits just those: half the VALU work of the whole-slot form, which every one of the
// 256 workgroups repeats on the same x on the critical path of the launch (the frame's timeline prices the split at
// 0.4-0.7 us per node). The finished bf16 pairs, not the floats, then cross between lanes m and m + 8 (DPP row_ror:8):
// six moves per k-step instead of eight, and a lane's B operand is [own pair 0, own pair 1, partner pair 0, partner pair 1]
// — element order 0..7 for the columns in use (m < 8); lanes m >= 8 end up with their halves swapped, columns nobody reads.
// Σx² is accumulated over the lane's own four elements; the caller adds the partner's sum once at the end (gemv_ss_half).
__device__ __forceinline__ uint32_t ror8u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, false); }
template <bool RMS>
__device__ __forceinline__ Split3 gemv_prep_half(bool valid, const float4& x, const float4& n, float& ss) {
#if Q3_ABLATE == 2
    float xv[4] = {1.f, 2.f, 3.f, 4.f};
    ss += valid ? 1.f : 0.f;
#else
    float xv[4] = {x.x, x.y, x.z, x.w};
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = valid ? xv[e] : 0.0f;
    if constexpr (RMS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ss = fmaf(xv[e], xv[e], ss);
        xv[0] *= n.x; xv[1] *= n.y; xv[2] *= n.z; xv[3] *= n.w;
    }
    Split3 s;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = xv[2 * i], b = xv[2 * i + 1];
        const uint32_t h = cvt_pk_bf16(a, b);
#if Q3_ABLATE == 1 || Q3_ABLATE == 2
        s.hi[i] = h; s.mid[i] = 0; s.lo[i] = 0;
#else
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const uint32_t m = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = cvt_pk_bf16(sa, sb);
#endif
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { s.hi[2 + i] = ror8u(s.hi[i]); s.mid[2 + i] = ror8u(s.mid[i]); s.lo[2 + i] = ror8u(s.lo[i]); }
    return s;
}
// Σx² of a column = the sums of its two half-slot lanes (m and m + 8)
__device__ __forceinline__ float gemv_ss_half(float ss) { return ss + __builtin_bit_cast(float, ror8u(__builtin_bit_cast(uint32_t, ss))); }

// four floats from the lane that loaded them (row-contiguous order) to the lane whose B-operand slot they fill: LDS crossbar, no LDS memory
__device__ __forceinline__ float4 x_to_b_order(const float4& t, int bsrc) {
    float4 r;
    r.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsrc, __builtin_bit_cast(int, t.x)));
    r.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsrc, __builtin_bit_cast(int, t.y)));
    r.z = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsrc, __builtin_bit_cast(int, t.z)));
    r.w = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsrc, __builtin_bit_cast(int, t.w)));
    return r;
}

// NWAVES waves split K; weights for up to G k-steps are requested up front (G KiB per wave in flight per
// matrix) before any of them is consumed, so a wave's whole slice is usually one HBM round trip.
//
// HALF (M <= 8): a 64-lane 16-byte load costs the CU's vector-memory pipe the same whether all of its lanes carry data or
// only the eight batch columns in use, and the B operand needs 32 bytes per (column, k-group). So the two halves of a
// column's 32 bytes are fetched by lanes m and m + 8 of the same 16-lane row in ONE instruction (lane m + 8 would
// otherwise idle) and exchanged with a DPP row rotate: one x (and one norm-weight) instruction per k-step instead of two.
// Lanes m >= 8 end up holding column m - 8 with its halves swapped — columns nobody reads.
template <int EPI, bool RMS, int NWAVES, bool HALF, bool CO = true>      // CO: see COAL below (off for M <= 2)
__global__ __launch_bounds__(NWAVES * 64) void k_gemv_mfma(Q3_LIN_PRE, LinArgs a_in) {
    Q3_LIN_APPLY(a, a_in);
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int G = (NW == 2 || RMS) ? 4 : 6;
    __shared__ __attribute__((aligned(16))) float red[NWAVES][NW][256];
    __shared__ float ssq[NWAVES][4][16];
    Q3T_DECL Q3T(0); Q3T_K(7, a.Kpad);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (Rotating the wave -> K-slice map with the workgroup index, so that the 256 CUs do not all walk the shared activation
    // vector in the same order, was tried against an L2-channel hot-spot theory: no change on any shape or on the frame —
    // profiles/r2_gemv_variants_M8.txt.)
    const int m = lane & 15, kg = lane >> 4;
    const int S = a.Kpad >> 5;                       // k-steps of 32
    const int s0 = (wave * S) / NWAVES, s1 = ((wave + 1) * S) / NWAVES;
    const size_t tile_base = (size_t)blockIdx.x * S * 64 + lane;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + tile_base;
    const u32x4_t* __restrict__ wp2 = NW == 2 ? reinterpret_cast<const u32x4_t*>(a.W2) + tile_base : wp;
    const int xrow = HALF ? (m & 7) : m, xhalf = HALF ? (m >> 3) * 4 : 0;
    const bool act = HALF ? xrow < a.M : m < a.M;
    // x (and the residual, y, the zero job) travel write-through / L1-bypassing: q3_kernels.h "activation transport"
    const __amdgpu_buffer_rsrc_t xres = act_rsrc(a.x);
    const int xr = ((act ? xrow : 0) * a.ldx + kg * 8 + xhalf) * 4;          // byte offsets into x
    const float* __restrict__ nwp = RMS ? a.norm_w + kg * 8 + xhalf : nullptr;
    // COAL: x requested in row-contiguous lane order and moved to B-operand order through the LDS crossbar (see k_gemv_sk2)
    // (M <= 2 keeps the B-operand order: one or two live rows are one or two sectors per quad either way, and the four crossbar
    // moves per k-step then only cost — B = 1 frame 2.569 -> 2.588 ms with them)
    constexpr bool COAL = CO && NWAVES <= 8;         // (the 8-wave HOIST schedule; !HALF: the two-instruction form of k_gemv_sk2)
    const int crow = HALF ? lane >> 3 : lane >> 2, cchunk = HALF ? lane & 7 : 2 * (lane & 3);
    const bool cact = crow < a.M;
    const int xc = ((cact ? crow : 0) * a.ldx + cchunk * 4) * 4;
    const int bsrc = HALF ? ((xrow * 8) + 2 * kg + (m >> 3)) * 4 : (m * 4 + kg) * 4;

    // epilogue operands (bias, residual) are requested up front so their round trip hides under the weight stream
    float pre_b = 0.0f, pre_r = 0.0f;
    {
        const int col = tid >> 4, n = blockIdx.x * 16 + (tid & 15);
        if (tid < 256 && col < a.M && n < a.N) {
            if (a.bias) pre_b = a.bias[n];
            if constexpr (EPI == EPI_RESID) pre_r = act_ld1(act_rsrc(a.resid), (col * a.ldr + n) * 4);
        }
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.0f;
    // Two schedules. HOIST (8-wave workgroups, 256 VGPRs per lane available): VMEM returns are counted in issue
    // order, so x / norm weight (L2 hits) are requested FIRST and the HBM weight tiles after them; the bf16x3 split
    // (the VALU-heavy part) of the whole group then runs under the weight latency and only the MFMAs wait for HBM.
    // 16-wave workgroups have 128 VGPRs per lane — not enough to hold a group's splits — and keep the interleaved
    // per-step order.
    // (With the half-row form the hoisted order also fits the 16-wave register budget; measured on the down-proj: 10.96 ->
    // 11.80 us hoisted, 11.27 us with only the x requests moved ahead of the weight requests — the interleaved order stays.)
    constexpr bool HOIST = NWAVES <= 8;
    if constexpr (HOIST) {
        // (Keeping a second group's loads in flight while the first is split and multiplied — a software pipeline over the
        // groups — was built and measured: qkv 6.22 -> 7.29 us, gate/up 13.24 -> 13.81 us at M = 8, the frame 1.4 % slower:
        // twice the live registers cost a resident workgroup per CU; profiles/r2_gemv_variants_M8.txt.)
        struct Grp { u32x4_t wa[G], wb[G]; float4 xa[G], xb[G], na[G], nb[G]; };
        auto load_group = [&](Grp& g, int sb) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);      // clamp: duplicate load, masked in consume
                const int ko = (s * 32 + kg * 8) < a.K ? s * 32 : 0;    // K tail (K % 32 != 0): clamp, masked in consume
                // lanes of unused batch columns issue no request — measured per variant: qkv 7.8 -> 7.4 us, code-predictor
                // qkv 5.2 -> 4.6 us at M = 8, but the two-instruction SwiGLU pair loses (13.7 -> 15.7 us) and stays unmasked
                const bool ld = (NW == 2 && !HALF) || act;
                if constexpr (COAL) {
                    const int kc = (s * 32 + cchunk * 4) < a.K ? s * 32 : 0;
                    g.xa[i] = cact ? act_ld4(xres, xc + kc * 4) : float4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (!HALF) g.xb[i] = cact ? act_ld4(xres, xc + kc * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
                } else {
                    g.xa[i] = ld ? act_ld4(xres, xr + ko * 4) : float4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (!HALF) g.xb[i] = ld ? act_ld4(xres, xr + ko * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (RMS) {
                    g.na[i] = *reinterpret_cast<const float4*>(nwp + ko);
                    if constexpr (!HALF) g.nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
#if Q3_ABLATE == 4
                g.wa[i] = u32x4_t{(unsigned)s, 1u, 2u, 3u}; g.wb[i] = g.wa[i];
#else
                g.wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
                if constexpr (NW == 2) g.wb[i] = Q3_WLOAD(wp2 + (size_t)s * 64);
#endif
            }
        };
        auto consume = [&](Grp& g, int sb) {
            Split3 sp[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = sb + i;      // s >= s1 (ragged last group): zero operand, its MFMAs add nothing — no branch,
                const bool valid = act && s < s1 && (s * 32 + kg * 8) < a.K;   // so the group stays one scheduling region
                if constexpr (COAL) { g.xa[i] = x_to_b_order(g.xa[i], bsrc); if constexpr (!HALF) g.xb[i] = x_to_b_order(g.xb[i], bsrc); }
                if constexpr (HALF) sp[i] = gemv_prep_half<RMS>(valid, g.xa[i], RMS ? g.na[i] : g.xa[i], ss);
                else sp[i] = gemv_prep<RMS>(valid, g.xa[i], g.xb[i], RMS ? g.na[i] : g.xa[i], RMS ? g.nb[i] : g.xb[i], ss);
            }
            __builtin_amdgcn_sched_barrier(0);      // all splits done before the first wait on a weight tile
            Q3T(1);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                acc0 = mfma3(g.wa[i], sp[i], acc0);
                if constexpr (NW == 2) acc1 = mfma3(g.wb[i], sp[i], acc1);
            }
        };
        Grp A;
        for (int sb = s0; sb < s1; sb += G) {
            load_group(A, sb);
            __builtin_amdgcn_sched_barrier(0);      // keep every load of the group issued before the first use
            consume(A, sb);
        }
    } else {
        for (int sb = s0; sb < s1; sb += G) {
            u32x4_t wa[G], wb[G];
            float4 xa[G], xb[G], na[G], nb[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
                wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
                if constexpr (NW == 2) wb[i] = Q3_WLOAD(wp2 + (size_t)s * 64);
                const int ko = (s * 32 + kg * 8) < a.K ? s * 32 : 0;
                // lanes of unused batch columns (m >= M) issue no request (down-proj at M = 8: 12.3 -> 11.5 us)
                xa[i] = act ? act_ld4(xres, xr + ko * 4) : float4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!HALF) xb[i] = act ? act_ld4(xres, xr + ko * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
                if constexpr (RMS) {
                    na[i] = *reinterpret_cast<const float4*>(nwp + ko);
                    if constexpr (!HALF) nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = sb + i;
                if constexpr (HALF) {      // the split stays outside the branch: its DPP moves read all lanes
                    const Split3 sp = gemv_prep_half<RMS>(act && s < s1 && (s * 32 + kg * 8) < a.K, xa[i], RMS ? na[i] : xa[i], ss);
                    if (s < s1) {
                        acc0 = mfma3(wa[i], sp, acc0);
                        if constexpr (NW == 2) acc1 = mfma3(wb[i], sp, acc1);
                    }
                } else if (s < s1) {
                    const bool valid = act && (s * 32 + kg * 8) < a.K;
                    const Split3 sp = gemv_prep<RMS>(valid, xa[i], xb[i], RMS ? na[i] : xa[i], RMS ? nb[i] : xb[i], ss);
                    acc0 = mfma3(wa[i], sp, acc0);
                    if constexpr (NW == 2) acc1 = mfma3(wb[i], sp, acc1);
                }
            }
        }
    }
#if Q3_ABLATE == 3
    if (wave == 0 && act) {
        float* yo = a.y + (size_t)m * a.ldy + blockIdx.x * 16 + kg * 4;
        yo[0] = acc0[0] + acc1[0] + ss; yo[1] = acc0[1]; yo[2] = acc0[2]; yo[3] = acc0[3];
    }
    return;
#endif
    Q3T_W(2);
    // side job AFTER the last load was issued: vmcnt retires in issue order, so a store issued ahead of the x loads would sit
    // in front of every wait for them (code-predictor gate/up: B operand ready 0.7 us later with the store first)
    zero_job(a.zero, a.zero_n, blockIdx.x, gridDim.x, tid, NWAVES * 64);
    // partial tile → LDS, layout [col m][row]: lane (m, kg) owns rows kg*4 .. kg*4+3
    *reinterpret_cast<f32x4_t*>(&red[wave][0][m * 16 + kg * 4]) = acc0;
    if constexpr (NW == 2) *reinterpret_cast<f32x4_t*>(&red[wave][1][m * 16 + kg * 4]) = acc1;
    if constexpr (RMS) {
        // (round 5) a column's four k-group lanes are added inside the wave, BEFORE the barrier — work the early waves do while
        // they wait for the last one anyway —, so the epilogue behind the barrier adds 8 partial sums per column instead of 32
        float sc = HALF ? gemv_ss_half(ss) : ss;
        sc += __shfl_xor(sc, 16); sc += __shfl_xor(sc, 32);
        if (kg == 0) ssq[wave][0][m] = sc;
    }
    Q3T(6);
    __syncthreads();
    Q3T(5);
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
                for (int w = 0; w < NWAVES; ++w) tot += ssq[w][0][col];
                const float den = sqrtf(tot / (float)a.K + a.eps);
                v = v / den;
                if constexpr (NW == 2) v2 = v2 / den;
            }
            const int n = blockIdx.x * 16 + row;
            if (n < a.N) {
                if (a.bias) v = v + pre_b;
                if constexpr (EPI == EPI_RESID) v = pre_r + v;
                if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
                if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
                act_st1(act_rsrc(a.y), (col * a.ldy + n) * 4, v);
            }
        }
    }
    Q3T(3); act_drain(); Q3T_W(4); Q3T_FLUSH(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Split-K in two for the narrow, long-K projections (o-proj, down-proj: N <= 2048, K >= 2048, epilogue none / +residual).
//
// What the per-node timeline of the frame shows for them (tools/trace_frame.py, profiles/r3_trace_frame_b8.txt): every
// workgroup of a GEMV reads ALL of x — M*K*4 bytes, 64-196 KB here, 4-6x its share of the weights — and the 32 CUs of an
// XCD pull those same bytes out of one L2 at the same moment: the B operand is ready 2.3-2.9 us after kernel entry
// (0.8 us for a kernel that reads little), and with N = 1024 the 4-row tiles that fill the chip pay a 16-block
// cross-lane reduction on top. Here a 16-row tile's K range is cut in two; grid (tiles, 2): a workgroup reads half of x,
// twice as many workgroups stream weights, and x traffic through the L2s halves (CP shapes: quarters, 4-row -> 16-row).
// The two halves meet in y through f32 atomic adds onto ZEROS (LinArgs::ksplit): two addends commute bit for bit, so the
// result does not depend on arrival order — no ticket, no fence, no second pass. The k = 0 half adds bias + residual.
// ------------------------------------------------------------------------------------------------
template <int EPI, int G, bool HALF, bool MB = false>       // MB: wide sessions, blockIdx.z = block of 16 activation rows
__global__ __launch_bounds__(512) void k_gemv_sk2(Q3_LIN_PRE, LinArgs a_in) {
    Q3_LIN_APPLY(a, a_in);
    a.W2 = nullptr; a.norm_w = nullptr; a.epi = EPI;      // this family's launcher sends the residual / bias / residual pitch in those preloaded slots (below): never read them as what they are named
    constexpr int NWAVES = 8;
    __shared__ __attribute__((aligned(16))) float red[NWAVES][256];
    Q3T_DECL Q3T(0); Q3T_K(7, a.Kpad);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int half = blockIdx.y;
    if constexpr (MB) {                              // each row block is its own two-addend sum (a separate instantiation: the pointer
                                                     // arithmetic on the preloaded arguments cost the M <= 16 kernel 0.4 us per launch)
        const int r0 = (int)blockIdx.z * 16;
        a.x += (size_t)r0 * a.ldx; a.y += (size_t)r0 * a.ldy;
        if (a.resid) a.resid += (size_t)r0 * a.ldr;
        a.M = (a.M - r0) < 16 ? (a.M - r0) : 16;
    }
    const int S = a.Kpad >> 5;                       // k-steps of 32
    const int h0 = (half * S) / 2, h1 = ((half + 1) * S) / 2, Sh = h1 - h0;
    const int s0 = h0 + (wave * Sh) / NWAVES, s1 = h0 + ((wave + 1) * Sh) / NWAVES;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + (size_t)blockIdx.x * S * 64 + lane;
    const int xrow = HALF ? (m & 7) : m, xhalf = HALF ? (m >> 3) * 4 : 0;
    const bool act = HALF ? xrow < a.M : m < a.M;
    const __amdgpu_buffer_rsrc_t xres = act_rsrc(a.x);       // write-through / L1-bypassing transport: q3_kernels.h
    const int xr = ((act ? xrow : 0) * a.ldx + kg * 8 + xhalf) * 4;          // byte offsets into x
    // COAL (round 5, HALF only): the x loads in ROW-CONTIGUOUS lane order. In B-operand order consecutive lanes belong to different
    // batch rows (4-12 KB apart), so the four lanes of a quad never share a 64-byte sector and the CU's texture path spends a
    // cycle per lane instead of one per quad on every x instruction — four times what the same bytes cost as weight tiles. Lane l
    // now asks for the 16 bytes of row l / 8, chunk l % 8 (a quad = one sector, an instruction = eight full 128-byte lines) and the
    // four floats travel to the lane that needs them — (m, kg) wants row m % 8, chunk 2 kg + m / 8 — through the LDS crossbar
    // (four ds_bpermute_b32 per k-step, no LDS memory, no barrier).
    // Full 16-column tiles (!HALF; M > 8 and the row blocks of wide sessions): two instructions per k-step, lane l asks for row
    // l / 4, chunks 2 (l % 4) and 2 (l % 4) + 1 — a quad = one row's 128-byte line, two half-used sectors per instruction instead of
    // four lanes x four sectors —, and both halves of lane (m, kg)'s slot come from lane 4 m + kg.
    constexpr bool COAL = true;
    const int crow = HALF ? lane >> 3 : lane >> 2, cchunk = HALF ? lane & 7 : 2 * (lane & 3);
    const bool cact = crow < a.M;
    const int xc = (__mul24(cact ? crow : 0, a.ldx) + cchunk * 4) * 4;      // (24-bit multiply-add: the 64-bit v_mad form took a pending load's register as its undefined high half and waited for it)
    const int bsrc = HALF ? ((xrow * 8) + 2 * kg + (m >> 3)) * 4 : (m * 4 + kg) * 4;

    // bias and residual of the k = 0 half, requested up front — and NOT combined here: `resid + bias` at this point made the compiler wait
    // (s_waitcnt vmcnt(0)) for both before the first x / weight request went out, a whole round trip at the head of waves 0-3 of every
    // k = 0 workgroup (the residual was written two nodes ago and comes back across the XCDs). They are added behind the barrier, in
    // the same order: (resid + bias) + sum.
    // This kernel has no second matrix, no fused norm and its epilogue is a template parameter, so launch_gemv_sk2 sends the RESIDUAL pointer in
    // the W2 slot, the BIAS pointer in the norm-weight slot and the residual's row pitch in the epi slot of the 14 preloaded dwords: the two
    // requests then need nothing from the argument struct, whose scalar load otherwise stands in front of them (and of the first x request
    // behind them) in waves 0-3 of every k = 0 workgroup.
    const float* __restrict__ rp = reinterpret_cast<const float*>(pW2);
    const float* __restrict__ bp = pnw;
    const int ldr_pre = pepi;
    float pre_b = 0.0f, pre_r = 0.0f;
    {
        const int col = tid >> 4, n = blockIdx.x * 16 + (tid & 15);
        if (half == 0 && tid < 256 && col < pM - (MB ? (int)blockIdx.z * 16 : 0) && n < pN) {
            if (bp) pre_b = bp[n];
            if constexpr (EPI == EPI_RESID) pre_r = act_ld1(act_rsrc(rp), ((col + (MB ? (int)blockIdx.z * 16 : 0)) * ldr_pre + n) * 4);
        }
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.0f;
    struct Grp { u32x4_t wa[G]; float4 xa[G], xb[G]; };
    Grp g;
    for (int sb = s0; sb < s1; sb += G) {
        // x first (L2), weights second (HBM): VMEM returns are counted in issue order (see k_gemv_mfma)
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
            const int ko = (s * 32 + kg * 8) < a.K ? s * 32 : 0;
            if constexpr (COAL) {
                const int kc = (s * 32 + cchunk * 4) < a.K ? s * 32 : 0;
                g.xa[i] = cact ? act_ld4(xres, xc + kc * 4) : float4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!HALF) g.xb[i] = cact ? act_ld4(xres, xc + kc * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
            } else {
                g.xa[i] = act ? act_ld4(xres, xr + ko * 4) : float4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!HALF) g.xb[i] = act ? act_ld4(xres, xr + ko * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
            g.wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        Split3 sp[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = sb + i;
            const bool valid = act && s < s1 && (s * 32 + kg * 8) < a.K;
            if constexpr (COAL) {
                g.xa[i] = x_to_b_order(g.xa[i], bsrc);
                if constexpr (!HALF) g.xb[i] = x_to_b_order(g.xb[i], bsrc);
            }
            if constexpr (HALF) sp[i] = gemv_prep_half<false>(valid, g.xa[i], g.xa[i], ss);
            else sp[i] = gemv_prep<false>(valid, g.xa[i], g.xb[i], g.xa[i], g.xb[i], ss);
        }
        __builtin_amdgcn_sched_barrier(0);
        Q3T(1);
#pragma unroll
        for (int i = 0; i < G; ++i) acc0 = mfma3(g.wa[i], sp[i], acc0);
    }
    Q3T_W(2);
    if constexpr (MB) zero_job(a.zero, a.zero_n, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z, tid, NWAVES * 64);
    else zero_job(a.zero, a.zero_n, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, tid, NWAVES * 64);
    *reinterpret_cast<f32x4_t*>(&red[wave][m * 16 + kg * 4]) = acc0;
    Q3T(6);
    __syncthreads();
    Q3T(5);
    if (tid < 256) {
        const int col = tid >> 4, row = tid & 15, n = blockIdx.x * 16 + row;
        if (col < a.M && n < a.N) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) v += red[w][tid];
            if (half == 0) { const float pre = EPI == EPI_RESID ? pre_r + pre_b : pre_b; v = pre + v; }
            unsafeAtomicAdd(&a.y[(size_t)col * a.ldy + n], v);
        }
    }
    Q3T(3); act_drain(); Q3T_W(4); Q3T_FLUSH(a, blockIdx.y * gridDim.x + blockIdx.x);
}

static hipError_t launch_gemv_sk2(const LinArgs& a, hipStream_t st) {
    if (a.norm_w || (a.epi != EPI_NONE && a.epi != EPI_RESID) || a.tiled != 1 || a.M > 64) return hipErrorInvalidValue;
    const int tiles = (a.N + 15) / 16, S = a.Kpad >> 5;
    if (S < 16) return hipErrorInvalidValue;
    const dim3 grid(tiles, 2, (a.M + 15) / 16), blk(512);     // M > 16 (wide sessions): one grid plane per 16 rows
    const bool g6 = ((S / 2) % 48) == 0;             // a wave's slice is a multiple of 6 k-steps (K = 3072, 6144): no ragged group
    const bool half = a.M <= 8;
    // preloaded slots of this family: W2 = residual, norm weight = bias, epi = residual row pitch (see the kernel)
#define Q3_SK2_PASS a.W, reinterpret_cast<const uint16_t*>(a.resid), a.x, a.bias, a.M, a.N, a.K, a.Kpad, a.ldx, a.ldr, a
#define Q3_SK2(E, GG, H) hipLaunchKernelGGL((k_gemv_sk2<E, GG, H>), grid, blk, 0, st, Q3_SK2_PASS)
    if (a.M > 16) {                                  // row blocks: full 16-column tiles only
        if (a.epi == EPI_RESID) { if (g6) hipLaunchKernelGGL((k_gemv_sk2<EPI_RESID, 6, false, true>), grid, blk, 0, st, Q3_SK2_PASS); else hipLaunchKernelGGL((k_gemv_sk2<EPI_RESID, 4, false, true>), grid, blk, 0, st, Q3_SK2_PASS); }
        else { if (g6) hipLaunchKernelGGL((k_gemv_sk2<EPI_NONE, 6, false, true>), grid, blk, 0, st, Q3_SK2_PASS); else hipLaunchKernelGGL((k_gemv_sk2<EPI_NONE, 4, false, true>), grid, blk, 0, st, Q3_SK2_PASS); }
        return hipGetLastError();
    }
    if (a.epi == EPI_RESID) {
        if (g6) { if (half) Q3_SK2(EPI_RESID, 6, true); else Q3_SK2(EPI_RESID, 6, false); }
        else    { if (half) Q3_SK2(EPI_RESID, 4, true); else Q3_SK2(EPI_RESID, 4, false); }
    } else {
        if (g6) { if (half) Q3_SK2(EPI_NONE, 6, true); else Q3_SK2(EPI_NONE, 6, false); }
        else    { if (half) Q3_SK2(EPI_NONE, 4, true); else Q3_SK2(EPI_NONE, 4, false); }
    }
#undef Q3_SK2
#undef Q3_SK2_PASS
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Third generation of the 16-row-tile kernel: x goes through a wave-private LDS staging buffer.
//
// Measured on MI355X (tools/bench_kernels.py with the -DQ3_ABLATE builds): in k_gemv_mfma the B-operand loads — two
// 16-B x loads and two 16-B norm-weight loads per lane per k-step, against ONE weight load — take 30-45 % of the
// kernel (removing them: qkv 7.8 -> 4.7 us, down 12.3 -> 6.6 us at M = 8) although they hit L2: every 64-lane
// dwordx4 load costs the CU's vector-memory pipe the same ~16 clocks whether it brings 1 KiB of fresh weights or
// 16 half-used cache lines of x, so 4 of every 5 VMEM instructions moved activations.
// Here a wave brings the x rows of a 128-float k-group in with row-contiguous loads (one instruction = two rows x
// 512 B, only rows < M), applies the norm weight and accumulates sum(x^2) on that coalesced form, parks the
// result in its own 8.25 KiB LDS buffer and reads the MFMA B operand back with two ds_read_b128 per k-step — the
// LDS pipe is separate from VMEM. Per 4 k-steps at M = 8: 4 (x) + 1 (norm) + 4 (weights) VMEM instructions
// instead of 20. Weight tiles and the next group's x are requested one group ahead (double buffered), so a wave
// keeps 8 KiB of weights in flight however long its K slice is, and one geometry (8 waves) serves every shape.
// ------------------------------------------------------------------------------------------------
constexpr int ZS = 132;              // staging row stride in floats: 128 + 4 pad → the 16 rows of one ds_read_b128
                                     // phase land on 16 distinct 4-bank groups
constexpr int ZB = 16 * ZS;          // floats per wave

struct XGroup { float4 x[8]; float4 nw; };    // rows (2r + lane/32), floats (lane%32)*4 .. +3 of the k-group

template <bool RMS>
__device__ __forceinline__ void xg_load(XGroup& g, __amdgpu_buffer_rsrc_t x, int ldx, const float* __restrict__ norm_w,
                                        int M, int K, int k0, int lane) {
    const int half = lane >> 5, c = k0 + (lane & 31) * 4;
    const bool kok = c < K;                                    // K % 4 == 0: a float4 is all in or all out
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (2 * r < M) {                                       // wave-uniform
            const int row = 2 * r + half;
            g.x[r] = (kok && row < M) ? act_ld4(x, (row * ldx + c) * 4) : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if constexpr (RMS) g.nw = kok ? *reinterpret_cast<const float4*>(norm_w + c) : float4{0.f, 0.f, 0.f, 0.f};
}

template <bool RMS>
__device__ __forceinline__ void xg_stage(const XGroup& g, float* __restrict__ zb, float (&ss)[8], int M, int lane) {
    const int half = lane >> 5, c4 = (lane & 31) * 4;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (2 * r < M) {
            float4 v = g.x[r];
            if constexpr (RMS) {
                ss[r] = fmaf(v.x, v.x, ss[r]); ss[r] = fmaf(v.y, v.y, ss[r]); ss[r] = fmaf(v.z, v.z, ss[r]); ss[r] = fmaf(v.w, v.w, ss[r]);
                v.x *= g.nw.x; v.y *= g.nw.y; v.z *= g.nw.z; v.w *= g.nw.w;
            }
            *reinterpret_cast<float4*>(zb + (2 * r + half) * ZS + c4) = v;
        }
    }
}

template <int NW>
__device__ __forceinline__ void wg_load(u32x4_t (&wa)[4], u32x4_t (&wb)[4], const u32x4_t* __restrict__ wp,
                                        const u32x4_t* __restrict__ wp2, int sb, int s1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);     // ragged last group: duplicate load, never consumed
        wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
        if constexpr (NW == 2) wb[i] = Q3_WLOAD(wp2 + (size_t)s * 64);
    }
}

template <int NW>
__device__ __forceinline__ void g_compute(const u32x4_t (&wa)[4], const u32x4_t (&wb)[4], const float* __restrict__ zrow,
                                          bool act, int sb, int s1, f32x4_t& acc0, f32x4_t& acc1) {
    Split3 sp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a0 = *reinterpret_cast<const float4*>(zrow + i * 32);
        const float4 a1 = *reinterpret_cast<const float4*>(zrow + i * 32 + 4);
        float xv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = act ? xv[e] : 0.0f;     // columns >= M read rows nobody staged
        sp[i] = split3(xv);
    }
    __builtin_amdgcn_sched_barrier(0);      // the splits (LDS + VALU) run under the weight latency; only the MFMAs wait on HBM
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (sb + i < s1) {                  // wave-uniform
            acc0 = mfma3(wa[i], sp[i], acc0);
            if constexpr (NW == 2) acc1 = mfma3(wb[i], sp[i], acc1);
        }
    }
}

template <int EPI, bool RMS>
__global__ __launch_bounds__(512) void k_gemv_lds(Q3_LIN_PRE, LinArgs a_in) {
    Q3_LIN_APPLY(a, a_in);
    constexpr int NWAVES = 8, G = 4;
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float lds[NWAVES * ZB + NWAVES * 16];
    Q3T_DECL Q3T(0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int S = a.Kpad >> 5;                       // k-steps of 32
    const int s0 = (wave * S) / NWAVES, s1 = ((wave + 1) * S) / NWAVES;
    const size_t tile_base = (size_t)blockIdx.x * S * 64 + lane;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + tile_base;
    const u32x4_t* __restrict__ wp2 = NW == 2 ? reinterpret_cast<const u32x4_t*>(a.W2) + tile_base : wp;
    const bool act = m < a.M;
    float* __restrict__ zb = lds + wave * ZB;
    const float* __restrict__ zrow = zb + m * ZS + kg * 8;
    float* __restrict__ ssq = lds + NWAVES * ZB;      // [NWAVES][16], outside the area `red` aliases
    const __amdgpu_buffer_rsrc_t xres = act_rsrc(a.x);       // write-through / L1-bypassing transport: q3_kernels.h

    // epilogue operands (bias, residual) are requested up front so their round trip hides under the weight stream
    float pre_b = 0.0f, pre_r = 0.0f;
    {
        const int col = tid >> 4, n = blockIdx.x * 16 + (tid & 15);
        if (tid < 256 && col < a.M && n < a.N) {
            if (a.bias) pre_b = a.bias[n];
            if constexpr (EPI == EPI_RESID) pre_r = act_ld1(act_rsrc(a.resid), (col * a.ldr + n) * 4);
        }
    }

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float ss[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    XGroup X0, X1;
    u32x4_t wa0[G], wb0[G], wa1[G], wb1[G];
    if (s0 < s1) {
        // x first, weights second: VMEM returns are counted in issue order, the staging must not wait for HBM.
        // (Requesting weights two groups ahead instead of one measured 5-10 % slower: more live registers, same latency.)
        xg_load<RMS>(X0, xres, a.ldx, a.norm_w, a.M, a.K, s0 * 32, lane);
        wg_load<NW>(wa0, wb0, wp, wp2, s0, s1);
        for (int sb = s0; sb < s1; sb += 2 * G) {
            xg_stage<RMS>(X0, zb, ss, a.M, lane);
            if (sb == s0) Q3T(1);
            const bool more1 = sb + G < s1;
            if (more1) {
                xg_load<RMS>(X1, xres, a.ldx, a.norm_w, a.M, a.K, (sb + G) * 32, lane);
                wg_load<NW>(wa1, wb1, wp, wp2, sb + G, s1);
            }
            __builtin_amdgcn_sched_barrier(0);
            g_compute<NW>(wa0, wb0, zrow, act, sb, s1, acc0, acc1);
            if (!more1) break;
            xg_stage<RMS>(X1, zb, ss, a.M, lane);
            if (sb + 2 * G < s1) {
                xg_load<RMS>(X0, xres, a.ldx, a.norm_w, a.M, a.K, (sb + 2 * G) * 32, lane);
                wg_load<NW>(wa0, wb0, wp, wp2, sb + 2 * G, s1);
            }
            __builtin_amdgcn_sched_barrier(0);
            g_compute<NW>(wa1, wb1, zrow, act, sb + G, s1, acc0, acc1);
        }
    }
    if constexpr (RMS) {
        // per-row sum(x^2) of this wave's K slice: lanes 0-31 hold row 2r, lanes 32-63 row 2r+1
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float v = ss[r];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            if ((lane & 31) == 0) ssq[wave * 16 + 2 * r + (lane >> 5)] = v;
        }
    }
    Q3T_W(2);
    zero_job(a.zero, a.zero_n, blockIdx.x, gridDim.x, tid, NWAVES * 64);
    __syncthreads();                                  // every wave is done with its staging buffer: `red` may alias it
    float* __restrict__ red = lds;                    // [NWAVES][NW][256], layout [col m][row]
    *reinterpret_cast<f32x4_t*>(&red[(wave * NW + 0) * 256 + m * 16 + kg * 4]) = acc0;
    if constexpr (NW == 2) *reinterpret_cast<f32x4_t*>(&red[(wave * NW + 1) * 256 + m * 16 + kg * 4]) = acc1;
    Q3T(6);
    __syncthreads();
    Q3T(5);
    if (tid < 256) {
        const int col = tid >> 4, row = tid & 15;
        if (col < a.M) {
            float v = 0.0f, v2 = 0.0f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) {
                v += red[(w * NW + 0) * 256 + tid];
                if constexpr (NW == 2) v2 += red[(w * NW + 1) * 256 + tid];
            }
            if constexpr (RMS) {
                float tot = 0.0f;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) tot += ssq[w * 16 + col];
                const float den = sqrtf(tot / (float)a.K + a.eps);
                v = v / den;
                if constexpr (NW == 2) v2 = v2 / den;
            }
            const int n = blockIdx.x * 16 + row;
            if (n < a.N) {
                if (a.bias) v = v + pre_b;
                if constexpr (EPI == EPI_RESID) v = pre_r + v;
                if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
                if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
                act_st1(act_rsrc(a.y), (col * a.ldy + n) * 4, v);
            }
        }
    }
    Q3T(3); act_drain(); Q3T_W(4); Q3T_FLUSH(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// SwiGLU pair with 24 rows per workgroup (the talker's gate/up: N = 6144 = 384 sixteen-row tiles on 256 CUs).
//
// One tile pair per workgroup gives 384 workgroups: half of the CUs get two (2 x 195 KB of weights + x), the others one,
// and the launch lasts as long as the loaded half (12.6 us in-kernel, per-CU fetch bound: the timeline prices a CU's
// bytes at ~18 us/MB). Here a workgroup takes ONE AND A HALF tiles of each matrix — 256 workgroups, one per CU, 260 KB
// each, and the x rows are read once per 24 weight rows instead of once per 16. Workgroup 2j owns tile 3j and rows 0-7 of
// tile 3j+1, workgroup 2j+1 rows 8-15 of tile 3j+1 and tile 3j+2. No second weight image: in the 16-row tile layout
// (slot = row + 16 * k-group) the eight rows of a half are the lanes (0-7 | 8-15) + 16*kg — four aligned 128-byte runs,
// whole L2 lines — so a half tile is fetched with half of the lanes and no wasted byte; its MFMA simply sees zeros in the
// other eight A-operand rows. 8 waves split K; groups of 4 k-steps; x / norm weight first, weights second; the bf16x3
// split shared by the four weight operands of a k-step.
// ------------------------------------------------------------------------------------------------
template <bool HALF, bool CO = true>
__global__ __launch_bounds__(512) void k_gemv_gu24(Q3_LIN_PRE, LinArgs a_in) {
    Q3_LIN_APPLY(a, a_in);
    constexpr int NWAVES = 8, G = 4;
    __shared__ __attribute__((aligned(16))) float red[NWAVES][4][256];        // [wave][gate full, gate half, up full, up half][col m][row]
    __shared__ float ssq[NWAVES][4][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int pair = blockIdx.x >> 1, odd = blockIdx.x & 1;
    const int tile_full = 3 * pair + (odd ? 2 : 0), tile_half = 3 * pair + 1;
    const bool half_lane = ((m >> 3) == odd);                               // this lane's weight row belongs to this workgroup's half of the shared tile
    const int S = a.Kpad >> 5;
    const int s0 = (wave * S) / NWAVES, s1 = ((wave + 1) * S) / NWAVES;
    const u32x4_t* __restrict__ gF = reinterpret_cast<const u32x4_t*>(a.W) + (size_t)tile_full * S * 64 + lane;
    const u32x4_t* __restrict__ gH = reinterpret_cast<const u32x4_t*>(a.W) + (size_t)tile_half * S * 64 + lane;
    const u32x4_t* __restrict__ uF = reinterpret_cast<const u32x4_t*>(a.W2) + (size_t)tile_full * S * 64 + lane;
    const u32x4_t* __restrict__ uH = reinterpret_cast<const u32x4_t*>(a.W2) + (size_t)tile_half * S * 64 + lane;
    const int xrow = HALF ? (m & 7) : m, xhalf = HALF ? (m >> 3) * 4 : 0;
    const bool act = HALF ? xrow < a.M : m < a.M;
    const __amdgpu_buffer_rsrc_t xres = act_rsrc(a.x);       // write-through / L1-bypassing transport: q3_kernels.h
    const int xr = ((act ? xrow : 0) * a.ldx + kg * 8 + xhalf) * 4;          // byte offsets into x
    const float* __restrict__ nwp = a.norm_w + kg * 8 + xhalf;
    constexpr bool COAL = CO;                        // see k_gemv_sk2: x in row-contiguous lane order, moved by the LDS crossbar (off for M <= 2)
    const int crow = HALF ? lane >> 3 : lane >> 2, cchunk = HALF ? lane & 7 : 2 * (lane & 3);
    const bool cact = crow < a.M;
    const int xc = ((cact ? crow : 0) * a.ldx + cchunk * 4) * 4;
    const int bsrc = HALF ? ((xrow * 8) + 2 * kg + (m >> 3)) * 4 : (m * 4 + kg) * 4;
    f32x4_t aGF = {0.f, 0.f, 0.f, 0.f}, aGH = aGF, aUF = aGF, aUH = aGF;
    float ss = 0.0f;
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    for (int sb = s0; sb < s1; sb += G) {
        float4 xa[G], xb[G], na[G], nb[G];
        u32x4_t wgf[G], wgh[G], wuf[G], wuh[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
            const int ko = s * 32;
            if constexpr (COAL) {
                xa[i] = cact ? act_ld4(xres, xc + ko * 4) : float4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!HALF) xb[i] = cact ? act_ld4(xres, xc + ko * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
            } else {
                xa[i] = act ? act_ld4(xres, xr + ko * 4) : float4{0.f, 0.f, 0.f, 0.f};
                if constexpr (!HALF) xb[i] = act ? act_ld4(xres, xr + ko * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
            }
            na[i] = *reinterpret_cast<const float4*>(nwp + ko);
            if constexpr (!HALF) nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
            wgf[i] = Q3_WLOAD(gF + (size_t)s * 64); wuf[i] = Q3_WLOAD(uF + (size_t)s * 64);
            wgh[i] = half_lane ? Q3_WLOAD(gH + (size_t)s * 64) : zero4;
            wuh[i] = half_lane ? Q3_WLOAD(uH + (size_t)s * 64) : zero4;
        }
        __builtin_amdgcn_sched_barrier(0);
        Split3 sp[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const bool valid = act && (sb + i) < s1;
            if constexpr (COAL) { xa[i] = x_to_b_order(xa[i], bsrc); if constexpr (!HALF) xb[i] = x_to_b_order(xb[i], bsrc); }
            if constexpr (HALF) sp[i] = gemv_prep_half<true>(valid, xa[i], na[i], ss);
            else sp[i] = gemv_prep<true>(valid, xa[i], xb[i], na[i], nb[i], ss);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            aGF = mfma3(wgf[i], sp[i], aGF); aUF = mfma3(wuf[i], sp[i], aUF);
            aGH = mfma3(wgh[i], sp[i], aGH); aUH = mfma3(wuh[i], sp[i], aUH);
        }
    }
    zero_job(a.zero, a.zero_n, blockIdx.x, gridDim.x, tid, NWAVES * 64);
    *reinterpret_cast<f32x4_t*>(&red[wave][0][m * 16 + kg * 4]) = aGF;
    *reinterpret_cast<f32x4_t*>(&red[wave][1][m * 16 + kg * 4]) = aGH;
    *reinterpret_cast<f32x4_t*>(&red[wave][2][m * 16 + kg * 4]) = aUF;
    *reinterpret_cast<f32x4_t*>(&red[wave][3][m * 16 + kg * 4]) = aUH;
    {
        float sc = HALF ? gemv_ss_half(ss) : ss;          // see k_gemv_mfma: the four k-group lanes of a column added before the barrier
        sc += __shfl_xor(sc, 16); sc += __shfl_xor(sc, 32);
        if (kg == 0) ssq[wave][0][m] = sc;
    }
    __syncthreads();
    // 24 rows x 16 columns: thread (col, r24); rows 0-15 = the full tile, 16-23 = this workgroup's half of the shared tile
    if (tid < 384) {
        const int col = tid / 24, r = tid - col * 24;
        if (col < a.M) {
            const bool full = r < 16;
            const int row16 = full ? r : (odd * 8 + (r - 16));          // row inside its 16-row tile
            const int which = full ? 0 : 1, idx = col * 16 + row16;
            float g = 0.0f, u = 0.0f, tot = 0.0f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) { g += red[w][which][idx]; u += red[w][2 + which][idx]; }
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) tot += ssq[w][0][col];
            const float den = sqrtf(tot / (float)a.K + a.eps);
            g = g / den; u = u / den;
            const int n = (full ? tile_full : tile_half) * 16 + row16;
            if (n < a.N) act_st1(act_rsrc(a.y), (col * a.ldy + n) * 4, (g / (1.0f + expf(-g))) * u);
        }
    }
    act_drain();
}

template <int EPI, bool RMS>
static hipError_t launch_gemv_t(const LinArgs& a, hipStream_t st) {
    const int tiles = (a.N + 15) / 16;
    const int S = a.Kpad >> 5;
    // Kernel choice, from tools/bench_kernels.py on MI355X at M = 8 (profiles/r1_gemv_microbench_*.log; run-to-run noise
    // is about +-0.5 us, so only consistent differences are encoded):
    //   * K <= 1024 (a wave's slice is one 4-step group): 8 waves, register-direct, everything requested at kernel
    //     start — code-predictor qkv 4.6, gate/up 5.4, lm_head 4.3 us (16 waves: 6.6 / 8.0 / 6.3; LDS-staged: 5.7 / 6.5 / 5.1);
    //   * K = 2048 with fewer than 256 tiles (CUs idle, latency-bound): the LDS-staged generation, flat in M —
    //     o-proj 5.9, codec head 6.2 us (register-direct 6.6 / 6.9);
    //   * K = 2048 with >= 256 tiles: 8 waves register-direct — qkv 7.4, gate/up 13.7 us (LDS-staged 8.4 / 17.8);
    //   * K >= 4096 (down-proj): 16 waves so that a wave's slice stays 12 steps — 11.5 us (8 waves 12.8, 4 waves 14.4).
    // Q3_GEMV_WAVES = 4 / 8 / 16 forces a register-direct geometry, 1 forces the LDS-staged kernel (tuning aid).
    bool big = S >= 96;
    static const int force = [] { const char* e = getenv("Q3_GEMV_WAVES"); return e ? atoi(e) : 0; }();   // tuning aid
    const bool lds_ok = a.K % 4 == 0 && S >= 8;
    // long-K projections (down-proj, S >= 96) beyond 8 tokens: the LDS-staged kernel is flat in M where the
    // register-direct one pays for every x row (talker down M = 16: 14.2 vs 16.2 us; code-predictor down 7.9 vs 8.8)
    static const bool no_lds = getenv("Q3_GEMV_NO_LDS") != nullptr;       // tuning aids
    static const bool big8 = getenv("Q3_GEMV_BIG8") != nullptr;
    const bool lds_pick = !no_lds && ((tiles < 256 && S > 32 && S <= 64) || (a.M > 8 && S >= 96));
    if (big8 && a.M <= 8) big = false;
    if (lds_ok && (force == 1 || (force == 0 && lds_pick))) {
        hipLaunchKernelGGL((k_gemv_lds<EPI, RMS>), dim3(tiles), dim3(512), 0, st, Q3_LIN_PASS(a));
        return hipGetLastError();
    }
    if (force == 8) big = false; else if (force == 16) big = true;
    // the two-matrix SwiGLU tile at K = 2048 (talker gate/up, 50 MB): 4 waves — three workgroups fit a CU and the stream
    // keeps more bytes in flight: 14.1 us at M = 8 against 15.9 (8 waves) / 16.5 (16 waves); M = 1: 13.3 / 14.4 / 15.8
    if constexpr (EPI == EPI_SWIGLU && RMS) {
        // 24 rows per workgroup where that is what puts one workgroup on every CU (the talker's gate/up: 384 tiles -> 256)
        static const bool no24 = getenv("Q3_GEMV_NO_GU24") != nullptr;      // A/B aid
        if (!no24 && force == 0 && tiles % 3 == 0 && tiles / 3 * 2 >= 224 && tiles / 3 * 2 <= 288 && S >= 64 && S % 8 == 0 && a.K == a.Kpad && !a.bias) {
            if (a.M <= 2) hipLaunchKernelGGL((k_gemv_gu24<true, false>), dim3(tiles / 3 * 2), dim3(512), 0, st, Q3_LIN_PASS(a));
            else if (a.M <= 8) hipLaunchKernelGGL((k_gemv_gu24<true>), dim3(tiles / 3 * 2), dim3(512), 0, st, Q3_LIN_PASS(a));
            else hipLaunchKernelGGL((k_gemv_gu24<false>), dim3(tiles / 3 * 2), dim3(512), 0, st, Q3_LIN_PASS(a));
            return hipGetLastError();
        }
    }
    const bool four = force == 4 || (force == 0 && EPI == EPI_SWIGLU && S == 64 && tiles >= 256);
    // M <= 8: half-row interleaved x loads (one x instruction per k-step). Q3_GEMV_NO_HALF=1 keeps the two-instruction form (A/B aid).
    static const bool no_half = getenv("Q3_GEMV_NO_HALF") != nullptr;
    const bool half = a.M <= 8 && !no_half;
    if (four) {
        if (half && a.M <= 2) hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 4, true, false>), dim3(tiles), dim3(4 * 64), 0, st, Q3_LIN_PASS(a));
        else if (half) hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 4, true>), dim3(tiles), dim3(4 * 64), 0, st, Q3_LIN_PASS(a));
        else hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 4, false>), dim3(tiles), dim3(4 * 64), 0, st, Q3_LIN_PASS(a));
        return hipGetLastError();
    }
    if (big) {
        if (half) hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 16, true>), dim3(tiles), dim3(16 * 64), 0, st, Q3_LIN_PASS(a));
        else hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 16, false>), dim3(tiles), dim3(16 * 64), 0, st, Q3_LIN_PASS(a));
    } else {
        if (half && a.M <= 2) hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 8, true, false>), dim3(tiles), dim3(8 * 64), 0, st, Q3_LIN_PASS(a));
        else if (half) hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 8, true>), dim3(tiles), dim3(8 * 64), 0, st, Q3_LIN_PASS(a));
        else hipLaunchKernelGGL((k_gemv_mfma<EPI, RMS, 8, false>), dim3(tiles), dim3(8 * 64), 0, st, Q3_LIN_PASS(a));
    }
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

// transposing butterfly over the 16 k-blocks of the 4x4x4 MFMA (lane = 4b + j; exchanges with lane ^ 32, 16, 8, 4): N live values
// are halved per step while N > 1, the remaining steps reduce the single survivor plainly
template <int V, int N, int OFF>
__device__ __forceinline__ void tsum_blocks(float (&v)[V], int lane, int& idx) {
    if constexpr (OFF >= 4) {
        if constexpr (N > 1) {
            const bool up = (lane & OFF) != 0;
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const float send = up ? v[i] : v[i + N / 2];
                const float keep = up ? v[i + N / 2] : v[i];
                v[i] = keep + __shfl_xor(send, OFF);
            }
            if (up) idx += N / 2;
            tsum_blocks<V, N / 2, OFF / 2>(v, lane, idx);
        } else {
            v[0] += __shfl_xor(v[0], OFF);
            tsum_blocks<V, 1, OFF / 2>(v, lane, idx);
        }
    }
}

template <int EPI, bool RMS, int MG, int G = 2>
__global__ __launch_bounds__(512) void k_gemv_mfma4(Q3_LIN_PRE, LinArgs a_in) {
    Q3_LIN_APPLY(a, a_in);
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    __shared__ float red[8][NW][MG][4][4];
    __shared__ float ssq[8][MG][4];
    Q3T_DECL Q3T(0);
    const int nwv = blockDim.x >> 6;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 3, kb = lane >> 2;
    const int S = a.Kpad >> 7;                       // k-steps of 128
    const int s0 = (wave * S) / nwv, s1 = ((wave + 1) * S) / nwv;
    const size_t tile_base = (size_t)blockIdx.x * S * 64 + lane;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + tile_base;
    const u32x4_t* __restrict__ wp2 = NW == 2 ? reinterpret_cast<const u32x4_t*>(a.W2) + tile_base : wp;
    const __amdgpu_buffer_rsrc_t xres = act_rsrc(a.x);       // write-through / L1-bypassing transport: q3_kernels.h
    int xr[MG]; bool act[MG];                                 // byte offsets into x
#pragma unroll
    for (int g = 0; g < MG; ++g) {
        const int m = g * 4 + j;
        act[g] = m < a.M;
        xr[g] = (act[g] ? m : 0) * a.ldx * 4;
    }
    const float* __restrict__ nwp = RMS ? a.norm_w : nullptr;

    float pre_b = 0.0f, pre_r = 0.0f;     // epilogue operands requested up front (see k_gemv_mfma)
    // (round 6) the +residual instances (o / down projections at M <= 2: 206 nodes of a single-utterance frame) have no second
    // matrix and no fused norm, so their launcher sends the RESIDUAL pointer in the W2 slot, the BIAS pointer in the norm-weight slot
    // and the residual's row pitch in the epi slot of the 14 preloaded dwords — as k_gemv_sk2 does: these two requests, and wave 0's
    // first x request behind them, then do not wait for the argument struct's scalar load
    const float* __restrict__ bp = EPI == EPI_RESID ? pnw : a.bias;
    const float* __restrict__ rp = EPI == EPI_RESID ? reinterpret_cast<const float*>(pW2) : a.resid;
    const int ldr_pre = EPI == EPI_RESID ? pepi : a.ldr;
    if constexpr (EPI == EPI_RESID) { a.W2 = nullptr; a.norm_w = nullptr; a.epi = EPI; }      // never read the repurposed slots as what they are named
    if (tid < 16 * MG) {
        const int m = tid >> 2, n = blockIdx.x * 4 + (tid & 3);
        if (m < pM && n < pN) {
            if (bp) pre_b = bp[n];
            if constexpr (EPI == EPI_RESID) pre_r = act_ld1(act_rsrc(rp), (m * ldr_pre + n) * 4);
        }
    }
    f32x4_t acc[NW][MG];
    float ss[MG];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int g = 0; g < MG; ++g) acc[w][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < MG; ++g) ss[g] = 0.0f;

    // Non-RMS variants hoist the x loads and the split ahead of the weight wait (see k_gemv_mfma); the RMS variants
    // (norm-weight loads on top: the M <= 2 gate/up, lm_head path) keep the interleaved per-step order — hoisting
    // measured 15-40 % slower there (register pressure).
    for (int sb = s0; sb < s1; sb += G) {
        if constexpr (!RMS) {
            // x / norm weight first, weight tiles second (see k_gemv_mfma): the split runs under the HBM latency
            u32x4_t wa[G], wb[G];
            float4 xa[G][MG], xb[G][MG], na[G], nb[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
                const int ko = (s * 128 + kb * 8) < a.K ? s * 128 + kb * 8 : 0;   // past K (K % 128 != 0): column 0, masked below — never past the row
#pragma unroll
                for (int g = 0; g < MG; ++g) {
                    xa[i][g] = act[g] ? act_ld4(xres, xr[g] + ko * 4) : float4{0.f, 0.f, 0.f, 0.f};     // unused columns: no request
                    xb[i][g] = act[g] ? act_ld4(xres, xr[g] + ko * 4 + 16) : float4{0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (RMS) {
                    na[i] = *reinterpret_cast<const float4*>(nwp + ko);
                    nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
#if Q3_ABLATE == 4
                wa[i] = u32x4_t{(unsigned)s, 1u, 2u, 3u}; wb[i] = wa[i];
#else
                wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
                if constexpr (NW == 2) wb[i] = Q3_WLOAD(wp2 + (size_t)s * 64);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            Split3 sp[G][MG];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = sb + i;
                const bool kok = s < s1 && (s * 128 + kb * 8) < a.K;     // ragged last group: zero operand, no branch
#pragma unroll
                for (int g = 0; g < MG; ++g)
                    sp[i][g] = gemv_prep<RMS>(act[g] && kok, xa[i][g], xb[i][g], RMS ? na[i] : xa[i][g], RMS ? nb[i] : xb[i][g], ss[g]);
            }
            __builtin_amdgcn_sched_barrier(0);
            Q3T(1);
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int g = 0; g < MG; ++g) {
                    acc[0][g] = mfma4_tile(wa[i], sp[i][g], acc[0][g]);
                    if constexpr (NW == 2) acc[1][g] = mfma4_tile(wb[i], sp[i][g], acc[1][g]);
                }
        } else {
            u32x4_t wa[G], wb[G];
            float4 xa[G][MG], xb[G][MG], na[G], nb[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);
                wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
                if constexpr (NW == 2) wb[i] = Q3_WLOAD(wp2 + (size_t)s * 64);
                const int ko = (s * 128 + kb * 8) < a.K ? s * 128 + kb * 8 : 0;   // past K (K % 128 != 0): column 0, masked below — never past the row
#pragma unroll
                for (int g = 0; g < MG; ++g) {
                    xa[i][g] = act_ld4(xres, xr[g] + ko * 4);
                    xb[i][g] = act_ld4(xres, xr[g] + ko * 4 + 16);
                }
                na[i] = *reinterpret_cast<const float4*>(nwp + ko);
                nb[i] = *reinterpret_cast<const float4*>(nwp + ko + 4);
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = sb + i;
                if (s < s1) {
                    const bool kok = (s * 128 + kb * 8) < a.K;
#pragma unroll
                    for (int g = 0; g < MG; ++g) {
                        const Split3 sp = gemv_prep<RMS>(act[g] && kok, xa[i][g], xb[i][g], na[i], nb[i], ss[g]);
                        acc[0][g] = mfma4_tile(wa[i], sp, acc[0][g]);
                        if constexpr (NW == 2) acc[1][g] = mfma4_tile(wb[i], sp, acc[1][g]);
                    }
                }
            }
        }
    }
#if Q3_ABLATE == 3
    if (wave == 0 && lane < 4 && act[0]) {
        float v = ss[0];
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int g = 0; g < MG; ++g) v += acc[w][g][0] + acc[w][g][1] + acc[w][g][2] + acc[w][g][3];
        a.y[(size_t)lane * a.ldy + blockIdx.x * 4] = v;
    }
    return;
#endif
    Q3T_W(2);
    zero_job(a.zero, a.zero_n, blockIdx.x, gridDim.x, tid, blockDim.x);
    // Sum the 16 k-blocks (lanes 4b + j, all b) of the V = 4 * NW * MG values a lane holds. A plain butterfly costs 4 exchanges
    // per value (32 at M = 8: 1 us of the launch in the frame's timeline); the TRANSPOSING butterfly hands half of the values
    // to the partner at every step and keeps the other half — V/2 + V/4 + ... exchanges, 8 at M = 8 — and leaves value `idx`
    // complete in the lanes whose block bits spell idx (replicated over the bits that were reduced plainly).
    constexpr int V = 4 * NW * MG;
    float vals[V];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) vals[(w * MG + g) * 4 + i] = acc[w][g][i];
    int idx = 0;
    tsum_blocks<V, V, 32>(vals, lane, idx);
    // canonical writer of (idx, j): the lane whose plainly-reduced block bits are zero
    constexpr int PLAIN_MASK = V >= 16 ? 0 : V == 8 ? 4 : V == 4 ? 12 : 28;
    constexpr int R = V > 16 ? V / 16 : 1;          // survivors per lane (V = 32: two neighbouring values)
    if ((lane & PLAIN_MASK) == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) (&red[wave][0][0][0][0])[(idx + r) * 4 + j] = vals[r];
    }
    if constexpr (RMS) {
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            const float v = sum_over_blocks(ss[g]);
            if (lane < 4) ssq[wave][g][lane] = v;
        }
    }
    Q3T(6);
    __syncthreads();
    Q3T(5);
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
            if (a.bias) v = v + pre_b;
            if constexpr (EPI == EPI_RESID) v = pre_r + v;
            if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
            if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
            act_st1(act_rsrc(a.y), (m * a.ldy + n) * 4, v);
        }
    }
    Q3T(3); act_drain(); Q3T_W(4); Q3T_FLUSH(a, blockIdx.x);
}

template <int EPI, bool RMS>
static hipError_t launch_gemv4_t(const LinArgs& a, hipStream_t st) {
    const int tiles = (a.N + 3) / 4;
    const int S = a.Kpad >> 7;
    const int nwv = S >= 8 ? 8 : (S >= 4 ? 4 : (S >= 2 ? 2 : 1));
    const int mg = (a.M + 3) / 4;
    // groups of 3 k-steps when a wave's slice is a multiple of 3 (K = 3072: 24 steps over 8 waves), so that no
    // group is ragged (a ragged group still pays its split + MFMAs on a zero operand)
    const bool g3 = (S % nwv == 0) && ((S / nwv) % 3 == 0) && !RMS && EPI != EPI_SWIGLU;
    if constexpr (EPI == EPI_RESID && !RMS) {      // preloaded slots of the +residual instances: W2 = residual, norm weight = bias, epi = residual row pitch (see the kernel)
#define Q3_G4R_PASS a.W, reinterpret_cast<const uint16_t*>(a.resid), a.x, a.bias, a.M, a.N, a.K, a.Kpad, a.ldx, a.ldr, a
        if (mg <= 1) {
            if (g3) hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 1, 3>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_G4R_PASS);
            else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 1>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_G4R_PASS);
        } else if (mg == 2) {
            if (g3) hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 2, 3>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_G4R_PASS);
            else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 2>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_G4R_PASS);
        } else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 4>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_G4R_PASS);
#undef Q3_G4R_PASS
        return hipGetLastError();
    }
    if (mg <= 1) {
        if (g3) hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 1, 3>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_LIN_PASS(a));
        else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 1>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_LIN_PASS(a));
    } else if (mg == 2) {
        if (g3) hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 2, 3>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_LIN_PASS(a));
        else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 2>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_LIN_PASS(a));
    } else hipLaunchKernelGGL((k_gemv_mfma4<EPI, RMS, 4>), dim3(tiles), dim3(nwv * 64), 0, st, Q3_LIN_PASS(a));
    return hipGetLastError();
}

hipError_t launch_gemv_tiled4(const LinArgs& a, hipStream_t st) {
    if (a.ksplit != 1) return hipErrorInvalidValue;
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

// ------------------------------------------------------------------------------------------------
// Wide batches (16 < M <= 64): one session carries up to 64 sequences, so a weight tile streamed from HBM once serves
// MT = ceil(M / 16) column tiles. A workgroup = 8 waves owns one 16-row weight tile (two for SwiGLU) and the waves
// split K as in the kernels above; per k-step a wave brings the 16*MT activation rows in with row-contiguous loads
// (one instruction = 8 rows x 128 B), applies the norm weight / accumulates sum(x^2) on that form, parks them in its
// own LDS staging buffer (36-float pitch: the 16 rows of a ds_read_b128 phase land on distinct 4-bank groups) and reads
// the MFMA B operand of every column tile back; the weight tiles of 4 k-steps are requested ahead, the x rows of the
// next k-step are in flight while the current one is multiplied. The launch is bound by the L2 -> CU traffic of x
// (M*K*4 bytes per workgroup against 32*K bytes of weights): ~2.4x the M = 8 time for 8x the rows.
// ------------------------------------------------------------------------------------------------
constexpr int WS = 36;                      // staging pitch in floats (32 k + 4 pad)

template <int EPI, bool RMS, int MT>
__global__ __launch_bounds__(512) void k_gemv_wide(Q3_LIN_PRE, LinArgs a_in) {
    Q3_LIN_APPLY(a, a_in);
    constexpr int NWAVES = 8, G = 4, NX = 2 * MT;           // NX x-load instructions per k-step (8 rows each)
    constexpr int NW = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int ZW = 16 * MT * WS;                        // staging floats per wave
    constexpr int RED = NWAVES * NW * MT * 256, STG = NWAVES * ZW;
    __shared__ __attribute__((aligned(16))) float lds[(RED > STG ? RED : STG) + NWAVES * 16 * MT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int S = a.Kpad >> 5;
    const int s0 = (wave * S) / NWAVES, s1 = ((wave + 1) * S) / NWAVES;
    const size_t tile_base = (size_t)blockIdx.x * S * 64 + lane;
    const u32x4_t* __restrict__ wp = reinterpret_cast<const u32x4_t*>(a.W) + tile_base;
    const u32x4_t* __restrict__ wp2 = NW == 2 ? reinterpret_cast<const u32x4_t*>(a.W2) + tile_base : wp;
    float* __restrict__ zb = lds + wave * ZW;
    float* __restrict__ ssq = lds + (RED > STG ? RED : STG);          // [NWAVES][16*MT], outside the aliased area
    const int xr8 = lane >> 3, xc = (lane & 7) * 4;                    // x loads: row xr8 + 8j, floats xc .. xc+3 of the k-step

    // epilogue operands requested up front: thread (col = tid >> 4 in 0..31, row = tid & 15) serves columns col + 32*j
    constexpr int NE = (MT + 1) / 2;
    float pre_b = 0.0f, pre_r[NE];
    {
        const int n = blockIdx.x * 16 + (tid & 15);
        if (n < a.N && a.bias) pre_b = a.bias[n];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int col = (tid >> 4) + 32 * j;
            pre_r[j] = 0.0f;
            if constexpr (EPI == EPI_RESID) if (col < a.M && n < a.N) pre_r[j] = a.resid[(size_t)col * a.ldr + n];
        }
    }
    f32x4_t acc[NW][MT];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[w][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float ss[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) ss[j] = 0.0f;

    auto xload = [&](float4 (&xv)[NX], float4& nv, int s) {
        const int k0 = s * 32 + xc;
        const bool kok = k0 < a.K;                                   // K % 4 == 0: a float4 is all in or all out
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int row = xr8 + 8 * j;
            xv[j] = (kok && row < a.M) ? *reinterpret_cast<const float4*>(a.x + (size_t)row * a.ldx + k0) : float4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (RMS) nv = kok ? *reinterpret_cast<const float4*>(a.norm_w + k0) : float4{0.f, 0.f, 0.f, 0.f};
    };
    for (int sb = s0; sb < s1; sb += G) {
        u32x4_t wa[G], wb[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = (sb + i) < s1 ? (sb + i) : (s1 - 1);       // ragged last group: duplicate load, never consumed
            wa[i] = Q3_WLOAD(wp + (size_t)s * 64);
            if constexpr (NW == 2) wb[i] = Q3_WLOAD(wp2 + (size_t)s * 64);
        }
        float4 xc0[NX], xc1[NX], n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
        xload(xc0, n0, sb);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = sb + i;
            if (s >= s1) break;                                      // wave-uniform
            float4 (&cur)[NX] = (i & 1) ? xc1 : xc0; float4 (&nxt)[NX] = (i & 1) ? xc0 : xc1;
            float4& ncur = (i & 1) ? n1 : n0; float4& nnxt = (i & 1) ? n0 : n1;
            if (i + 1 < G && s + 1 < s1) xload(nxt, nnxt, s + 1);    // next k-step's rows in flight under this one's MFMAs
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                float4 v = cur[j];
                if constexpr (RMS) {
                    ss[j] = fmaf(v.x, v.x, ss[j]); ss[j] = fmaf(v.y, v.y, ss[j]); ss[j] = fmaf(v.z, v.z, ss[j]); ss[j] = fmaf(v.w, v.w, ss[j]);
                    v.x *= ncur.x; v.y *= ncur.y; v.z *= ncur.z; v.w *= ncur.w;
                }
                *reinterpret_cast<float4*>(zb + (xr8 + 8 * j) * WS + xc) = v;
            }
            // wave-private buffer: the writes above are visible to this wave's reads below in program order (LDS is in-order per wave)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float* zr = zb + (16 * t + m) * WS + kg * 8;
                const float4 b0 = *reinterpret_cast<const float4*>(zr), b1 = *reinterpret_cast<const float4*>(zr + 4);
                const float xv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                const Split3 sp = split3(xv);
                acc[0][t] = mfma3(wa[i], sp, acc[0][t]);
                if constexpr (NW == 2) acc[1][t] = mfma3(wb[i], sp, acc[1][t]);
            }
        }
    }
    zero_job(a.zero, a.zero_n, blockIdx.x, gridDim.x, tid, NWAVES * 64);      // behind the loads of the main loop (vmcnt retires in order)
    if constexpr (RMS) {
        // per-row sum(x^2) of this wave's K slice: the 8 lanes with equal lane >> 3 share a row
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            float v = ss[j];
            v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            if ((lane & 7) == 0) ssq[wave * 16 * MT + xr8 + 8 * j] = v;
        }
    }
    __syncthreads();                                  // every wave is done with its staging buffer: `red` aliases it
    float* __restrict__ red = lds;                    // [NWAVES][NW][MT][256], layout [col m][row]
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int t = 0; t < MT; ++t)
            *reinterpret_cast<f32x4_t*>(&red[((wave * NW + w) * MT + t) * 256 + m * 16 + kg * 4]) = acc[w][t];
    __syncthreads();
    const int row = tid & 15, n = blockIdx.x * 16 + row;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int col = (tid >> 4) + 32 * j;              // batch column 0 .. 16*MT-1
        if (col >= 16 * MT || col >= a.M) continue;
        const int t = col >> 4, idx = (col & 15) * 16 + row;
        float v = 0.0f, v2 = 0.0f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            v += red[((w * NW + 0) * MT + t) * 256 + idx];
            if constexpr (NW == 2) v2 += red[((w * NW + 1) * MT + t) * 256 + idx];
        }
        if constexpr (RMS) {
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) tot += ssq[w * 16 * MT + col];
            const float den = sqrtf(tot / (float)a.K + a.eps);
            v = v / den;
            if constexpr (NW == 2) v2 = v2 / den;
        }
        if (n < a.N) {
            if (a.bias) v = v + pre_b;
            if constexpr (EPI == EPI_RESID) v = pre_r[j] + v;
            if constexpr (EPI == EPI_SILU) v = v / (1.0f + expf(-v));
            if constexpr (EPI == EPI_SWIGLU) v = (v / (1.0f + expf(-v))) * v2;
            a.y[(size_t)col * a.ldy + n] = v;
        }
    }
}

template <int EPI, bool RMS>
static hipError_t launch_gemv_wide_t(const LinArgs& a, hipStream_t st) {
    const int tiles = (a.N + 15) / 16, mt = (a.M + 15) / 16;
    if (mt <= 2) hipLaunchKernelGGL((k_gemv_wide<EPI, RMS, 2>), dim3(tiles), dim3(512), 0, st, Q3_LIN_PASS(a));
    else if (mt == 3) hipLaunchKernelGGL((k_gemv_wide<EPI, RMS, 3>), dim3(tiles), dim3(512), 0, st, Q3_LIN_PASS(a));
    else hipLaunchKernelGGL((k_gemv_wide<EPI, RMS, 4>), dim3(tiles), dim3(512), 0, st, Q3_LIN_PASS(a));
    return hipGetLastError();
}
static hipError_t launch_gemv_wide(const LinArgs& a, hipStream_t st) {
    if (a.tiled != 1 || a.Kpad % 32 != 0 || a.K % 8 != 0 || a.K > a.Kpad || a.ldx % 4 != 0 || a.M < 17 || a.M > 64 || a.N < 1) return hipErrorInvalidValue;
    const bool rms = a.norm_w != nullptr;
    switch (a.epi) {
        case EPI_NONE: return rms ? launch_gemv_wide_t<EPI_NONE, true>(a, st) : launch_gemv_wide_t<EPI_NONE, false>(a, st);
        case EPI_RESID: return rms ? hipErrorInvalidValue : launch_gemv_wide_t<EPI_RESID, false>(a, st);
        case EPI_SILU: return rms ? hipErrorInvalidValue : launch_gemv_wide_t<EPI_SILU, false>(a, st);
        case EPI_SWIGLU: return rms ? launch_gemv_wide_t<EPI_SWIGLU, true>(a, st) : launch_gemv_wide_t<EPI_SWIGLU, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemv_tiled(const LinArgs& a, hipStream_t st) {
    if (a.M > 16) {
        if (a.ksplit == 2) return launch_gemv_sk2(a, st);
        if (a.ksplit > 1) return hipErrorInvalidValue;
        static const bool no_gemm = getenv("Q3_WIDE_NO_GEMM") != nullptr;      // A/B aid: wide sessions on k_gemv_wide
        if (a.ws && !no_gemm) { const hipError_t e = launch_gemm_wide(a, st); if (e != hipErrorNotSupported) return e; }
        return launch_gemv_wide(a, st);
    }
    if (a.ksplit == 2) return launch_gemv_sk2(a, st);
    if (a.ksplit != 1) return hipErrorInvalidValue;
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
