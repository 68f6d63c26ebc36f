// XCD-local exchange probe on gfx950 (development aid, round 5): what does a producer -> consumer hand-over cost INSIDE one
// kernel when all the workgroups involved sit on the same XCD and therefore share one L2?
// Motivation: a dependent edge of the frame costs ~1.1 us of launch gap + ~1.3 us until the consumer's first loads of the
// producer's output land (the data crosses XCDs through memory) + ~1.3 us store -> visible. The q|k|v projection -> attention
// edge is special: attention of KV group g needs only the 512 rows (2 q heads, k, v) of group g — 32 sixteen-row tiles = the
// 32 CUs of ONE XCD, and there are 8 KV groups = 8 XCDs. If workgroup b runs on XCD b % 8 (the dispatcher's round-robin the
// vocoder's tile order already leans on for speed), a fused kernel could hand q|k|v to the attention through the XCD's own L2.
//   L  XCD-local: group = blockIdx % 8; plain stores + s_waitcnt vmcnt(0) (a store is acknowledged by the L2), arrival counter
//      bumped AND polled with non-sc1 atomics (they execute in that L2; an sc0 load may hit the CU's own L1 — the first cut of
//      this probe polled with sc0 loads and spun on a stale line for ever), data read with PLAIN loads from addresses this CU has
//      not read before in this kernel (every round publishes into a fresh region, as a layer's q|k|v are fresh addresses for
//      the attention that follows: nothing stale can sit in the CU's L1, which the dispatch invalidated)
//   A  the same grouping with agent-scope everything (sc1 write-through stores, sc1 atomics / polls / loads): what the hand-over
//      costs when it must be correct whatever XCD a workgroup runs on
//   X  groups of 32 CONSECUTIVE workgroups (every group spread over all 8 XCDs), agent scope: a cross-XCD hand-over inside a kernel
//   W  groups of 32 consecutive workgroups with the XCD-local operations: WRONG on purpose — shows that locality is what makes L work
// Every round is verified (each consumer checks the 4096 values its group published); every spin is bounded. Also prints the
// XCC_ID census: which XCD each blockIdx really ran on.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_local.hip -o xcd_local
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned SPIN_LIMIT = 1u << 16;
constexpr int WGS = 256, NG = 8, PER = 32, SLICE = 128;           // 8 groups of 32 workgroups, 128 floats published per workgroup
constexpr int RING = 1024;                                        // regions (rounds <= RING: no address is reused inside a launch)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u;
}

struct Ctl {
    unsigned cnt[NG][32];           // arrival counter per group, one 128-byte line each
    unsigned fail, bad, pad[30];
    unsigned xcc[WGS];
    unsigned long long t_first[WGS], t_last[WGS];
};

// LOCAL: sc0 loads / plain stores / L2 atomics; else agent scope (sc1).  SPREAD: group = blockIdx / 32 instead of blockIdx % 8
template <bool LOCAL, bool SPREAD>
__global__ __launch_bounds__(512) void k_rounds(float* buf, Ctl* c, int rounds) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const int g = SPREAD ? b / PER : b % NG, t = SPREAD ? b % PER : b / NG;
    if (tid == 0) { c->xcc[b] = xcc_id(); c->t_first[b] = __builtin_readcyclecounter(); }
    __shared__ int ok;
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)c, 0, 0x7fffffff, 0x00020000);
    unsigned bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        float* slot = buf + ((size_t)(r % RING) * NG + g) * PER * SLICE;    // this group's 4096 floats, a fresh region every round
        if (tid < SLICE) {
            const float v = (float)(r * 1024 + t * 4 + (tid & 3));
            if (LOCAL) slot[t * SLICE + tid] = v;
            else __hip_atomic_store(slot + t * SLICE + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (LOCAL) __hip_atomic_fetch_add(&c->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&c->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int good = 1; unsigned spins = 0;
            const unsigned want = (unsigned)(r * PER);
            const int off = (int)((char*)&c->cnt[g][0] - (char*)c);
            for (;;) {
                (void)off; (void)cr;
                unsigned v;
                if (LOCAL) {      // a returning atomic add of zero, written as asm: LLVM turns an idempotent atomicrmw into a LOAD, which may hit the L1
                    unsigned* p = &c->cnt[g][0]; const unsigned zero = 0;
                    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(zero) : "memory");
                } else v = __builtin_amdgcn_raw_buffer_load_b32(cr, off, 0, 16);
                if (v >= want) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { good = 0; break; }
            }
            ok = good;
        }
        __syncthreads();
        if (!ok) { if (tid == 0) __hip_atomic_store(&c->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        // consume: the group's 4096 floats, 8 per thread
        const int base = (int)((size_t)((r % RING) * NG + g) * PER * SLICE * 4);
        const f32x4_t v0 = __builtin_bit_cast(f32x4_t, LOCAL ? __builtin_amdgcn_raw_buffer_load_b128(br, base + tid * 16, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(br, base + tid * 16, 0, 16));
        const f32x4_t v1 = __builtin_bit_cast(f32x4_t, LOCAL ? __builtin_amdgcn_raw_buffer_load_b128(br, base + (512 + tid) * 16, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(br, base + (512 + tid) * 16, 0, 16));
        const int w0 = (tid * 4) / SLICE, w1 = ((512 + tid) * 4) / SLICE;         // writer workgroups (index inside the group)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bad += v0[e] != (float)(r * 1024 + w0 * 4 + e);
            bad += v1[e] != (float)(r * 1024 + w1 * 4 + e);
        }
    }
    if (bad) atomicAdd(&c->bad, bad);
    if (tid == 0) c->t_last[b] = __builtin_readcyclecounter();
}

int main() {
    float* buf; Ctl* c;
    const size_t BUF = (size_t)RING * NG * PER * SLICE * 4;
    CK(hipMalloc(&buf, BUF)); CK(hipMalloc(&c, sizeof(Ctl)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 1000;
    struct V { const char* name; int kind; };
    const V vs[] = {{"L  XCD-local (group = blockIdx % 8): plain st, L2 atomic, sc0 ld        ", 0},
                    {"A  same groups, agent scope: sc1 st / atomic / ld                       ", 1},
                    {"X  groups of 32 consecutive workgroups (8 XCDs each), agent scope       ", 2},
                    {"W  groups of 32 consecutive workgroups with XCD-local operations (WRONG) ", 3}};
    for (const V& v : vs) {
        double best = 1e30; unsigned fail = 0, bad = 0;
        std::vector<Ctl> h(1);
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(buf, 0, BUF)); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (v.kind == 0) hipLaunchKernelGGL((k_rounds<true, false>), dim3(WGS), dim3(512), 0, 0, buf, c, R);
            else if (v.kind == 1) hipLaunchKernelGGL((k_rounds<false, false>), dim3(WGS), dim3(512), 0, 0, buf, c, R);
            else if (v.kind == 2) hipLaunchKernelGGL((k_rounds<false, true>), dim3(WGS), dim3(512), 0, 0, buf, c, R);
            else hipLaunchKernelGGL((k_rounds<true, true>), dim3(WGS), dim3(512), 0, 0, buf, c, R);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), c, sizeof(Ctl), hipMemcpyDeviceToHost));
            fail |= h[0].fail; bad += h[0].bad;
            if (ms * 1e3 / R < best) best = ms * 1e3 / R;
        }
        printf("%s %6.3f us per hand-over round%s%s\n", v.name, best, fail ? "   [SPIN LIMIT HIT]" : "", bad ? "   [STALE / WRONG VALUES READ]" : "");
        if (v.kind == 0) {
            int match = 0; int per_xcc[16] = {0};
            for (int b = 0; b < WGS; ++b) { match += (int)h[0].xcc[b] == b % NG; per_xcc[h[0].xcc[b] & 15]++; }
            printf("#  XCC_ID census: %d of %d workgroups ran on XCD blockIdx %% 8; workgroups per XCD:", match, WGS);
            for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
            printf("\n");
        }
        fflush(stdout);
    }
    return 0;
}
