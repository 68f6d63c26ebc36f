// Grid-barrier / persistent-stage probe on gfx950 (development aid): what does ONE dependent stage of
// the code predictor cost when the stages live inside one persistent launch instead of one kernel each?
//   A  flat device-scope counter barrier (release fence -> arrive -> relaxed sc1 poll -> acquire fence), us per barrier
//   B  XCD-hierarchical barrier (per-XCC counter, XCD leader -> top counter -> per-XCC generation word)
//   C  stage emulation, persistent: every workgroup streams its share of the stage's weights (nt, requested BEFORE the
//      barrier of the previous stage: the prefetch a kernel boundary cannot do), reads the whole activation vector
//      (32 KB = 8 rows x 1024 f32, written by all workgroups in the previous stage) with sc1 loads, reduces, publishes
//      its 128-byte slice with write-through (sc1) stores, drains, arrives.  us per stage.
//   D  the same stage as ONE KERNEL PER STAGE replayed from a hipGraph (what the engine does today).
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned SPIN_LIMIT = 1u << 20;      // every spin is bounded (one budget per kernel and polling lane): a stranded block ends the kernel with a code

struct Bar {                                   // every polled word in its own 128-byte line
    unsigned flat; unsigned pad0[31];
    unsigned top; unsigned pad1[31];
    unsigned xc[8][32];                        // per-XCC arrival counters
    unsigned gen[8][32];                       // per-XCC generation words
    unsigned census[8][32];                    // workgroups per XCC (filled before the first hierarchical barrier)
    unsigned fail; unsigned pad2[31];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u;
}
__shared__ unsigned spent;                      // spins used by this block's polling lane over the whole kernel (zeroed at kernel start)
__device__ __forceinline__ bool poll_ge(unsigned* p, unsigned target, Bar* b) {
    if (spent > SPIN_LIMIT) return false;
    unsigned spins = spent;
    while (__hip_atomic_load(p, RLX_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { __hip_atomic_store(&b->fail, 1u, RLX_AGENT); spent = spins; return false; }
    }
    spent = spins;
    return true;
}
// FENCES: 0 = none (payload is sc1 both sides), 1 = release before arrive + acquire after the poll (plain payload)
template <int FENCES>
__device__ __forceinline__ void bar_flat(Bar* b, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __hip_atomic_fetch_add(&b->flat, 1u, RLX_AGENT);
        poll_ge(&b->flat, epoch * gridDim.x, b);
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
template <int FENCES>
__device__ __forceinline__ void bar_xcd(Bar* b, unsigned epoch, unsigned x, unsigned n_in_xcd, unsigned n_xcd) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned old = __hip_atomic_fetch_add(&b->xc[x][0], 1u, RLX_AGENT);
        if (old + 1 == epoch * n_in_xcd) {                          // last arriver of this XCD: represent it at the top
            __hip_atomic_fetch_add(&b->top, 1u, RLX_AGENT);
            poll_ge(&b->top, epoch * n_xcd, b);
            __hip_atomic_store(&b->gen[x][0], epoch, RLX_AGENT);
        } else {
            poll_ge(&b->gen[x][0], epoch, b);
        }
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int KIND, int FENCES>
__global__ __launch_bounds__(256) void k_barriers(Bar* b, int n, unsigned* sink) {
    if (threadIdx.x == 0) spent = 0;
    unsigned x = 0, nin = 0, nx = 0;
    if (KIND == 1) {
        x = xcc_id();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&b->census[x][0], 1u, RLX_AGENT);
        bar_flat<0>(b, 1);
        nin = __hip_atomic_load(&b->census[x][0], RLX_AGENT);
        for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(&b->census[i][0], RLX_AGENT) ? 1u : 0u;
    }
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) bar_flat<FENCES>(b, (unsigned)i + 1 + 0);
        else bar_xcd<FENCES>(b, (unsigned)i + 1, x, nin, nx);
    }
    if (sink && threadIdx.x == 9999) sink[0] = 1;
}

// ---- stage emulation -------------------------------------------------------------------------------------
constexpr int XN = 8 * 1024;                  // activation floats per stage (8 rows x 1024)
// 16-byte sc1 (device-coherent) load through a buffer descriptor: compiler-counted, so the four loads of a lane are in
// flight together (aux 16 = sc1)
__device__ __forceinline__ f32x4_t ld_sc1(__amdgpu_buffer_rsrc_t r, int float_off) {
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 16));
}
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, RLX_AGENT); }

// NLW = 16-byte weight loads per lane per stage (512 threads: NLW x 8 KB per workgroup)
template <int NLW, int KIND>
__global__ __launch_bounds__(512) void k_persist(const u32x4_t* __restrict__ w, size_t stage_vec, int w_slots, float* xring, Bar* b,
                                                 int stages, float* out) {
    __shared__ float red[8];
    const int tid = threadIdx.x, wave = tid >> 6, nwg = gridDim.x;
    if (tid == 0) spent = 0;
    unsigned x = 0, nin = 0, nx = 0;
    if (KIND == 1) {
        x = xcc_id();
        if (tid == 0) __hip_atomic_fetch_add(&b->census[x][0], 1u, RLX_AGENT);
        bar_flat<0>(b, 1);
        nin = __hip_atomic_load(&b->census[x][0], RLX_AGENT);
        for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(&b->census[i][0], RLX_AGENT) ? 1u : 0u;
    }
    const int slice = XN / nwg;               // floats this workgroup publishes per stage
    u32x4_t wv[NLW];
    auto wload = [&](int s) {
        const u32x4_t* p = w + (size_t)(s % w_slots) * stage_vec + (size_t)blockIdx.x * NLW * 512 + tid;
#pragma unroll
        for (int i = 0; i < NLW; ++i) wv[i] = __builtin_nontemporal_load(p + (size_t)i * 512);
    };
    wload(0);
    float carry = 0.f;
    for (int s = 0; s < stages; ++s) {
        // the whole activation vector of this stage: 2048 float4, 4 per lane, sc1 (written by other CUs / XCDs)
        const __amdgpu_buffer_rsrc_t xin = __builtin_amdgcn_make_buffer_rsrc((void*)(xring + (size_t)(s & 1) * XN), 0, XN * 4, 0x00020000);
        float acc = carry;
        f32x4_t xv[XN / 4 / 512];
#pragma unroll
        for (int i = 0; i < XN / 4 / 512; ++i) xv[i] = ld_sc1(xin, (i * 512 + tid) * 4);
#pragma unroll
        for (int i = 0; i < XN / 4 / 512; ++i) acc += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
        unsigned h = 0;
#pragma unroll
        for (int i = 0; i < NLW; ++i) h ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
        acc += (h == 0x12345u) ? 1.f : 0.f;
        // wave reduce + cross-wave through LDS (the GEMV's K-split reduction)
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        if ((tid & 63) == 0) red[wave] = acc;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += red[i];
        carry = tot * 1e-30f;
        float* xout = xring + (size_t)((s + 1) & 1) * XN + (size_t)blockIdx.x * slice;
        if (wave == 0) {                      // ONE wave publishes (write-through) and drains before the arrive
            if (tid < slice) st_sc1(xout + tid, tot * 1e-30f + 1.0f);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (s + 1 < stages) {
            wload(s + 1);                     // the other waves request the NEXT stage's weights before the barrier
        }
        if (KIND == 0) bar_flat<0>(b, (unsigned)s + 1 + (KIND == 1 ? 1u : 0u));
        else bar_xcd<0>(b, (unsigned)s + 1, x, nin, nx);
        if (wave == 0 && s + 1 < stages) wload(s + 1);
    }
    if (tid == 0) out[blockIdx.x] = carry;
}

// the same stage as its own kernel (plain loads / stores; the kernel boundary is the barrier)
// ROT: every workgroup starts its walk over the shared activation vector at a different offset (L2 channel hot-spot probe)
template <int NLW, int ROT>
__global__ __launch_bounds__(512) void k_stage(const u32x4_t* __restrict__ w, const float* __restrict__ xin, float* xout, int nwg_slice) {
    __shared__ float red[8];
    const int tid = threadIdx.x, wave = tid >> 6;
    u32x4_t wv[NLW];
    const u32x4_t* p = w + (size_t)blockIdx.x * NLW * 512 + tid;
#pragma unroll
    for (int i = 0; i < NLW; ++i) wv[i] = __builtin_nontemporal_load(p + (size_t)i * 512);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < XN / 4 / 512; ++i) {
        const int slot = ROT ? ((i * 512 + tid + (int)blockIdx.x * 67) & (XN / 4 - 1)) : (i * 512 + tid);
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(xin + slot * 4);
        acc += v[0] + v[1] + v[2] + v[3];
    }
    unsigned h = 0;
#pragma unroll
    for (int i = 0; i < NLW; ++i) h ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
    acc += (h == 0x12345u) ? 1.f : 0.f;
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[wave] = acc;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += red[i];
    if (tid < nwg_slice) xout[(size_t)blockIdx.x * nwg_slice + tid] = tot * 1e-30f + 1.0f;
}

template <class F> static int timed(const char* name, int units, int reps, hipStream_t st, F run, Bar* bar) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(bar, 0, sizeof(Bar), st)); run(); CK(hipStreamSynchronize(st));
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
        CK(hipEventRecord(e0, st)); run(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; if (ms < best) best = ms;
    }
    Bar h; CK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    printf("%-64s mean %7.2f  best %7.2f us/unit%s\n", name, sum / reps * 1e3 / units, best * 1e3 / units, h.fail ? "  [SPIN LIMIT HIT]" : "");
    CK(hipGetLastError());
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    Bar* bar; CK(hipMalloc(&bar, sizeof(Bar)));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    const int NB = 200, REPS = 10;
    for (int wgs : {64, 128, 256, 512}) {
        char nm[128];
        snprintf(nm, sizeof nm, "A flat counter, no fences          %4d WGs x 256", wgs);
        if (timed(nm, NB, REPS, st, [&] { hipLaunchKernelGGL((k_barriers<0, 0>), dim3(wgs), dim3(256), 0, st, bar, NB, sink); }, bar)) return 1;
        snprintf(nm, sizeof nm, "A flat counter, release+acquire    %4d WGs x 256", wgs);
        if (timed(nm, NB, REPS, st, [&] { hipLaunchKernelGGL((k_barriers<0, 1>), dim3(wgs), dim3(256), 0, st, bar, NB, sink); }, bar)) return 1;
        snprintf(nm, sizeof nm, "B XCD-hierarchical, no fences      %4d WGs x 256", wgs);
        if (timed(nm, NB, REPS, st, [&] { hipLaunchKernelGGL((k_barriers<1, 0>), dim3(wgs), dim3(256), 0, st, bar, NB, sink); }, bar)) return 1;
        snprintf(nm, sizeof nm, "B XCD-hierarchical, release+acquire %4d WGs x 256", wgs);
        if (timed(nm, NB, REPS, st, [&] { hipLaunchKernelGGL((k_barriers<1, 1>), dim3(wgs), dim3(256), 0, st, bar, NB, sink); }, bar)) return 1;
    }
    // stage emulation: 256 workgroups x 512 threads, NLW x 8 KB of weights per workgroup per stage
    const size_t W_BYTES = (size_t)1 << 30;
    u32x4_t* w; float *xring, *out;
    CK(hipMalloc(&w, W_BYTES)); CK(hipMemset(w, 1, W_BYTES)); CK(hipMalloc(&xring, 2 * XN * 4)); CK(hipMemset(xring, 0, 2 * XN * 4)); CK(hipMalloc(&out, 4096));
    const int STAGES = 200, WGS = 256;
#define STAGE_CASE(NLW)                                                                                                                   \
    {                                                                                                                                      \
        const size_t stage_vec = (size_t)WGS * NLW * 512; const int slots = (int)(W_BYTES / 16 / stage_vec); char nm[128];                \
        snprintf(nm, sizeof nm, "C persistent stage, flat barrier   %5.1f MB weights/stage", stage_vec * 16 / 1e6);                       \
        if (timed(nm, STAGES, REPS, st, [&] { hipLaunchKernelGGL((k_persist<NLW, 0>), dim3(WGS), dim3(512), 0, st, w, stage_vec, slots, xring, bar, STAGES, out); }, bar)) return 1; \
        snprintf(nm, sizeof nm, "C persistent stage, XCD barrier    %5.1f MB weights/stage", stage_vec * 16 / 1e6);                       \
        if (timed(nm, STAGES, REPS, st, [&] { hipLaunchKernelGGL((k_persist<NLW, 1>), dim3(WGS), dim3(512), 0, st, w, stage_vec, slots, xring, bar, STAGES, out); }, bar)) return 1; \
        for (int rot = 0; rot < 2; ++rot) {                                                                                                \
        hipGraph_t g; hipGraphExec_t ge;                                                                                                   \
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));                                                                         \
        for (int s = 0; s < STAGES; ++s) {                                                                                                 \
            if (rot) hipLaunchKernelGGL((k_stage<NLW, 1>), dim3(WGS), dim3(512), 0, st, w + (size_t)(s % slots) * stage_vec, xring + (size_t)(s & 1) * XN, xring + (size_t)((s + 1) & 1) * XN, XN / WGS); \
            else hipLaunchKernelGGL((k_stage<NLW, 0>), dim3(WGS), dim3(512), 0, st, w + (size_t)(s % slots) * stage_vec, xring + (size_t)(s & 1) * XN, xring + (size_t)((s + 1) & 1) * XN, XN / WGS); \
        }                                                                                                                                  \
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));                                             \
        snprintf(nm, sizeof nm, "D one kernel per stage (hipGraph)%s %5.1f MB weights/stage", rot ? ", rotated x" : "            ", stage_vec * 16 / 1e6); \
        if (timed(nm, STAGES, REPS, st, [&] { (void)hipGraphLaunch(ge, st); }, bar)) return 1;                                             \
        hipGraphExecDestroy(ge); hipGraphDestroy(g);                                                                                       \
        }                                                                                                                                  \
    }
    STAGE_CASE(1) STAGE_CASE(2) STAGE_CASE(4) STAGE_CASE(6) STAGE_CASE(8)
    return 0;
}
