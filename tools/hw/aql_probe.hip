// AQL chain probe on gfx950 (development aid): what does ONE dependent stage of the code predictor cost
// when the packets of the chain are written by us into our own HSA queue instead of being replayed by hipGraph?
//   D   one kernel per stage from a hipGraph (what the engine does today; plain loads / stores)
//   Q1  own AQL queue, barrier bit set, acquire/release fences = AGENT (should equal D: it is what HIP writes)
//   Q2  own AQL queue, barrier bit set, fences = NONE; the kernels carry the coherence themselves: x read with sc1 loads, y
//       published with write-through (sc1) stores + vmcnt(0) before the wave ends
//   Q3  own AQL queue, barrier bit CLEAR: stage s+1 is dispatched while stage s runs, requests its weights at once, and only
//       its x read waits — one lane per workgroup polls the arrival counter of stage s (relaxed sc1 load + s_sleep), producers
//       publish y write-through, drain, and arrive.  In-order dispatch (one queue) places every producer before its consumer.
//   Q3b the same with the arrival counter sharded 8 / 32 ways (blockIdx % NS; lanes 0..NS-1 of the polling wave watch one each)
//   Q3c no atomics: every producer workgroup stores its own flag word (sc1) after draining; the consumer's first wave polls the
//       1 KB flag array with one 16-byte sc1 load per lane
//   Q3d the data is the flag: x travels as 8-byte {tag = stage, value} granules (one sc1 store each, no drain, no flag); every
//       consumer thread re-reads its sixteen granules until all tags match (one sentinel granule polled first)
//   Q5  TWO AQL queues, stage s in queue s % 2 (one queue never overlaps its own packets — see the overlap check printed first —
//       so the run-ahead comes from the second queue): stage s+1 starts when stage s-1 ends and waits for stage s by polling
//   hipExtLaunchKernel(..., hipExtAnyOrderLaunch) is documented as unsupported on gfx9 (hip_ext.h:67), hence HSA.
// Every spin is bounded; every chain is verified (x carries the stage index, so ONE stale 16-byte read anywhere changes the
// final value).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=15 aql_probe.hip -o aql_probe -lhsa-runtime64
//        hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=15 --cuda-device-only --no-gpu-bundle-output -c aql_probe.hip -o aql_probe.hsaco
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int XN = 8 * 1024;                  // activation floats per stage (8 rows x 1024)
constexpr unsigned SPIN_LIMIT = 1u << 18;

struct StageArgs {                            // one 64-byte kernarg block per stage (no hidden arguments are used)
    const u32x4_t* w;                         // this stage's weights
    const float* xin;                         // XN floats
    float* xout;                              // XN floats
    unsigned* cnt_prev;                       // arrival counter of the previous stage (Q3)
    unsigned* cnt_mine;                       // arrival counter of this stage (Q3)
    unsigned* fail;                           // spin-limit flag
    unsigned target_prev;                     // workgroups of the previous stage (0 = first stage: do not wait)
    int slice;                                // floats this workgroup publishes
    unsigned pad[2];
};

__device__ __forceinline__ f32x4_t ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// MODE 0: plain x loads / y stores (the kernel boundary's fences do the coherence)
// MODE 1: sc1 loads / write-through stores + drain (no fence needed at the boundary)
// MODE 2: MODE 1 + wait for the previous stage's arrivals before reading x, arrive after publishing
// MODE 3: MODE 2 with `target_prev >> 16` counter shards (128 bytes apart), low 16 bits = arrivals per shard
// MODE 4: MODE 2 with one flag word per producer workgroup instead of a counter (target_prev = epoch, 256 producers)
// MODE 5: granule transport (xin / xout = 2 * XN granules; target_prev = expected tag, 0 for the first stage)
typedef __attribute__((ext_vector_type(4))) unsigned int gr2_t;       // two granules: {tag, value, tag, value}
template <int NLW>
__device__ __forceinline__ void stage_granules(const u32x4_t* w, const float* xin, float* xout, unsigned* fail, unsigned tag_prev, int slice) {
    __shared__ float red[8];
    const int tid = threadIdx.x, wave = tid >> 6;
    u32x4_t wv[NLW];
    const u32x4_t* p = w + (size_t)blockIdx.x * NLW * 512 + tid;
#pragma unroll
    for (int i = 0; i < NLW; ++i) wv[i] = __builtin_nontemporal_load(p + (size_t)i * 512);
    constexpr int NG = XN / 2 / 512;                              // 16-byte granule pairs per thread (8)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xin, 0, XN * 8, 0x00020000);
    gr2_t g[NG];
    if (tag_prev) {                                               // sentinel: the granule pair this thread reads last
        unsigned spins = 0;
        for (;;) {
            const gr2_t v = __builtin_bit_cast(gr2_t, __builtin_amdgcn_raw_buffer_load_b128(xr, ((NG - 1) * 512 + tid) * 16, 0, 16));
            if (__all(v[0] == tag_prev && v[2] == tag_prev)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { if (tid == 0) __hip_atomic_store(fail, 1u, RLX_AGENT); break; }
        }
    }
    for (unsigned spins = 0;;) {
#pragma unroll
        for (int i = 0; i < NG; ++i) g[i] = __builtin_bit_cast(gr2_t, __builtin_amdgcn_raw_buffer_load_b128(xr, (i * 512 + tid) * 16, 0, 16));
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NG; ++i) ok &= (g[i][0] == tag_prev) & (g[i][2] == tag_prev);
        if (__all(ok) || !tag_prev) break;
        asm volatile("" ::: "memory");                            // the loads must be re-issued: nothing else in this loop touches memory
        if (++spins > SPIN_LIMIT) { if (tid == 0) __hip_atomic_store(fail, 1u, RLX_AGENT); break; }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) { const unsigned v1 = g[i][1], v3 = g[i][3]; acc += __uint_as_float(v1) + __uint_as_float(v3); }   // (bit_cast of a vector ELEMENT reads element 0)
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
    const float y = tot * (1.0f / XN) + 1.0f;
    if (tid < slice) {
        unsigned long long* go = reinterpret_cast<unsigned long long*>(xout) + (size_t)blockIdx.x * slice + tid;
        __hip_atomic_store(go, ((unsigned long long)__builtin_bit_cast(unsigned, y) << 32) | (unsigned long long)(tag_prev + 1), RLX_AGENT);
    }
}
template <int NLW, int MODE>
__device__ __forceinline__ void stage_body(const u32x4_t* w, const float* xin, float* xout, unsigned* cnt_prev, unsigned* cnt_mine, unsigned* fail,
                                           unsigned target_prev, int slice, unsigned aux) {
    StageArgs a; a.pad[0] = aux; a.w = w; a.xin = xin; a.xout = xout; a.cnt_prev = cnt_prev; a.cnt_mine = cnt_mine; a.fail = fail; a.target_prev = target_prev; a.slice = slice;
    __shared__ float red[8];
    __shared__ int ok;
    const int tid = threadIdx.x, wave = tid >> 6;
    u32x4_t wv[NLW];
    const u32x4_t* p = a.w + (size_t)blockIdx.x * NLW * 512 + tid;
#pragma unroll
    for (int i = 0; i < NLW; ++i) wv[i] = __builtin_nontemporal_load(p + (size_t)i * 512);
    if (MODE == 2) {
        if (tid == 0) {
            int good = 1;
            if (a.target_prev) {
                unsigned spins = 0;
                while (__hip_atomic_load(a.cnt_prev, RLX_AGENT) < a.target_prev) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { __hip_atomic_store(a.fail, 1u, RLX_AGENT); good = 0; break; }
                }
            }
            ok = good;
        }
        __syncthreads();
    }
    if (MODE == 3) {
        if (wave == 0 && a.target_prev) {
            const unsigned ns = a.target_prev >> 16, per = a.target_prev & 0xffffu;
            unsigned spins = 0;
            for (;;) {
                const unsigned v = (unsigned)tid < ns ? __hip_atomic_load(a.cnt_prev + tid * 32, RLX_AGENT) : per;
                if (__all(v >= per)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { if (tid == 0) __hip_atomic_store(a.fail, 1u, RLX_AGENT); break; }
            }
        }
        __syncthreads();
    }
    if (MODE == 4) {
        if (wave == 0 && a.target_prev) {
            const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)a.cnt_prev, 0, 1024, 0x00020000);
            unsigned spins = 0;
            for (;;) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(fr, tid * 16, 0, 16);
                if (__all(v[0] == a.target_prev && v[1] == a.target_prev && v[2] == a.target_prev && v[3] == a.target_prev)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { if (tid == 0) __hip_atomic_store(a.fail, 1u, RLX_AGENT); break; }
            }
        }
        __syncthreads();
    }
    float acc = 0.f;
    f32x4_t xv[XN / 4 / 512];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < XN / 4 / 512; ++i) xv[i] = *reinterpret_cast<const f32x4_t*>(a.xin + (i * 512 + tid) * 4);
    } else {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xin, 0, XN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < XN / 4 / 512; ++i) xv[i] = ld_sc1(xr, (i * 512 + tid) * 16);
    }
#pragma unroll
    for (int i = 0; i < XN / 4 / 512; ++i) acc += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
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
    const float y = tot * (1.0f / XN) + 1.0f;                     // x == s everywhere  ->  y == s + 1 exactly
    float* xo = a.xout + (size_t)blockIdx.x * a.slice;
    if (MODE == 0) {
        if (tid < a.slice) xo[tid] = y;
    } else {
        if (wave == 0) {
            if (tid < a.slice) __hip_atomic_store(xo + tid, y, RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 2 && tid == 0) __hip_atomic_fetch_add(a.cnt_mine, 1u, RLX_AGENT);
            if (MODE == 3 && tid == 0) __hip_atomic_fetch_add(a.cnt_mine + (blockIdx.x % (unsigned)a.pad[0]) * 32, 1u, RLX_AGENT);
            if (MODE == 4 && tid == 0) __hip_atomic_store(a.cnt_mine + blockIdx.x, (unsigned)a.pad[0], RLX_AGENT);
        }
    }
}

// scalar arguments (15 dwords) so that the kernel-argument preload covers them all, as in the product kernels
#define STAGE_ARGS const u32x4_t* w, const float* xin, float* xout, unsigned* cnt_prev, unsigned* cnt_mine, unsigned* fail, unsigned target_prev, int slice, unsigned aux
#define STAGE_PASS w, xin, xout, cnt_prev, cnt_mine, fail, target_prev, slice, aux
#define STAGE_KERNELS(NLW)                                                                                                     \
    extern "C" __global__ __launch_bounds__(512) void k_plain_##NLW(STAGE_ARGS) { stage_body<NLW, 0>(STAGE_PASS); }         \
    extern "C" __global__ __launch_bounds__(512) void k_sc1_##NLW(STAGE_ARGS) { stage_body<NLW, 1>(STAGE_PASS); }           \
    extern "C" __global__ __launch_bounds__(512) void k_poll_##NLW(STAGE_ARGS) { stage_body<NLW, 2>(STAGE_PASS); }           \
    extern "C" __global__ __launch_bounds__(512) void k_shard_##NLW(STAGE_ARGS) { stage_body<NLW, 3>(STAGE_PASS); }          \
    extern "C" __global__ __launch_bounds__(512) void k_flags_##NLW(STAGE_ARGS) { stage_body<NLW, 4>(STAGE_PASS); }          \
    extern "C" __global__ __launch_bounds__(512) void k_gran_##NLW(STAGE_ARGS) { stage_granules<NLW>(w, xin, xout, fail, target_prev, slice); }
STAGE_KERNELS(1) STAGE_KERNELS(2) STAGE_KERNELS(4) STAGE_KERNELS(6) STAGE_KERNELS(8)

// overlap check: k_wait (dispatched FIRST) can only finish if k_set (dispatched SECOND, barrier bit clear) runs beside it
extern "C" __global__ __launch_bounds__(512) void k_wait(unsigned* flag, unsigned* fail, unsigned* spins_out) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, RLX_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(fail, 1u, RLX_AGENT); break; }
        }
        if (blockIdx.x == 0) spins_out[0] = spins;
    }
}
extern "C" __global__ __launch_bounds__(512) void k_set(unsigned* flag, unsigned* fail, unsigned* spins_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, RLX_AGENT);
}

#ifndef __HIP_DEVICE_COMPILE__
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define HK(x) do { hsa_status_t e_ = (x); if (e_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(e_, &m); printf("%s: %s (line %d)\n", #x, m, __LINE__); return 1; } } while (0)

static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_status_t pick_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
    return HSA_STATUS_SUCCESS;
}
struct KSym { uint64_t object; uint32_t kernarg, group, priv; };

static int get_kernel(hsa_executable_t ex, const char* name, KSym* k) {
    hsa_executable_symbol_t s; std::string n = std::string(name) + ".kd";
    HK(hsa_executable_get_symbol_by_name(ex, n.c_str(), &g_gpu, &s));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k->object));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k->kernarg));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k->group));
    HK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k->priv));
    return 0;
}

// writes n dispatch packets — stage i into queue i % nq — and rings each doorbell once; waits for the last packet of every queue
static int run_chain(hsa_queue_t** qs, int nq, hsa_signal_t* dones, const KSym& k, const char* kargs_dev, int n, int wgs, bool barrier, int fence, double* us,
                     uint32_t lds_pad = 0, int acq_fence = -1) {      // acq_fence >= 0: acquire scope set apart from the release scope (`fence`)
    uint64_t base[4], cnt[4] = {0, 0, 0, 0};
    for (int j = 0; j < nq; ++j) {
        const uint64_t mine = (uint64_t)((n - j + nq - 1) / nq);
        hsa_signal_store_relaxed(dones[j], 1);
        base[j] = hsa_queue_add_write_index_relaxed(qs[j], mine);
        while (base[j] + mine - hsa_queue_load_read_index_scacquire(qs[j]) > qs[j]->size) { }
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        const int j = i % nq; hsa_queue_t* q = qs[j];
        hsa_kernel_dispatch_packet_t pk; memset(&pk, 0, sizeof pk);
        const bool first = i < nq, last = i >= n - nq;
        // the first packet acquires at system scope (the host's memsets), the last releases at system scope (the host reads)
        const int acq = first ? HSA_FENCE_SCOPE_SYSTEM : (acq_fence >= 0 ? acq_fence : fence), rel = last ? HSA_FENCE_SCOPE_SYSTEM : fence;
        pk.setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        pk.workgroup_size_x = 512; pk.workgroup_size_y = 1; pk.workgroup_size_z = 1;
        pk.grid_size_x = (uint32_t)wgs * 512; pk.grid_size_y = 1; pk.grid_size_z = 1;
        pk.private_segment_size = k.priv; pk.group_segment_size = k.group > lds_pad ? k.group : lds_pad;
        pk.kernel_object = k.object; pk.kernarg_address = (void*)(kargs_dev + (size_t)i * 64);
        pk.completion_signal = last ? dones[j] : hsa_signal_t{0};
        const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier || last || first ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                        (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        hsa_kernel_dispatch_packet_t* dst = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base[j] + cnt[j]++) & (q->size - 1));
        memcpy((char*)dst + 4, (char*)&pk + 4, sizeof pk - 4);
        __atomic_store_n(&dst->full_header, (uint32_t)hdr | ((uint32_t)pk.setup << 16), __ATOMIC_RELEASE);
    }
    for (int j = 0; j < nq; ++j) hsa_signal_store_screlease(qs[j]->doorbell_signal, (hsa_signal_value_t)(base[j] + cnt[j] - 1));
    for (int j = 0; j < nq; ++j) {
        hsa_signal_value_t v = hsa_signal_wait_scacquire(dones[j], HSA_SIGNAL_CONDITION_LT, 1, 5000000000ull, HSA_WAIT_STATE_ACTIVE);   // bounded: 5 s
        if (v >= 1) { printf("TIMEOUT waiting for the chain\n"); return 2; }
    }
    const auto t1 = std::chrono::steady_clock::now();
    *us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    return 0;
}

int main(int argc, char** argv) {
    const char* hsaco = argc > 1 ? argv[1] : "aql_probe.hsaco";
    hipStream_t st; CK(hipStreamCreate(&st));                   // HIP first: it initialises ROCr
    HK(hsa_init());
    HK(hsa_iterate_agents(pick_gpu, nullptr));
    if (!g_have) { printf("no GPU agent\n"); return 1; }
    // code object
    FILE* f = fopen(hsaco, "rb"); if (!f) { printf("cannot open %s\n", hsaco); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(sz); if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 1; fclose(f);
    hsa_code_object_reader_t rd; HK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd));
    hsa_executable_t ex; HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    HK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    HK(hsa_executable_freeze(ex, nullptr));
    hsa_queue_t* qs[2]; hsa_signal_t dones[2];
    for (int j = 0; j < 2; ++j) { HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &qs[j])); HK(hsa_signal_create(1, 0, nullptr, &dones[j])); }
    hsa_queue_t* q = qs[0]; hsa_signal_t done = dones[0];

    for (int mode = 0; mode < 3; ++mode) {   // does a packet really start beside its predecessor?  0: same queue, barrier clear, fences SYSTEM; 1: fences NONE; 2: two queues
        unsigned* ov; CK(hipMalloc(&ov, 4096)); CK(hipMemset(ov, 0, 4096)); CK(hipDeviceSynchronize());
        struct { unsigned *flag, *fail, *spins; } oa = {ov, ov + 32, ov + 64};
        char* ka; CK(hipMalloc(&ka, 128)); CK(hipMemcpy(ka, &oa, sizeof oa, hipMemcpyHostToDevice)); CK(hipMemcpy(ka + 64, &oa, sizeof oa, hipMemcpyHostToDevice));
        KSym kw, ks2; if (get_kernel(ex, "k_wait", &kw) || get_kernel(ex, "k_set", &ks2)) return 1;
        hsa_signal_store_relaxed(dones[0], 1); hsa_signal_store_relaxed(dones[1], 1);
        const int fence = mode == 1 ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_SYSTEM;
        for (int i = 0; i < 2; ++i) {
            hsa_queue_t* qq = mode == 2 ? qs[i] : qs[0];
            const uint64_t at = hsa_queue_add_write_index_relaxed(qq, 1);
            hsa_kernel_dispatch_packet_t pk; memset(&pk, 0, sizeof pk);
            const KSym& k = i ? ks2 : kw;
            pk.setup = 1; pk.workgroup_size_x = 512; pk.workgroup_size_y = 1; pk.workgroup_size_z = 1;
            pk.grid_size_x = (i ? 1 : 256) * 512; pk.grid_size_y = 1; pk.grid_size_z = 1;
            pk.private_segment_size = k.priv; pk.group_segment_size = k.group; pk.kernel_object = k.object; pk.kernarg_address = ka + i * 64;
            pk.completion_signal = mode == 2 ? dones[i] : (i ? dones[0] : hsa_signal_t{0});
            const uint16_t hdr = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((i ? 0 : 1) << HSA_PACKET_HEADER_BARRIER) |
                                            ((i ? fence : HSA_FENCE_SCOPE_SYSTEM) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
            hsa_kernel_dispatch_packet_t* dst = (hsa_kernel_dispatch_packet_t*)qq->base_address + (at & (qq->size - 1));
            memcpy((char*)dst + 4, (char*)&pk + 4, sizeof pk - 4);
            __atomic_store_n(&dst->full_header, (uint32_t)hdr | ((uint32_t)pk.setup << 16), __ATOMIC_RELEASE);
            if (mode == 2 || i == 1) hsa_signal_store_screlease(qq->doorbell_signal, (hsa_signal_value_t)at);
        }
        for (int j = 0; j < (mode == 2 ? 2 : 1); ++j)
            if (hsa_signal_wait_scacquire(dones[j], HSA_SIGNAL_CONDITION_LT, 1, 5000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) { printf("overlap check: TIMEOUT\n"); return 2; }
        CK(hipDeviceSynchronize());
        unsigned h[96]; CK(hipMemcpy(h, ov, sizeof h, hipMemcpyDeviceToHost));
        printf("# overlap check (%s): a 256-workgroup kernel waits for a flag that only the NEXT packet sets: %s (%u polls)\n",
               mode == 0 ? "same queue, barrier bit clear, acquire SYSTEM" : mode == 1 ? "same queue, barrier bit clear, acquire NONE" : "two queues",
               h[32] ? "spin limit hit -> the packets did NOT overlap" : "released -> the packets ran side by side", h[64]);
    }
    const size_t W_BYTES = (size_t)1 << 30;
    u32x4_t* w; float *xring, *gring; unsigned *cnt, *fail; char* kargs;
    const int STAGES = 2000, WGS = 256, REPS = 5;
    const size_t CNT_BYTES = (size_t)(STAGES + 1) * 32 * 128;      // up to 32 shard lines (or one 1 KB flag array) per stage
    CK(hipMalloc(&w, W_BYTES)); CK(hipMemset(w, 1, W_BYTES)); CK(hipMalloc(&xring, 2 * XN * 4)); CK(hipMalloc(&gring, 2 * XN * 8));
    CK(hipMalloc(&cnt, CNT_BYTES)); CK(hipMalloc(&fail, 128)); CK(hipMalloc(&kargs, (size_t)STAGES * 64));
    printf("# %d stages per chain, %d workgroups x 512 threads, x = 32 KB per stage; us per stage (mean of %d chains / best)\n", STAGES, WGS, REPS);
    static_assert(sizeof(StageArgs) == 64, "kernarg block");

    for (int nlw : {1, 2, 4, 6, 8}) {
        const size_t stage_vec = (size_t)WGS * nlw * 512; const int slots = (int)(W_BYTES / 16 / stage_vec);
        std::vector<StageArgs> ha(STAGES);
        // kind: 0 plain / sc1 / single counter, 3 = sharded counters (ns), 4 = flags, 5 = granules
        auto build = [&](int kind, int ns) -> int {
            for (int s = 0; s < STAGES; ++s) {
                StageArgs& A = ha[s]; memset(&A, 0, sizeof A);
                A.w = w + (size_t)(s % slots) * stage_vec; A.fail = fail; A.slice = XN / WGS;
                if (kind == 5) {
                    A.xin = gring + (size_t)(s & 1) * XN * 2; A.xout = gring + (size_t)((s + 1) & 1) * XN * 2; A.target_prev = (unsigned)s;   // tag of stage s-1's output = s
                } else {
                    A.xin = xring + (size_t)(s & 1) * XN; A.xout = xring + (size_t)((s + 1) & 1) * XN;
                    const size_t per_stage = 32 * 32;              // words
                    A.cnt_prev = cnt + (size_t)s * per_stage; A.cnt_mine = cnt + (size_t)(s + 1) * per_stage;
                    if (kind == 3) { A.target_prev = s ? ((unsigned)ns << 16) | (unsigned)(WGS / ns) : 0u; A.pad[0] = (unsigned)ns; }
                    else if (kind == 4) { A.target_prev = s ? (unsigned)s : 0u; A.pad[0] = (unsigned)(s + 1); }   // flag value = stage + 1
                    else A.target_prev = s ? (unsigned)WGS : 0u;
                }
            }
            CK(hipMemcpy(kargs, ha.data(), (size_t)STAGES * 64, hipMemcpyHostToDevice));
            return 0;
        };
        const double mb = stage_vec * 16 / 1e6;
        auto reset = [&]() -> int {
            CK(hipMemset(xring, 0, 2 * XN * 4)); CK(hipMemset(gring, 0, 2 * XN * 8)); CK(hipMemset(cnt, 0, CNT_BYTES)); CK(hipMemset(fail, 0, 128)); CK(hipDeviceSynchronize());
            return 0;
        };
        auto verify = [&](const char* what, bool gran) -> int {
            std::vector<float> hx(XN * 2); unsigned hf = 0;
            if (gran) CK(hipMemcpy(hx.data(), gring + (size_t)(STAGES & 1) * XN * 2, XN * 8, hipMemcpyDeviceToHost));
            else CK(hipMemcpy(hx.data(), xring + (size_t)(STAGES & 1) * XN, XN * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
            int bad = 0; for (int i = 0; i < XN; ++i) bad += (gran ? hx[2 * i + 1] : hx[i]) != (float)STAGES;
            if (bad || hf) printf("    !! %s: %d of %d final values wrong (x[0] = %g, want %d)%s\n", what, bad, XN, gran ? hx[1] : hx[0], STAGES, hf ? "  [SPIN LIMIT HIT]" : "");
            return 0;
        };
        // D: hipGraph
        {
            if (build(0, 0)) return 1;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int s = 0; s < STAGES; ++s) {
                void* fn = nlw == 1 ? (void*)k_plain_1 : nlw == 2 ? (void*)k_plain_2 : nlw == 4 ? (void*)k_plain_4 : nlw == 6 ? (void*)k_plain_6 : (void*)k_plain_8;
                StageArgs& A = ha[s];
                void* args[] = {&A.w, &A.xin, &A.xout, &A.cnt_prev, &A.cnt_mine, &A.fail, &A.target_prev, &A.slice, &A.pad[0]};
                CK(hipLaunchKernel(fn, dim3(WGS), dim3(512), args, 0, st));
            }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (reset()) return 1;
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); verify("D warmup", false);
            double sum = 0, best = 1e30;
            for (int r = 0; r < REPS; ++r) {
                if (reset()) return 1;
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms * 1e3; if (ms * 1e3 < best) best = ms * 1e3;
            }
            verify("D", false);
            printf("D   hipGraph, one kernel per stage                     %5.1f MB/stage   %6.2f / %6.2f\n", mb, sum / REPS / STAGES, best / STAGES);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
        // G2: the two-queue form expressed in HIP alone: ONE hipGraph, captured from two streams (fork / join by events), stage s on
        // stream s % 2 — two independent chains, no edge between them; the data dependence is carried by the tagged granules
        for (int nchain = 2; nchain <= 3; ++nchain) {
            if (build(5, 0)) return 1;
            hipStream_t sb[3]; hipEvent_t ef, ej[3];
            for (int j = 0; j < 3; ++j) { CK(hipStreamCreate(&sb[j])); CK(hipEventCreateWithFlags(&ej[j], hipEventDisableTiming)); }
            CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            CK(hipEventRecord(ef, st));
            for (int j = 1; j < nchain; ++j) CK(hipStreamWaitEvent(sb[j], ef, 0));
            for (int s = 0; s < STAGES; ++s) {
                void* fn = nlw == 1 ? (void*)k_gran_1 : nlw == 2 ? (void*)k_gran_2 : nlw == 4 ? (void*)k_gran_4 : nlw == 6 ? (void*)k_gran_6 : (void*)k_gran_8;
                StageArgs& A = ha[s];
                void* args[] = {&A.w, &A.xin, &A.xout, &A.cnt_prev, &A.cnt_mine, &A.fail, &A.target_prev, &A.slice, &A.pad[0]};
                CK(hipLaunchKernel(fn, dim3(WGS), dim3(512), args, 0, (s % nchain) ? sb[s % nchain] : st));
            }
            for (int j = 1; j < nchain; ++j) { CK(hipEventRecord(ej[j], sb[j])); CK(hipStreamWaitEvent(st, ej[j], 0)); }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            if (reset()) return 1;
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); verify("G2 warmup", true);
            double sum = 0, best = 1e30;
            for (int r = 0; r < REPS; ++r) {
                if (reset()) return 1;
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms * 1e3; if (ms * 1e3 < best) best = ms * 1e3;
            }
            verify("G2", true);
            printf("G%d  hipGraph, %d independent chains, tagged granules       %5.1f MB/stage   %6.2f / %6.2f\n", nchain, nchain, mb, sum / REPS / STAGES, best / STAGES);
            fflush(stdout);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
            for (int j = 0; j < 3; ++j) { (void)hipStreamDestroy(sb[j]); (void)hipEventDestroy(ej[j]); }
        }
        struct V { const char* name; const char* kernel; int kind, ns; bool barrier; int fence; uint32_t lds; int nq; int acq = -1; };
        const V vs[] = {
            {"Q1  AQL, barrier, fences AGENT, plain ld/st            ", "k_plain", 0, 0, true, HSA_FENCE_SCOPE_AGENT},
            // the two halves of the boundary priced apart, PLAIN loads and stores: every workgroup of every stage reads the whole x
            // ring slot (so every XCD's L2 and every CU's L1 holds it) and two stages later other workgroups have rewritten it — a
            // stale hit anywhere changes the final value and the verification reports it
            {"Q1b AQL, barrier, acquire NONE + release AGENT, plain   ", "k_plain", 0, 0, true, HSA_FENCE_SCOPE_AGENT, 0, 0, HSA_FENCE_SCOPE_NONE},
            {"Q1c AQL, barrier, acquire AGENT + release NONE, plain   ", "k_plain", 0, 0, true, HSA_FENCE_SCOPE_NONE, 0, 0, HSA_FENCE_SCOPE_AGENT},
            {"Q2  AQL, barrier, fences NONE, sc1 ld/st               ", "k_sc1", 0, 0, true, HSA_FENCE_SCOPE_NONE},
            {"Q2a AQL, barrier, fences AGENT, sc1 ld/st              ", "k_sc1", 0, 0, true, HSA_FENCE_SCOPE_AGENT},
            {"Q3  AQL, NO barrier, one counter, poll + arrive        ", "k_poll", 0, 0, false, HSA_FENCE_SCOPE_NONE},
            {"Q3b AQL, NO barrier, 8 counter shards                  ", "k_shard", 3, 8, false, HSA_FENCE_SCOPE_NONE},
            {"Q3b AQL, NO barrier, 32 counter shards                 ", "k_shard", 3, 32, false, HSA_FENCE_SCOPE_NONE},
            {"Q3c AQL, NO barrier, per-workgroup flags               ", "k_flags", 4, 0, false, HSA_FENCE_SCOPE_NONE},
            {"Q3d AQL, NO barrier, tagged granules                   ", "k_gran", 5, 0, false, HSA_FENCE_SCOPE_NONE},
            {"Q3e AQL, barrier, fences NONE, 32 shards (no overlap)  ", "k_shard", 3, 32, true, HSA_FENCE_SCOPE_NONE},
            {"Q3f = Q3c, 80 KB LDS per workgroup (2 stages resident) ", "k_flags", 4, 0, false, HSA_FENCE_SCOPE_NONE, 80 * 1024},
            {"Q3g = Q3d, 80 KB LDS per workgroup (2 stages resident) ", "k_gran", 5, 0, false, HSA_FENCE_SCOPE_NONE, 80 * 1024},
            {"Q2b = Q2, 80 KB LDS per workgroup                      ", "k_sc1", 0, 0, true, HSA_FENCE_SCOPE_NONE, 80 * 1024},
            {"Q5c TWO queues alternating, per-workgroup flags        ", "k_flags", 4, 0, true, HSA_FENCE_SCOPE_NONE, 0, 2},
            {"Q5d TWO queues alternating, tagged granules            ", "k_gran", 5, 0, true, HSA_FENCE_SCOPE_NONE, 0, 2},
            {"Q5b TWO queues alternating, 32 counter shards          ", "k_shard", 3, 32, true, HSA_FENCE_SCOPE_NONE, 0, 2},
        };
        for (const V& v : vs) {
            char kn[64]; KSym k; snprintf(kn, sizeof kn, "%s_%d", v.kernel, nlw); if (get_kernel(ex, kn, &k)) return 1;
            if (build(v.kind, v.ns)) return 1;
            double us = 0, sum = 0, best = 1e30;
            if (reset()) return 1;
            int rc = run_chain(qs, v.nq ? v.nq : 1, dones, k, kargs, STAGES, WGS, v.barrier, v.fence, &us, v.lds, v.acq); if (rc) return rc;
            verify("warmup", v.kind == 5);
            for (int r = 0; r < REPS; ++r) {
                if (reset()) return 1;
                rc = run_chain(qs, v.nq ? v.nq : 1, dones, k, kargs, STAGES, WGS, v.barrier, v.fence, &us, v.lds, v.acq); if (rc) return rc;
                sum += us; if (us < best) best = us;
            }
            verify(v.name, v.kind == 5);
            printf("%s %5.1f MB/stage   %6.2f / %6.2f\n", v.name, mb, sum / REPS / STAGES, best / STAGES);
            fflush(stdout);
        }
    }
    hsa_queue_destroy(qs[0]); hsa_queue_destroy(qs[1]);
    return 0;
}
#endif
