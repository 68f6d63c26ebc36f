// Launch / memory-hop floor probe on gfx950 (development aid): per-node time of hipGraph-replayed chains of
//   empty kernels, 1-hop (one 16-B nt load per lane then a store), 2-hop (index load → data load → store) kernels
// at several grid shapes. Build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(int* sink) { if (sink && threadIdx.x == 9999) sink[0] = 1; }
__global__ void k_hop1(const u32x4_t* w, float* out, size_t per_block_vec) {
    const u32x4_t v = __builtin_nontemporal_load(w + blockIdx.x * per_block_vec + threadIdx.x);
    if ((v[0] ^ v[1] ^ v[2] ^ v[3]) == 0x12345u) out[blockIdx.x] = 1.f;
}
template <int NL>
__global__ void k_stream(const u32x4_t* w, float* out, size_t per_block_vec) {   // NL 16-B loads per lane in flight
    u32x4_t acc = {0, 0, 0, 0};
    const u32x4_t* p = w + blockIdx.x * per_block_vec + threadIdx.x;
    u32x4_t v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = __builtin_nontemporal_load(p + (size_t)i * blockDim.x);
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1.f;
}
__global__ void k_hop2(const int* idx, const u32x4_t* w, float* out, size_t per_block_vec) {
    const int j = idx[blockIdx.x];
    const u32x4_t v = __builtin_nontemporal_load(w + (size_t)j * per_block_vec + threadIdx.x);
    if ((v[0] ^ v[1] ^ v[2] ^ v[3]) == 0x12345u) out[blockIdx.x] = 1.f;
}

template <class F> static int chain(const char* name, int nodes, int reps, F launch) {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) launch(st, i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us/node\n", name, ms * 1e3 / (nodes * reps));
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    return 0;
}

int main() {
    const size_t W_BYTES = (size_t)1 << 30;     // 1 GiB of "weights" so consecutive nodes never hit L2 / MALL
    u32x4_t* w; float* out; int* idx;
    CK(hipMalloc(&w, W_BYTES)); CK(hipMemset(w, 1, W_BYTES)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&idx, 1 << 20));
    std::vector<int> h(1 << 18); for (size_t i = 0; i < h.size(); ++i) h[i] = (int)(i % 1024);
    CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int NODES = 200, REPS = 20;
    struct G { int wgs, thr; } grids[] = {{1, 64}, {64, 256}, {256, 256}, {256, 512}, {256, 1024}, {512, 512}, {1024, 256}, {2048, 256}};
    for (auto g : grids) {
        char nm[96]; snprintf(nm, sizeof nm, "empty            %5d WGs x %4d thr", g.wgs, g.thr);
        if (chain(nm, NODES, REPS, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty, dim3(g.wgs), dim3(g.thr), 0, st, (int*)nullptr); })) return 1;
    }
    for (auto g : grids) {
        char nm[96]; snprintf(nm, sizeof nm, "1 hop (16 B/lane) %5d WGs x %4d thr", g.wgs, g.thr);
        const size_t pbv = g.thr; const size_t node_vec = pbv * g.wgs; const int slots = (int)(W_BYTES / 16 / node_vec);
        if (chain(nm, NODES, REPS, [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_hop1, dim3(g.wgs), dim3(g.thr), 0, st, w + (size_t)(i % slots) * node_vec, out, pbv); })) return 1;
    }
    for (auto g : grids) {
        char nm[96]; snprintf(nm, sizeof nm, "2 hops            %5d WGs x %4d thr", g.wgs, g.thr);
        const size_t pbv = g.thr;
        if (chain(nm, NODES, REPS, [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_hop2, dim3(g.wgs), dim3(g.thr), 0, st, idx + (i % 64) * 1024, w, out, pbv); })) return 1;
    }
    // streaming N MiB per node with NL loads in flight per lane
    struct S { int wgs, thr, nl; } ss[] = {{256, 256, 4}, {256, 256, 8}, {256, 256, 16}, {256, 512, 4}, {256, 512, 8}, {256, 512, 16}, {256, 1024, 4}, {256, 1024, 8},
                                           {512, 256, 8}, {512, 256, 16}, {512, 512, 8}, {1024, 256, 8}, {1024, 256, 16}, {2048, 256, 16}};
    for (auto s : ss) {
        char nm[96];
        const size_t pbv = (size_t)s.thr * s.nl, node_vec = pbv * s.wgs; const double mb = node_vec * 16 / 1e6; const int slots = (int)(W_BYTES / 16 / node_vec);
        snprintf(nm, sizeof nm, "stream %5.1f MB %4d WGs x %4d thr x %2d ld", mb, s.wgs, s.thr, s.nl);
        auto L = [&](hipStream_t st, int i) {
            const u32x4_t* p = w + (size_t)(i % slots) * node_vec;
            switch (s.nl) { case 4: hipLaunchKernelGGL(k_stream<4>, dim3(s.wgs), dim3(s.thr), 0, st, p, out, pbv); break;
                            case 8: hipLaunchKernelGGL(k_stream<8>, dim3(s.wgs), dim3(s.thr), 0, st, p, out, pbv); break;
                            default: hipLaunchKernelGGL(k_stream<16>, dim3(s.wgs), dim3(s.thr), 0, st, p, out, pbv); }
        };
        if (chain(nm, NODES, REPS, L)) return 1;
    }
    // Does data read by one kernel stay in the L2 / Infinity Cache for the next? Same streaming kernels, but consecutive
    // nodes alternate between only 2 (or 16) regions instead of walking 1 GiB: "warm" per-node time vs the cold one above.
    for (int regions : {1, 2, 16}) {
        for (auto s : {S{256, 512, 4}, S{256, 512, 8}, S{512, 512, 8}}) {
            char nm[96];
            const size_t pbv = (size_t)s.thr * s.nl, node_vec = pbv * s.wgs; const double mb = node_vec * 16 / 1e6;
            snprintf(nm, sizeof nm, "warm x%-2d %5.1f MB %4d WGs x %4d thr x %2d ld", regions, mb, s.wgs, s.thr, s.nl);
            auto L = [&](hipStream_t st, int i) {
                const u32x4_t* p = w + (size_t)(i % regions) * node_vec;
                switch (s.nl) { case 4: hipLaunchKernelGGL(k_stream<4>, dim3(s.wgs), dim3(s.thr), 0, st, p, out, pbv); break;
                                default: hipLaunchKernelGGL(k_stream<8>, dim3(s.wgs), dim3(s.thr), 0, st, p, out, pbv); }
            };
            if (chain(nm, NODES, REPS, L)) return 1;
        }
    }
    return 0;
}
