// Probe of v_mfma_f32_4x4x4_16b_bf16 operand/result layout on gfx950 (development aid). Build: hipcc --offload-arch=gfx950 -O3 mfma4_layout.hip -o mfma4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
static uint16_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
__global__ void k(const s16x4* a, const s16x4* b, f32x4* d) {
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}
int main() {
    // assumed: lane = 4*blk + i holds A_blk[i][k=0..3]; lane = 4*blk + j holds B_blk[k=0..3][j]; D: lane = 4*blk + j, reg i
    uint16_t ha[64][4], hb[64][4]; float A[16][4][4], B[16][4][4];
    for (int blk = 0; blk < 16; ++blk) for (int r = 0; r < 4; ++r) for (int kk = 0; kk < 4; ++kk) {
        A[blk][r][kk] = (float)(1 + r + 4 * kk) * (blk % 3 + 1);        // small ints: exact in bf16
        B[blk][kk][r] = (float)((r == kk ? 2 : 1) + (blk & 1)) ;         // B[blk][k][j]
    }
    for (int l = 0; l < 64; ++l) for (int kk = 0; kk < 4; ++kk) { ha[l][kk] = bf(A[l / 4][l % 4][kk]); hb[l][kk] = bf(B[l / 4][kk][l % 4]); }
    s16x4 *da, *db; f32x4* dd; float hd[64][4];
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int blk = l / 4, j = l % 4; float e = 0;
        for (int kk = 0; kk < 4; ++kk) e += A[blk][r][kk] * B[blk][kk][j];
        if (e != hd[l][r]) { if (bad < 12) printf("lane %d reg %d: got %g expected %g\n", l, r, hd[l][r], e); ++bad; }
    }
    printf("assumed layout mismatches: %d / 256\n", bad);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l][0], hd[l][1], hd[l][2], hd[l][3]);
    return 0;
}
