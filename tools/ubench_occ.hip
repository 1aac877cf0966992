// ubench_occ.hip — how VALU throughput on gfx950 depends on (a) waves per SIMD, (b) dependent vs independent
// streams, (c) an LDS read + wait in front of each block of math (the shape of the SPH pair loops).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
// MODE 0: 32 independent fma per iteration (8 regs x 4);  1: 32 fma in ONE dependent chain;
// MODE 2: ds_read_b64 (lane-dependent address) -> wait -> 32 fma (4 chains of 8);  3: same but next read prefetched
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed, int lds_pad_words) {
    extern __shared__ float2 lds[];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float2(seed + i, seed - i);
    __syncthreads();
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3f;
    const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
    unsigned idx = threadIdx.x * 7u;
    float2 nxt = lds[idx & 1023];
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = fmaf(a[i], c1, c2);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 32; r++) a[0] = fmaf(a[0], c1, c2);
        } else if (MODE == 2) {
            idx = idx * 5u + 1u;
            float2 q = lds[idx & 1023];
#pragma unroll
            for (int r = 0; r < 8; r++) { a[0] = fmaf(a[0], c1, q.x); a[1] = fmaf(a[1], c1, q.y); a[2] = fmaf(a[2], c1, q.x); a[3] = fmaf(a[3], c1, q.y); }
        } else {
            idx = idx * 5u + 1u;
            float2 q = nxt;
            nxt = lds[idx & 1023];
#pragma unroll
            for (int r = 0; r < 8; r++) { a[0] = fmaf(a[0], c1, q.x); a[1] = fmaf(a[1], c1, q.y); a[2] = fmaf(a[2], c1, q.x); a[3] = fmaf(a[3], c1, q.y); }
        }
    }
    float s = nxt.x;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, float *d, int blocks_per_cu) {
    // occupancy is limited through dynamic LDS: 160 KiB / blocks_per_cu per block
    size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 1024;
    if (lds < 8192) lds = 8192;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu * 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 1.0001f, 0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 1.0001f, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = blocks * 4.0 * ITERS * 32.0;   // fma wave-instructions
    printf("%-44s waves/SIMD %d  %8.3f ms  %6.2f ns*SIMD per fma wave-instr\n", name, blocks_per_cu, ms, ms * 1e6 * 1024 / instr);
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 4 * 256 * 4);
    for (int occ : {1, 2, 4, 8}) run<0>("independent fma", d, occ);
    for (int occ : {1, 2, 4, 8}) run<1>("dependent fma chain", d, occ);
    for (int occ : {1, 2, 4, 8}) run<2>("lds read -> wait -> 32 fma (4 chains)", d, occ);
    for (int occ : {1, 2, 4, 8}) run<3>("prefetched lds read, 32 fma (4 chains)", d, occ);
    return 0;
}
