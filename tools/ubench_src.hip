// ubench_src.hip — does the VALU issue rate on gfx950 depend on how many VGPR source operands an instruction reads?
// (cost model for the SPH pair loops: their instructions read 2-3 VGPRs, tools/ubench_valu's read one.)
// 2048 blocks x 256 threads, 8 independent accumulators per lane, ITERS x 8 instructions of one kind per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[8], b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = 1.0f - 1e-7f * (threadIdx.x + i); c[i] = 1e-9f * (threadIdx.x + 3 * i); }
    const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c1), "s"(c2));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c[i]));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(c2));
            if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, 1.0 clamp" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(c1));
            if (KIND == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
            if (KIND == 8) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (KIND == 9) asm volatile("v_min_f32 %0, 0, %0" : "+v"(a[i]));
            if (KIND == 10) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));   // dst != src
            if (KIND == 11) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "v"(b[(i + 1) & 7]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, float *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 2048;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = blocks * 4.0 * ITERS * 8.0;
    printf("%-44s %8.3f ms  %6.3f ns*SIMD per wave-instr\n", name, ms, ms * 1e6 * 1024 / instr);
}
int main() {
    float *d; (void)hipMalloc(&d, 2048 * 256 * 4);
    run<0>("fma  v = v*s + s        (1 VGPR src)", d);
    run<6>("mul  v = v*s            (1 VGPR src)", d);
    run<9>("min  v = min(0, v)      (1 VGPR src)", d);
    run<2>("mul  v = v*v            (2 VGPR src)", d);
    run<3>("add  v = v+v            (2 VGPR src)", d);
    run<8>("sub  v = v-v            (2 VGPR src)", d);
    run<4>("fma  v = v*v + s        (2 VGPR src)", d);
    run<5>("fma  v = v*v + 1 clamp  (2 VGPR src)", d);
    run<1>("fma  v = v*v + v        (3 VGPR src)", d);
    run<7>("fmac v += v*v           (3 VGPR src)", d);
    run<10>("mul  d = v*v  (dst!=src, 2 VGPR src)", d);
    run<11>("fma  d = v*v + v (dst!=src, 3 VGPR)", d);
    return 0;
}
