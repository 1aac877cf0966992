// ubench_valu.hip — VALU issue-cost microbenchmark for gfx950 (cost model for the pair loops).
// Each kernel runs a long dependent-free stream of one instruction kind on 8 independent registers per lane,
// 1024 blocks x 256 threads (every SIMD saturated), and reports SIMD-cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[8]; v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
    const v2f q1 = {c1, c1}, q2 = {c2, c2};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) a[i] = fmaf(a[i], c1, c2);
            if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], q1, q2);
            if (KIND == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if (KIND == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (KIND == 4) a[i] = a[i] * c1;
            if (KIND == 5) p[i] = p[i] * q1;
            if (KIND == 6) a[i] = fminf(a[i], c1) + 0.0f * it;
            if (KIND == 7) { unsigned u = __float_as_uint(a[i]); u = (u & (u - 1)) | 0x3f800000u; a[i] = __uint_as_float(u); }
            if (KIND == 8) { unsigned u = __float_as_uint(a[i]); a[i] = __uint_as_float((unsigned)__ffs((int)u) + 0x3f800000u); }
            if (KIND == 9) a[i] = (a[i] < c1) ? a[i] + c2 : c1;
            if (KIND == 10) a[i] = __builtin_amdgcn_rsqf(a[i]);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, float *d, int instr_per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * 4.0, instr = waves * ITERS * 8.0 * instr_per_iter;
    double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;   // at the nominal 2.4 GHz; real clock may be lower
    printf("%-28s %8.3f ms  %6.2f SIMD-cycles(@2.4GHz)/wave-instr\n", name, ms, simd_cycles / instr);
}
int main() {
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    run<0>("v_fma_f32", d, 1);
    run<1>("v_pk_fma_f32", d, 1);
    run<4>("v_mul_f32", d, 1);
    run<5>("v_pk_mul_f32", d, 1);
    run<2>("v_sqrt_f32", d, 1);
    run<3>("v_rcp_f32", d, 1);
    run<10>("v_rsq_f32", d, 1);
    run<6>("v_min+fma(2 instr)", d, 2);
    run<7>("and/add/or (3 int instr)", d, 3);
    run<8>("ffbl+add (2 instr)", d, 2);
    run<9>("cmp+add+cndmask (3 instr)", d, 3);
    return 0;
}
