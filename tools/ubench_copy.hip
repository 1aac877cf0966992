// ubench_copy.hip — which streaming-copy kernel shape reaches the most on this part (the copy leg of sph_box_calibrate)?
// hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
template <int U, bool NT> void run(const char *name, const v4f *s, v4f *d, size_t n4, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, s, d, n4);
    hipEventRecord(e0, 0);
    const int reps = 40;
    for (int k = 0; k < reps; k++) hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, s, d, n4);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %6d : %8.1f GB/s (read + write)\n", name, grid, 2.0 * n4 * 16 * reps / (ms * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    v4f *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<1, false>("1 load / thread / trip", s, d, n4, grid);
        run<4, false>("4 loads / thread / trip", s, d, n4, grid);
        run<4, true>("4 loads, nontemporal", s, d, n4, grid);
        run<8, false>("8 loads / thread / trip", s, d, n4, grid);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e0, 0);
    for (int k = 0; k < 20; k++) hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyDtoD                              : %8.1f GB/s (read + write)\n", 2.0 * bytes * 20 / (ms * 1e-3) / 1e9);
    return 0;
}
