// ubench_ldsatomic.hip — what does an LDS atomic cost on gfx950?  Each lane of every wave does N operations on pseudo-random slots of a
// 4 KB LDS array: ds_add_f32, ds_add_u32, ds_write_b32, ds_read_b32 (returning).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ldsatomic.hip -o tools/ubench_ldsatomic
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int n) {
    __shared__ float a[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) a[i] = 0.0f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.0f;
    const unsigned base = (unsigned)(size_t)a;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        const unsigned addr = base + ((s >> 10) & 1023u) * 4u;
        const float v = (float)(s & 255u);
        if (OP == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(v) : "memory");
        else if (OP == 1) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(s) : "memory");
        else if (OP == 2) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
        else { float r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); acc += r; }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = a[threadIdx.x] + acc;
}
int main() {
    float *out;
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 256, grid = 4096;
    const char *names[4] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_read_b32 + wait"};
    for (int op = 0; op < 4; op++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, n);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, n);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, n);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per CU: grid * 4 waves * n / 256 CUs; cycles at 2.4 GHz
            const double per = ms * 1e-3 * 2.4e9 / ((double)grid * 4 * n / 256.0);
            if (rep) printf("%-20s %8.3f ms  = %6.1f cycles per wave-instruction and CU\n", names[op], ms, per);
        }
    }
    return 0;
}
