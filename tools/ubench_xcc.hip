// tools/ubench_xcc.hip — which XCD does workgroup i of a small grid run on (s_getreg XCC_ID), alone and behind another
// stream's kernel?  The grid barrier of k_rebuild assumes "workgroup i on XCD i mod 8" and checks it with this register.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_xcc.hip -o tools/ubench_xcc && tools/ubench_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_xcc(unsigned *out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ void k_busy(float *x, int iters) {
    float a = x[threadIdx.x];
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    x[threadIdx.x] = a;
}
int main() {
    const int n = 48;
    unsigned *d; float *f;
    hipMalloc(&d, n * 4); hipMalloc(&f, 1024 * 4); hipMemset(f, 0, 4096);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<unsigned> h(n);
    for (int rep = 0; rep < 4; rep++) {
        if (rep >= 2) hipLaunchKernelGGL(k_busy, dim3(37 + 11 * rep), dim3(256), 0, s2, f, 200000);      // another stream keeps some slots busy
        if (rep == 1 || rep == 3) hipLaunchKernelGGL(k_busy, dim3(3 + rep), dim3(256), 0, s1, f, 10);       // a small kernel just before, same stream
        hipLaunchKernelGGL(k_xcc, dim3(n), dim3(256), 0, s1, d);
        hipStreamSynchronize(s1);
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        printf("rep %d raw[0]=0x%x:", rep, h[0]);
        int bad = 0;
        for (int i = 0; i < n; i++) { printf(" %u", h[i] & 15u); bad += ((h[i] & 15u) != (h[i & 7] & 15u)); }
        printf("  | workgroups not on the XCD of workgroup (i mod 8): %d\n", bad);
        hipDeviceSynchronize();
    }
    return 0;
}
