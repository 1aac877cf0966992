// ubench_launch.hip — what does a kernel that returns at once cost inside a captured graph, as a function of its grid
// size and static LDS?  (The per-step rebuild kernels are such launches in most steps.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k_gate(const unsigned *flag, float *out) {
    __shared__ float buf[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if (*flag == 0u) return;
    buf[threadIdx.x] = (float)blockIdx.x;      // never runs: keeps the LDS allocation alive
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[255 - threadIdx.x];
}

__global__ __launch_bounds__(256) void k_work(float *a, int n) {      // a short real kernel between the gates
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = a[i] * 1.0001f + 1.0f;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS>
int run(hipStream_t st, unsigned *flag, float *out, float *a, int n, int grid, int chain) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
    for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_gate<LDS>, dim3(grid), dim3(256), 0, st, flag, out);
    hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 20; r++) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    const int reps = 200;
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("lds %6d B  grid %6d  chain %2d : %8.2f us per graph\n", LDS, grid, chain, ms * 1e3 / reps);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    return 0;
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *flag;
    float *out, *a;
    const int n = 1 << 21;
    CHECK(hipMalloc(&flag, 4));
    CHECK(hipMemset(flag, 0, 4));
    CHECK(hipMalloc(&out, (size_t)65536 * 256 * 4));
    CHECK(hipMalloc(&a, (size_t)n * 4));
    CHECK(hipMemset(a, 0, (size_t)n * 4));
    const int grids[] = {8, 256, 1024, 2048, 8192, 32768};
    for (int chain : {0, 1, 6}) {
        for (int grid : grids) {
            if (chain == 0 && grid != 8) continue;
            if (run<0>(st, flag, out, a, n, grid, chain)) return 1;
            if (chain && run<32768>(st, flag, out, a, n, grid, chain)) return 1;
        }
    }
    return 0;
}
