// tools/ubench_l1inv.hip — which instruction makes a workgroup see, with a PLAIN load, data that a workgroup on ANOTHER
// compute unit of the SAME XCD has written since it last read the line?  (The L2 is shared inside an XCD, so what this
// measures is the per-CU vector L1.)  Workgroups 0 and 8 of a 16-workgroup grid run on one XCD (tools/ubench_xcc).
//   reader (wg 0): reads data[0..63] (line now in its L1), raises `ready`, waits for `written`, executes the candidate
//                  invalidate, reads data[] again with plain loads and counts stale values
//   writer (wg 8): waits for `ready`, writes new values, waits for its stores (vmcnt), raises `written`
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(volatile unsigned *flags, unsigned *data, unsigned *stale, unsigned round) {
    const unsigned wg = blockIdx.x, t = threadIdx.x;
    if (wg == 0) {
        unsigned a = data[t];                                  // plain load: the line is in this CU's L1 now
        __syncthreads();
        if (t == 0) __hip_atomic_store((unsigned *)&flags[0], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) while (__hip_atomic_load((unsigned *)&flags[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != round) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        if (t == 0) {
            if (MODE == 1) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 2) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 3) asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 4) asm volatile("buffer_inv\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        unsigned b;
        if (MODE == 5) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(data + t) : "memory");
        else if (MODE == 6) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(data + t) : "memory");
        else b = data[t];
        if (b != round * 1000u + t) atomicAdd(stale, 1u);
        if (a == 0xffffffffu) data[t + 4096] = a;              // (keeps `a` alive)
    } else if (wg == 8) {
        if (t == 0) while (__hip_atomic_load((unsigned *)&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != round) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        data[t] = round * 1000u + t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store((unsigned *)&flags[32], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int MODE> void run(const char *name, unsigned *flags, unsigned *data, unsigned *stale) {
    unsigned total = 0;
    for (unsigned r = 1; r <= 200; r++) {
        hipMemset(stale, 0, 4);
        hipLaunchKernelGGL(k<MODE>, dim3(16), dim3(64), 0, 0, flags, data, stale, r + 1000u * MODE);
        hipDeviceSynchronize();
        unsigned h = 0; hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost);
        total += h;
    }
    printf("%-28s stale values in 200 rounds x 64 lanes: %u\n", name, total);
}
int main() {
    unsigned *flags, *data, *stale;
    hipMalloc(&flags, 4096); hipMalloc(&data, 65536); hipMalloc(&stale, 4);
    hipMemset(flags, 0, 4096); hipMemset(data, 0, 65536);
    run<0>("nothing", flags, data, stale);
    run<1>("buffer_inv sc0", flags, data, stale);
    run<2>("buffer_inv sc1", flags, data, stale);
    run<3>("buffer_inv sc0 sc1", flags, data, stale);
    run<4>("buffer_inv (no bits)", flags, data, stale);
    run<5>("plain L1 + load sc0", flags, data, stale);
    run<6>("plain L1 + load nt", flags, data, stale);
    return 0;
}
