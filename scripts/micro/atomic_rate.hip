// Microbenchmark: rate of device-scope atomicAdd with return on one / eight addresses from
// many resident waves (one atomic per wave per iteration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* ctr, int nctr, int iters, unsigned* sink)
{
    unsigned acc = 0;
    const int which = (blockIdx.x % nctr) * 64; // counters 256 B apart
    for (int i = 0; i < iters; ++i) {
        unsigned v = 0;
        if (threadIdx.x == 0) v = atomicAdd(&ctr[which], 64u);
        acc += __shfl(v, 0, 64);
    }
    if (acc == 0xdeadbeef) sink[0] = acc;
}
int main()
{
    unsigned *ctr, *sink;
    hipMalloc(&ctr, 4096 * 4); hipMalloc(&sink, 4);
    hipMemset(ctr, 0, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nctr : {1, 8, 64}) for (int waves : {1024, 7168}) {
        const int iters = 8;
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, ctr, nctr, iters, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, ctr, nctr, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("counters=%d waves=%d atomics=%d time=%.1f us -> %.1f ns per atomic (aggregate)\n", nctr, waves, waves * iters, ms * 1e3, ms * 1e6 / (waves * iters));
    }
    return 0;
}
