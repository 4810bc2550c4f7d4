// microbenchmark: cost of a device-wide barrier inside one kernel (cooperative groups grid.sync() and a hand-rolled
// atomic counter barrier) vs the kernel boundary, for 64-thread blocks. hipcc --offload-arch=gfx950 -O3 gridsync.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_cg(int iters, double *buf) {
    cg::grid_group g = cg::this_grid();
    double x = buf[blockIdx.x];
    for (int i = 0; i < iters; ++i) {
        x = x * 1.0000001 + 1.0;
        if (threadIdx.x == 0) buf[blockIdx.x] = x;
        g.sync();
        x += buf[(blockIdx.x + 1) % gridDim.x] * 1e-9;
    }
    if (threadIdx.x == 0) buf[blockIdx.x] = x;
}
__global__ void k_atomic(int iters, double *buf, unsigned *bar) {
    double x = buf[blockIdx.x];
    for (int i = 0; i < iters; ++i) {
        x = x * 1.0000001 + 1.0;
        if (threadIdx.x == 0) {
            __hip_atomic_store(&buf[blockIdx.x], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default? use explicit builtin below
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * gridDim.x;
            long long spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 100000000LL) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        x += __hip_atomic_load(&buf[(blockIdx.x + 1) % gridDim.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 1e-9;
    }
    if (threadIdx.x == 0) buf[blockIdx.x] = x;
}
__global__ void k_one(double *buf) {
    double x = buf[blockIdx.x];
    x = x * 1.0000001 + 1.0 + buf[(blockIdx.x + 1) % gridDim.x] * 1e-9;
    if (threadIdx.x == 0) buf[blockIdx.x] = x;
}
int main() {
    const int iters = 2000;
    for (int blocks : {64, 256, 1024, 2048}) {
        double *buf; unsigned *bar;
        hipMalloc(&buf, sizeof(double) * blocks); hipMemset(buf, 0, sizeof(double) * blocks);
        hipMalloc(&bar, 4); hipMemset(bar, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        int it = iters;
        void *args[] = {&it, &buf};
        hipEventRecord(e0);
        hipError_t err = hipLaunchCooperativeKernel((void *)k_cg, dim3(blocks), dim3(64), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %4d: grid.sync %s %.2f us/iter", blocks, hipGetErrorString(err), ms * 1e3 / iters);
        hipMemset(bar, 0, 4);
        void *args2[] = {&it, &buf, &bar};
        hipEventRecord(e0);
        err = hipLaunchCooperativeKernel((void *)k_atomic, dim3(blocks), dim3(64), args2, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf(" | atomic barrier %s %.2f us/iter", hipGetErrorString(err), ms * 1e3 / iters);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(64), 0, 0, buf);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf(" | kernel boundary %.2f us/iter\n", ms * 1e3 / iters);
        hipFree(buf); hipFree(bar);
    }
    return 0;
}
