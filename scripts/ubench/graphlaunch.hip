// microbenchmark: dependent tiny kernels back to back -- stream launches vs one hipGraph of the same chain.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_one(double *buf) {
    double x = buf[blockIdx.x];
    x = x * 1.0000001 + 1.0 + buf[(blockIdx.x + 1) % gridDim.x] * 1e-9;
    if (threadIdx.x == 0) buf[blockIdx.x] = x;
}
int main() {
    const int iters = 2000;
    for (int blocks : {64, 256, 1024}) {
        double *buf;
        hipMalloc(&buf, sizeof(double) * blocks); hipMemset(buf, 0, sizeof(double) * blocks);
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(64), 0, s, buf);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(64), 0, s, buf);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %4d: stream %.2f us/kernel", blocks, ms * 1e3 / iters);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(64), 0, s, buf);
        hipStreamEndCapture(s, &g);
        hipError_t err = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf(" | graph (%s) %.2f us/kernel\n", hipGetErrorString(err), ms * 1e3 / iters);
        hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(buf);
    }
    return 0;
}
