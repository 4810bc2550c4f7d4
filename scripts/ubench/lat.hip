// micro-benchmark: dependent-issue latency of f64 VALU ops on gfx950 (one wave, s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b, int iters) {
    double x = a + threadIdx.x, y0 = b, y1 = b + 1, y2 = b + 2, y3 = b + 3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (MODE == 0) { x = x + b; asm volatile("" : "+v"(x)); }                       // dependent adds
            if (MODE == 1) { x = __builtin_fma(x, b, a); asm volatile("" : "+v"(x)); }      // dependent fmas
            if (MODE == 2) {                                                                // add + 1 independent
                x = x + b; asm volatile("" : "+v"(x));
                y0 = __builtin_fma(y0, b, a); asm volatile("" : "+v"(y0));
            }
            if (MODE == 3) {                                                                // add + 2 independent
                x = x + b; asm volatile("" : "+v"(x));
                y0 = __builtin_fma(y0, b, a); asm volatile("" : "+v"(y0));
                y1 = __builtin_fma(y1, b, a); asm volatile("" : "+v"(y1));
            }
            if (MODE == 4) {                                                                // add + 4 independent
                x = x + b; asm volatile("" : "+v"(x));
                y0 = __builtin_fma(y0, b, a); asm volatile("" : "+v"(y0));
                y1 = __builtin_fma(y1, b, a); asm volatile("" : "+v"(y1));
                y2 = __builtin_fma(y2, b, a); asm volatile("" : "+v"(y2));
                y3 = __builtin_fma(y3, b, a); asm volatile("" : "+v"(y3));
            }
            if (MODE == 5) {                                                                // 4 independent chains of adds
                x = x + b; asm volatile("" : "+v"(x));
                y0 = y0 + b; asm volatile("" : "+v"(y0));
                y1 = y1 + b; asm volatile("" : "+v"(y1));
                y2 = y2 + b; asm volatile("" : "+v"(y2));
            }
            if (MODE == 6) { x = __builtin_amdgcn_rsq(x); asm volatile("" : "+v"(x)); }     // dependent rsq
            if (MODE == 7) {                                                                // independent rsq x4
                y0 = __builtin_amdgcn_rsq(y0); asm volatile("" : "+v"(y0));
                y1 = __builtin_amdgcn_rsq(y1); asm volatile("" : "+v"(y1));
                y2 = __builtin_amdgcn_rsq(y2); asm volatile("" : "+v"(y2));
                y3 = __builtin_amdgcn_rsq(y3); asm volatile("" : "+v"(y3));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x + y0 + y1 + y2 + y3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
int run(const char *name, int ops_per_iter, int waves_per_simd) {
    double *out; long long *cyc;
    const int blocks = 256 * 4 * waves_per_simd;   // one wave per block
    CHECK(hipMalloc(&out, sizeof(double) * 64 * blocks));
    CHECK(hipMalloc(&cyc, sizeof(long long)));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, cyc, 1.0, 1e-9, 10);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    k<MODE><<<blocks, 64>>>(out, cyc, 1.0, 1e-9, iters);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; CHECK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    double n = (double)iters * 64;
    printf("%-28s waves/SIMD %d: %.2f ns per group (%d ops), s_memtime ticks/group %.2f\n", name, waves_per_simd,
           ms * 1e6 / n, ops_per_iter, (double)c / n);
    hipFree(out); hipFree(cyc);
    return 0;
}
int main() {
    for (int w : {1, 2}) {
        run<0>("dep add", 1, w);
        run<1>("dep fma", 1, w);
        run<2>("dep add + 1 indep fma", 2, w);
        run<3>("dep add + 2 indep fma", 3, w);
        run<4>("dep add + 4 indep fma", 5, w);
        run<5>("4 indep add chains", 4, w);
        run<6>("dep rsq", 1, w);
        run<7>("4 indep rsq", 4, w);
    }
    return 0;
}
