// microbenchmark: cost of the pair "finish" arithmetic (in-range sqrt + reciprocal sequences, 22 f64 VALU per body) for
// NB bodies per wave, one wave per SIMD (1024 blocks) or two, compiler schedule vs explicit stage-major interleave.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pairmix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double sqrt_inrange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double rcp_inrange(double p) {
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0);
    return __builtin_fma(e, r, r);
}
#define ANCHOR(x) asm volatile("" : "+v"(x))

template <int NB, int MODE>
__global__ void __launch_bounds__(64) k(int iters, const double *in, double *out, long long *cyc) {
    double xi[NB], yi[NB], zi[NB], acc = 0.0;
    for (int b = 0; b < NB; ++b) { xi[b] = in[b * 3]; yi[b] = in[b * 3 + 1]; zi[b] = in[b * 3 + 2]; }
    double px = in[30] + threadIdx.x, py = in[31], pz = in[32], mu = in[33];
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        px += 1e-3;
        double dx[NB], dy[NB], dz[NB], n2[NB], c[3 * NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            dx[b] = px - xi[b]; dy[b] = py - yi[b]; dz[b] = pz - zi[b];
            n2[b] = dx[b] * dx[b] + dy[b] * dy[b] + dz[b] * dz[b];
        }
        if (MODE == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double inv = rcp_inrange(n2[b] * sqrt_inrange(n2[b]));
                const double s = mu * inv;
                c[3 * b] = dx[b] * s; c[3 * b + 1] = dy[b] * s; c[3 * b + 2] = dz[b] * s;
            }
        } else {   // stage-major: every stage of the sequences for all NB bodies before the next stage
            double y[NB], g[NB], h[NB], r[NB], d[NB], p[NB], q[NB], e[NB];
#define STAGE(stmt) _Pragma("unroll") for (int b = 0; b < NB; ++b) { stmt; } _Pragma("unroll") for (int b = 0; b < NB; ++b) { ANCHOR(g[b]); ANCHOR(h[b]); }
#pragma unroll
            for (int b = 0; b < NB; ++b) { y[b] = __builtin_amdgcn_rsq(n2[b]); g[b] = 0; h[b] = 0; }
            STAGE(g[b] = n2[b] * y[b]; h[b] = y[b] * 0.5)
            STAGE(r[b] = __builtin_fma(-h[b], g[b], 0.5); ANCHOR(r[b]))
            STAGE(g[b] = __builtin_fma(g[b], r[b], g[b]); h[b] = __builtin_fma(h[b], r[b], h[b]))
            STAGE(d[b] = __builtin_fma(-g[b], g[b], n2[b]); ANCHOR(d[b]))
            STAGE(g[b] = __builtin_fma(d[b], h[b], g[b]))
            STAGE(d[b] = __builtin_fma(-g[b], g[b], n2[b]); ANCHOR(d[b]))
            STAGE(g[b] = __builtin_fma(d[b], h[b], g[b]))
            STAGE(p[b] = n2[b] * g[b]; ANCHOR(p[b]))
            STAGE(q[b] = __builtin_amdgcn_rcp(p[b]); ANCHOR(q[b]))
            STAGE(e[b] = __builtin_fma(-p[b], q[b], 1.0); ANCHOR(e[b]))
            STAGE(q[b] = __builtin_fma(q[b], e[b], q[b]); ANCHOR(q[b]))
            STAGE(e[b] = __builtin_fma(-p[b], q[b], 1.0); ANCHOR(e[b]))
            STAGE(q[b] = __builtin_fma(q[b], e[b], q[b]); ANCHOR(q[b]))
            STAGE(e[b] = __builtin_fma(-p[b], q[b], 1.0); ANCHOR(e[b]))
            STAGE(q[b] = __builtin_fma(e[b], q[b], q[b]); ANCHOR(q[b]))
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double s = mu * q[b];
                c[3 * b] = dx[b] * s; c[3 * b + 1] = dy[b] * s; c[3 * b + 2] = dz[b] * s;
            }
        }
#pragma unroll
        for (int q2 = 0; q2 < 3 * NB; ++q2) acc += c[q2];     // (stands in for the LDS writes; 3 adds per body)
        if (MODE == 2) {                                       // idle about as long as the arithmetic took
#pragma unroll
            for (int z = 0; z < 12; ++z) __builtin_amdgcn_s_sleep(1);   // 12 x 64 cycles
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 5) cyc[0] = c1 - c0;
}

template <int NB, int MODE>
void run(const char *name, int blocks, const double *din, double *dout, long long *dcyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NB, MODE>), dim3(blocks), dim3(64), 0, 0, iters, din, dout, dcyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB, MODE>), dim3(blocks), dim3(64), 0, 0, iters, din, dout, dcyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s NB=%d blocks=%4d: %7.1f cycles per tile, %6.1f per body, kernel %.3f ms (clock %.2f GHz)\n", name, NB, blocks,
           (double)cyc / iters, (double)cyc / iters / NB, ms, cyc / (ms * 1e6));
}
int main() {
    std::vector<double> h(64, 0.0);
    for (int i = 0; i < 30; ++i) h[i] = 1.0 + 0.37 * i;
    h[30] = 100.0; h[31] = 50.0; h[32] = -20.0; h[33] = 2.5;
    double *din, *dout; long long *dcyc;
    hipMalloc(&din, 64 * 8); hipMalloc(&dout, 4096 * 64 * 8); hipMalloc(&dcyc, 8);
    hipMemcpy(din, h.data(), 64 * 8, hipMemcpyHostToDevice);
    run<5, 0>("compiler order", 1024, din, dout, dcyc);
    run<5, 1>("stage-major", 1024, din, dout, dcyc);
    run<4, 0>("compiler order", 1024, din, dout, dcyc);
    run<4, 1>("stage-major", 1024, din, dout, dcyc);
    run<2, 0>("compiler order", 1024, din, dout, dcyc);
    run<5, 0>("compiler order, 2 waves/SIMD", 2048, din, dout, dcyc);
    run<5, 1>("stage-major, 2 waves/SIMD", 2048, din, dout, dcyc);
    run<5, 0>("compiler order, 1/4 chip", 256, din, dout, dcyc);
    run<5, 2>("50% duty, 1 wave/SIMD", 1024, din, dout, dcyc);
    run<5, 2>("50% duty, 2 waves/SIMD", 2048, din, dout, dcyc);
    run<5, 2>("50% duty, 3 waves/SIMD", 3072, din, dout, dcyc);
    run<5, 2>("50% duty, 4 waves/SIMD", 4096, din, dout, dcyc);
    return 0;
}
