// micro-benchmark: one wave summing 64-double LDS rows with a dependent v_add_f64 chain; how should the
// ds_read_b128 be interleaved with the adds? (one wave per CU, s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int tiles) {
    __shared__ __attribute__((aligned(16))) double C[3][48 * 66];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3 * 48 * 66; i += 64) (&C[0][0])[i] = 1e-3 * i;
    __syncthreads();
    const int ch = lane < 48 ? lane : 47;
    double acc = 0.0;
    double2 q[32];
    long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {          // 8 reads then 16 adds, chunked (chain_full_pf)
        const double *row0 = &C[0][ch * 66];
#pragma unroll
        for (int k = 0; k < 16; ++k) q[k] = *reinterpret_cast<const double2 *>(row0 + 2 * k);
        for (int t = 0; t < tiles; ++t) {
            const double *row = &C[t % 3][ch * 66], *rn = &C[(t + 1) % 3][ch * 66];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[16 + k] = *reinterpret_cast<const double2 *>(row + 32 + 2 * k);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc += q[k].x; acc += q[k].y; }
#pragma unroll
            for (int k = 0; k < 8; ++k) q[24 + k] = *reinterpret_cast<const double2 *>(row + 48 + 2 * k);
#pragma unroll
            for (int k = 8; k < 16; ++k) { acc += q[k].x; acc += q[k].y; }
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = *reinterpret_cast<const double2 *>(rn + 2 * k);
#pragma unroll
            for (int k = 16; k < 24; ++k) { acc += q[k].x; acc += q[k].y; }
#pragma unroll
            for (int k = 0; k < 8; ++k) q[8 + k] = *reinterpret_cast<const double2 *>(rn + 16 + 2 * k);
#pragma unroll
            for (int k = 24; k < 32; ++k) { acc += q[k].x; acc += q[k].y; }
        }
    }
    if (MODE == 1) {          // fine interleave: one read (16 elements ahead) after every 2 adds, pinned
        const double *row0 = &C[0][ch * 66];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = *reinterpret_cast<const double2 *>(row0 + 2 * k);
        for (int t = 0; t < tiles; ++t) {
            const double *row = &C[t % 3][ch * 66], *rn = &C[(t + 1) % 3][ch * 66];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int kn = k + 8;                       // element pair read now, used 16 adds later
                if (kn < 32) q[kn % 16 + 0] = q[kn % 16 + 0];
                double2 nv = kn < 32 ? *reinterpret_cast<const double2 *>(row + 2 * kn)
                                     : *reinterpret_cast<const double2 *>(rn + 2 * (kn - 32));
                acc += q[k % 16].x; asm volatile("" : "+v"(acc));
                acc += q[k % 16].y; asm volatile("" : "+v"(acc));
                q[(k + 8) % 16] = nv;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (MODE == 2) {          // adds only (no LDS): the floor
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { acc += 1e-9; asm volatile("" : "+v"(acc)); }
        }
    }
    if (MODE == 3) {          // reads only
        for (int t = 0; t < tiles; ++t) {
            const double *row = &C[t % 3][ch * 66];
#pragma unroll
            for (int k = 0; k < 32; ++k) q[k] = *reinterpret_cast<const double2 *>(row + 2 * k);
#pragma unroll
            for (int k = 0; k < 32; ++k) asm volatile("" :: "v"(q[k].x), "v"(q[k].y));
        }
    }
    if (MODE >= 10) {         // 32 ds_read_b128 with only (MODE-10)*16 lanes active + adds (chunked)
        if (lane < (MODE - 10) * 16) {
            const double *row0 = &C[0][ch * 66];
#pragma unroll
            for (int k = 0; k < 16; ++k) q[k] = *reinterpret_cast<const double2 *>(row0 + 2 * k);
            for (int t = 0; t < tiles; ++t) {
                const double *row = &C[t % 3][ch * 66], *rn = &C[(t + 1) % 3][ch * 66];
#pragma unroll
                for (int k = 0; k < 8; ++k) q[16 + k] = *reinterpret_cast<const double2 *>(row + 32 + 2 * k);
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc += q[k].x; acc += q[k].y; }
#pragma unroll
                for (int k = 0; k < 8; ++k) q[24 + k] = *reinterpret_cast<const double2 *>(row + 48 + 2 * k);
#pragma unroll
                for (int k = 8; k < 16; ++k) { acc += q[k].x; acc += q[k].y; }
#pragma unroll
                for (int k = 0; k < 8; ++k) q[k] = *reinterpret_cast<const double2 *>(rn + 2 * k);
#pragma unroll
                for (int k = 16; k < 24; ++k) { acc += q[k].x; acc += q[k].y; }
#pragma unroll
                for (int k = 0; k < 8; ++k) q[8 + k] = *reinterpret_cast<const double2 *>(rn + 16 + 2 * k);
#pragma unroll
                for (int k = 24; k < 32; ++k) { acc += q[k].x; acc += q[k].y; }
            }
        }
    }
    if (MODE == 5) {          // ds_read_b64: 64 reads only
        for (int t = 0; t < tiles; ++t) {
            const double *row = &C[t % 3][ch * 66];
            double r[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) r[k] = row[k];
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("" :: "v"(r[k]));
        }
    }
    if (MODE == 4) {          // ds_read_b64 x64 interleaved 1:1 with adds (8 ahead)
        for (int t = 0; t < tiles; ++t) {
            const double *row = &C[t % 3][ch * 66];
            double r[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) r[k] = row[k];
#pragma unroll
            for (int k = 0; k < 64; ++k) { acc += r[k]; }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc + q[0].x;
    if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
int run(const char *name) {
    double *out; long long *cyc;
    CHECK(hipMalloc(&out, sizeof(double) * 64 * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long)));
    const int tiles = 3000;
    k<MODE><<<256, 64>>>(out, cyc, 10);
    CHECK(hipDeviceSynchronize());
    k<MODE><<<256, 64>>>(out, cyc, tiles);
    CHECK(hipDeviceSynchronize());
    long long c; CHECK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    printf("%-44s %.0f cycles per 64-element tile\n", name, (double)c / tiles);
    hipFree(out); hipFree(cyc);
    return 0;
}
int main() {
    run<2>("64 dependent adds only");
    run<3>("32 ds_read_b128 only");
    run<0>("chunked: 8 reads / 16 adds");
    run<1>("fine interleave: 1 read per 2 adds (pinned)");
    run<5>("64 ds_read_b64 only");
    run<11>("chunked, 16 lanes active");
    run<12>("chunked, 32 lanes active");
    run<13>("chunked, 48 lanes active");
    run<14>("chunked, 64 lanes active");
    return 0;
}
