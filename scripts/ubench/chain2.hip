// micro-benchmark: can a single wave overlap its ds_read_b128 with a dependent v_add_f64 chain if the instruction order
// is pinned in assembly? One wave per CU, 48 chains of 64 doubles per tile from LDS (row stride 66 doubles), cycles by
// s_memtime and wall time by HIP events (the two do not tick alike: see profiles/r02_step_kernel_evidence.md).
// The tile bodies are generated (scripts/ubench/gen_chain2.py) into chain2_*.inc.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int tiles, const double *gbuf) {
    __shared__ __attribute__((aligned(16))) double C[3][48 * 66];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3 * 48 * 66; i += 64) (&C[0][0])[i] = 1e-3 * i;
    __syncthreads();
    const int ch = lane < 48 ? lane : 47;
    double acc = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        const unsigned addr = (unsigned)(size_t)(&C[t % 3][ch * 66]);   // LDS byte address of this lane's row
        if (MODE == 0) {
#include "chain2_r8.inc"
        } else if (MODE == 1) {
#include "chain2_r4.inc"
        } else if (MODE == 2) {
#include "chain2_r16.inc"
        } else if (MODE == 3) {
#include "chain2_m8.inc"
        } else if (MODE == 4) {
#include "chain2_adds.inc"
        } else if (MODE == 5) {
#include "chain2_reads.inc"
        } else if (MODE == 6) {
#include "chain2_b64.inc"
        } else {
            // the same rows in global memory (27 KB per CU: L1 / L2 resident), address = this lane's row of tile t % 3
            const double *addr = gbuf + ((size_t)blockIdx.x * 3 + t % 3) * 48 * 66 + ch * 66;
            if (MODE == 7) {
#include "chain2_g16.inc"
            } else {
#include "chain2_greads.inc"
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
int run(const char *name) {
    double *out; long long *cyc;
    CHECK(hipMalloc(&out, sizeof(double) * 64 * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long)));
    const int tiles = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double *gbuf;
    CHECK(hipMalloc(&gbuf, sizeof(double) * 256 * 3 * 48 * 66));
    CHECK(hipMemset(gbuf, 0, sizeof(double) * 256 * 3 * 48 * 66));
    k<MODE><<<256, 64>>>(out, cyc, 10, gbuf);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE><<<256, 64>>>(out, cyc, tiles, gbuf);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CHECK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    printf("%-58s %6.0f s_memtime ticks, %6.1f ns per 64-element tile\n", name, (double)c / tiles, ms * 1e6 / tiles);
    hipFree(out); hipFree(cyc);
    return 0;
}
int main() {
    run<4>("64 dependent v_add_f64 only");
    run<5>("32 ds_read_b128 only");
    run<0>("asm: add, add, read; ring of 8 reads in flight");
    run<1>("asm: add, add, read; ring of 4");
    run<2>("asm: add, add, read; ring of 16");
    run<3>("asm: add, read, add (hi half copied out); ring of 8");
    run<6>("asm: 64 ds_read_b64, add, read; ring of 16");
    run<8>("32 global_load_dwordx4 only (L1/L2 resident rows)");
    run<7>("asm: add, add, global_load; ring of 16");
    return 0;
}
