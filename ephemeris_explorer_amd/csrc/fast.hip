// fast.hip -- the OPT-IN paths of the massive-body step that give up the reference's summation order (EPH_PATH_FAST), its IEEE
// square root and division as well (EPH_PATH_FAST_RSQ), or f64 pair arithmetic (EPH_PATH_F32_PAIRS, BASELINE configs[4]'s
// precision). They exist to measure what bit-exactness costs; the default path stays the ordered one (step_wg.hip, step_wave.hip).
// Compiled once per evaluation order of the point-mass term (pair_ns.h).
#include <algorithm>
#include <cstdlib>

#include "pair_ns.h"

namespace eph {
namespace EPH_PV_NS {

// ------------------------------------------------------------------------------------------------------
// OPT-IN FAST PATH (eph_nbody_set_path(.., EPH_PATH_FAST)): the same pair arithmetic (IEEE sqrt / divide, no
// contraction), but NOT the reference's summation order -- SURVEY §7 "hard parts", north_star's "tile-parallel
// partial sums". It exists to measure what bit-exactness costs; the default path stays the ordered one.
//
// Work split: lane = target body (a block of 64 consecutive bodies per wave), every lane of a wave works on the SAME
// source body, fetched with scalar loads (s_load_dwordx8 through the constant address space: no LDS, no vector
// loads, no transposition in the loop). The sources are cut into S slices; wave (block, slice) accumulates its
// slice in source order into three registers per lane. A workgroup = 4 slices of one block (one wave per SIMD).
// The S partial sums of a body are combined in slice order
// by a second small launch (k_fast_finish), which also does the Cowell velocity, the solout sample and the
// predictor -- deterministic: the value never depends on which wave finishes first.
//   a_i = ((p_0 + p_1) + ... + p_{S-1}),  p_s = ((0 + c(i, j0)) + c(i, j0 + 1)) + ...   (j over slice s, j != i)
// ------------------------------------------------------------------------------------------------------
constexpr int kFastWaves = 4;                          // waves (= slices) per workgroup
constexpr int kFastMaxSlices = 64;

// 1 / r^3 WITHOUT the IEEE square root and division (EPH_PATH_FAST_RSQ): y = v_rsq_f64(n2) refined by two Newton steps
// (relative error ~1e-16, not correctly rounded), then y * y * y. 15 VALU operations instead of 22 and one
// transcendental instead of two. n2 = 0 (the body itself) gives NaN here too; the caller masks that source.
__device__ __forceinline__ double inv_r3_approx(double n2) {
    double y = __builtin_amdgcn_rsq(n2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double a = n2 * y;
        const double r = __builtin_fma(-a, 0.5 * y, 0.5);      // 0.5 * (1 - n2 * y^2)
        y = __builtin_fma(y, r, y);
    }
    return y * y * y;
}
template <bool DIAG, int kFastUnroll, bool APPROX>
__device__ __forceinline__ void fast_slice(const __attribute__((address_space(4))) Body4 *src, int j0, int j1, int n, int i,
                                           double xi, double yi, double zi, double &ax, double &ay, double &az) {
    auto fetch = [&](int j, Body4 (&p)[kFastUnroll]) {
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) { p[u].x = src[j + u].x; p[u].y = src[j + u].y; p[u].z = src[j + u].z; p[u].mu = src[j + u].mu; }
    };
    Body4 nxt[kFastUnroll];
    fetch(j0, nxt);
    for (int j = j0; j < j1; j += kFastUnroll) {       // j1 - j0 is a multiple of kFastUnroll; sources >= n are padding
        Body4 pj[kFastUnroll];
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) pj[u] = nxt[u];
        fetch(min(j + kFastUnroll, j1 - kFastUnroll), nxt);   // next group's scalar loads in flight under this one's arithmetic
        PairPre pre[kFastUnroll];
        unsigned worst = 0u;
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) {
            pre[u] = pair_pre(xi, yi, zi, pj[u]);
            worst = max(worst, range_key(pre[u].n2));
        }
        double c[3 * kFastUnroll];
        if (APPROX) {
#pragma unroll
            for (int u = 0; u < kFastUnroll; ++u) {
                const double sc = pj[u].mu * inv_r3_approx(pre[u].n2);
                c[3 * u] = pre[u].dx * sc; c[3 * u + 1] = pre[u].dy * sc; c[3 * u + 2] = pre[u].dz * sc;
            }
        } else if (__builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0) {
#pragma unroll
            for (int u = 0; u < kFastUnroll; ++u) pair_finish<true>(pre[u], pj[u].mu, c[3 * u], c[3 * u + 1], c[3 * u + 2]);
        } else {
#pragma unroll
            for (int u = 0; u < kFastUnroll; ++u) pair_finish<false>(pre[u], pj[u].mu, c[3 * u], c[3 * u + 1], c[3 * u + 2]);
        }
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) {
            if (DIAG && j + u == i) continue;          // the body itself (n2 = 0 -> NaN): not a source
            // padding rows are zeros at the ORIGIN with mu = 0: a real body sitting exactly there (the central body of a
            // heliocentric system) would get n2 = 0 -> 0 * inf = NaN from them. Wave-uniform test, last slice only.
            if (j + u >= n) continue;
            ax = ax + c[3 * u];
            ay = ay + c[3 * u + 1];
            az = az + c[3 * u + 2];
        }
    }
}

// partial: [S][3][npad] scratch. Two launches per step: the kernel boundary is the release/acquire between the
// slice sums and their combination. (First version: one launch with a per-block arrival ticket, the last workgroup
// of a block combining -- measured 66 / 96 / 166 us per step at 16 / 32 / 64 slices, N = 4096: the agent-scope
// fence each workgroup needs before its ticket costs ~0.13 us and they serialise; gpurun_out r02a.)
template <int UNROLL, bool APPROX>
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_partial(int n, int npad, const Body4 *__restrict__ pos,
                                                                  int S, int slice_len, double *__restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs_per_block = S / kFastWaves;
    const int block = blockIdx.x / wgs_per_block;
    const int slice = (blockIdx.x % wgs_per_block) * kFastWaves + wave;
    const int i = block * 64 + lane;
    const int ic = min(i, n - 1);
    const auto *src = (const __attribute__((address_space(4))) Body4 *)(unsigned long long)pos;
    const double xi = pos[ic].x, yi = pos[ic].y, zi = pos[ic].z;
    const int j0 = slice * slice_len, j1 = min(j0 + slice_len, npad);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j0 < j1) {
        if (j0 < block * 64 + 64 && j1 > block * 64) fast_slice<true, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        else fast_slice<false, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
    }
    double *pp = partial + (size_t)slice * 3 * npad + i;
    pp[0] = ax;
    pp[(size_t)npad] = ay;
    pp[(size_t)2 * npad] = az;
}

// ------------------------------------------------------------------------------------------------------
// OPT-IN MIXED PRECISION (eph_nbody_set_path(.., EPH_PATH_F32_PAIRS); BASELINE.json configs[4] "65 536-body f32 system"):
// the pair arithmetic in binary32 -- differences of positions rounded to f32, n2 by fma, v_rsq_f32 + one Newton step,
// ((mu y) y) y, the three products -- two sources at a time in the packed f32 instructions (v_pk_add / v_pk_mul /
// v_pk_fma_f32: the only VALU form that runs at twice the f64 rate), the contributions of FOUR consecutive sources summed in
// binary32, converted and ACCUMULATED in f64 in the fast path's slice order; Cowell, predictor and the whole integrator state
// stay f64 (a twelfth-order multistep recurrence cannot hold its state in binary32). The reference has no
// f32 path (ephemeris/src/propagators/nbody.rs:13,19): no parity claim, never the default, for large systems only.
// Supported magnitudes: positions and mu must be representable in binary32 (|x| < 3.4e38, mu >= 1.2e-38 or 0); separations from
// 0 (two bodies coinciding after rounding: their mutual term is dropped) to 1.8e19 length units (beyond: the term is ~0) --
// the reference's km and N-body units are far inside, SI metres at heliocentric distances (1e12-1e13) still are.
// ------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
struct BodyF { float x, y, z, mu; };
__global__ void __launch_bounds__(256) k_pos_to_f32(int n, int npad, const Body4 *__restrict__ pos, BodyF *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    BodyF b{0.f, 0.f, 0.f, 0.f};
    if (i < n) { const Body4 p = pos[i]; b = BodyF{(float)p.x, (float)p.y, (float)p.z, (float)p.mu}; }
    out[i] = b;
}
template <bool MASKED>
__device__ __forceinline__ void f32_slice(const __attribute__((address_space(4))) BodyF *src, int j0, int j1, int n, int i,
                                          float xi, float yi, float zi, double &ax, double &ay, double &az) {
    constexpr int U = 4;                               // sources per iteration: two packed pairs
    const v2f x2{xi, xi}, y2{yi, yi}, z2{zi, zi};
    for (int j = j0; j < j1; j += U) {                 // j1 - j0 is a multiple of U; sources >= n are padding
        BodyF p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { p[u].x = src[j + u].x; p[u].y = src[j + u].y; p[u].z = src[j + u].z; p[u].mu = src[j + u].mu; }
        v2f cx[2], cy[2], cz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const BodyF &pa = p[2 * h], &pb = p[2 * h + 1];
            const v2f dx = v2f{pa.x, pb.x} - x2, dy = v2f{pa.y, pb.y} - y2, dz = v2f{pa.z, pb.z} - z2;
            v2f n2 = dx * dx;
            n2 = __builtin_elementwise_fma(dy, dy, n2);
            n2 = __builtin_elementwise_fma(dz, dz, n2);
            // two bodies whose positions coincide after rounding to binary32 (n2 = 0) and separations beyond 1.8e19 length units
            // (n2 = inf) would turn the sum into NaN: clamped (one v_med3_f32 each), their terms come out as 0 resp. negligible
            n2 = v2f{__builtin_amdgcn_fmed3f(n2.x, 1.0e-37f, 3.0e38f), __builtin_amdgcn_fmed3f(n2.y, 1.0e-37f, 3.0e38f)};
            v2f y{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)};
            const v2f hn = n2 * v2f{0.5f, 0.5f};
            const v2f r = __builtin_elementwise_fma(-(hn * y), y, v2f{0.5f, 0.5f});   // 0.5 (1 - n2 y^2)
            y = __builtin_elementwise_fma(y, r, y);
            // (mu y) y y, not mu (y y y): y^3 alone leaves binary32's normal range for separations above 2e12 length units
            const v2f sc = ((v2f{pa.mu, pb.mu} * y) * y) * y;
            cx[h] = dx * sc; cy[h] = dy * sc; cz[h] = dz * sc;
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((j + u == i) || (j + u >= n)) {    // the body itself (its clamped term is not a source); padding rows
                    cx[u >> 1][u & 1] = 0.0f; cy[u >> 1][u & 1] = 0.0f; cz[u >> 1][u & 1] = 0.0f;
                }
            }
        }
        // the four contributions summed in binary32 (their own rounding is 2^-24 of each term, as is the sum's), ONE conversion and
        // one f64 addition per component and four sources: the f64 side was a third of the loop's issue cycles (6 cvt + 6 adds per two)
        const v2f sx = cx[0] + cx[1], sy = cy[0] + cy[1], sz = cz[0] + cz[1];
        ax = ax + (double)(sx.x + sx.y);
        ay = ay + (double)(sy.x + sy.y);
        az = az + (double)(sz.x + sz.y);
    }
}
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_partial_f32(int n, int npad, const BodyF *__restrict__ posf, int S,
                                                                      int slice_len, double *__restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs_per_block = S / kFastWaves;
    const int block = blockIdx.x / wgs_per_block;
    const int slice = (blockIdx.x % wgs_per_block) * kFastWaves + wave;
    const int i = block * 64 + lane;
    const int ic = min(i, n - 1);
    const auto *src = (const __attribute__((address_space(4))) BodyF *)(unsigned long long)posf;
    const float xi = posf[ic].x, yi = posf[ic].y, zi = posf[ic].z;
    const int j0 = slice * slice_len, j1 = min(j0 + slice_len, npad);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j0 < j1) {
        if ((j0 < block * 64 + 64 && j1 > block * 64) || j1 > n) f32_slice<true>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        else f32_slice<false>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
    }
    double *pp = partial + (size_t)slice * 3 * npad + i;
    pp[0] = ax;
    pp[(size_t)npad] = ay;
    pp[(size_t)2 * npad] = az;
}

// thread per (component, body): partial sums combined in slice order, then the rest of the fused step
template <int L>
__global__ void __launch_bounds__(256) k_fast_finish(const LmArgs a, int S, const double *__restrict__ partial) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * a.npad) return;
    const int cc = t / a.npad, my_i = t % a.npad;               // consecutive threads = consecutive bodies: coalesced
    if (my_i >= a.n) return;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + my_i;
    double yv[L], av[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = a.Y[slot * lvl + off];
        av[j] = j > 0 ? a.A[slot * lvl + off] : 0.0;
    }
    // all loads of a group of 16 slices in flight before the ordered adds (one load per add would pay the memory
    // latency S times: measured 11 us for this kernel at S = 32)
    double anew = 0.0;
    for (int base = 0; base < S; base += 16) {
        double pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = base + u < S ? partial[(size_t)(base + u) * lvl + off] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (base + u < S) anew = anew + pv[u];
    }
    a.A[(size_t)a.cur * lvl + off] = anew;
    {
        double prev[L];
#pragma unroll
        for (int j = 0; j < L - 1; ++j) prev[j] = av[j + 1];
        prev[L - 1] = 0.0;
        a.V[off] = lm_cowell<L>(anew, prev, yv[0], yv[1], a.cw, a.h, a.hc);
    }
    maybe_sample(a.samp, my_i, cc, a.step, yv[0]);
    if (a.do_predict) {
        av[0] = anew;
        const double ynext = lm_predict<L>(yv, av, a.wa, a.wb, a.hh);
        const int nslot = (a.cur + L - 1) % L;
        a.Y[(size_t)nslot * lvl + off] = ynext;
        reinterpret_cast<double *>(a.pos_next + my_i)[cc] = ynext;
    }
}


int lm_step_fast(hipStream_t s, const LmArgs &a, double *partial, int S, int unroll, bool approx, float *posf) {
    int slice_len = (a.npad + S - 1) / S;
    const int un = approx ? 4 : unroll;
    slice_len = (slice_len + un - 1) / un * un;
    const dim3 pgrid((unsigned)(a.npad / 64 * (S / kFastWaves))), pblock(64 * kFastWaves);
    if (posf) {                                                         // EPH_PATH_F32_PAIRS
        BodyF *pf = reinterpret_cast<BodyF *>(posf);
        hipLaunchKernelGGL(k_pos_to_f32, dim3((unsigned)((a.npad + 255) / 256)), dim3(256), 0, s, a.n, a.npad, a.pos_cur, pf);
        hipLaunchKernelGGL(k_fast_partial_f32, pgrid, pblock, 0, s, a.n, a.npad, (const BodyF *)pf, S, slice_len, partial);
    } else if (approx)
        hipLaunchKernelGGL((k_fast_partial<4, true>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    else if (unroll == 8 && a.npad % 8 == 0)
        hipLaunchKernelGGL((k_fast_partial<8, false>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    else
        hipLaunchKernelGGL((k_fast_partial<4, false>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    const dim3 grid((3 * a.npad + 255) / 256), block(256);
    if (a.L == 12) hipLaunchKernelGGL(k_fast_finish<12>, grid, block, 0, s, a, S, partial);
    else if (a.L == 13) hipLaunchKernelGGL(k_fast_finish<13>, grid, block, 0, s, a, S, partial);
    else return EPH_ERR_UNSUPPORTED;
    return launched("k_fast_partial / k_fast_finish");
}

}  // namespace EPH_PV_NS
}  // namespace eph
