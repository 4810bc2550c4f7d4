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
        // the three keys pair_finish<true> requires its caller to have wave-tested (pair_term.h): n2 in range, and -- for the division
        // forms -- the smallest |d_c| and mu inside the wrapper-free division's range, as step_wave.hip and step_wg.hip fold them
        unsigned worst = 0u, low = ~0u;
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) {
            pre[u] = pair_pre(xi, yi, zi, pj[u]);
            worst = max(worst, max(range_key(pre[u].n2), mu_key(pj[u].mu)));
            low = min(low, pre[u].lo);
        }
        worst = max(worst, low_key(low));
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

// EPH_PATH_FAST_RSQ, round 5: the path that promises neither the reference's order NOR its IEEE operations is written the way an
// unconstrained kernel would be -- every multiply-add fused, 1 / r^3 from v_rsq_f64 and ONE third-order correction
// (e = 1 - n2 y^2; y <- y (1 + e/2 + 3 e^2/8): seed error 2^-23 -> 2^-68), 17 f64 operations + the transcendental per
// interaction instead of 25 + 1 unfused. (Round 4's form measured 28.3 us per step at N = 4096 against the exact path's 36: "what
// bit-exactness costs" looked like 20 %; it flattered the exact path)
template <bool DIAG, bool PAD, int U>
__device__ __forceinline__ void fast_slice_rsq(const __attribute__((address_space(4))) Body4 *src, int j0, int j1, int n, int i,
                                               double xi, double yi, double zi, double &ax, double &ay, double &az) {
    auto fetch = [&](int j, Body4 (&p)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) { p[u].x = src[j + u].x; p[u].y = src[j + u].y; p[u].z = src[j + u].z; p[u].mu = src[j + u].mu; }
    };
    Body4 nxt[U];
    fetch(j0, nxt);
    for (int j = j0; j < j1; j += U) {                 // j1 - j0 is a multiple of U; sources >= n are padding
        Body4 pj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pj[u] = nxt[u];
        fetch(min(j + U, j1 - U), nxt);                // next group's scalar loads in flight under this one's arithmetic
        double dx[U], dy[U], dz[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dx[u] = pj[u].x - xi; dy[u] = pj[u].y - yi; dz[u] = pj[u].z - zi;
            const double n2 = __builtin_fma(dz[u], dz[u], __builtin_fma(dy[u], dy[u], dx[u] * dx[u]));
            const double y0 = __builtin_amdgcn_rsq(n2);
            const double e = __builtin_fma(-(n2 * y0), y0, 1.0);
            const double c = __builtin_fma(e, 0.375, 0.5) * e;
            y[u] = __builtin_fma(y0, c, y0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double sc = (pj[u].mu * y[u]) * (y[u] * y[u]);
            if (DIAG) sc = (j + u == i) ? 0.0 : sc;      // the body itself (n2 = 0 -> NaN): not a source
            if (PAD && j + u >= n) continue;            // padding rows (wave-uniform; only the slice that reaches past n is compiled with it)
            ax = __builtin_fma(dx[u], sc, ax);
            ay = __builtin_fma(dy[u], sc, ay);
            az = __builtin_fma(dz[u], sc, az);
        }
        // (pinning the prefetched group at the end of the trip -- the compiler sinks its scalar loads to the top of the next one --
        // was measured: no change, four waves per SIMD cover the scalar-cache latency)
    }
}

// partial: [S][3][npad] scratch. The TWO-launch form (EPH_FAST_FUSED=0): the kernel boundary is the release/acquire between the
// slice sums and their combination (k_fast_finish). The one-launch form is k_fast_step below.
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
        const bool diag = j0 < block * 64 + 64 && j1 > block * 64;
        if constexpr (APPROX) {
            const bool pad = j1 > n;                    // wave-uniform
            if (diag && pad) fast_slice_rsq<true, true, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else if (diag) fast_slice_rsq<true, false, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else if (pad) fast_slice_rsq<false, true, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else fast_slice_rsq<false, false, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        } else {
            if (diag) fast_slice<true, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else fast_slice<false, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        }
    }
    double *pp = partial + (size_t)slice * 3 * npad + i;
    pp[0] = ax;
    pp[(size_t)npad] = ay;
    pp[(size_t)2 * npad] = az;
}

// ------------------------------------------------------------------------------------------------------
// OPT-IN MIXED PRECISION (eph_nbody_set_path(.., EPH_PATH_F32_PAIRS); BASELINE.json configs[4] "65 536-body f32 system"):
// the pair arithmetic in binary32 -- differences of positions rounded to f32, n2 by fma, v_rsq_f32 (1 ulp; a Newton step on top
// bought nothing measurable: the error of this path is the rounding of the POSITIONS to binary32, 1.3e-6 of the accelerations
// with and without it), ((mu y) y) y -- two sources at a time in the packed f32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32:
// the only VALU form that runs at twice the f64 rate), the contributions of 32 consecutive sources accumulated by packed fma
// in binary32 (even sources in one half, odd ones in the other), then converted and ACCUMULATED in f64 in the fast path's slice
// order; Cowell, predictor and the whole integrator state stay f64 (a twelfth-order multistep recurrence cannot hold its state in
// binary32). The reference has no f32 path (ephemeris/src/propagators/nbody.rs:13,19): no parity claim,
// never the default, for large systems only.
// Supported magnitudes: positions and mu must be representable in binary32 (|x| < 3.4e38, mu >= 1.2e-38 or 0). Two bodies that
// coincide after rounding to binary32 (n2 = 0, y = inf): mu y^3 is clamped to the largest finite value and multiplies a zero
// separation -- their mutual term is dropped, and so is a body's term with itself (no source is masked); a massless source there
// (0 * inf = NaN) leaves the clamp as 0 (v_med3_f32 returns the minimum of the other two for a NaN). Separations beyond 1.8e19 length units (n2 = inf, y = 0): the term is 0. The
// reference's km and N-body units are far inside, SI metres at heliocentric distances (1e12-1e13) still are.
// Layout of the binary32 copy: sources in PAIRS, {x_a, x_b, y_a, y_b, z_a, z_b, mu_a, mu_b} -- the operand pairs of the packed
// instructions arrive in consecutive scalar registers straight from s_load_dwordx16 (the AoS form needed 13 s_mov per 4 sources).
// ------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef EPH_F32_GROUP
#define EPH_F32_GROUP 32
#endif
constexpr int kF32Group = EPH_F32_GROUP;               // sources per conversion to f64 (slices are multiples of it)
// rows [lo, lo + cnt) of the binary32 copy (a sharded handle converts the rows it owns: they are what it sends)
__global__ void __launch_bounds__(256) k_pos_to_f32(int n, int lo, int cnt, const Body4 *__restrict__ pos, float *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const int i = lo + t;
    float x = 0.f, y = 0.f, z = 0.f, mu = 0.f;
    if (i < n) { const Body4 p = pos[i]; x = (float)p.x; y = (float)p.y; z = (float)p.z; mu = (float)p.mu; }
    float *o = out + (size_t)(i >> 1) * 8 + (i & 1);
    o[0] = x; o[2] = y; o[4] = z; o[6] = mu;
}
__device__ __forceinline__ void f32_slice(const __attribute__((address_space(4))) float *src, int j0, int j1,
                                          float xi, float yi, float zi, double &ax, double &ay, double &az) {
    const v2f x2{xi, xi}, y2{yi, yi}, z2{zi, zi};
    const auto *pg = src + (size_t)j0 * 4;
    for (int jg = j0; jg < j1; jg += kF32Group, pg += 4 * kF32Group) {   // j1 - j0 is a multiple of kF32Group
        v2f sx{0.f, 0.f}, sy{0.f, 0.f}, sz{0.f, 0.f};
#pragma unroll
        for (int u = 0; u < kF32Group; u += 2) {
            const auto *p = pg + 4 * u;                // the pair (jg + u, jg + u + 1)
            const v2f dx = v2f{p[0], p[1]} - x2, dy = v2f{p[2], p[3]} - y2, dz = v2f{p[4], p[5]} - z2;
            v2f n2 = dx * dx;
            n2 = __builtin_elementwise_fma(dy, dy, n2);
            n2 = __builtin_elementwise_fma(dz, dz, n2);
            const v2f y{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)};
            // (mu y) y y, not mu (y y y): y^3 alone leaves binary32's normal range for separations above 2e12 length units
            v2f sc = ((v2f{p[6], p[7]} * y) * y) * y;
            // No source is masked: the body itself and whatever coincides with it have d = 0, n2 = 0, y = inf -- mu y^3 = inf is
            // clamped to a finite value and multiplies the zero separation, a massless one's 0 * inf = NaN comes out of the
            // clamp as 0; padding rows (position 0, mu 0) contribute 0 * y^3 = 0, or that NaN when the body sits at the origin.
            sc = v2f{__builtin_amdgcn_fmed3f(sc.x, 0.0f, 3.0e38f), __builtin_amdgcn_fmed3f(sc.y, 0.0f, 3.0e38f)};
            sx = __builtin_elementwise_fma(dx, sc, sx);
            sy = __builtin_elementwise_fma(dy, sc, sy);
            sz = __builtin_elementwise_fma(dz, sc, sz);
        }
        // one conversion and one f64 addition per component and group (per four sources: the f64 side was a fifth of the loop)
        ax = ax + (double)(sx.x + sx.y);
        ay = ay + (double)(sy.x + sy.y);
        az = az + (double)(sz.x + sz.y);
    }
}
// block0: the first 64-body block of TARGETS of this launch (a sharded handle sums its own bodies; the slices are cut on global
// source indices, so a body's sum does not depend on how many ranks share the system)
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_partial_f32(int n, int npad, const float *__restrict__ posf, int S,
                                                                      int slice_len, double *__restrict__ partial, int block0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs_per_block = S / kFastWaves;
    const int block = block0 + blockIdx.x / wgs_per_block;
    const int slice = (blockIdx.x % wgs_per_block) * kFastWaves + wave;
    const int i = block * 64 + lane;
    const int ic = min(i, n - 1);
    const auto *src = (const __attribute__((address_space(4))) float *)(unsigned long long)posf;
    const float *own = posf + (size_t)(ic >> 1) * 8 + (ic & 1);
    const float xi = own[0], yi = own[2], zi = own[4];
    const int j0 = slice * slice_len, j1 = min(j0 + slice_len, npad);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j0 < j1) f32_slice(src, j0, j1, xi, yi, zi, ax, ay, az);
    double *pp = partial + (size_t)slice * 3 * npad + i;
    pp[0] = ax;
    pp[(size_t)npad] = ay;
    pp[(size_t)2 * npad] = az;
}

// one (component, body): partial sums combined in slice order, then the rest of the fused step. COHERENT: the partial sums were
// written by OTHER workgroups of this launch (agent-scope write-through stores): read them with agent-scope loads
template <int L, bool COHERENT>
__device__ __forceinline__ void fast_finish_one(const LmArgs &a, int S, const double *__restrict__ partial, int cc, int my_i) {
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + my_i;
    // EVERY slice's load in flight before the first ordered add (S <= kFastMaxSlices; groups of 16 behind one another paid the
    // memory latency S / 16 times: 7.3 us for this kernel at S = 64, round 5). In the one-launch form (COHERENT) the registers of
    // this epilogue set the occupancy of the whole kernel -- the rsq form's slice loop wants four waves per SIMD -- so the sums are
    // taken in groups of 32 (two latencies at 64 slices) and the history is loaded only afterwards: ~80 instead of 186 VGPRs.
    constexpr int G = COHERENT ? 32 : kFastMaxSlices;
    double anew = 0.0;
#pragma unroll
    for (int g0 = 0; g0 < kFastMaxSlices; g0 += G) {
        if (g0 >= S) break;
        double pv[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if constexpr (COHERENT) pv[u] = g0 + u < S ? __hip_atomic_load(partial + (size_t)(g0 + u) * lvl + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            else pv[u] = g0 + u < S ? partial[(size_t)(g0 + u) * lvl + off] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < G; ++u)
            if (g0 + u < S) anew = anew + pv[u];
    }
    if constexpr (COHERENT) asm volatile("" : "+v"(anew) :: "memory");      // (the history loads below stay below)
    double yv[L], av[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = a.Y[slot * lvl + off];
        av[j] = j > 0 ? a.A[slot * lvl + off] : 0.0;
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
// thread per (component, body): the second launch of the two-launch form
template <int L>
__global__ void __launch_bounds__(256) k_fast_finish(const LmArgs a, int S, const double *__restrict__ partial) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int span = a.hi - a.lo;                               // the launch's bodies [lo, hi): all of them unless the handle is sharded
    if (t >= 3 * span) return;
    fast_finish_one<L, false>(a, S, partial, t / span, a.lo + t % span);     // consecutive threads = consecutive bodies: coalesced
}

// ONE launch per step (round 6): the workgroup that arrives LAST at a block of 64 targets combines the block's S slice sums, in
// slice order, and does Cowell / sample / predictor for those bodies -- the same arithmetic in the same order as k_fast_finish, so
// the result does not depend on which workgroup that is (tests/test_gpu_fast.py: bit-identical to the two-launch form and from run
// to run). Inter-workgroup hand-off by the write-through recipe of MI355X_MICROARCH.md (splitk-seam / publish-large): the slice
// sums leave as agent-scope (sc1) stores, every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup's barrier, ONE relaxed
// agent-scope ticket per workgroup; the last arriver reads the sums with agent-scope (sc1) loads and puts the ticket back to 0 for
// the next step. No agent-scope fence anywhere: round 2's single-launch form had one release fence (buffer_wbl2) per workgroup in
// front of its ticket and they serialised (66 / 96 / 166 us per step at 16 / 32 / 64 slices).
__device__ __forceinline__ void publish_slice(double *pp, size_t npad, double ax, double ay, double az) {
    __hip_atomic_store(pp, ax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pp + npad, ay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pp + 2 * npad, az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores have left before the barrier
}
template <int L>
__device__ __forceinline__ void arrive_and_finish(const LmArgs &a, int S, const double *partial, unsigned *tickets, int block,
                                                  int wgs_per_block) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(tickets + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)(wgs_per_block - 1);
        if (last) __hip_atomic_store(tickets + block, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // every other arrival is in
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int t = threadIdx.x;                                  // 192 of the 256 threads: (component, body of the block)
    const int my_i = block * 64 + (t & 63);
    if (t < 192 && my_i >= a.lo && my_i < a.hi) fast_finish_one<L, true>(a, S, partial, t >> 6, my_i);
}
template <int UNROLL, bool APPROX, int L>
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_step(const LmArgs a, int S, int slice_len, double *__restrict__ partial,
                                                               unsigned *__restrict__ tickets) {
    const int n = a.n, npad = a.npad;
    const Body4 *__restrict__ pos = a.pos_cur;
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
        const bool diag = j0 < block * 64 + 64 && j1 > block * 64;
        if constexpr (APPROX) {
            const bool pad = j1 > n;                    // wave-uniform
            if (diag && pad) fast_slice_rsq<true, true, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else if (diag) fast_slice_rsq<true, false, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else if (pad) fast_slice_rsq<false, true, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else fast_slice_rsq<false, false, UNROLL>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        } else {
            if (diag) fast_slice<true, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
            else fast_slice<false, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        }
    }
    publish_slice(partial + (size_t)slice * 3 * npad + i, (size_t)npad, ax, ay, az);
    arrive_and_finish<L>(a, S, partial, tickets, block, wgs_per_block);
}
template <int L>
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_step_f32(const LmArgs a, const float *__restrict__ posf, int S, int slice_len,
                                                                   double *__restrict__ partial, unsigned *__restrict__ tickets, int block0) {
    const int n = a.n, npad = a.npad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs_per_block = S / kFastWaves;
    const int block = block0 + blockIdx.x / wgs_per_block;
    const int slice = (blockIdx.x % wgs_per_block) * kFastWaves + wave;
    const int i = block * 64 + lane;
    const int ic = min(i, n - 1);
    const auto *src = (const __attribute__((address_space(4))) float *)(unsigned long long)posf;
    const float *own = posf + (size_t)(ic >> 1) * 8 + (ic & 1);
    const float xi = own[0], yi = own[2], zi = own[4];
    const int j0 = slice * slice_len, j1 = min(j0 + slice_len, npad);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j0 < j1) f32_slice(src, j0, j1, xi, yi, zi, ax, ay, az);
    publish_slice(partial + (size_t)slice * 3 * npad + i, (size_t)npad, ax, ay, az);
    arrive_and_finish<L>(a, S, partial, tickets, block, wgs_per_block);
}

// f32_stage (EPH_PATH_F32_PAIRS only): 0 = the whole step; 1 = only the binary32 copy of rows [conv_lo, conv_lo + conv_cnt) (a sharded
// handle: its own rows, which it then all-gathers); 2 = the step on a copy that is already complete
int lm_step_fast(hipStream_t s, const LmArgs &a, double *partial, int S, int unroll, bool approx, float *posf, int f32_stage,
                 int conv_lo, int conv_cnt, unsigned *tickets) {
    int slice_len = (a.npad + S - 1) / S;
    const int un = posf ? kF32Group : approx ? 4 : unroll;   // (8 sources per trip spill 149 SGPRs: the prefetched group is 64 of the 102)
    slice_len = (slice_len + un - 1) / un * un;
    const int block0 = a.lo / 64, nblocks = (a.hi - a.lo + 63) / 64;    // (a.lo is a multiple of 64 on a sharded handle, else 0)
    if (!posf && (a.lo != 0 || a.hi != a.n)) return EPH_ERR_UNSUPPORTED;
    if (a.L != 12 && a.L != 13) return EPH_ERR_UNSUPPORTED;
    const dim3 pgrid((unsigned)(nblocks * (S / kFastWaves))), pblock(64 * kFastWaves);
    const bool fused = tickets != nullptr;                              // one launch per step: the last workgroup of a block finishes it
    if (posf) {                                                         // EPH_PATH_F32_PAIRS
        if (f32_stage != 2 && conv_cnt > 0)
            hipLaunchKernelGGL(k_pos_to_f32, dim3((unsigned)((conv_cnt + 255) / 256)), dim3(256), 0, s, a.n, conv_lo, conv_cnt, a.pos_cur, posf);
        if (f32_stage == 1) return launched("k_pos_to_f32");
        if (fused) {
            if (a.L == 12) hipLaunchKernelGGL(k_fast_step_f32<12>, pgrid, pblock, 0, s, a, (const float *)posf, S, slice_len, partial, tickets, block0);
            else hipLaunchKernelGGL(k_fast_step_f32<13>, pgrid, pblock, 0, s, a, (const float *)posf, S, slice_len, partial, tickets, block0);
            return launched("k_fast_step_f32");
        }
        hipLaunchKernelGGL(k_fast_partial_f32, pgrid, pblock, 0, s, a.n, a.npad, (const float *)posf, S, slice_len, partial, block0);
    } else if (fused) {
#define EPH_FAST_STEP(U, AP) \
        do { if (a.L == 12) hipLaunchKernelGGL((k_fast_step<U, AP, 12>), pgrid, pblock, 0, s, a, S, slice_len, partial, tickets); \
             else hipLaunchKernelGGL((k_fast_step<U, AP, 13>), pgrid, pblock, 0, s, a, S, slice_len, partial, tickets); } while (0)
        if (approx) EPH_FAST_STEP(4, true);
        else if (unroll == 8 && a.npad % 8 == 0) EPH_FAST_STEP(8, false);
        else EPH_FAST_STEP(4, false);
#undef EPH_FAST_STEP
        return launched("k_fast_step");
    } else if (approx)
        hipLaunchKernelGGL((k_fast_partial<4, true>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    else if (unroll == 8 && a.npad % 8 == 0)
        hipLaunchKernelGGL((k_fast_partial<8, false>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    else
        hipLaunchKernelGGL((k_fast_partial<4, false>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    const dim3 grid((3 * (a.hi - a.lo) + 255) / 256), block(256);
    if (a.L == 12) hipLaunchKernelGGL(k_fast_finish<12>, grid, block, 0, s, a, S, partial);
    else hipLaunchKernelGGL(k_fast_finish<13>, grid, block, 0, s, a, S, partial);
    return launched("k_fast_partial / k_fast_finish");
}

}  // namespace EPH_PV_NS
}  // namespace eph
