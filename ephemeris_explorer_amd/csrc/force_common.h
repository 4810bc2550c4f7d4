// force_common.h -- device helpers shared by the step kernels that do not depend on the evaluation order of the point-mass term:
// the ordered accumulation of LDS rows, the linear-multistep formulas, the solout sample.
#pragma once
#include <hip/hip_runtime.h>

#include "eph_internal.h"

namespace eph {

__device__ __forceinline__ void wave_lds_fence() {
    // same-wave LDS hand-off (lane-per-source writes -> lane-per-chain reads): DS ops of one wave execute in
    // order; this only stops the compiler from moving them across.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------
// Ordered accumulation of one 64-wide row of the contribution tile (phase B of wave_force).
// The row is read with ds_read_b128 in four 16-element chunks, the next chunk's reads in flight while the
// current one is added, so the dependent v_add_f64 chain never waits on LDS latency.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_chunk(const double *row, int c, double2 (&r)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const double2 *>(row + c * 16 + 2 * k);   // ds_read_b128
}
__device__ __forceinline__ double add_chunk(const double2 (&r)[8], double acc) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc = acc + r[k].x;
        acc = acc + r[k].y;
    }
    return acc;
}

// full tile, none of the wave's bodies inside it: 64 plain ordered adds
__device__ __forceinline__ double chain_full(const double *row, double acc) {
    double2 ra[8], rb[8];
    load_chunk(row, 0, ra);
    load_chunk(row, 1, rb);
    acc = add_chunk(ra, acc);
    load_chunk(row, 2, ra);
    acc = add_chunk(rb, acc);
    load_chunk(row, 3, rb);
    acc = add_chunk(ra, acc);
    return add_chunk(rb, acc);
}

// The tile that holds the wave's own bodies (and/or the ragged last tile). The wave's BPW bodies are consecutive
// and BPW-aligned, so they occupy exactly one BPW-wide group `gself` of the tile: groups before it are
// "sources before the body" for every chain, groups after it "sources after the body"; only inside that one group
// does a chain skip its own body, close the lower sum and restart from V::default(). cnt = valid sources.
// li = index of this lane's own body inside the tile.
template <int BPW>
__device__ __forceinline__ void chain_masked(const double *row, int cnt, int gself, int li, double &acc,
                                             double &accL) {
    double2 ra[8], rb[8];
    load_chunk(row, 0, ra);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double2(&cur)[8] = (c & 1) ? rb : ra;
        double2(&nxt)[8] = (c & 1) ? ra : rb;
        if (c * 16 >= cnt) break;                       // wave-uniform
        if (c < 3 && (c + 1) * 16 < cnt) load_chunk(row, c + 1, nxt);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int jl = c * 16 + e;
            const double v = (e & 1) ? cur[e >> 1].y : cur[e >> 1].x;
            if (jl >= cnt) continue;                    // wave-uniform
            if (jl / BPW != gself) {                    // wave-uniform
                acc = acc + v;
            } else {
                const bool self = (jl == li);
                const double t = acc + v;               // NaN on the self lane, discarded
                accL = self ? acc : accL;               // lower chain complete
                acc = self ? 0.0 : t;                   // upper chain starts from V::default()
            }
        }
    }
}

// SRKN stage update of one (body, component) behind its force evaluation   symplectic.rs:90-97
__device__ __forceinline__ void kick_drift_one(const KickDrift &kd, size_t o, int body, int comp, double a) {
    const double vn = kd.v[o] + a * kd.hb;            // *dy = *dy + *ddy * (h * C::B[s])
    kd.v[o] = vn;
    const double yn = kd.y[o] + vn * kd.ha;           // *y = *y + *dy * (h * C::A[s])
    kd.y[o] = yn;
    reinterpret_cast<double *>(kd.pos_out + body)[comp] = yn;   // mu is already in both packed buffers
}

// ------------------------------------------------------------------------------------------------------
// Linear multistep pieces shared by the per-step and the persistent kernels.
//   predictor  ELM2::advance           integration/src/multistep/second_order/mod.rs:93-121
//   velocity   Cowell::update_velocity integration/src/multistep/second_order/cowell.rs:19-53
// yv[j], av[j] = position / acceleration component of level (newest - j).
// ------------------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ double lm_predict(const double (&yv)[L], const double (&av)[L], const double *wa,
                                             const double *wb, double hh) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        s1 = s1 + yv[j] * wa[j];   // *sum1 = *sum1 + *y * (1.0 * Ratio::from_int(-ALPHA[j+1]))
        s2 = s2 + av[j] * wb[j];   // *sum2 = *sum2 + *ddy * (1.0 * Ratio::from_int(BETA_N[j+1]))
    }
    return s1 + s2 * hh;           // *y = *sum1 + *sum2 * (h * h * Ratio::from_recip(BETA_D))
}

// a_new = acceleration of the new level; av[0..L-2] = the L-1 levels before it
template <int L>
__device__ __forceinline__ double lm_cowell(double a_new, const double (&av)[L], double y_new, double y_prev,
                                            const double *cw, double h, double hc) {
    double s = 0.0;
    s = s + a_new * cw[0];
#pragma unroll
    for (int j = 1; j < L; ++j) s = s + av[j - 1] * cw[j];
    return (y_new - y_prev) / h + s * hc;   // *dy = (*y - *ym1) / h + *work * (h * Ratio::from_recip(BETA_D))
}

__device__ __forceinline__ void maybe_sample(const SampleArgs &sa, int body, int comp, uint32_t step, double y) {
    if (!sa.period) return;
    const uint32_t m = sa.period[body];
    if (m == 0) return;
    const uint32_t t = sa.phase[body] + step;
    if (t % m == 0) sa.log[(sa.offset[body] + (uint64_t)(t / m - 1)) * 3 + comp] = y;
}

// the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2] on both halves of the double)
__device__ __forceinline__ double dpp_xor1(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
    const int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    return __hiloint2double(h2, l2);
}
// A workgroup-wide barrier for LDS hand-offs only: __syncthreads() also waits for every outstanding GLOBAL access
// (vmcnt(0)) -- the solout's sample stores would stall all eight waves for a memory round trip at every sampled step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// hipGetLastError after a launch -> status
inline int launched(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(what, e);
        return EPH_ERR_HIP;
    }
    return EPH_OK;
}

}  // namespace eph
