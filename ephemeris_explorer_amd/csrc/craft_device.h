// craft_device.h -- device-side types and helpers of the massless path that do not depend on the evaluation order of the
// point-mass term: the ephemeris table entry, a timeline segment, the argument block of the sweep kernels, DVec3 helpers, the
// correctly rounded pow of the step-size controller, the segment lookup of a UniformSpline. Shared by craft_sweep.hip (compiled
// once per evaluation order, pair_ns.h) and craft.hip (compiled once). Reference citations: see craft.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "eph_internal.h"
#include "ieee_seq.h"

namespace eph {

constexpr int kCraftWaveMax = 12288;                 // batches up to this size run one wave per spacecraft (measured crossover)

struct BodyEntry {            // one massive body's UniformSpline on the device
    double start, interval, mu;
    long long npoly;
    long long coeff_off;      // index of polynomial 0 in the coefficient / ncoef arrays
    double span;              // interval * (double)npoly, the product UniformSpline::span() forms on every lookup
    double rinv;              // rcp_refined(interval), filled on the device (k_body_reciprocals); +0.0 = the interval is out of
                              //   range for the wrapper-free division (pair_term.h in_range_div): use the compiler's
    const double *rows;       // device address of polynomial 0's coefficient row (= coeffs + coeff_off * 24): the sweep's body loop forms a
                              //   row address from the entry alone (no table base to keep in SGPRs, half the scalar address arithmetic)
};
struct SegmentDev {           // Segment<DVec3, ReferenceFrame>
    double start, end;
    double ax, ay, az;
    int is_burn, ref;         // ref: body index, -1 = inertial
};
struct CraftArgs {
    long long n_craft;
    int n_bodies;
    const BodyEntry *bodies;     // the table in the order Bodies::acceleration visits it (the ephemeris's own, or the batch's permuted copy)
    const BodyEntry *bodies_by_index;   // the ephemeris's table: a burn's reference body is an index into THIS one
    const double *coeffs;     // [poly][8][3]
    const int *ncoef;
    // per craft (SoA)
    double *time, *y /*[6][n]*/, *next_h, *klast /*[6][n] FSAL carry: k[S-1]*/, *last_knot_t;
    double *kfirst;           // [6][n] k[0] of an FSAL pair between calls: after a FAILED attempt (EvalFailed inside the stages) the reference's
                              // next advance swaps k[0] and k[S-1] once more (explicit.rs:76-79), which brings back whatever k[0] held
    int retry;                // step the craft whose last step returned a StepError (eph_craft_batch_retry_failed), else they are skipped
    unsigned *n_attempts, *rk_i, *steps;
    int *cur_seg, *status, *nknots;
    const long long *seg_off;
    const SegmentDev *segs;
    // knots: [max_knots][n] and [max_knots][6][n]
    double *knot_t, *knot_y;
    int max_knots;
    // method + controller
    ErkCoeffs rk;
    const ErkCoeffs *rkd;     // the same table in device memory (k_craft_wave indexes it by stage at run time)
    double h_init, h_max, tol_pos, tol_vel, fac_min, fac_max, fac;
    unsigned n_max;
    double t_end;
    unsigned step_limit;      // accepted steps this call may take per craft (0 = until t_end)
    unsigned long long *queue;   // k_craft_propagate's work queue: the next craft nobody has started (set by craft_launch)
    const int *perm;             // lane / queue position -> craft (null: identity). Heterogeneous batches: craft sorted by their
                                 // dynamical time at creation, so that the lanes of a wave carry craft of similar step counts
};

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
__device__ __forceinline__ double length_recip(V3 a) { return 1.0 / sqrt(dot(a, a)); }
__device__ __forceinline__ bool try_normalize(V3 a, V3 &out) {
    const double rcp = length_recip(a);
    if (isfinite(rcp) && rcp > 0.0) { out = scale(a, rcp); return true; }
    return false;
}


// ---- powf of the step-size controller ------------------------------------------------------------------------
// `err.pow(-k.inv())` (integration/src/runge_kutta/mod.rs:239) is the platform libm's pow in the reference -- the only
// operation on the path whose bits depend on the platform. Evaluated here correctly rounded in double-double
// arithmetic (log: atanh series, exp: Taylor series, ~100 bits), the same operation sequence the oracle pins.
struct DD { double hi, lo; };
#define EPH_POW_CONST __device__ const
typedef DD EPH_POW_DD;
#include "cr_pow_tables.inc"
__device__ __forceinline__ DD dd_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ DD dd_quick(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ DD dd_two_prod(double a, double b) { const double p = a * b; return {p, __builtin_fma(a, b, -p)}; }
__device__ __forceinline__ DD dd_add(DD a, DD b) {
    DD s = dd_two_sum(a.hi, b.hi);
    const DD t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick(s.hi, s.lo);
}
__device__ __forceinline__ DD dd_mul(DD a, DD b) {
    DD p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick(p.hi, p.lo);
}
__device__ __forceinline__ DD dd_mul_d(DD a, double b) {
    DD p = dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return dd_quick(p.hi, p.lo);
}
__device__ __forceinline__ DD dd_neg(DD a) { return {-a.hi, -a.lo}; }
__device__ __noinline__ DD dd_div(DD a, DD b) {
    const double q1 = a.hi / b.hi;
    DD r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    const DD q = dd_quick(q1, q2);
    return dd_add(q, DD{q3, 0.0});
}
__device__ __noinline__ double cr_pow(double x, double y) {
    if (isnan(x) || isnan(y)) return __builtin_nan("");
    if (x == 0.0) return y < 0.0 ? __builtin_inf() : 0.0;
    if (isinf(x)) return y < 0.0 ? 0.0 : __builtin_inf();
    if (x < 0.0) return __builtin_nan("");
    const DD ln2 = {0x1.62e42fefa39efp-1, 0x1.abc9e3b39803fp-56};
    int e;
    double m = frexp(x, &e);
    if (m < 0x1.6a09e667f3bcdp-1) { m *= 2.0; e -= 1; }
    const DD s = dd_div(DD{m - 1.0, 0.0}, dd_two_sum(m, 1.0));
    const DD s2 = dd_mul(s, s);
    // atanh(s)/s = sum_k s2^k / (2k+1), Horner over the double-double table (remainder < 2^-120)
    // (both series fully unrolled: the table entries become literals. As run-time loops they paid a scalar-cache round trip per
    // three terms, waited for on the spot -- half of the function's time on the single wave k_craft_wave is)
    DD sum = {eph_pow_atanh[EPH_POW_TERMS - 1].hi, eph_pow_atanh[EPH_POW_TERMS - 1].lo};
#pragma unroll
    for (int k = EPH_POW_TERMS - 2; k >= 0; --k) sum = dd_add(dd_mul(sum, s2), DD{eph_pow_atanh[k].hi, eph_pow_atanh[k].lo});
    DD lg = dd_mul(dd_mul_d(s, 2.0), sum);
    lg = dd_add(dd_mul_d(ln2, (double)e), lg);
    const DD z = dd_mul_d(lg, y);
    if (z.hi > 709.0) return __builtin_inf();
    if (z.hi < -745.0) return 0.0;
    const double kf = nearbyint(z.hi / ln2.hi);
    const DD r = dd_add(z, dd_neg(dd_mul_d(ln2, kf)));
    // exp(r) = sum_n r^n / n!, Horner over the double-double table, |r| <= ln2/2
    DD ex = {eph_pow_invfact[EPH_POW_TERMS - 1].hi, eph_pow_invfact[EPH_POW_TERMS - 1].lo};
#pragma unroll
    for (int n = EPH_POW_TERMS - 2; n >= 0; --n) ex = dd_add(dd_mul(ex, r), DD{eph_pow_invfact[n].hi, eph_pow_invfact[n].lo});
    return ldexp(ex.hi + ex.lo, (int)kf);
}
// UniformSpline::get_polynomial: false = None
__device__ __forceinline__ bool spline_locate(const BodyEntry &b, double at, long long &idx, double &tau) {
    const double local = at - b.start;
    const double span = b.interval * (double)b.npoly;
    if (__builtin_signbit(local) || local > span) return false;
    const double c = ceil(local / b.interval);
    const unsigned long long ci = c <= 0.0 ? 0ull : (c >= 18446744073709551616.0 ? ~0ull : (unsigned long long)c);
    const unsigned long long i = ci == 0 ? 0 : ci - 1;
    if (i >= (unsigned long long)b.npoly) return false;
    idx = (long long)i;
    tau = (local - b.interval * (double)i) / b.interval;
    return true;
}

// UniformSpline::get_polynomial for the sweep kernels: the span product comes precomputed with the table entry and
// the f64 <-> u64 conversions take the one-instruction 32-bit forms when every lane's segment count fits (always, in
// practice); same values as spline_locate.
__device__ __forceinline__ bool locate_quot_ok(double a) {   // numerator usable by div_refined: +-0 or in in_range_div
    return a == 0.0 || in_range_div(a);
}
__device__ __forceinline__ bool spline_locate_fast(const BodyEntry &b, double at, long long &idx, double &tau) {
    const double local = at - b.start;
    if (__builtin_signbit(local) || local > b.span) return false;
    // the two divisions by the interval share its refined reciprocal (table entry, wave-uniform) whenever the wrappers of the
    // compiler's division would have been no-ops for every lane: the same quotients, 6 operations instead of 24 + two v_rcp_f64
    const bool shared = __double_as_longlong(b.rinv) != 0;
    const bool fast1 = shared && __builtin_amdgcn_ballot_w64(!locate_quot_ok(local)) == 0;
    const double c = ceil(fast1 ? div_refined(local, b.interval, b.rinv) : local / b.interval);
    unsigned long long i;
    double fi;
    if (__builtin_amdgcn_ballot_w64(!(c < 2147483648.0)) == 0) {      // also false for NaN
        const unsigned ci = c <= 0.0 ? 0u : (unsigned)c;
        const unsigned i32 = ci == 0 ? 0u : ci - 1u;
        i = i32;
        fi = (double)i32;
    } else {
        const unsigned long long ci = c <= 0.0 ? 0ull : (c >= 18446744073709551616.0 ? ~0ull : (unsigned long long)c);
        i = ci == 0 ? 0 : ci - 1;
        fi = (double)i;
    }
    if (i >= (unsigned long long)b.npoly) return false;
    idx = (long long)i;
    const double rem = local - b.interval * fi;
    const bool fast2 = shared && __builtin_amdgcn_ballot_w64(!locate_quot_ok(rem)) == 0;
    tau = fast2 ? div_refined(rem, b.interval, b.rinv) : rem / b.interval;
    return true;
}

}  // namespace eph
