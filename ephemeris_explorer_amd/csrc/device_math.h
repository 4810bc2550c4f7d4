// device_math.h -- IEEE-exact f64 sqrt / reciprocal sequences without the range-scaling wrappers (device only).
//
// sqrt and the reciprocal on the path must be the IEEE correctly rounded results (the CPU's sqrtsd / divsd). The
// compiler's f64 expansions are: v_rsq_f64 / v_rcp_f64 seed + fma refinement, wrapped in range scaling (v_ldexp,
// v_div_scale, v_div_fmas, v_div_fixup) that only acts for operands near the ends of the exponent range.
// `*_inrange` are exactly those refinement sequences without the wrappers: bit-identical whenever the scaling
// would have been a no-op, which in_range() guarantees (x in [2^-300, 2^300), so x*sqrt(x) in [2^-450, 2^450)).
// tests/test_gpu_parity.py::test_inrange_sqrt_and_reciprocal_sequences_are_ieee checks them against the host.
#pragma once
#include <hip/hip_runtime.h>

namespace eph {

// Which evaluation order of the point-mass term this build carries (see the table further down).
#ifndef EPH_PAIR_VARIANT
#define EPH_PAIR_VARIANT 0
#endif
static_assert(EPH_PAIR_VARIANT >= 0 && EPH_PAIR_VARIANT <= 6, "EPH_PAIR_VARIANT must be 0..6");
constexpr int kPairVariant = EPH_PAIR_VARIANT;
// Guarded operand range of the wrapper-free sequences. Variants 0-3 (one reciprocal): n2 in [2^-300, 2^300), biased
// exponent in [723, 1323). Variants 4-6 (true divisions by p = n2*sqrt(n2) through a shared refined reciprocal,
// div_refined below) need p in [2^-200, 2^200): n2 in [2^-133, 2^133), biased exponent in [890, 1156).
constexpr unsigned kRangeBase = kPairVariant >= 4 ? 0x37A00000u : 0x2D300000u;
constexpr unsigned kRangeSpan = kPairVariant >= 4 ? 0x10A00000u : 0x25800000u;
__device__ __forceinline__ bool in_range(double n2) {
    // n2 >= 0; NaN/inf/0/denormals are out
    return (unsigned)(__double2hiint(n2) - kRangeBase) < kRangeSpan;
}
// the same test for several operands at once: in range iff max over the operands of range_key() < kRangeSpan
// (one integer add and one max per operand instead of a compare and boolean plumbing)
__device__ __forceinline__ unsigned range_key(double n2) { return (unsigned)(__double2hiint(n2) - kRangeBase); }
__device__ __forceinline__ double sqrt_inrange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
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

// THE point-mass kernel of the path, 1 / r^3 from n2 = |d|^2, in the one place it is defined for every kernel
// (force kernels, k_lm_small, the spacecraft sweeps). It restates `particular`'s acceleration_paired /
// acceleration_at, whose source (git rev d490707a) is not on disk -- DESIGN.md §2: "parity unpinned" at this one
// boundary. The evaluation order is therefore a BUILD FLAG, not an edit: -DEPH_PAIR_VARIANT=k selects
//     0 (default, the published crate's form)   inv = 1 / (n2 * sqrt(n2))
//     1                                         r = sqrt(n2) ; inv = 1 / (r * r * r)
//     2                                         s = 1 / sqrt(n2) ; inv = s * s * s
//     3                                         inv = (1 / n2) * (1 / sqrt(n2))
// each followed by a = d * (mu * inv), and the DIVISION forms with p = n2 * sqrt(n2) and three true divisions
// (glam's DVec3 / f64 is component-wise):
//     4   a = (d * mu) / p     Rust `dir * mu / (mag_2 * mag_2.sqrt())`, the published crate's documented scalar form
//     5   a = d * (mu / p)
//     6   a = (d / p) * mu     a paired routine sharing `dir / p` between the two masses
// with IEEE sqrt / divide in every form (the CPU restatement the tests check against has the same seven). `python -m ephemeris_explorer_amd.build --pair-variant k` builds
// libephemeris_amd_pv<k>.so; tests/test_gpu_variants.py checks each against the oracle in the same variant. Kernels reach
// the term through pair_den / pair_apply at the end of this file.
// Only variant 0 has the hand-interleaved tile pipeline of wave_force (kernels.hip pair_stage); the others run
// the same kernels with the compiler's schedule.
// in_range(n2) keeps every intermediate of every variant inside the exponent range where the stripped sequences
// equal the compiler's IEEE expansions: n2 in [2^-300, 2^300) => sqrt in [2^-150, 2^150), products and reciprocals
// within [2^-450, 2^450].
// Variant 0 with the reciprocal's seed taken from the square root's own refinement instead of a second quarter-rate
// v_rcp_f64 (saves a transcendental and one fma per interaction). WHY THE RESULT IS RN(1 / p), p = RN(x RN(sqrt x)),
// for every in-range x -- u = 2^-53, all fma single-rounded, "d_k" = a rounding error with |d_k| <= u:
//  (1) Seed. y = v_rsq_f64(x) = (1 + e0) / sqrt(x); the ISA documents |e0| <= 2^29 ulp = 2^-23, measured max 2^-24.2
//      (scripts/probe_rcp_edge.py, 4e6 operands). The coupled step gives h = (1 + e_h) / (2 sqrt x) with
//      e_h = -(3/2) e0^2 - d_1/2 + d_4 + O(e0^3), so |e_h| <= 1.5 * 2^-46 + 1.5 u < 2^-45.3 (measured max 2^-47.8).
//  (2) q0 = RN(RN(RN(h h) h) 8) = (1 + 3 e_h + d + d') / (x sqrt x), and p = x sqrt(x) (1 + eta)(1 + d_p) with |eta| <= u
//      (g is the correctly rounded root: the lines up to `g = ...` ARE the compiler's sqrt expansion). Against 1 / p:
//      q0 = (1 + eps) / p,  |eps| <= 3 |e_h| + 4u + (second order) < 2^-43.5.
//  (3) Newton step. e = RN(1 - p q0) = -eps (1 + d_a) (the product is exact inside the fma), q1 = RN(q0 + q0 e)
//      = RN((1 - eps^2 - eps d_a (1 + eps)) / p): q1 is the rounding of a value within 2^-86 (relative) of 1 / p,
//      so |rho| <= 2^-54 p' + 2^-85 for rho = p q1 - 1, p' in [1, 2) the significand of p.
//  (4) Residual step. p q1 is a 106-bit product within 2^-52 of 1, so 1 - p q1 = -rho is a multiple of 2^-105 below
//      2^-52: representable, the fma returns it exactly; the last fma rounds v = q1 (1 - rho) = (1 - rho^2) / p ONCE.
//      v < 1 / p, so RN(v) = RN(1 / p) unless a rounding boundary m (odd multiple of half an ulp) lies in [v, 1 / p).
//      Scale p' into [1, 2), 1 / p' into (1/2, 1]: m p' is a multiple of 2^-106 and != 1, so 1 / p' - m = j 2^-106 / p'
//      with an integer j >= 1, while 1 / p' - v = rho^2 / p' < 2^-106 (p'/2)^2 (1 + 2^-29) / p'. A boundary can be
//      crossed only if j = 1 and p' > 2 - 2^-28. Writing p' = 2 - k 2^-52: the boundary just below 1 / p' is
//      m = (2^53 + k) 2^-54 for odd k, with m p' = 1 - k^2 2^-106, i.e. j = k^2; for even k the nearest boundary is
//      half an ulp away. So the ONLY significand for which the residual step can fail is k = 1, all ones (Markstein's
//      exception): there 1 / p = m + 2^-107, an iterate from below rounds to q1 = 2^-(E+1), and v is an exact tie.
//  (5) That significand cannot occur. x -> 4x maps g -> 2g and p -> 8p exactly, so which significands p can take
//      just below a power of two depends only on the binade of p modulo 3; enumerating the x around (2^(E+1))^(2/3)
//      for the three classes (tests/exceptional_operands.py) gives p = 2^(E+1) - k ulp with smallest k = 2, 3, 2:
//      never 1. Those operands (every binade, k <= 64: 19 618 of them) are in
//      tests/test_gpu_parity.py::test_inrange_sqrt_and_reciprocal_sequences_are_ieee, next to the random sweeps
//      (eph_debug_inv_r3_sweep: 2.7e11 operands) that the argument above makes redundant but that stay as a guard
//      against a transcription slip.
// (The other variants and div_refined keep the compiler's own v_rcp_f64-seeded expansion, for which the same all-ones
// exception exists in principle; on this hardware 1/b, 3/b and (2-ulp)/b come out correctly rounded for all-ones b in
// every binade -- the same test -- because of where v_rcp_f64's seed falls, a measured property, not a theorem.)
#ifndef EPH_RCP_SEED_FROM_RSQ
#define EPH_RCP_SEED_FROM_RSQ 1
#endif
__device__ __forceinline__ double inv_r3_seeded(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);                         // sqrt(x), correctly rounded (sqrt_inrange)
    const double p = x * g;
    double q = (h * h) * h * 8.0;                       // ~ 1 / p
    double e = __builtin_fma(-p, q, 1.0);
    q = __builtin_fma(q, e, q);
    e = __builtin_fma(-p, q, 1.0);
    return __builtin_fma(e, q, q);
}
__device__ __forceinline__ double inv_r3_inrange(double n2) {
    if constexpr (kPairVariant == 0 && EPH_RCP_SEED_FROM_RSQ) return inv_r3_seeded(n2);
    if constexpr (kPairVariant == 1) { const double r = sqrt_inrange(n2); return rcp_inrange(r * r * r); }
    else if constexpr (kPairVariant == 2) { const double s = rcp_inrange(sqrt_inrange(n2)); return s * s * s; }
    else if constexpr (kPairVariant == 3) return rcp_inrange(n2) * rcp_inrange(sqrt_inrange(n2));
    else return rcp_inrange(n2 * sqrt_inrange(n2));
}
__device__ __forceinline__ double inv_r3_ieee(double n2) {
    if constexpr (kPairVariant == 1) { const double r = sqrt(n2); return 1.0 / (r * r * r); }
    else if constexpr (kPairVariant == 2) { const double s = 1.0 / sqrt(n2); return s * s * s; }
    else if constexpr (kPairVariant == 3) return (1.0 / n2) * (1.0 / sqrt(n2));
    else return 1.0 / (n2 * sqrt(n2));
}

// a / b, IEEE correctly rounded, with the reciprocal refinement shared between numerators: the compiler's f64
// division is  r = rcp(b) + two Newton steps;  q = a*r;  e = fma(-b, q, a);  q = fma(e, r, q)  inside the scaling
// wrappers. rcp_refined(b) is the first half, div_refined the second; bit-identical to a / b whenever div_scale /
// div_fixup would have been no-ops: b in [2^-200, 2^200) and a either 0 or in that range too (quotient within
// [2^-400, 2^400)). tests/test_gpu_craft.py::test_shared_reciprocal_division_is_ieee.
__device__ __forceinline__ bool in_range_div(double x) {   // biased exponent in [823, 1223)
    return (unsigned)(__double2hiint(x) - 0x33700000) < 0x19000000u;
}
__device__ __forceinline__ double rcp_refined(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double div_refined(double a, double b, double r) {
    const double q = a * r;
    const double e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, r, q);
}
// a / b with the shared reciprocal where it is exact (b_ok: b in range), the compiler's division otherwise
__device__ __forceinline__ double div_shared(double a, double b, double r, bool b_ok) {
    if (b_ok && (a == 0.0 || in_range_div(a))) return div_refined(a, b, r);
    return a / b;
}

// ---- the point-mass term behind one interface (every kernel family calls these two) ---------------------------
// pair_den<FAST>(n2): what the two directions of a pair share -- variants 0-3: v = 1/r^3; variants 4-6: v = p =
// n2*sqrt(n2) and, in the FAST form, r = rcp_refined(p). pair_apply<FAST>(den, d, mu): the acceleration d-wards of a
// mass mu. FAST = the wrapper-free sequences (caller guarantees in_range(n2) for the whole wave); !FAST = the
// compiler's IEEE sqrt / divide. Quotients in the FAST form go through div_refined when the numerator is in
// in_range_div (a signed zero numerator returns itself: x/p = x for p > 0), through the compiler's division otherwise.
struct PairDen { double v, r; };
template <bool FAST>
__device__ __forceinline__ PairDen pair_den(double n2) {
    PairDen d;
    if constexpr (kPairVariant <= 3) {
        d.v = FAST ? inv_r3_inrange(n2) : inv_r3_ieee(n2);
        d.r = 0.0;
    } else if constexpr (FAST) {
        d.v = n2 * sqrt_inrange(n2);
        d.r = rcp_refined(d.v);
    } else {
        d.v = n2 * sqrt(n2);
        d.r = 0.0;
    }
    return d;
}
// numerator a of a FAST quotient is usable by div_refined: a signed zero (handled by a select) or in in_range_div
__device__ __forceinline__ bool quot_ok(double a) { return a == 0.0 || in_range_div(a); }
// a / p through the shared reciprocal, branch-free: valid when quot_ok(a). p > 0, so the quotient carries a's sign: for a = -0
// the refinement's residual step turns the -0 into +0 (fma(+0, r, -0)), and putting a's sign back (one v_bfi_b32 instead of a
// compare and two selects) is the identity for every other in-range a.
__device__ __forceinline__ double quot_fast(double a, const PairDen &d) {
    return __builtin_copysign(div_refined(a, d.v, d.r), a);
}
template <bool FAST>
__device__ __forceinline__ void pair_apply(const PairDen &d, double dx, double dy, double dz, double mu, double &cx,
                                           double &cy, double &cz) {
    if constexpr (kPairVariant <= 3) {
        const double s = mu * d.v;
        cx = dx * s;
        cy = dy * s;
        cz = dz * s;
    } else if constexpr (!FAST) {
        if constexpr (kPairVariant == 4) { cx = (dx * mu) / d.v; cy = (dy * mu) / d.v; cz = (dz * mu) / d.v; }
        else if constexpr (kPairVariant == 5) { const double s = mu / d.v; cx = dx * s; cy = dy * s; cz = dz * s; }
        else { cx = (dx / d.v) * mu; cy = (dy / d.v) * mu; cz = (dz / d.v) * mu; }
    } else {
        // straight-line fast quotients for every lane; lanes whose numerator leaves the guarded range (denormal-scale
        // products, |a| < 2^-200 or >= 2^200) redo theirs with the compiler's division behind ONE wave-uniform branch
        // that a tile of ordinary operands never takes
        if constexpr (kPairVariant == 4) {
            const double nx = dx * mu, ny = dy * mu, nz = dz * mu;
            cx = quot_fast(nx, d); cy = quot_fast(ny, d); cz = quot_fast(nz, d);
            const bool bad = !(quot_ok(nx) && quot_ok(ny) && quot_ok(nz));
            // (wave-uniform, and with an empty volatile asm inside: otherwise the compiler flattens the branch into selects and every
            // lane pays the three IEEE divisions as well -- 76.8 instead of 62.3 us per step at N = 4096; the compiler's quotient is the same value
            // for the lanes that were in range)
            if (__builtin_amdgcn_ballot_w64(bad) != 0) { asm volatile(""); cx = nx / d.v; cy = ny / d.v; cz = nz / d.v; }
        } else if constexpr (kPairVariant == 5) {
            double s;                                  // (one quotient: plain lane branches measured faster here, 42.9 vs 50.8 us)
            if (mu == 0.0) s = mu;
            else if (in_range_div(mu)) s = div_refined(mu, d.v, d.r);
            else s = mu / d.v;
            cx = dx * s; cy = dy * s; cz = dz * s;
        } else {
            double qx = quot_fast(dx, d), qy = quot_fast(dy, d), qz = quot_fast(dz, d);
            const bool bad = !(quot_ok(dx) && quot_ok(dy) && quot_ok(dz));
            if (__builtin_amdgcn_ballot_w64(bad) != 0) { asm volatile(""); qx = dx / d.v; qy = dy / d.v; qz = dz / d.v; }
            cx = qx * mu; cy = qy * mu; cz = qz * mu;
        }
    }
}

}  // namespace eph
