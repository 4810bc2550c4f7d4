// pair_term.h -- THE point-mass term of the path, in the one place it is defined for every kernel family (force kernels,
// k_lm_small, the fast path, the spacecraft sweeps). INCLUDED INSIDE namespace eph::pv<k> by the translation units that are
// compiled once per evaluation order (-DEPH_PAIR_VARIANT=k; pair_ns.h): the order is a compile-time constant in the kernels
// and a RUN-TIME choice of the library (eph_set_pair_variant -> which namespace's launchers a handle uses).
//
// It restates `particular`'s acceleration_paired / acceleration_at::<false> (call sites ephemeris/src/propagators/nbody.rs:29,
// ephemeris_explorer/src/dynamics/spacecraft.rs:73), whose source (git rev d490707a, Cargo.lock:4277-4285) is not on disk --
// "parity unpinned" at this one boundary; tools/identify_pair_variant.py names the order from a print-out of
// the real crate. With d = p_other - p_self, n2 = d.x*d.x + d.y*d.y + d.z*d.z (glam, left to right), p = n2 * sqrt(n2):
//     0 (default, the published crate's form)   inv = 1 / p                       a = d * (mu * inv)
//     1                                         r = sqrt(n2) ; inv = 1 / (r * r * r)
//     2                                         s = 1 / sqrt(n2) ; inv = s * s * s
//     3                                         inv = (1 / n2) * (1 / sqrt(n2))
//     4   a = (d * mu) / p     Rust `dir * mu / (mag_2 * mag_2.sqrt())`, the published crate's documented scalar form (glam's
//                              DVec3 / f64 is component-wise: three true divisions)
//     5   a = d * (mu / p)
//     6   a = (d / p) * mu     a paired routine sharing `dir / p` between the two masses
// IEEE sqrt / divide in every form; the CPU restatement the tests check against (oracle/) has the same seven.
//
// FAST forms: the wrapper-free sequences of ieee_seq.h, valid for a whole wave when the caller's range test passes:
//   * n2 inside the guarded range (in_range / range_key): orders 0-3 [2^-300, 2^300); orders 4-6 [2^-133, 2^133), so that
//     p is in [2^-200, 2^200);
//   * orders 4 and 6 divide the components of d (times mu): every |d_c| >= 2^-500, tested on the squares pair_pre forms anyway
//     (PairPre::lo = the smallest high word of d.x^2, d.y^2, d.z^2; low_key). A zero component goes to the IEEE path: the
//     refinement would return +0 for a -0 numerator;
//   * orders 4 and 5 divide mu (times d): mu in [2^-200, 2^200) (mu_key; order 5 also takes +0, for which the sequence
//     returns +0 like the division).
// Quotients then are  t = RN(a r); e = a - p t (one fma); q = RN(t + e r)  with r = RN(1 / p) -- Markstein's division step on a
// CORRECTLY ROUNDED reciprocal (inv_r3_seeded below). tests/division_hard_cases.py builds the numerators whose quotient lies
// closest to a rounding boundary for a given p (|a/p - midpoint| = k 2^-106 / p', k = 1, 2, ...) and runs them through this
// sequence in exact arithmetic and on the device (tests/test_gpu_variants.py::test_seeded_quotient_on_hard_cases): identical
// to IEEE division on all of them and on 10^9 random operands.

constexpr int kPairVariant = EPH_PAIR_VARIANT;
static_assert(kPairVariant >= 0 && kPairVariant <= 6, "EPH_PAIR_VARIANT must be 0..6");
constexpr bool kDivForm = kPairVariant >= 4;                      // three true divisions per term
constexpr bool kNeedLow = kPairVariant == 4 || kPairVariant == 6;   // the components of d are numerators
constexpr unsigned kRangeBase = kDivForm ? 0x37A00000u : 0x2D300000u;   // biased exponent 890 / 723
constexpr unsigned kRangeSpan = kDivForm ? 0x10A00000u : 0x25800000u;   // ... up to 1156 / 1323
constexpr unsigned kLowSq = 0x01700000u;                          // high word of 2^-1000: |d_c| >= 2^-500
__device__ __forceinline__ bool in_range(double n2) {             // n2 >= 0; NaN / inf / 0 / denormals are out
    return (unsigned)(__double2hiint(n2) - kRangeBase) < kRangeSpan;
}
// the same test for several operands at once: all in range iff the max of the keys < kRangeSpan
__device__ __forceinline__ unsigned range_key(double n2) { return (unsigned)(__double2hiint(n2) - kRangeBase); }
// 0 when every square of a component is >= 2^-1000, >= 2^31 otherwise (lo = min over the interactions of PairPre::lo)
__device__ __forceinline__ unsigned low_key(unsigned lo) { return kNeedLow ? min(lo, kLowSq) - kLowSq : 0u; }
// 0 when mu may go through the shared reciprocal, kRangeSpan otherwise
__device__ __forceinline__ unsigned mu_key(double mu) {
    if constexpr (kPairVariant == 4) return in_range_div(mu) ? 0u : kRangeSpan;
    else if constexpr (kPairVariant == 5) return (__double_as_longlong(mu) == 0 || in_range_div(mu)) ? 0u : kRangeSpan;
    else return 0u;
}

// The reciprocal of p = n2 * sqrt(n2) with its seed taken from the square root's own refinement instead of a second quarter-rate
// v_rcp_f64 (saves a transcendental and one fma per interaction); orders 0 and 4-6 use it. WHY THE RESULT IS RN(1 / p), p = RN(x RN(sqrt x)),
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
// (Orders 1-3 and div_refined with rcp_refined keep the compiler's own v_rcp_f64-seeded expansion, for which the same all-ones
// exception exists in principle; on this hardware 1/b, 3/b and (2-ulp)/b come out correctly rounded for all-ones b in
// every binade -- the same test -- because of where v_rcp_f64's seed falls, a measured property, not a theorem.)
__device__ __forceinline__ double inv_r3_seeded(double x, double *p_out = nullptr) {
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
    if (p_out) *p_out = p;
    return __builtin_fma(e, q, q);
}
__device__ __forceinline__ double inv_r3_inrange(double n2) {
    if constexpr (kPairVariant == 1) { const double r = sqrt_inrange(n2); return rcp_inrange(r * r * r); }
    else if constexpr (kPairVariant == 2) { const double s = rcp_inrange(sqrt_inrange(n2)); return s * s * s; }
    else if constexpr (kPairVariant == 3) return rcp_inrange(n2) * rcp_inrange(sqrt_inrange(n2));
    else return inv_r3_seeded(n2);
}
__device__ __forceinline__ double inv_r3_ieee(double n2) {
    if constexpr (kPairVariant == 1) { const double r = sqrt(n2); return 1.0 / (r * r * r); }
    else if constexpr (kPairVariant == 2) { const double s = 1.0 / sqrt(n2); return s * s * s; }
    else if constexpr (kPairVariant == 3) return (1.0 / n2) * (1.0 / sqrt(n2));
    else return 1.0 / (n2 * sqrt(n2));
}

// ---- differences and squared distance ---------------------------------------------------------------------------------------
struct PairPre { double dx, dy, dz, n2; unsigned lo; };
__device__ __forceinline__ PairPre pair_pre(double dx, double dy, double dz) {
    PairPre p;
    p.dx = dx; p.dy = dy; p.dz = dz;
    const double sx = dx * dx, sy = dy * dy, sz = dz * dz;
    p.n2 = sx + sy + sz;                              // glam DVec3::length_squared, left to right
    p.lo = kNeedLow ? min(min((unsigned)__double2hiint(sx), (unsigned)__double2hiint(sy)), (unsigned)__double2hiint(sz)) : 0u;
    return p;
}
__device__ __forceinline__ PairPre pair_pre(double xi, double yi, double zi, const Body4 &pj) {
    return pair_pre(pj.x - xi, pj.y - yi, pj.z - zi);
}

// ---- the term: both halves of a pair share `den` -------------------------------------------------------------------------------
// pair_den<FAST>(n2): orders 0-3: v = 1/r^3; orders 4-6: v = p and, FAST, r = RN(1/p).
// pair_apply<FAST>(den, d, mu): the acceleration d-wards of a mass mu. pair_apply<true> tests ITS numerators itself (lane test,
// wave-uniform branch to the compiler's division): callers only guarantee in_range(n2) for the whole wave. The hot loops of the
// workgroup kernel use pair_finish_staged below instead, with every test folded into one ballot per tile.
struct PairDen { double v, r; };
template <bool FAST>
__device__ __forceinline__ PairDen pair_den(double n2) {
    PairDen d;
    if constexpr (!kDivForm) {
        d.v = FAST ? inv_r3_inrange(n2) : inv_r3_ieee(n2);
        d.r = 0.0;
    } else if constexpr (FAST) {
        d.r = inv_r3_seeded(n2, &d.v);
    } else {
        d.v = n2 * sqrt(n2);
        d.r = 0.0;
    }
    return d;
}
template <bool FAST>
__device__ __forceinline__ void pair_apply(const PairDen &d, double dx, double dy, double dz, double mu, double &cx,
                                           double &cy, double &cz) {
    if constexpr (!kDivForm) {
        const double s = mu * d.v;
        cx = dx * s;
        cy = dy * s;
        cz = dz * s;
    } else if constexpr (!FAST) {
        if constexpr (kPairVariant == 4) { cx = (dx * mu) / d.v; cy = (dy * mu) / d.v; cz = (dz * mu) / d.v; }
        else if constexpr (kPairVariant == 5) { const double s = mu / d.v; cx = dx * s; cy = dy * s; cz = dz * s; }
        else { cx = (dx / d.v) * mu; cy = (dy / d.v) * mu; cz = (dz / d.v) * mu; }
    } else {
        // straight-line quotients for every lane; a numerator outside the guarded range anywhere in the wave (a zero component,
        // a massless source, denormal-scale products) sends the wave through the compiler's division behind ONE uniform branch
        // that ordinary operands never take. (The empty asm keeps the branch a branch: flattened into selects every lane would
        // pay the three IEEE divisions as well -- round 3 measured 76.8 against 62.3 us per step.)
        if constexpr (kPairVariant == 4) {
            const double nx = dx * mu, ny = dy * mu, nz = dz * mu;
            cx = div_refined(nx, d.v, d.r); cy = div_refined(ny, d.v, d.r); cz = div_refined(nz, d.v, d.r);
            const bool bad = !(in_range_div(nx) && in_range_div(ny) && in_range_div(nz));
            if (__builtin_amdgcn_ballot_w64(bad) != 0) { asm volatile(""); cx = nx / d.v; cy = ny / d.v; cz = nz / d.v; }
        } else if constexpr (kPairVariant == 5) {
            double s = div_refined(mu, d.v, d.r);
            if (__builtin_amdgcn_ballot_w64(mu_key(mu) != 0u) != 0) { asm volatile(""); s = mu / d.v; }
            cx = dx * s; cy = dy * s; cz = dz * s;
        } else {
            double qx = div_refined(dx, d.v, d.r), qy = div_refined(dy, d.v, d.r), qz = div_refined(dz, d.v, d.r);
            const bool bad = !(in_range_div(dx) && in_range_div(dy) && in_range_div(dz));
            if (__builtin_amdgcn_ballot_w64(bad) != 0) { asm volatile(""); qx = dx / d.v; qy = dy / d.v; qz = dz / d.v; }
            cx = qx * mu; cy = qy * mu; cz = qz * mu;
        }
    }
}
// one directed term. FAST: the CALLER has tested range_key(n2), low_key(lo) and mu_key(mu) for the whole wave.
template <bool FAST>
__device__ __forceinline__ void pair_finish(const PairPre &p, double mu, double &cx, double &cy, double &cz) {
    if constexpr (!kDivForm) {
        const double s = mu * (FAST ? inv_r3_inrange(p.n2) : inv_r3_ieee(p.n2));
        cx = p.dx * s;
        cy = p.dy * s;
        cz = p.dz * s;
    } else if constexpr (!FAST) {
        pair_apply<false>(pair_den<false>(p.n2), p.dx, p.dy, p.dz, mu, cx, cy, cz);
    } else {
        double pp;
        const double r = inv_r3_seeded(p.n2, &pp);
        if constexpr (kPairVariant == 4) {
            cx = div_refined(p.dx * mu, pp, r); cy = div_refined(p.dy * mu, pp, r); cz = div_refined(p.dz * mu, pp, r);
        } else if constexpr (kPairVariant == 5) {
            const double s = div_refined(mu, pp, r);
            cx = p.dx * s; cy = p.dy * s; cz = p.dz * s;
        } else {
            cx = div_refined(p.dx, pp, r) * mu; cy = div_refined(p.dy, pp, r) * mu; cz = div_refined(p.dz, pp, r) * mu;
        }
    }
}

// The FAST term for M interactions at once, STAGE by stage with the VALU order pinned (sched_barrier between stages): the M
// operations of a stage are independent, so a wave covers part of the dependent latency on its own instead of leaving all of it
// to the other waves of its SIMD. Same operations as pair_finish<true>, same bits. Orders 0 and 4-6 (the seeded reciprocal);
// orders 1-3 take the compiler's schedule. Measured at N = 4096 (order 0): 36.9 against 37.3 us per step
// (profiles/r03_step_kernel_evidence.md section 6); the division forms: profiles/r04_pair_variants.md.
constexpr int kSchedMask = 0x4 | 0x10 | 0x80;   // SALU, VMEM, DS may cross a sched_barrier; VALU stays pinned
constexpr bool kPairStaged = kPairVariant == 0 || kDivForm;
template <int M>
__device__ __forceinline__ void pair_finish_staged(const PairPre (&pre)[M], const double (&mu)[M], double (&c)[3 * M]) {
    double x[M], g[M], h[M], r[M], d[M], p[M], q[M], e[M];
#define EPH_STAGE(body) _Pragma("unroll") for (int k = 0; k < M; ++k) { body; } __builtin_amdgcn_sched_barrier(kSchedMask)
    EPH_STAGE(x[k] = pre[k].n2; q[k] = __builtin_amdgcn_rsq(x[k]));
    EPH_STAGE(g[k] = x[k] * q[k]; h[k] = q[k] * 0.5);
    EPH_STAGE(r[k] = __builtin_fma(-h[k], g[k], 0.5));
    EPH_STAGE(g[k] = __builtin_fma(g[k], r[k], g[k]); h[k] = __builtin_fma(h[k], r[k], h[k]));
    EPH_STAGE(d[k] = __builtin_fma(-g[k], g[k], x[k]));
    EPH_STAGE(g[k] = __builtin_fma(d[k], h[k], g[k]));
    EPH_STAGE(d[k] = __builtin_fma(-g[k], g[k], x[k]); q[k] = h[k] * h[k]);
    EPH_STAGE(g[k] = __builtin_fma(d[k], h[k], g[k]); q[k] = q[k] * h[k]);
    EPH_STAGE(p[k] = x[k] * g[k]; q[k] = q[k] * 8.0);
    EPH_STAGE(e[k] = __builtin_fma(-p[k], q[k], 1.0));
    EPH_STAGE(q[k] = __builtin_fma(q[k], e[k], q[k]));
    EPH_STAGE(e[k] = __builtin_fma(-p[k], q[k], 1.0));
    EPH_STAGE(q[k] = __builtin_fma(e[k], q[k], q[k]));                      // RN(1 / p)
    if constexpr (kPairVariant == 0) {
        EPH_STAGE(q[k] = mu[k] * q[k]);
        EPH_STAGE(c[3 * k] = pre[k].dx * q[k]; c[3 * k + 1] = pre[k].dy * q[k]; c[3 * k + 2] = pre[k].dz * q[k]);
    } else if constexpr (kPairVariant == 5) {                                // d * (mu / p)
        EPH_STAGE(g[k] = mu[k] * q[k]);
        EPH_STAGE(e[k] = __builtin_fma(-p[k], g[k], mu[k]));
        EPH_STAGE(g[k] = __builtin_fma(e[k], q[k], g[k]));
        EPH_STAGE(c[3 * k] = pre[k].dx * g[k]; c[3 * k + 1] = pre[k].dy * g[k]; c[3 * k + 2] = pre[k].dz * g[k]);
    } else if constexpr (kPairVariant == 4) {                                // (d * mu) / p, component-wise
        double n[3 * M], t[3 * M];
        EPH_STAGE(n[3 * k] = pre[k].dx * mu[k]; n[3 * k + 1] = pre[k].dy * mu[k]; n[3 * k + 2] = pre[k].dz * mu[k]);
        EPH_STAGE(t[3 * k] = n[3 * k] * q[k]; t[3 * k + 1] = n[3 * k + 1] * q[k]; t[3 * k + 2] = n[3 * k + 2] * q[k]);
        EPH_STAGE(n[3 * k] = __builtin_fma(-p[k], t[3 * k], n[3 * k]); n[3 * k + 1] = __builtin_fma(-p[k], t[3 * k + 1], n[3 * k + 1]);
                  n[3 * k + 2] = __builtin_fma(-p[k], t[3 * k + 2], n[3 * k + 2]));
        EPH_STAGE(c[3 * k] = __builtin_fma(n[3 * k], q[k], t[3 * k]); c[3 * k + 1] = __builtin_fma(n[3 * k + 1], q[k], t[3 * k + 1]);
                  c[3 * k + 2] = __builtin_fma(n[3 * k + 2], q[k], t[3 * k + 2]));
    } else {                                                                 // (d / p) * mu
        double t[3 * M], f[3 * M];
        EPH_STAGE(t[3 * k] = pre[k].dx * q[k]; t[3 * k + 1] = pre[k].dy * q[k]; t[3 * k + 2] = pre[k].dz * q[k]);
        EPH_STAGE(f[3 * k] = __builtin_fma(-p[k], t[3 * k], pre[k].dx); f[3 * k + 1] = __builtin_fma(-p[k], t[3 * k + 1], pre[k].dy);
                  f[3 * k + 2] = __builtin_fma(-p[k], t[3 * k + 2], pre[k].dz));
        EPH_STAGE(t[3 * k] = __builtin_fma(f[3 * k], q[k], t[3 * k]); t[3 * k + 1] = __builtin_fma(f[3 * k + 1], q[k], t[3 * k + 1]);
                  t[3 * k + 2] = __builtin_fma(f[3 * k + 2], q[k], t[3 * k + 2]));
        EPH_STAGE(c[3 * k] = t[3 * k] * mu[k]; c[3 * k + 1] = t[3 * k + 1] * mu[k]; c[3 * k + 2] = t[3 * k + 2] * mu[k]);
    }
#undef EPH_STAGE
}
