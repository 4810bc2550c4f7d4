// craft_sweep.hip -- the massless sweep's kernels: one device thread (k_craft_propagate, k_craft_queue) or one wave
// (k_craft_wave) per spacecraft runs the whole adaptive embedded Runge-Kutta loop against the massive bodies' piecewise-polynomial
// ephemeris. Compiled once per evaluation order of the point-mass term (pair_ns.h); the host side of the batch, the event
// search and the evaluators are in craft.hip. Mirrors (paths relative to the reference repository root):
//   SpacecraftPropagator::{new, step, reset_integrator}, SpacecraftModel, Timeline, CubicHermiteSplineSolout
//                                                       ephemeris/src/propagators/spacecraft.rs:58-332,415-695
//   AdaptiveRungeKuttaIntegrator::advance, IController::step, PreviousStep
//                                                       integration/src/runge_kutta/mod.rs:188-285,396-440
//   ERK::{advance, error, undo_step}                    integration/src/runge_kutta/explicit.rs:54-141
//   Bodies::acceleration, GravitationalBody::acceleration_at, TNB, ReferenceFrame, AbsTol
//                                                       ephemeris_explorer/src/dynamics/spacecraft.rs:70-74,218-293,609-641
// Same f64 operations in the same order as the CPU path (-ffp-contract=off).
#include <algorithm>

#include "pair_ns.h"
#include "craft_device.h"

namespace eph {
namespace EPH_PV_NS {

#ifndef EPH_CRAFT_SCALAR_ROWS
#define EPH_CRAFT_SCALAR_ROWS 1
#endif
// One body's term of Bodies::acceleration (dynamics/spacecraft.rs:70-74,222-228): segment lookup, Horner, point mass.
__device__ __forceinline__ bool body_term(const CraftArgs &a, const BodyEntry &be, double t, const V3 &pos, V3 &term) {
    long long idx;
    double tau;
    if (!spline_locate_fast(be, t, idx, tau)) return false;
    // eval_slice_horner over all kDiv rows: rows >= ncoef are +0.0 in the device table (eph_ephemeris_create), so
    // the leading steps give 0*tau + 0 = +0, the state the reference's Horner starts from -- same bits, no
    // ncoef load, no loop, and twelve 16-byte loads in flight at once
    V3 bp = {0.0, 0.0, 0.0};
    const long long row = be.coeff_off + idx;         // (in the wave-per-craft form of > 64 bodies the lanes differ in be too)
    const long long row0 = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)row) |
                           ((long long)__builtin_amdgcn_readfirstlane((int)(row >> 32)) << 32);
    if (EPH_CRAFT_SCALAR_ROWS && __builtin_amdgcn_ballot_w64(row != row0) == 0) {
        // every lane of the wave is inside the SAME polynomial (craft of one sweep started together: the usual case): its 24
        // coefficients come through the scalar cache into SGPRs instead of 64 lanes x 192 B through the vector L1
        const auto *cs = (const __attribute__((address_space(4))) double *)(unsigned long long)(a.coeffs + row0 * kDiv * 3);
#pragma unroll
        for (int k = kDiv - 1; k >= 0; --k) {
            bp.x = bp.x * tau + cs[k * 3 + 0];
            bp.y = bp.y * tau + cs[k * 3 + 1];
            bp.z = bp.z * tau + cs[k * 3 + 2];
        }
    } else {
        const double2 *co = reinterpret_cast<const double2 *>(a.coeffs + row * kDiv * 3);
        double c[kDiv * 3];
#pragma unroll
        for (int q = 0; q < kDiv * 3 / 2; ++q) { const double2 v = co[q]; c[2 * q] = v.x; c[2 * q + 1] = v.y; }
#pragma unroll
        for (int k = kDiv - 1; k >= 0; --k) {
            bp.x = bp.x * tau + c[k * 3 + 0];
            bp.y = bp.y * tau + c[k * 3 + 1];
            bp.z = bp.z * tau + c[k * 3 + 2];
        }
    }
    const V3 d = sub(bp, pos);                        // acceleration_at::<false>: dir = body - at
    const double n2 = dot(d, d);
    // the point-mass term in the build's evaluation order, IEEE sqrt and divide (pair_term.h)
    if (__builtin_amdgcn_ballot_w64(!in_range(n2)) == 0) pair_apply<true>(pair_den<true>(n2), d.x, d.y, d.z, be.mu, term.x, term.y, term.z);
    else pair_apply<false>(pair_den<false>(n2), d.x, d.y, d.z, be.mu, term.x, term.y, term.z);
    return true;
}
// k_craft_wave: lane b always evaluates body b, so the body's table entry, the refined reciprocal of its spline
// interval and the coefficients of the polynomial it is currently in stay in the lane's registers; the polynomial is
// reloaded only when the segment index changes (every few hundred steps). Used when n_bodies <= 64.
struct LaneBody {
    BodyEntry be;
    double r;                 // rcp_refined(be.interval)
    bool b_ok;                // interval in range for the wrapper-free division
    long long idx;            // segment whose coefficients are in c (-1: none)
    double c[kDiv * 3];
};
__device__ __forceinline__ bool body_term_cached(const CraftArgs &a, LaneBody &lb, double t, const V3 &pos, V3 &term) {
    // UniformSpline::get_polynomial, the two divisions by the interval through the shared reciprocal (same quotients)
    const BodyEntry &b = lb.be;
    const double local = t - b.start;
    if (__builtin_signbit(local) || local > b.span) return false;
    const double cq = ceil(div_shared(local, b.interval, lb.r, lb.b_ok));
    unsigned long long i;
    double fi;
    if (__builtin_amdgcn_ballot_w64(!(cq < 2147483648.0)) == 0) {
        const unsigned ci = cq <= 0.0 ? 0u : (unsigned)cq;
        const unsigned i32 = ci == 0 ? 0u : ci - 1u;
        i = i32;
        fi = (double)i32;
    } else {
        const unsigned long long ci = cq <= 0.0 ? 0ull : (cq >= 18446744073709551616.0 ? ~0ull : (unsigned long long)cq);
        i = ci == 0 ? 0 : ci - 1;
        fi = (double)i;
    }
    if (i >= (unsigned long long)b.npoly) return false;
    const long long idx = (long long)i;
    const double tau = div_shared(local - b.interval * fi, b.interval, lb.r, lb.b_ok);
    if (idx != lb.idx) {
        const double2 *co = reinterpret_cast<const double2 *>(a.coeffs + (b.coeff_off + idx) * kDiv * 3);
#pragma unroll
        for (int q = 0; q < kDiv * 3 / 2; ++q) { const double2 v = co[q]; lb.c[2 * q] = v.x; lb.c[2 * q + 1] = v.y; }
        lb.idx = idx;
    }
    V3 bp = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = kDiv - 1; k >= 0; --k) {
        bp.x = bp.x * tau + lb.c[k * 3 + 0];
        bp.y = bp.y * tau + lb.c[k * 3 + 1];
        bp.z = bp.z * tau + lb.c[k * 3 + 2];
    }
    const V3 d = sub(bp, pos);
    const double n2 = dot(d, d);
    if (__builtin_amdgcn_ballot_w64(!in_range(n2)) == 0) pair_apply<true>(pair_den<true>(n2), d.x, d.y, d.z, b.mu, term.x, term.y, term.z);
    else pair_apply<false>(pair_den<false>(n2), d.x, d.y, d.z, b.mu, term.x, term.y, term.z);
    return true;
}
__device__ __forceinline__ unsigned div_key(double x) { return (unsigned)(__double2hiint(x) - 0x33700000); }    // in_range_div: key < 0x19000000
// The same term with the lookup SPECULATIVE and branch-free, like the thread-per-craft kernels' locate_spec (every guarded choice of
// body_term_cached assumed -- the lane's own reciprocal for both quotients, 32-bit segment count, in-range squared distance -- and ONE
// ballot at the end; whatever does not hold sends the wave through body_term_cached itself, which gives the same bits). On the single
// wave this kernel is, every branch of the guarded form is a bubble: a ship's step 32.0-32.7 -> 30.7-31.1 us (64 craft: 27.5 -> 26.1).
__device__ __forceinline__ bool body_term_wave(const CraftArgs &a, LaneBody &lb, double t, const V3 &pos, V3 &term) {
    const BodyEntry &b = lb.be;
    const double local = t - b.start;
    const double cq = ceil(div_refined(local, b.interval, lb.r));
    const unsigned ci = (unsigned)fmin(fmax(cq, 0.0), 2147483648.0);
    const unsigned i32 = ci == 0 ? 0u : ci - 1u;
    const double rem = local - b.interval * (double)i32;
    const double tau = div_refined(rem, b.interval, lb.r);
    bool bad = !lb.b_ok | (max(div_key(local), div_key(rem)) >= 0x19000000u) | (local > b.span) |
               ((unsigned long long)i32 >= (unsigned long long)b.npoly);
    const long long idx = (long long)i32;
    if (!bad && idx != lb.idx) {                      // (the lane's polynomial changed: every few hundred steps)
        const double2 *co = reinterpret_cast<const double2 *>(a.coeffs + (b.coeff_off + idx) * kDiv * 3);
#pragma unroll
        for (int q = 0; q < kDiv * 3 / 2; ++q) { const double2 v = co[q]; lb.c[2 * q] = v.x; lb.c[2 * q + 1] = v.y; }
        lb.idx = idx;
    }
    V3 bp = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = kDiv - 1; k >= 0; --k) {
        bp.x = bp.x * tau + lb.c[k * 3 + 0];
        bp.y = bp.y * tau + lb.c[k * 3 + 1];
        bp.z = bp.z * tau + lb.c[k * 3 + 2];
    }
    const V3 d = sub(bp, pos);
    const double n2 = dot(d, d);
    pair_apply<true>(pair_den<true>(n2), d.x, d.y, d.z, b.mu, term.x, term.y, term.z);
    bad = bad | !in_range(n2);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
        asm volatile("");
        return body_term_cached(a, lb, t, pos, term);
    }
    return true;
}
__device__ __forceinline__ double lane_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// ---- Bodies::acceleration of the thread-per-craft kernels, PIPELINED over the bodies (round 6) -----------------------------------
// Round 5's loop evaluated one body_term after the other: per (stage, body) two dependent scalar-memory round trips (the table
// entry, then the coefficient row whose address needs the segment index), a dozen taken branches between small blocks (every
// guarded fast path its own block, laid out behind the slow ones) and one dependent chain at a time; inlined thirteen times it was
// 82 KB of code for the 13-stage pair -- more than the instruction cache. Now, per body b:
//   P(b)  Horner over the coefficient row that was PREFETCHED into SGPRs while body b-1's term was computed (three independent
//         chains); lanes inside OTHER polynomials than the first lane's (a long sweep's craft drift apart in time) load their own rows,
//         out of line;
//   R(b)  difference and squared distance, the point-mass term of body b -- one dependent chain -- TOGETHER WITH the segment lookup
//         of body b+1 (a second, independent chain) in ONE basic block; the table entry and the row of body b+1 are requested here
//         and arrive under the term's arithmetic.
// The lookup is speculative and branch-free (every guarded choice of spline_locate_fast assumed; one ballot says whether all held);
// whatever does not fit -- a lane outside its spline (-> EvalFailed), an interval or operand outside the guarded ranges, more than
// 2^31 polynomials -- goes through ONE out-of-line copy of the plain IEEE path per kernel (body_position_generic / pair_generic),
// which produces the same bits (ieee_seq.h: the guarded sequences ARE the compiler's expansions without their no-op wrappers).
// Same operations in the same order per body, terms added in a.bodies order.
__device__ __forceinline__ BodyEntry entry_uniform(const BodyEntry *table, int b) {
    // the body's table entry is the same for every lane: scalar loads through the constant address space
    const int bu = __builtin_amdgcn_readfirstlane(b);
    const auto *bc = (const __attribute__((address_space(4))) BodyEntry *)(unsigned long long)(table + bu);
    BodyEntry be;
    be.start = bc->start; be.interval = bc->interval; be.mu = bc->mu; be.npoly = bc->npoly;
    be.coeff_off = bc->coeff_off; be.span = bc->span; be.rinv = bc->rinv; be.rows = bc->rows;
    return be;
}
__device__ __forceinline__ long long uniform64(long long v) {
    return (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v) | ((long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32);
}
__device__ __forceinline__ long long lane64(long long v, int l) {
    return (long long)(unsigned)__builtin_amdgcn_readlane((int)v, l) | ((long long)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32);
}
struct RowS { double c[kDiv * 3]; };                  // one polynomial's 24 coefficients, wave-uniform: SGPRs
// EPH_CRAFT_ROW_SPLIT=1 loads it in two parts -- the 16 high coefficients (Horner's first five steps) a body ahead, the 8 low ones at
// the start of the body's own Horner pass -- so that 32 instead of 48 SGPRs stay occupied across the previous body's term and none of
// the row is spilled to VGPR lanes. Measured on one box (scripts/ab_craft.sh, 262 144 craft x 0.25 d): 31.1 ms against 29.9-30.0 for
// the whole row a body ahead (8 v_readlane per term, but no exposed scalar-cache latency in front of Horner): off.
#ifndef EPH_CRAFT_ROW_SPLIT
#define EPH_CRAFT_ROW_SPLIT 0
#endif
constexpr int kRowLo = EPH_CRAFT_ROW_SPLIT ? 8 : 0;   // coefficients [0, kRowLo) are loaded late
__device__ __forceinline__ void row_uniform_hi(const double *rows, unsigned i0, RowS &r) {
    const auto *cs = (const __attribute__((address_space(4))) double *)(unsigned long long)(rows + (size_t)i0 * (kDiv * 3));
#pragma unroll
    for (int k = kRowLo; k < kDiv * 3; ++k) r.c[k] = cs[k];
}
__device__ __forceinline__ void row_uniform_lo(const double *rows, unsigned i0, RowS &r) {
    const auto *cs = (const __attribute__((address_space(4))) double *)(unsigned long long)(rows + (size_t)i0 * (kDiv * 3));
#pragma unroll
    for (int k = 0; k < kRowLo; ++k) r.c[k] = cs[k];
}
// eval_slice_horner over all kDiv rows: rows >= ncoef are +0.0 in the device table (eph_ephemeris_create), so the leading steps give
// 0*tau + 0 = +0, the state the reference's Horner starts from -- same bits, no ncoef load, no loop
// ... and the first step without its product: the reference starts from T::default() = +0 (trajectory.rs:404-407), and (+0) * tau is +0
// for every tau the speculative lookup lets through (tau > 0: the remainder of a positive `local`), so step one is (+0) + c -- kept as
// an addition, because (+0) + (-0) is +0, not -0. Three multiplications less per term; same bits.
__device__ __forceinline__ V3 horner_row(const RowS &r, double tau) {
    V3 bp = {0.0 + r.c[(kDiv - 1) * 3 + 0], 0.0 + r.c[(kDiv - 1) * 3 + 1], 0.0 + r.c[(kDiv - 1) * 3 + 2]};
#pragma unroll
    for (int k = kDiv - 2; k >= 0; --k) {
        bp.x = bp.x * tau + r.c[k * 3 + 0];
        bp.y = bp.y * tau + r.c[k * 3 + 1];
        bp.z = bp.z * tau + r.c[k * 3 + 2];
    }
    return bp;
}
// UniformSpline::get_polynomial, speculatively: spline_locate_fast with every guarded choice assumed (shared reciprocal for both
// quotients, 32-bit segment count). Straight-line. When every lane's assumptions hold and it is inside the spline (and the entry
// passes entry_fast), tau and idx are spline_locate_fast's. The tests are folded: both
// numerators in the guarded range of the shared-reciprocal division (one max over their range keys; a negative, zero or NaN `local`
// has a key outside it, so the sign test is implied -- t == start exactly, where local = rem = +0 is a legal numerator, takes the
// out-of-line path), not beyond the span, and the segment inside the table (which also catches a count that was clamped at 2^31:
// entry_fast requires npoly < 2^31).
// Returns the wave's ballot of lanes for which an assumption does NOT hold (0 = every lane fine), as the OR of one ballot per test: a
// ballot of a single compare is the compare's own result mask, a ballot of a combined per-lane flag costs a select and a second compare.
__device__ __forceinline__ unsigned long long locate_spec(const BodyEntry &b, double at, double &tau, unsigned &idx) {
    const double local = at - b.start;
    const double c = ceil(div_refined(local, b.interval, b.rinv));
    const unsigned ci = (unsigned)fmin(fmax(c, 0.0), 2147483648.0);       // c <= 0 (and NaN) -> 0, as `c <= 0.0 ? 0u : (unsigned)c`
    const unsigned i32 = ci == 0 ? 0u : ci - 1u;
    const double rem = local - b.interval * (double)i32;
    tau = div_refined(rem, b.interval, b.rinv);
    idx = i32;
    return __builtin_amdgcn_ballot_w64(max(div_key(local), div_key(rem)) >= 0x19000000u) | __builtin_amdgcn_ballot_w64(local > b.span) |
           __builtin_amdgcn_ballot_w64(i32 >= (unsigned)b.npoly);
}
// wave-uniform, one scalar compare: the entry's refined reciprocal is +0.0 unless the interval is in the guarded range of the
// shared-reciprocal division AND the segment count fits 31 bits (k_body_reciprocals); a nonzero one is a normal number
__device__ __forceinline__ bool entry_fast(const BodyEntry &b) { return __double2hiint(b.rinv) != 0; }
// the out-of-line IEEE path (cold): UniformSpline::get_polynomial + eval with the compiler's divisions, per-lane coefficient loads
struct BodyPos { double x, y, z; int located; };
// (reads the table entry itself: the hot loop then keeps nothing of an entry alive for this call's sake)
__device__ __noinline__ BodyPos body_position_generic(const double *coeffs, const BodyEntry *table, int b, double t) {
    BodyEntry be = table[b];
    be.span = be.interval * (double)be.npoly;
    const long long coeff_off = be.coeff_off;
    long long idx;
    double tau;
    if (!spline_locate(be, t, idx, tau)) return BodyPos{0.0, 0.0, 0.0, 0};
    const double *co = coeffs + (coeff_off + idx) * kDiv * 3;
    V3 bp = {0.0, 0.0, 0.0};
    for (int k = kDiv - 1; k >= 0; --k) {
        bp.x = bp.x * tau + co[k * 3 + 0];
        bp.y = bp.y * tau + co[k * 3 + 1];
        bp.z = bp.z * tau + co[k * 3 + 2];
    }
    return BodyPos{bp.x, bp.y, bp.z, 1};
}
// Horner with PER-LANE coefficient loads (cold): the lanes of a wave that are inside many different polynomials of one body -- craft
// whose epochs have drifted far apart (an undealt heterogeneous batch in the work-queue kernel: lanes pick up new craft at any time)
__device__ __noinline__ V3 horner_lane_rows(const double *rows, unsigned idx, double tau) {
    const double2 *co = reinterpret_cast<const double2 *>(rows + (size_t)idx * (kDiv * 3));
    double c[kDiv * 3];
#pragma unroll
    for (int q = 0; q < kDiv * 3 / 2; ++q) { const double2 v = co[q]; c[2 * q] = v.x; c[2 * q + 1] = v.y; }
    V3 bp = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = kDiv - 1; k >= 0; --k) {
        bp.x = bp.x * tau + c[k * 3 + 0];
        bp.y = bp.y * tau + c[k * 3 + 1];
        bp.z = bp.z * tau + c[k * 3 + 2];
    }
    return bp;
}
__device__ __noinline__ V3 pair_generic(double n2, double dx, double dy, double dz, double mu) {
    V3 term;
    pair_apply<false>(pair_den<false>(n2), dx, dy, dz, mu, term.x, term.y, term.z);
    return term;
}
__device__ __forceinline__ bool bodies_acceleration(const CraftArgs &a, double t, const V3 &pos, V3 &acc) {
    const int nb = a.n_bodies;
    if (nb <= 0) return true;
    // A lane outside a body's spline (EvalFailed) does NOT leave the loop: it is flagged and carries harmless numbers to the end (a
    // divergent exit would make every loop-carried value divergent in the compiler's eyes -- table entries and coefficient rows in
    // VGPRs, per-lane loads -- although they are the same for every lane). It stays IN the ballots: where its numbers fail a test the
    // wave takes the out-of-line IEEE path for that body, which gives the other lanes the same bits -- a slow last evaluation for a
    // wave that holds a failing craft, three scalar operations less per term for everybody else.
    bool failed = false;
    // prologue: body 0's entry, lookup and the high part of its row
    BodyEntry be = entry_uniform(a.bodies, 0);
    double tau;
    unsigned idx;
    unsigned long long bad = locate_spec(be, t, tau, idx);
    bool all_good = entry_fast(be) & (bad == 0);                                                        // wave-uniform
    unsigned i0 = (unsigned)__builtin_amdgcn_readfirstlane((int)idx) & -(unsigned)all_good;            // (branch-free: row 0 of a body is always a valid address)
    const double *rows = be.rows;                     // (a row address from the entry alone: no table base in the loop)
    RowS cs;
    row_uniform_hi(rows, i0, cs);
    for (int b = 0; b < nb; ++b) {
        // ---- P(b): the body's position at t
        row_uniform_lo(rows, i0, cs);
        V3 bp;
        if (__builtin_expect(all_good, 1)) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(idx != i0) == 0, 1)) {
                bp = horner_row(cs, tau);             // every lane inside the SAME polynomial (craft of one sweep started together)
            } else {
                // the lanes inside the prefetched polynomial take it from the SGPRs; the others load their own rows, out of line.
                // (A waterfall -- one scalar-cache pass per distinct polynomial -- was measured: unbounded it costs the work-queue
                // kernel, whose waves can hold 64 craft at 64 epochs, a factor of three on the undealt mixed population (1010 ms per
                // sweep; round 4: 350); bounded to 3 / 2 / 1 passes 413 / 370 / 329 ms, the dealt sweeps unchanged by the bound.)
                bp = horner_row(cs, tau);
                if (idx != i0) bp = horner_lane_rows(rows, idx, tau);
            }
        } else {
            asm volatile("");
            const BodyPos g = body_position_generic(a.coeffs, a.bodies, b, t);
            failed = failed | !g.located;
            bp = V3{g.x, g.y, g.z};
        }
        // ---- R(b): the term of body b, with the lookup of body b + 1 beside it and the next row / entry requested
        const double mu = be.mu;
        const V3 d = sub(bp, pos);                    // acceleration_at::<false>: dir = body - at
        const double n2 = dot(d, d);
        const bool more = b + 1 < nb;
        // (the last body's block looks up entry 0 once more and discards it: a lookup behind `if (more)` was measured -- the branch
        // cuts the block in three, the lookup no longer overlaps the term and the allocation changes: 40.5 against 29.2 ms)
        // (ONE entry in SGPRs at a time, its latency under body b's term; requested a body earlier -- before Horner -- the second entry
        // costs the row its registers: 32.4 against 27.5 ms)
        be = entry_uniform(a.bodies, more ? b + 1 : 0);
        bad = locate_spec(be, t, tau, idx);
        // the point-mass term in the build's evaluation order, IEEE sqrt and divide (pair_term.h): the wrapper-free sequences for
        // every lane; a squared distance outside the guarded range anywhere in the wave sends it through the compiler's expansions
        const PairDen den = pair_den<true>(n2);
        all_good = more & entry_fast(be) & (bad == 0);
        i0 = (unsigned)__builtin_amdgcn_readfirstlane((int)idx) & -(unsigned)all_good;
        rows = be.rows;
        row_uniform_hi(rows, i0, cs);
        V3 term;
        pair_apply<true>(den, d.x, d.y, d.z, mu, term.x, term.y, term.z);
        // (the empty asm keeps the straight-line term IN this block, beside the lookup of the next body: two independent chains; without
        // it the compiler sinks the term behind the range test into a block of its own)
        asm volatile("" : "+v"(term.x), "+v"(term.y), "+v"(term.z));
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!in_range(n2)) != 0, 0)) {
            asm volatile("");
            term = pair_generic(n2, d.x, d.y, d.z, mu);
        }
        acc = add(acc, term);
    }
    return !failed;
}
constexpr int kRedRow = kTile + 2;                    // LDS row of the wave variant's contribution tile

// The manoeuvre's acceleration in its frame (ConstantThrust in ReferenceFrame::Relative(body) -> TNB of (craft - body), or inertial):
// ephemeris/src/propagators/spacecraft.rs:319-331, ephemeris_explorer/src/dynamics/spacecraft.rs:240-293. false = EvalFailed.
__device__ __forceinline__ bool burn_acceleration(const BodyEntry *bodies_by_index, const double *coeffs, const int *ncoef, int ref, double ax,
                                                  double ay, double az, double t, const V3 &pos, const V3 &vel, V3 &man) {
    const V3 thrust = {ax, ay, az};
    if (ref >= 0) {                                   // ReferenceFrame::Relative -> TNB::try_new(sv - ref.state_vector(t))
        const BodyEntry be = bodies_by_index[ref];
        long long idx;
        double tau;
        if (!spline_locate(be, t, idx, tau)) return false;
        const double *co = coeffs + (be.coeff_off + idx) * kDiv * 3;
        const int nc = ncoef[be.coeff_off + idx];
        double rp[3], rv[3];
        for (int c = 0; c < 3; ++c) {                 // Polynomial::eval_and_deriv
            const double first = nc ? co[c] : 0.0;
            const double last = nc ? co[(nc - 1) * 3 + c] : 0.0;
            double e = last, d = last;
            for (int k = nc - 2; k >= 1; --k) {
                e = e * tau + co[k * 3 + c];
                d = d * tau + e;
            }
            e = e * tau + first;
            rp[c] = e;
            rv[c] = d / be.interval;
        }
        const V3 rel_p = sub(pos, V3{rp[0], rp[1], rp[2]}), rel_v = sub(vel, V3{rv[0], rv[1], rv[2]});
        V3 x, yv;
        if (!try_normalize(rel_v, x)) return false;
        if (!try_normalize(cross(rel_p, rel_v), yv)) return false;
        const V3 xy = cross(x, yv);
        const V3 z = scale(xy, length_recip(xy));
        V3 r = scale(x, thrust.x);                    // DMat3::from_cols(x, z, y).mul_vec3(thrust)
        r = add(r, scale(z, thrust.y));
        r = add(r, scale(yv, thrust.z));
        man = r;
    } else {                                          // TNB::IDENTITY.mul_vec3(thrust)
        V3 r = scale(V3{1.0, 0.0, 0.0}, thrust.x);
        r = add(r, scale(V3{0.0, 1.0, 0.0}, thrust.y));
        r = add(r, scale(V3{0.0, 0.0, 1.0}, thrust.z));
        man = r;
    }
    return true;
}
struct BurnAcc { double x, y, z; int ok; };
__device__ __noinline__ BurnAcc burn_acceleration_cold(const BodyEntry *bodies_by_index, const double *coeffs, const int *ncoef, int ref,
                                                       double ax, double ay, double az, double t, double px, double py, double pz, double vx,
                                                       double vy, double vz) {
    V3 man = {0.0, 0.0, 0.0};
    const bool ok = burn_acceleration(bodies_by_index, coeffs, ncoef, ref, ax, ay, az, t, V3{px, py, pz}, V3{vx, vy, vz}, man);
    return BurnAcc{man.x, man.y, man.z, ok ? 1 : 0};
}

// What the thread-per-craft kernels keep of the current timeline segment in registers: whether it is a burn, and where the rest is
// (the thrust, its frame and the bounds are read back in the burn branch only: ten registers less across every body loop; the
// segment's end is the kernels' `bound`)
struct SegLight { int is_burn; const SegmentDev *full; };
__device__ __forceinline__ const SegmentDev &seg_full(const SegmentDev &sg) { return sg; }
__device__ __forceinline__ SegmentDev seg_full(const SegLight &sg) { return *sg.full; }

// FirstOrderODE::eval for SpacecraftModel (spacecraft.rs:297-308). Returns false for EvalFailed.
// WAVE = false: one thread per spacecraft, the bodies in a loop. WAVE = true: one WAVE per spacecraft (every lane
// carries the same craft state): lane b evaluates body b, the terms go through LDS and lanes 0..2 add them in body
// order -- the same chain of f64 additions -- then the sum is broadcast. `red` = 3 x kRedRow doubles of LDS.
template <bool WAVE, typename Seg>
__device__ __forceinline__ bool craft_rhs(const CraftArgs &a, const Seg &sg, double t, const double (&y)[6],
                                          double (&dy)[6], double *red, LaneBody *lb = nullptr) {
    const V3 pos = {y[0], y[1], y[2]}, vel = {y[3], y[4], y[5]};
    V3 acc = {0.0, 0.0, 0.0};
    if (WAVE) {
        const int lane = threadIdx.x;
        for (int b0 = 0; b0 < a.n_bodies; b0 += kTile) {
            const int b = b0 + lane;
            V3 term = {0.0, 0.0, 0.0};
            bool located = true;
            if (b < a.n_bodies) {
                if (a.n_bodies <= kTile) located = body_term_wave(a, *lb, t, pos, term);
                else {
                    const BodyEntry be = a.bodies[b];
                    located = body_term(a, be, t, pos, term);
                }
            }
            if (__builtin_amdgcn_ballot_w64(!located)) return false;
            red[lane] = term.x;                       // lanes past the last body contribute +0.0 (exact to add)
            red[kRedRow + lane] = term.y;
            red[2 * kRedRow + lane] = term.z;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int left = a.n_bodies - b0;
            const int cnt = ((left < kTile ? left : kTile) + 15) & ~15;
            double sum = lane == 0 ? acc.x : (lane == 1 ? acc.y : acc.z);
            if (lane < 3) {
                const double *row = red + lane * kRedRow;
                for (int c = 0; c < cnt; c += 16) {
                    double2 r[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const double2 *>(row + c + 2 * k);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        sum = sum + r[k].x;
                        sum = sum + r[k].y;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            acc.x = lane_bcast(sum, 0);
            acc.y = lane_bcast(sum, 1);
            acc.z = lane_bcast(sum, 2);
        }
    } else {
        if (!bodies_acceleration(a, t, pos, acc)) return false;   // Bodies::acceleration: the order of a.bodies (eph_craft_batch_set_body_order)
    }
    V3 man = {0.0, 0.0, 0.0};
    if (sg.is_burn) {
        // (thread-per-craft kernels: out of line -- the frame arithmetic with its three IEEE reciprocal square roots, inlined once per
        // stage, was 33 of the 13-stage kernel's 81 KB of code, and burns are minutes of a sweep's days. No measurable difference on
        // a sweep without burns: 25.5-25.6 against 25.6-25.7 ms -- the instruction cache was not the limit.)
        const SegmentDev f = seg_full(sg);
        if (WAVE) {
            if (!burn_acceleration(a.bodies_by_index, a.coeffs, a.ncoef, f.ref, f.ax, f.ay, f.az, t, pos, vel, man)) return false;
        } else {
            const BurnAcc r = burn_acceleration_cold(a.bodies_by_index, a.coeffs, a.ncoef, f.ref, f.ax, f.ay, f.az, t, pos.x, pos.y, pos.z,
                                                     vel.x, vel.y, vel.z);
            if (!r.ok) return false;
            man = V3{r.x, r.y, r.z};
        }
    }
    const V3 tot = add(acc, man);
    dy[0] = vel.x; dy[1] = vel.y; dy[2] = vel.z;
    dy[3] = tot.x; dy[4] = tot.y; dy[5] = tot.z;
    return true;
}

// ---- stage storage of the thread-per-craft kernels ---------------------------------------------------------------------------------
// k_s = (dy.position, dy.velocity) of stage s, and dy.position = y.velocity (spacecraft.rs:303-305): the VELOCITY half of every stage
// derivative is a copy of the stage state's velocity, the ACCELERATION half is the force sum. Round 5 kept both in VGPRs: 13 stages x 6
// doubles = 156 of a lane's 256 registers at two waves per SIMD, and the allocator spilled 100 VGPRs to scratch (counter traffic 5.9 x
// algorithmic: 4.63 GB per launch against 0.78 GB). Round 6: the velocity halves live in LDS, one 512-byte row of 64 lanes per
// (stage, component) -- conflict-free ds_read_b64 / ds_write_b64, no barrier (a lane only ever touches its own column) -- and the
// acceleration halves in registers. 13 stages x 3 x 512 B = 19 968 B per wave; the CU's 160 KB hold the eight waves of two per SIMD.
// A 16-stage pair keeps its last three velocity halves in registers (kCraftLdsStages). LDS traffic is ~350 8-byte accesses per lane
// and attempt against ~31 000 FP64 instructions: nothing. Storage only -- every operand and every operation is the reference's.
// (tuning: -DEPH_CRAFT_LDS_STAGES=8 -DEPH_CRAFT_OCC2_WAVES=3 holds the big-batch kernel to THREE waves per SIMD -- 168 VGPRs, 12 KB of
// LDS per wave, the compiler spilling the rest: 144 spilled VGPRs, 25.6-25.8 against 24.9-25.0 ms, 3 % slower; round 5's same
// experiment on the old structure: 16 % slower)
#ifndef EPH_CRAFT_LDS_STAGES
#define EPH_CRAFT_LDS_STAGES 13
#endif
constexpr int kCraftLdsStages = EPH_CRAFT_LDS_STAGES;
template <int S, bool NYS> struct CraftStages {
    static constexpr int LS = NYS ? 0 : (S < kCraftLdsStages ? S : kCraftLdsStages);      // stages whose velocity half is in LDS
    static constexpr int RS = NYS ? 0 : S - LS;                                            // ... in registers
};
#define EPH_CRAFT_STAGE_STORAGE                                                                            \
    constexpr int LS = CraftStages<S, NYS>::LS;                                                            \
    __shared__ double kv_lds[(LS ? LS : 1) * 3 * 64];                                                      \
    const int lane_ = threadIdx.x;                                                                         \
    double ka[S][3];                                          /* acceleration halves (ERKNG: dk[s]) */     \
    double kvr[CraftStages<S, NYS>::RS ? CraftStages<S, NYS>::RS : 1][3];
#define KV_GET(j, d) ((j) < LS ? kv_lds[((j) * 3 + (d)) * 64 + lane_] : kvr[(j) >= LS ? (j) - LS : 0][d])
#define KV_PUT(j, d, v)                                                                                    \
    do {                                                                                                   \
        if ((j) < LS) kv_lds[((j) * 3 + (d)) * 64 + lane_] = (v);                                          \
        else kvr[(j) >= LS ? (j) - LS : 0][d] = (v);                                                       \
    } while (0)
// the FSAL pairs' k[S-1] / k[0] between calls (a.klast / a.kfirst: [6][n], velocity half first)
#define EPH_CRAFT_LOAD_FSAL(i_)                                                                            \
    _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                        \
        if (!NYS) { KV_PUT(S - 1, d, a.klast[d * n + (i_)]); KV_PUT(0, d, a.kfirst[d * n + (i_)]); }       \
        ka[S - 1][d] = a.klast[((NYS ? 0 : 3) + d) * n + (i_)];                                            \
        ka[0][d] = a.kfirst[((NYS ? 0 : 3) + d) * n + (i_)];                                               \
    }
#define EPH_CRAFT_STORE_FSAL(i_)                                                                           \
    _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                        \
        if (!NYS) { a.klast[d * n + (i_)] = KV_GET(S - 1, d); a.kfirst[d * n + (i_)] = KV_GET(0, d); }     \
        a.klast[((NYS ? 0 : 3) + d) * n + (i_)] = ka[S - 1][d];                                            \
        a.kfirst[((NYS ? 0 : 3) + d) * n + (i_)] = ka[0][d];                                               \
    }

// NYS = false: ERK pair on the 6-vector (explicit.rs).  NYS = true: ERKNG pair on SecondOrderState<[DVec3; 1]>
// (nystrom/explicit_generalized.rs:97-170, the app's Fine45): k[s][0..2] hold dk[s].
// Register budget (OCC = waves per SIMD the allocation is held to): the 13- and 16-stage pairs need > 256 registers
// for the stage derivatives, which leaves ONE wave per SIMD and every L1 hit of the coefficient loads exposed
// (measured, rocprofv3: 48 % of wave time parked on s_waitcnt, 46 % issuing). With more than one wave of craft per
// SIMD, two resident waves with the surplus k[][] spilled to scratch are faster (262144 craft: 1.96e8 vs 1.2e8
// craft-steps/s); with fewer, the unconstrained allocation is (65536 craft: 1.14e8 vs 1.01e8). craft_launch picks
// by batch size. (Also measured: a runtime stage loop around ONE copy of the right-hand side, stage combinations
// selected by a uniform switch -- 4x less code than this unrolled form, which exceeds the instruction cache -- is
// slower, 1.03e8 / 1.12e8: every k[][] element then stays live across the loop and the allocator spills more.)
// The sweep, STATIC form: craft i on thread i for the whole call. The right form when every craft takes about the same
// number of attempts (the north star's sweep: one transfer arc +- 100 km, max / mean attempts per wave 1.09) or when the
// batch fits the chip at once -- which a heterogeneous batch becomes, wave by wave, once its craft are dealt to the lanes by the
// time scale of their orbits (craft_sort: a.perm; the knot slabs keep lane columns). k_craft_queue below is the form for
// heterogeneous batches that were not dealt (craft_launch chooses).
template <int S, bool FSAL, bool NYS = false, int OCC = 1>
#ifndef EPH_CRAFT_OCC2_WAVES
#define EPH_CRAFT_OCC2_WAVES 2
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC == 2 ? EPH_CRAFT_OCC2_WAVES : OCC, OCC == 2 ? EPH_CRAFT_OCC2_WAVES : OCC)))
k_craft_propagate(const CraftArgs a) {
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.n_craft) return;
    const long long i = a.perm ? a.perm[slot] : slot;
    const long long n = a.n_craft;
    int status = a.status[i];
    if (status != EPH_OK && status != EPH_KNOTS_FULL && !a.retry) return;    // a failed craft stays failed until the batch is re-armed
    status = EPH_OK;

    double time = a.time[i], y[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) y[d] = a.y[d * n + i];
    double next_h = a.next_h[i];
    unsigned n_att = a.n_attempts[i], rk_i = a.rk_i[i], steps = a.steps[i];
    int cur = a.cur_seg[i], nk = a.nknots[i];
    double last_knot = a.last_knot_t[i];
    const SegmentDev *segs = a.segs + a.seg_off[i];
    SegLight sg{segs[cur].is_burn, segs + cur};
    double bound = segs[cur].end;
    EPH_CRAFT_STAGE_STORAGE
    if (FSAL) { EPH_CRAFT_LOAD_FSAL(i) }
    const int lower = a.rk.order < a.rk.order_embedded ? a.rk.order : a.rk.order_embedded;

    unsigned taken = 0;
    while (!(last_knot >= a.t_end) && !(a.step_limit && taken >= a.step_limit)) {   // has_reached: solution.end() >= time
        if (nk >= a.max_knots) { status = EPH_KNOTS_FULL; break; }
        // SpacecraftPropagator::step: advance_timeline + reset_integrator  spacecraft.rs:606-609
        if (time >= bound) {                          // (bound = the segment's end)
            cur += 1;
            sg = SegLight{segs[cur].is_burn, segs + cur};
            bound = segs[cur].end;
            next_h = a.h_init;
            n_att = 0;
            rk_i = 0;
        }
        // AdaptiveRungeKuttaIntegrator::advance  mod.rs:414-439
        // PreviousStep: the state is not copied -- the newest knot of the slab IS the state before this step (knot 0 is the initial
        // state, and reset_knots keeps the newest), so a rejection reads it back from there
        const double prev_t = time;
        double prev_klast[6];
#pragma unroll
        for (int d = 0; d < 3; ++d) { prev_klast[d] = (FSAL && !NYS) ? KV_GET(S - 1, d) : 0.0; prev_klast[3 + d] = FSAL ? ka[S - 1][d] : 0.0; }
        const unsigned prev_i = rk_i;
        bool failed = false;
        for (;;) {
            if (n_att > a.n_max) { status = EPH_MAX_ITERATIONS_REACHED; failed = true; break; }
            if (time + next_h > bound) next_h = bound - time;
            const double h = next_h;
            if (time >= bound) { status = EPH_BOUND_REACHED; failed = true; break; }
            if (time + h == time) { status = EPH_STEP_SIZE_UNDERFLOW; failed = true; break; }
#define EPH_RK a.rk
#define EPH_ATTEMPT_PART 1
#include "craft_attempt.inc"
#undef EPH_ATTEMPT_PART
            if (!ok) { status = EPH_EVAL_FAILED; failed = true; break; }
#define EPH_ATTEMPT_PART 2
#include "craft_attempt.inc"
#undef EPH_ATTEMPT_PART
#undef EPH_RK
            time = time + h;
            rk_i += 1;
            n_att += 1;
            // AbsTol::err_over_tol
            const double pm = fmax(fabs(e[0] / a.tol_pos), fmax(fabs(e[1] / a.tol_pos), fabs(e[2] / a.tol_pos)));
            const double vm = fmax(fabs(e[3] / a.tol_vel), fmax(fabs(e[4] / a.tol_vel), fabs(e[5] / a.tol_vel)));
            const double err = fmax(pm, vm);
            // IController::step  mod.rs:225-243
            const double m = a.fac * cr_pow(err, -(1.0 / (double)lower));
            const double c = m < a.fac_min ? a.fac_min : (m > a.fac_max ? a.fac_max : m);
            const double nh = next_h * c;
            next_h = nh > a.h_max ? a.h_max : nh;
            if (err <= 1.0) break;
            time = prev_t;                            // PreviousStep::restore
#pragma unroll
            for (int d = 0; d < 6; ++d) y[d] = a.knot_y[((long long)(nk - 1) * 6 + d) * n + slot];
            rk_i = prev_i;
            if (FSAL) {
#pragma unroll
                for (int d = 0; d < 3; ++d) { if (!NYS) KV_PUT(S - 1, d, prev_klast[d]); ka[S - 1][d] = prev_klast[3 + d]; }
            }
        }
        if (failed) break;
        steps += 1;
        taken += 1;
        // CubicHermiteSplineSolout::solout: push (t, r, v)
        a.knot_t[(long long)nk * n + slot] = time;      // (knot slabs are in LANE order: coalesced whatever the deal)
#pragma unroll
        for (int d = 0; d < 6; ++d) a.knot_y[((long long)nk * 6 + d) * n + slot] = y[d];
        nk += 1;
        last_knot = time;
    }

    a.time[i] = time;
#pragma unroll
    for (int d = 0; d < 6; ++d) a.y[d * n + i] = y[d];
    a.next_h[i] = next_h;
    a.n_attempts[i] = n_att;
    a.rk_i[i] = rk_i;
    a.steps[i] = steps;
    a.cur_seg[i] = cur;
    a.nknots[i] = nk;
    a.last_knot_t[i] = last_knot;
    a.status[i] = status;
    if (FSAL) { EPH_CRAFT_STORE_FSAL(i) }
}

// The sweep, QUEUE form, for batches whose craft need very different numbers of attempts (adaptive step counts differ by
// more than 10x between a low orbit and a heliocentric cruise: bench.py --population mixed, max / mean attempts per wave
// 2.9). A PERSISTENT grid (craft_launch sizes it to what the chip holds) and a work queue: lane L starts with craft L; a
// lane whose craft is finished (reached t_end, took its steps, failed, or filled its knot slab) stores it and takes the
// next unstarted craft from an atomic counter (a.queue), so a wave does not idle 63 lanes waiting for its slowest craft.
// For the same reason the loop is FLAT: one iteration = one ATTEMPT of every active lane; accepting (knot) or rejecting
// (restore) is lane-local bookkeeping after it, so an accepted lane does not sit through its neighbours' retries.
// Measured (MI355X, Verner87, 524 288 mixed craft x 2 d): static 605 ms, queue with whole steps per iteration 492 ms, this
// form 370 ms; on the homogeneous sweep it is 10 % slower than the static kernel (42.6 vs 38.7 ms) -- hence two kernels.
// Craft are independent and every craft's operations are the reference's in the reference's order, so which lane
// integrates a craft, and when, does not touch a bit of its result (tests/test_gpu_craft.py runs both forms).
template <int S, bool FSAL, bool NYS = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_craft_queue(const CraftArgs a) {
    const long long n = a.n_craft;
    long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // queue position = the craft's column in the knot slabs
    const bool first_in_range = col < n;
    long long i = first_in_range && a.perm ? a.perm[col] : col;       // the craft at it
    const int lower = a.rk.order < a.rk.order_embedded ? a.rk.order : a.rk.order_embedded;

    // the craft this lane is integrating (registers); `have` = it still has work in this call
    int status = EPH_OK;
    double time = 0.0, y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, next_h = 0.0, last_knot = 0.0, bound = 0.0;
    unsigned n_att = 0, rk_i = 0, steps = 0, taken = 0;
    int cur = 0, nk = 0;
    const SegmentDev *segs = a.segs;
    SegmentDev sg{};
    EPH_CRAFT_STAGE_STORAGE
    double prev_t = 0.0, prev_klast[6];                 // (PreviousStep's state: the slab's newest knot, as in k_craft_propagate)
    unsigned prev_i = 0;
    bool in_step = false;                               // between a step's prologue and its acceptance
#pragma unroll
    for (int d = 0; d < 6; ++d) prev_klast[d] = 0.0;

    auto load = [&]() -> bool {                         // craft i -> registers; false: nothing to do for it
        status = a.status[i];
        if (status != EPH_OK && status != EPH_KNOTS_FULL && !a.retry) return false;    // a failed craft stays failed until re-armed
        status = EPH_OK;
        time = a.time[i];
#pragma unroll
        for (int d = 0; d < 6; ++d) y[d] = a.y[d * n + i];
        next_h = a.next_h[i];
        n_att = a.n_attempts[i]; rk_i = a.rk_i[i]; steps = a.steps[i];
        cur = a.cur_seg[i]; nk = a.nknots[i];
        last_knot = a.last_knot_t[i];
        segs = a.segs + a.seg_off[i];
        sg = segs[cur];
        bound = sg.end;
        if (FSAL) { EPH_CRAFT_LOAD_FSAL(i) }
        taken = 0;
        in_step = false;
        return true;
    };
    auto store = [&]() {
        a.time[i] = time;
#pragma unroll
        for (int d = 0; d < 6; ++d) a.y[d * n + i] = y[d];
        a.next_h[i] = next_h;
        a.n_attempts[i] = n_att;
        a.rk_i[i] = rk_i;
        a.steps[i] = steps;
        a.cur_seg[i] = cur;
        a.nknots[i] = nk;
        a.last_knot_t[i] = last_knot;
        a.status[i] = status;
        if (FSAL) { EPH_CRAFT_STORE_FSAL(i) }
    };

    bool have = first_in_range && load();
    bool drained = !first_in_range;                     // this lane will get no more craft
    for (;;) {
        // ---- lanes without work take the next craft from the queue
        while (!have && !drained) {
            col = (long long)atomicAdd(a.queue, 1ull);
            if (col >= n) { drained = true; break; }
            i = a.perm ? a.perm[col] : col;
            have = load();
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0) break;          // every lane of the wave is out of work
        if (have) {
            // ---- between steps: is this craft done?  (has_reached: solution.end() >= time; the step budget; the slab)
            if (!in_step) {
                bool done = last_knot >= a.t_end || (a.step_limit && taken >= a.step_limit);
                if (!done && nk >= a.max_knots) { status = EPH_KNOTS_FULL; done = true; }
                if (done) { store(); have = false; continue; }
                // SpacecraftPropagator::step: advance_timeline + reset_integrator  spacecraft.rs:606-609
                if (time >= sg.end) {
                    cur += 1;
                    sg = segs[cur];
                    bound = sg.end;
                    next_h = a.h_init;
                    n_att = 0;
                    rk_i = 0;
                }
                // AdaptiveRungeKuttaIntegrator::advance  mod.rs:414-439: PreviousStep
                prev_t = time;
#pragma unroll
                for (int d = 0; d < 3; ++d) { prev_klast[d] = (FSAL && !NYS) ? KV_GET(S - 1, d) : 0.0; prev_klast[3 + d] = FSAL ? ka[S - 1][d] : 0.0; }
                prev_i = rk_i;
                in_step = true;
            }
            // ---- one attempt. The method table through a pointer the optimiser cannot see through: hoisted out of the
            // persistent loop, the kernel-argument copy (a.rk) pins ~130 coefficients in SGPRs and spills 268 of them to
            // VGPR lanes; re-read per attempt they are scalar-cache hits
            const ErkCoeffs *rkp = a.rkd;
            asm volatile("" : "+s"(rkp));
#define EPH_RK (*rkp)
            bool failed = false;
            if (n_att > a.n_max) { status = EPH_MAX_ITERATIONS_REACHED; failed = true; }
            if (!failed && time + next_h > bound) next_h = bound - time;
            const double h = next_h;
            if (!failed && time >= bound) { status = EPH_BOUND_REACHED; failed = true; }
            if (!failed && time + h == time) { status = EPH_STEP_SIZE_UNDERFLOW; failed = true; }
            if (!failed) {
#define EPH_ATTEMPT_PART 1
#include "craft_attempt.inc"
#undef EPH_ATTEMPT_PART
                if (!ok) { status = EPH_EVAL_FAILED; failed = true; }
            }
            if (failed) { store(); have = false; continue; }   // a StepError ends the craft (state as the reference leaves it)
#define EPH_ATTEMPT_PART 2
#include "craft_attempt.inc"
#undef EPH_ATTEMPT_PART
#undef EPH_RK
            time = time + h;
            rk_i += 1;
            n_att += 1;
            // AbsTol::err_over_tol
            const double pm = fmax(fabs(e[0] / a.tol_pos), fmax(fabs(e[1] / a.tol_pos), fabs(e[2] / a.tol_pos)));
            const double vm = fmax(fabs(e[3] / a.tol_vel), fmax(fabs(e[4] / a.tol_vel), fabs(e[5] / a.tol_vel)));
            const double err = fmax(pm, vm);
            // IController::step  mod.rs:225-243
            const double m = a.fac * cr_pow(err, -(1.0 / (double)lower));
            const double c = m < a.fac_min ? a.fac_min : (m > a.fac_max ? a.fac_max : m);
            const double nh = next_h * c;
            next_h = nh > a.h_max ? a.h_max : nh;
            if (err <= 1.0) {
                // accepted. CubicHermiteSplineSolout::solout: push (t, r, v)
                steps += 1;
                taken += 1;
                a.knot_t[(long long)nk * n + col] = time;
#pragma unroll
                for (int d = 0; d < 6; ++d) a.knot_y[((long long)nk * 6 + d) * n + col] = y[d];
                nk += 1;
                last_knot = time;
                in_step = false;
            } else {
                time = prev_t;                            // PreviousStep::restore
#pragma unroll
                for (int d = 0; d < 6; ++d) y[d] = a.knot_y[((long long)(nk - 1) * 6 + d) * n + col];
                rk_i = prev_i;
                if (FSAL) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { if (!NYS) KV_PUT(S - 1, d, prev_klast[d]); ka[S - 1][d] = prev_klast[3 + d]; }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_craft_wave: ONE WAVE per spacecraft -- the form for few spacecraft (the app's handful of ships), where the
// thread form leaves the chip empty and needs ~0.27 ms per step of a craft (13 stages x 32 body terms in sequence).
// Every lane carries the same craft state; in the right-hand side lane b evaluates body b and lanes 0..2 add the
// terms in body order (craft_rhs<true>). The stage derivatives live in LDS (uniform, read back as broadcasts), so
// the stage loop is a run-time loop around one copy of the right-hand side: the code fits the instruction cache
// (with stages unrolled a single wave spends its time fetching ~100 KB of instructions per attempt: measured
// 50 us per step) and one kernel serves every method. Same operations in the same order as k_craft_propagate.
// ------------------------------------------------------------------------------------------------------
template <bool NYS>
__global__ void __launch_bounds__(64) k_craft_wave(const CraftArgs a) {
    __shared__ __attribute__((aligned(16))) double K[16 * 6];          // k[s][d] (ERK) / dk[s][0..2] (ERKNG)
    __shared__ __attribute__((aligned(16))) double red[3 * kRedRow];
    // ERK pairs (round 6): the stage combinations are LANE-parallel. y_i = y + sum_j k_j (h A_ij) is a chain of additions in j
    // order per component -- the reference's order -- but the products are independent: lane j forms k_j[d] * (h A_sj) for the six
    // components, lanes 0..5 each add one component's products in j order, and the six sums are broadcast. The run-time loop it
    // replaces paid a scalar load of A[s][j] and six LDS reads per (s, j), one after the other, on the single wave this kernel is:
    // a third of a step. The method's tables sit in LDS (Al, Bl, Cl, El), loaded once per launch.
    __shared__ __attribute__((aligned(16))) double Pl[16 * 6], Pe[16 * 6];
    __shared__ __attribute__((aligned(16))) double Al[16 * 16], Bl[16], Cl[16], El[16];
    const long long i = blockIdx.x, n = a.n_craft;
    const int lane = threadIdx.x;
    if (!NYS) {
        const double *ga = &a.rkd->A[0][0];
#pragma unroll
        for (int q = 0; q < 4; ++q) Al[lane * 4 + q] = ga[lane * 4 + q];
        if (lane < 16) { Bl[lane] = a.rkd->B[lane]; Cl[lane] = a.rkd->C[lane]; El[lane] = a.rkd->E[lane]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    int status = a.status[i];
    if (status != EPH_OK && status != EPH_KNOTS_FULL && !a.retry) return;    // a failed craft stays failed until the batch is re-armed
    status = EPH_OK;
    const auto *rc = (const __attribute__((address_space(4))) ErkCoeffs *)(unsigned long long)a.rkd;
    const int S = a.rk.stages;
    const bool FSAL = a.rk.fsal != 0;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto put_k = [&](int s, const double (&v)[6]) {                // lane 0 writes row s
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < 6; ++d) K[s * 6 + d] = v[d];
        }
        wave_sync();
    };

    double time = a.time[i], y[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) y[d] = a.y[d * n + i];
    double next_h = a.next_h[i];
    unsigned n_att = a.n_attempts[i], rk_i = a.rk_i[i], steps = a.steps[i];
    int cur = a.cur_seg[i], nk = a.nknots[i];
    double last_knot = a.last_knot_t[i];
    const SegmentDev *segs = a.segs + a.seg_off[i];
    SegmentDev sg = segs[cur];
    double bound = sg.end;
    if (FSAL) {
        double kl[6], kf[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) { kl[d] = a.klast[d * n + i]; kf[d] = a.kfirst[d * n + i]; }
        put_k(S - 1, kl);
        put_k(0, kf);
    }
    const int lower = a.rk.order < a.rk.order_embedded ? a.rk.order : a.rk.order_embedded;
    LaneBody lb;
    lb.be = a.bodies[lane < a.n_bodies ? lane : 0];
    lb.r = rcp_refined(lb.be.interval);
    lb.b_ok = in_range_div(lb.be.interval);
    lb.idx = -1;
#pragma unroll
    for (int q = 0; q < kDiv * 3; ++q) lb.c[q] = 0.0;

    unsigned taken = 0;
    while (!(last_knot >= a.t_end) && !(a.step_limit && taken >= a.step_limit)) {   // has_reached: solution.end() >= time
        if (nk >= a.max_knots) { status = EPH_KNOTS_FULL; break; }
        if (time >= sg.end) {                         // advance_timeline + reset_integrator  spacecraft.rs:606-609
            cur += 1;
            sg = segs[cur];
            bound = sg.end;
            next_h = a.h_init;
            n_att = 0;
            rk_i = 0;
        }
        // AdaptiveRungeKuttaIntegrator::advance  mod.rs:414-439
        const double prev_t = time;
        double prev_y[6], prev_klast[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) { prev_y[d] = y[d]; prev_klast[d] = FSAL ? K[(S - 1) * 6 + d] : 0.0; }
        const unsigned prev_i = rk_i;
        bool failed = false;
        for (;;) {
            if (n_att > a.n_max) { status = EPH_MAX_ITERATIONS_REACHED; failed = true; break; }
            if (time + next_h > bound) next_h = bound - time;
            const double h = next_h;
            if (time >= bound) { status = EPH_BOUND_REACHED; failed = true; break; }
            if (time + h == time) { status = EPH_STEP_SIZE_UNDERFLOW; failed = true; break; }
            bool ok = true;
            for (int sv = 0; sv < S; ++sv) {          // ERK::advance explicit.rs:72-106 / ERKNG::advance :97-143
                const int s = __builtin_amdgcn_readfirstlane(sv);
                if (FSAL && s == 0 && rk_i > 0) {     // self.k.swap(0, STAGES - 1); continue
                    double k0[6], kl[6];
#pragma unroll
                    for (int d = 0; d < 6; ++d) { k0[d] = K[d]; kl[d] = K[(S - 1) * 6 + d]; }
                    wave_sync();
                    put_k(0, kl);
                    put_k(S - 1, k0);
                    continue;
                }
                if (!ok) continue;
                const double ti = time + h * (NYS ? rc->C[s] : Cl[s]);
                double yi[6], out[6];
                if (NYS) {
                    const double hc = h * rc->C[s];
#pragma unroll
                    for (int d = 0; d < 3; ++d) { yi[d] = y[d] + y[3 + d] * hc; yi[3 + d] = y[3 + d]; }
                    for (int jv = 0; jv < s; ++jv) {
                        const int j = __builtin_amdgcn_readfirstlane(jv);
                        const double hhap = h * h * rc->A[s][j], hav = h * rc->A2[s & 7][j & 7];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const double kj = K[j * 6 + d];
                            yi[d] = yi[d] + kj * hhap;
                            yi[3 + d] = yi[3 + d] + kj * hav;
                        }
                    }
                    ok = craft_rhs<true>(a, sg, ti, yi, out, red, &lb);
                    // (an Err leaves dk[s] as `self.dk[s].zero()` made it: explicit_generalized.rs:109-111)
                    const double dk[6] = {ok ? out[3] : 0.0, ok ? out[4] : 0.0, ok ? out[5] : 0.0, 0.0, 0.0, 0.0};
                    put_k(s, dk);
                    continue;
                }
                {
                    if (lane < s) {                   // lane j: the products k_j[d] * (h A_sj)
                        const double ha = h * Al[s * 16 + lane];
#pragma unroll
                        for (int d = 0; d < 6; ++d) Pl[lane * 6 + d] = K[lane * 6 + d] * ha;
                    }
                    wave_sync();
                    // lane d < 6: y_i[d] = (..((y[d] + p_0[d]) + p_1[d]) + ..) in j order
                    double sum = lane == 0 ? y[0] : lane == 1 ? y[1] : lane == 2 ? y[2] : lane == 3 ? y[3] : lane == 4 ? y[4] : y[5];
                    if (lane < 6) {
                        const double *col = Pl + lane;
#pragma unroll 4
                        for (int j = 0; j < s; ++j) sum = sum + col[j * 6];
                    }
                    wave_sync();
#pragma unroll
                    for (int d = 0; d < 6; ++d) yi[d] = lane_bcast(sum, d);
                }
#if defined(EPH_EXPERIMENTS) && defined(EPH_WAVE_RHS2)
                { double o2[6]; (void)craft_rhs<true>(a, sg, ti + 1e-3, yi, o2, red, &lb); asm volatile("" ::"v"(o2[3]), "v"(o2[4]), "v"(o2[5])); }   // TIMING: one evaluation more per stage
#endif
                ok = craft_rhs<true>(a, sg, ti, yi, out, red, &lb);
                if (!ok) {                            // an Err leaves k[s] as `self.k[s].zero()` made it  explicit.rs:92
#pragma unroll
                    for (int d = 0; d < 6; ++d) out[d] = 0.0;
                }
                put_k(s, out);
            }
            if (!ok) { status = EPH_EVAL_FAILED; failed = true; break; }
            double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (NYS) {
#pragma unroll
                for (int d = 0; d < 3; ++d) y[d] = y[d] + y[3 + d] * h;
                for (int sv = 0; sv < S; ++sv) {
                    const int s = __builtin_amdgcn_readfirstlane(sv);
                    const double hhbp = h * h * rc->B[s], hbv = h * rc->B2[s & 7];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const double ks = K[s * 6 + d];
                        y[d] = y[d] + ks * hhbp;
                        y[3 + d] = y[3 + d] + ks * hbv;
                    }
                }
                for (int sv = 0; sv < S; ++sv) {
                    const int s = __builtin_amdgcn_readfirstlane(sv);
                    const double hhep = h * h * rc->E[s], hev = h * rc->E2[s & 7];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const double ks = K[s * 6 + d];
                        e[d] = e[d] + ks * hhep;
                        e[3 + d] = e[3 + d] + ks * hev;
                    }
                }
            } else {
                // the solution update y += sum_s k_s (h B_s) and RKEmbedded::error e = sum_s k_s (h E_s), lane-parallel like the stage
                // combinations: lane s forms both products, lanes 0..5 add y's components, lanes 8..13 the error's, in s order
                if (lane < S) {
                    const double hb = h * Bl[lane], he = h * El[lane];
#pragma unroll
                    for (int d = 0; d < 6; ++d) {
                        const double ks = K[lane * 6 + d];
                        Pl[lane * 6 + d] = ks * hb;
                        Pe[lane * 6 + d] = ks * he;
                    }
                }
                wave_sync();
                {
                    const int dsel = lane & 7;
                    double sum = lane >= 8 ? 0.0 : dsel == 0 ? y[0] : dsel == 1 ? y[1] : dsel == 2 ? y[2] : dsel == 3 ? y[3] : dsel == 4 ? y[4] : y[5];
                    if (lane < 14 && dsel < 6) {
                        const double *col = (lane >= 8 ? Pe : Pl) + dsel;
#pragma unroll 4
                        for (int sv = 0; sv < S; ++sv) sum = sum + col[sv * 6];
                    }
                    wave_sync();
#pragma unroll
                    for (int d = 0; d < 6; ++d) { y[d] = lane_bcast(sum, d); e[d] = lane_bcast(sum, 8 + d); }
                }
            }
            time = time + h;
            rk_i += 1;
            n_att += 1;
            // AbsTol::err_over_tol
            const double pm = fmax(fabs(e[0] / a.tol_pos), fmax(fabs(e[1] / a.tol_pos), fabs(e[2] / a.tol_pos)));
            const double vm = fmax(fabs(e[3] / a.tol_vel), fmax(fabs(e[4] / a.tol_vel), fabs(e[5] / a.tol_vel)));
            const double err = fmax(pm, vm);
            // IController::step  mod.rs:225-243
#if defined(EPH_EXPERIMENTS) && defined(EPH_WAVE_POW2)
            { const double m2 = cr_pow(err * 1.0000001, -(1.0 / (double)lower)); asm volatile("" ::"v"(m2)); }   // TIMING: one pow more per attempt
#endif
            const double m = a.fac * cr_pow(err, -(1.0 / (double)lower));
            const double c = m < a.fac_min ? a.fac_min : (m > a.fac_max ? a.fac_max : m);
            const double nh = next_h * c;
            next_h = nh > a.h_max ? a.h_max : nh;
            if (err <= 1.0) break;
            time = prev_t;                            // PreviousStep::restore
#pragma unroll
            for (int d = 0; d < 6; ++d) y[d] = prev_y[d];
            rk_i = prev_i;
            if (FSAL) {
                wave_sync();
                put_k(S - 1, prev_klast);
            }
        }
        if (failed) break;
        steps += 1;
        taken += 1;
        if (lane == 0) {                              // CubicHermiteSplineSolout::solout: push (t, r, v)
            a.knot_t[(long long)nk * n + i] = time;
#pragma unroll
            for (int d = 0; d < 6; ++d) a.knot_y[((long long)nk * 6 + d) * n + i] = y[d];
        }
        nk += 1;
        last_knot = time;
    }

    if (lane == 0) {
        a.time[i] = time;
#pragma unroll
        for (int d = 0; d < 6; ++d) a.y[d * n + i] = y[d];
        a.next_h[i] = next_h;
        a.n_attempts[i] = n_att;
        a.rk_i[i] = rk_i;
        a.steps[i] = steps;
        a.cur_seg[i] = cur;
        a.nknots[i] = nk;
        a.last_knot_t[i] = last_knot;
        a.status[i] = status;
        if (FSAL) {
#pragma unroll
            for (int d = 0; d < 6; ++d) { a.klast[d * n + i] = K[(S - 1) * 6 + d]; a.kfirst[d * n + i] = K[d]; }
        }
    }
}


// ---- launch: the form (wave per craft | thread per craft, static or work queue, one or two waves per SIMD) is the host's
// choice (craft.hip craft_launch_plan); this picks the instantiation for the method's stage count
int craft_launch(hipStream_t s, const CraftArgs &a, const CraftLaunch &how) {
    if (how.wave_form) {
        const dim3 grid((unsigned)a.n_craft), block(64);
        if (a.rk.nystrom) hipLaunchKernelGGL(k_craft_wave<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(k_craft_wave<false>, grid, block, 0, s, a);
        return launched("k_craft_wave");
    }
    const long long waves = (a.n_craft + 63) / 64;
    const int S = a.rk.stages;
    const bool F = a.rk.fsal != 0;
    if (how.queue) {
        const dim3 grid((unsigned)how.resident_waves), block(64);
        if (a.rk.nystrom) {
            if (S == 7 && F) hipLaunchKernelGGL((k_craft_queue<7, true, true>), grid, block, 0, s, a);
            else return EPH_ERR_UNSUPPORTED;
        } else if (S == 6 && !F) hipLaunchKernelGGL((k_craft_queue<6, false>), grid, block, 0, s, a);
        else if (S == 7 && F) hipLaunchKernelGGL((k_craft_queue<7, true>), grid, block, 0, s, a);
        else if (S == 7 && !F) hipLaunchKernelGGL((k_craft_queue<7, false>), grid, block, 0, s, a);
        else if (S == 9 && !F) hipLaunchKernelGGL((k_craft_queue<9, false>), grid, block, 0, s, a);
        else if (S == 13 && !F) hipLaunchKernelGGL((k_craft_queue<13, false>), grid, block, 0, s, a);
        else if (S == 16 && !F) hipLaunchKernelGGL((k_craft_queue<16, false>), grid, block, 0, s, a);
        else return EPH_ERR_UNSUPPORTED;
        return launched("k_craft_queue");
    }
    const dim3 grid((unsigned)waves), block(64);
    const bool occ2 = how.occ2;
#define EPH_CRAFT_CASE(S_, F_)                                                                     \
    do {                                                                                           \
        if (occ2) hipLaunchKernelGGL((k_craft_propagate<S_, F_, false, 2>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((k_craft_propagate<S_, F_, false, 1>), grid, block, 0, s, a);      \
    } while (0)
    if (a.rk.nystrom) {
        if (S == 7 && F) hipLaunchKernelGGL((k_craft_propagate<7, true, true, 2>), grid, block, 0, s, a);
        else return EPH_ERR_UNSUPPORTED;
    } else if (S == 6 && !F) EPH_CRAFT_CASE(6, false);
    else if (S == 7 && F) EPH_CRAFT_CASE(7, true);
    else if (S == 7 && !F) EPH_CRAFT_CASE(7, false);
    else if (S == 9 && !F) EPH_CRAFT_CASE(9, false);
    else if (S == 13 && !F) EPH_CRAFT_CASE(13, false);
    else if (S == 16 && !F) EPH_CRAFT_CASE(16, false);
    else return EPH_ERR_UNSUPPORTED;
#undef EPH_CRAFT_CASE
    return launched("k_craft_propagate");
}

}  // namespace EPH_PV_NS
}  // namespace eph
