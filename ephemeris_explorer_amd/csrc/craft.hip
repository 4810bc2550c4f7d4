// craft.hip -- the massless path: a batch of independent spacecraft, one device thread each, propagated with an
// adaptive embedded explicit Runge-Kutta pair against the massive bodies' piecewise-polynomial ephemeris.
//
// Mirrors (paths relative to the reference repository root):
//   SpacecraftPropagator::{new, step, reset_integrator}, SpacecraftModel, Timeline, CubicHermiteSplineSolout
//                                                       ephemeris/src/propagators/spacecraft.rs:58-332,415-695
//   AdaptiveRungeKuttaIntegrator::advance, IController::step, PreviousStep
//                                                       integration/src/runge_kutta/mod.rs:188-285,396-440
//   ERK::{advance, error, undo_step}                    integration/src/runge_kutta/explicit.rs:54-141
//   Bodies::acceleration, GravitationalBody::acceleration_at, TNB, ReferenceFrame, AbsTol
//                                                       ephemeris_explorer/src/dynamics/spacecraft.rs:70-74,218-293,609-641
//   UniformSpline::{position, state_vector}             ephemeris/src/trajectory.rs:459-470,551-617
// glam::DVec3 operations (crate glam 0.30.10, not on disk) are restated from the published crate.
// Same f64 operations in the same order as the CPU path; the one libm call on the path, powf in the step-size
// controller, is evaluated correctly rounded in double-double arithmetic on both sides.
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "craft_device.h"
#include "eph_debug.h"
#include "host.h"

namespace eph {

__global__ void k_debug_pow(long long n, const double *__restrict__ x, double y, double *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cr_pow(x[i], y);
}

// the reciprocal of every body's spline interval, formed once where the sweep kernels would form it (same instructions as
// LaneBody::r): the table entry carries it to spline_locate_fast
__global__ void k_body_reciprocals(int n, BodyEntry *bodies) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const double iv = bodies[b].interval;
    // +0.0 = "take the plain IEEE lookup": the interval outside the guarded range of the shared-reciprocal division, or more than
    // 2^31 - 1 segments (the sweep's speculative lookup converts the segment count in 32 bits)
    bodies[b].rinv = in_range_div(iv) && (unsigned long long)bodies[b].npoly < 0x80000000ull ? rcp_refined(iv) : 0.0;
}

__global__ void k_debug_div(long long n, const double *__restrict__ a, const double *__restrict__ b,
                            double *__restrict__ fast, double *__restrict__ ieee) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = div_shared(a[i], b[i], rcp_refined(b[i]), in_range_div(b[i]));
    ieee[i] = a[i] / b[i];
}
// ------------------------------------------------------------------------------------------------------
// SpacecraftSolout events (the app's solout: ephemeris_explorer/src/dynamics/spacecraft.rs:77-221,296-451,539-586):
// after every accepted step, sphere-of-influence crossings of every body and the apsides relative to the current
// sphere's body are searched on the step's CubicHermite by sign test + bisection (<= 100 halvings, 1e-3 s).
// One thread per craft walks its new segments in order (the transition list is sequential state). Bodies are
// visited in body order (the reference iterates an EntityHashMap, whose order is unspecified).
// ------------------------------------------------------------------------------------------------------
struct EventArgs {
    long long n_craft;
    int n_bodies;
    const BodyEntry *bodies;
    const double *coeffs;
    const int *ncoef;
    const double *soi;            // [n_bodies] sphere radii (inf for the root)
    const int *nknots;
    const double *knot_t, *knot_y;
    int *ev_seg;                  // next segment (knot pair k, k+1) to examine; -1 = new_solution not yet run
    int *ntr, *nap, *ev_status;
    double *tr_time; int *tr_body;                              // [max_tr][n]
    double *ap_time, *ap_dist; int *ap_body, *ap_kind;          // [max_ap][n]
    int max_tr, max_ap;
    const int *slot_of;           // craft -> its column in the knot slabs (null: identity)
};
struct Hermite { double b0; V3 a0, a1, a2, a3; };
__device__ __forceinline__ V3 hermite_pos(const Hermite &h, double t) {      // CubicHermite::eval  trajectory.rs:681-688
    const double dt = t - h.b0;
    return add(scale(add(scale(add(scale(h.a3, dt), h.a2), dt), h.a1), dt), h.a0);
}
__device__ __forceinline__ V3 hermite_vel(const Hermite &h, double t) {      // eval_derivative :690-697
    const double dt = t - h.b0;
    return add(scale(add(scale(scale(h.a3, dt), 3.0), scale(h.a2, 2.0)), dt), h.a1);
}
__device__ __forceinline__ bool ev_body_pos(const EventArgs &a, int b, double t, V3 &out) {
    const BodyEntry be = a.bodies[b];
    long long idx;
    double tau;
    if (!spline_locate(be, t, idx, tau)) return false;
    const double *co = a.coeffs + (be.coeff_off + idx) * kDiv * 3;
    const int nc = a.ncoef[be.coeff_off + idx];
    V3 bp = {0.0, 0.0, 0.0};
    for (int k = nc - 1; k >= 0; --k) {
        bp.x = bp.x * tau + co[k * 3 + 0];
        bp.y = bp.y * tau + co[k * 3 + 1];
        bp.z = bp.z * tau + co[k * 3 + 2];
    }
    out = bp;
    return true;
}
__device__ __forceinline__ bool ev_body_sv(const EventArgs &a, int b, double t, V3 &pos, V3 &vel) {
    const BodyEntry be = a.bodies[b];
    long long idx;
    double tau;
    if (!spline_locate(be, t, idx, tau)) return false;
    const double *co = a.coeffs + (be.coeff_off + idx) * kDiv * 3;
    const int nc = a.ncoef[be.coeff_off + idx];
    double rp[3], rv[3];
    for (int c = 0; c < 3; ++c) {                     // Polynomial::eval_and_deriv
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
    pos = {rp[0], rp[1], rp[2]};
    vel = {rv[0], rv[1], rv[2]};
    return true;
}
// soi_distance_squared_at :77-83 (RADIAL = false) / radial_velocity_at :85-89 (RADIAL = true)
template <bool RADIAL>
__device__ __forceinline__ bool event_f(const EventArgs &a, const Hermite &h, int body, double t, double &out) {
    if (!RADIAL) {
        V3 bp;
        if (!ev_body_pos(a, body, t, bp)) return false;
        const V3 d = sub(hermite_pos(h, t), bp);
        const double r = a.soi[body];
        out = dot(d, d) - r * r;
        return true;
    }
    V3 bp, bv;
    if (!ev_body_sv(a, body, t, bp, bv)) return false;
    const V3 rp = sub(hermite_pos(h, t), bp), rv = sub(hermite_vel(h, t), bv);
    out = dot(rp, rv);
    return true;
}
__device__ __forceinline__ double f64_signum(double x) { return x != x ? x : copysign(1.0, x); }
// find_zero_crossing + find_root_bisection :112-162
template <bool RADIAL>
__device__ bool find_zero_crossing(const EventArgs &a, const Hermite &h, int body, double t0, double t1, double &time,
                                   bool &ascending) {
    double f0, f1;
    if (!event_f<RADIAL>(a, h, body, t0, f0) || !event_f<RADIAL>(a, h, body, t1, f1)) return false;
    if (f64_signum(f0) == f64_signum(f1)) return false;
    double x0 = t0, x1 = t1, g0 = f0;
    for (int it = 0; it < 100; ++it) {
        const double mid = x0 + (x1 - x0) / 2.0;
        double f_mid = 0.0;
        event_f<RADIAL>(a, h, body, mid, f_mid);
        if (f64_signum(g0) != f64_signum(f_mid)) x1 = mid;
        else { x0 = mid; g0 = f_mid; }
        if (fabs(x1 - x0) < 1e-3) {
            time = x0;
            ascending = __builtin_signbit(f0);
            return true;
        }
    }
    return false;
}
// find_soi :172-185,208-221: inside iff d2 < r*r; the closest wins, the first on ties
__device__ int soi_at_except(const EventArgs &a, double t, V3 position, int except) {
    int best = -1;
    double best_d2 = 0.0;
    for (int b = 0; b < a.n_bodies; ++b) {
        if (b == except) continue;
        V3 bp;
        if (!ev_body_pos(a, b, t, bp)) continue;
        const V3 d = sub(position, bp);
        const double d2 = dot(d, d), r = a.soi[b];
        if (!(d2 < r * r)) continue;
        if (best < 0 || d2 < best_d2) { best = b; best_d2 = d2; }
    }
    return best;
}
// SoiTransitions::insert :332-339 on the craft's column of the slab; false = slab full
__device__ bool tr_insert(const EventArgs &a, long long i, int &ntr, double time, int body) {
    const long long n = a.n_craft;
    int lo = 0, hi = ntr;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        const double tm = a.tr_time[(long long)mid * n + i];
        if (tm == time) { a.tr_body[(long long)mid * n + i] = body; return true; }
        if (tm < time) lo = mid + 1; else hi = mid;
    }
    if (lo > 0 && a.tr_body[(long long)(lo - 1) * n + i] == body) return true;
    if (ntr >= a.max_tr) return false;
    for (int k = ntr; k > lo; --k) {
        a.tr_time[(long long)k * n + i] = a.tr_time[(long long)(k - 1) * n + i];
        a.tr_body[(long long)k * n + i] = a.tr_body[(long long)(k - 1) * n + i];
    }
    a.tr_time[(long long)lo * n + i] = time;
    a.tr_body[(long long)lo * n + i] = body;
    ntr += 1;
    return true;
}
__device__ bool ap_insert(const EventArgs &a, long long i, int &nap, double time, double dist, int body, int kind) {
    const long long n = a.n_craft;
    int lo = 0, hi = nap;
    bool found = false;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        const double tm = a.ap_time[(long long)mid * n + i];
        if (tm == time) { lo = mid; found = true; break; }
        if (tm < time) lo = mid + 1; else hi = mid;
    }
    if (!found) {
        if (nap >= a.max_ap) return false;
        for (int k = nap; k > lo; --k) {
            a.ap_time[(long long)k * n + i] = a.ap_time[(long long)(k - 1) * n + i];
            a.ap_dist[(long long)k * n + i] = a.ap_dist[(long long)(k - 1) * n + i];
            a.ap_body[(long long)k * n + i] = a.ap_body[(long long)(k - 1) * n + i];
            a.ap_kind[(long long)k * n + i] = a.ap_kind[(long long)(k - 1) * n + i];
        }
        nap += 1;
    }
    a.ap_time[(long long)lo * n + i] = time;
    a.ap_dist[(long long)lo * n + i] = dist;
    a.ap_body[(long long)lo * n + i] = body;
    a.ap_kind[(long long)lo * n + i] = kind;
    return true;
}
// WAVE = true (few spacecraft): one wave per craft, every lane in the same state; the sign tests of the SOI search
// -- two body evaluations per body and step, almost never followed by a crossing -- run with lane b on body b, and
// only the bodies whose sign changes go through the (wave-uniform) bisection, in body order.
template <bool WAVE>
__global__ void __launch_bounds__(64) k_craft_events(const EventArgs a) {
    const long long i = WAVE ? (long long)blockIdx.x : (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_craft) return;
    const long long n = a.n_craft;
    if (a.ev_status[i] != EPH_OK) return;
    int seg = a.ev_seg[i], ntr = a.ntr[i], nap = a.nap[i];
    const int nk = a.nknots[i];
    bool full = false;
    const long long col = a.slot_of ? a.slot_of[i] : i;
    auto knot = [&](int k, int d) { return a.knot_y[((long long)k * 6 + d) * n + col]; };
    if (seg < 0) {                                    // new_solution :525-537: the sphere the craft starts in
        const int cur = soi_at_except(a, a.knot_t[col], V3{knot(0, 0), knot(0, 1), knot(0, 2)}, -1);
        if (cur >= 0) full = !tr_insert(a, i, ntr, a.knot_t[col], cur);
        seg = 0;
    }
    for (; !full && seg + 1 < nk; ++seg) {            // solout :539-586 for the step that produced knot seg + 1
        // a slab counts as full while fewer than two entries are free at a step boundary, so that draining
        // (eph_craft_batch_reset_events) resumes exactly at a step; a step needing more than that (several
        // crossings at once into a nearly full slab) still reports EVENTS_FULL, from inside the step
        if (ntr + 2 > a.max_tr || nap + 2 > a.max_ap) { full = true; break; }
        const double t0 = a.knot_t[(long long)seg * n + col], t1 = a.knot_t[(long long)(seg + 1) * n + col];
        const V3 p0 = {knot(seg, 0), knot(seg, 1), knot(seg, 2)}, d0 = {knot(seg, 3), knot(seg, 4), knot(seg, 5)};
        const V3 p1 = {knot(seg + 1, 0), knot(seg + 1, 1), knot(seg + 1, 2)};
        const V3 d1 = {knot(seg + 1, 3), knot(seg + 1, 4), knot(seg + 1, 5)};
        Hermite h;                                    // CubicHermite::new  trajectory.rs:645-679
        h.b0 = t0; h.a0 = p0; h.a1 = d0;
        const double dt = t1 - t0;
        if (dt == 0.0 && p0.x == p1.x && p0.y == p1.y && p0.z == p1.z && d0.x == d1.x && d0.y == d1.y && d0.z == d1.z) {
            h.a2 = {0.0, 0.0, 0.0};
            h.a3 = {0.0, 0.0, 0.0};
        } else {
            const double dt_recip = 1.0 / dt;
            const double dt_recip_2 = dt_recip * dt_recip;
            const double dt_recip_3 = dt_recip * dt_recip_2;
            const V3 dt_val = sub(p1, p0);
            h.a2 = sub(scale(scale(dt_val, dt_recip_2), 3.0), scale(add(scale(d0, 2.0), d1), dt_recip));
            h.a3 = add(scale(scale(dt_val, dt_recip_3), -2.0), scale(add(d0, d1), dt_recip_2));
        }
        auto crossing = [&](int b) {                   // one body's find_soi_crossing and its consequence
            double time;
            bool asc;
            if (!find_zero_crossing<false>(a, h, b, t0, t1, time, asc)) return;
            if (!asc) full = !tr_insert(a, i, ntr, time, b);      // Descending: entered b's sphere
            else {
                const int entered = soi_at_except(a, time, hermite_pos(h, time), b);
                if (entered >= 0) full = !tr_insert(a, i, ntr, time, entered);
            }
        };
        if (WAVE) {
            for (int b0 = 0; b0 < a.n_bodies && !full; b0 += kTile) {
                const int b = b0 + (int)threadIdx.x;
                bool cross = false;
                if (b < a.n_bodies) {                  // the sign test of find_zero_crossing, lane b on body b
                    double f0, f1;
                    cross = event_f<false>(a, h, b, t0, f0) && event_f<false>(a, h, b, t1, f1) &&
                            f64_signum(f0) != f64_signum(f1);
                }
                unsigned long long mask = __builtin_amdgcn_ballot_w64(cross);
                while (mask && !full) {                // body order
                    const int lb = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    crossing(b0 + lb);
                }
            }
        } else {
            for (int b = 0; b < a.n_bodies && !full; ++b) crossing(b);
        }
        if (full) break;
        int lo = 0, hi = ntr, i0 = -1;                // transitions.starting_at(t0) :326-329
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            const double tm = a.tr_time[(long long)mid * n + i];
            if (tm == t0) { i0 = mid; break; }
            if (tm < t0) lo = mid + 1; else hi = mid;
        }
        if (i0 < 0) i0 = lo == 0 ? 0 : lo - 1;
        for (int q = i0; q < ntr && !full; ++q) {
            const double t = a.tr_time[(long long)q * n + i];
            const int soi = a.tr_body[(long long)q * n + i];
            const double ta = t0 > t ? t0 : t;
            const double tb = q + 1 < ntr ? a.tr_time[(long long)(q + 1) * n + i] : t1;
            double time;
            bool asc;
            if (!find_zero_crossing<true>(a, h, soi, ta, tb, time, asc)) continue;
            V3 bp;
            if (!ev_body_pos(a, soi, time, bp)) continue;
            const V3 d = sub(bp, hermite_pos(h, time));          // distance_at  dynamics/mod.rs:141-146
            full = !ap_insert(a, i, nap, time, sqrt(dot(d, d)), soi, asc ? 0 : 1);
        }
        if (full) break;
    }
    a.ev_seg[i] = seg;
    a.ntr[i] = ntr;
    a.nap[i] = nap;
    if (full) a.ev_status[i] = EPH_EVENTS_FULL;
}

// eph_craft_batch_reset_events: keeps the newest transition (the sphere the craft is in -- what
// SoiTransitions::starting_at needs for the next step), drops the older ones and all apsides, clears EVENTS_FULL
__global__ void __launch_bounds__(256) k_craft_reset_events(long long n, int *ntr, int *nap, int *ev_status,
                                                            double *tr_time, int *tr_body) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = ntr[i];
    if (k > 1) {
        tr_time[i] = tr_time[(long long)(k - 1) * n + i];
        tr_body[i] = tr_body[(long long)(k - 1) * n + i];
        ntr[i] = 1;
    }
    nap[i] = 0;
    if (ev_status[i] == EPH_EVENTS_FULL) ev_status[i] = EPH_OK;
}

// eph_craft_batch_reset_knots: the newest knot of every craft becomes knot 0 of an otherwise empty slab (the next
// CubicHermiteSpline piece starts where the drained one ended), a KNOTS_FULL status is cleared, and the event
// search's segment cursor moves with the knots.
__global__ void __launch_bounds__(256) k_craft_reset_knots(long long n, int *nknots, int *status, double *knot_t,
                                                           double *knot_y, int *ev_seg, const int *slot_of) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long col = slot_of ? slot_of[i] : i;
    const int nk = nknots[i];
    if (nk > 1) {
        knot_t[col] = knot_t[(long long)(nk - 1) * n + col];
#pragma unroll
        for (int d = 0; d < 6; ++d) knot_y[(long long)d * n + col] = knot_y[((long long)(nk - 1) * 6 + d) * n + col];
        nknots[i] = 1;
        if (ev_seg && ev_seg[i] >= 0) ev_seg[i] = max(ev_seg[i] - (nk - 1), 0);
    }
    if (status[i] == EPH_KNOTS_FULL) status[i] = EPH_OK;
}

// The debug window's interpolation-error scan (ephemeris_explorer/src/ui/windows/debug.rs:182-238): re-integrate
// the massive bodies and, after every step, compare each body's position with its UniformSpline at that epoch;
// keep the maximum of `position.distance(traj_position) * 1e3` (metres) per body. Thread per body; err[b] < 0 marks
// "no entry yet" (EntityHashMap::entry(..).or_insert).
__global__ void __launch_bounds__(256) k_interp_error(int n, int npad, const double *__restrict__ Y, double t,
                                                      const BodyEntry *__restrict__ bodies,
                                                      const double *__restrict__ coeffs, const int *__restrict__ ncoef,
                                                      double *err, int *failed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const BodyEntry be = bodies[b];
    long long idx;
    double tau;
    if (!spline_locate(be, t, idx, tau)) { *failed = 1; return; }   // traj.position(epoch).unwrap()
    const double *co = coeffs + (be.coeff_off + idx) * kDiv * 3;
    const int nc = ncoef[be.coeff_off + idx];
    V3 tp = {0.0, 0.0, 0.0};
    for (int k = nc - 1; k >= 0; --k) {
        tp.x = tp.x * tau + co[k * 3 + 0];
        tp.y = tp.y * tau + co[k * 3 + 1];
        tp.z = tp.z * tau + co[k * 3 + 2];
    }
    const V3 d = sub(V3{Y[b], Y[npad + b], Y[2 * npad + b]}, tp);
    const double e = sqrt(dot(d, d)) * 1e3;
    const double cur = err[b];
    err[b] = cur < 0.0 ? e : fmax(cur, e);
}

// CubicHermiteSpline::state_vector  trajectory.rs:766-797, CubicHermite::{new, eval, eval_derivative} :645-696
__global__ void __launch_bounds__(256) k_hermite_eval(long long nk, const double *__restrict__ t,
                                                      const double *__restrict__ pos, const double *__restrict__ vel,
                                                      long long m, const double *__restrict__ at,
                                                      double *__restrict__ op, double *__restrict__ ov,
                                                      uint8_t *__restrict__ inside) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const double x = at[q];
    long long lo = 0, hi = nk;
    long long hit = -1;
    while (lo < hi) {                                 // binary_search_by(|(t, _)| t.cmp(&at))
        const long long mid = lo + (hi - lo) / 2;
        const double tm = t[mid];
        if (tm == x) { hit = mid; break; }
        if (tm < x) lo = mid + 1; else hi = mid;
    }
    if (hit >= 0) {
        for (int c = 0; c < 3; ++c) { op[q * 3 + c] = pos[hit * 3 + c]; if (ov) ov[q * 3 + c] = vel[hit * 3 + c]; }
        inside[q] = 1;
        return;
    }
    if (lo == 0 || lo >= nk) {                        // i.checked_sub(1)? / self.0.get(i + 1)?
        for (int c = 0; c < 3; ++c) { op[q * 3 + c] = 0.0; if (ov) ov[q * 3 + c] = 0.0; }
        inside[q] = 0;
        return;
    }
    const long long i = lo - 1;
    const double b0 = t[i], dt = t[i + 1] - b0;
    const double dt_recip = 1.0 / dt;
    const double dt_recip_2 = dt_recip * dt_recip;
    const double dt_recip_3 = dt_recip * dt_recip_2;
    const double s = x - b0;
    for (int c = 0; c < 3; ++c) {
        const double v0 = pos[i * 3 + c], v1 = pos[(i + 1) * 3 + c], d0 = vel[i * 3 + c], d1 = vel[(i + 1) * 3 + c];
        const double dt_val = v1 - v0;
        const double a2 = dt_val * dt_recip_2 * 3.0 - (d0 * 2.0 + d1) * dt_recip;
        const double a3 = dt_val * dt_recip_3 * -2.0 + (d0 + d1) * dt_recip_2;
        op[q * 3 + c] = (((a3 * s + a2) * s) + d0) * s + v0;
        if (ov) ov[q * 3 + c] = ((a3 * s * 3.0 + a2 * 2.0) * s) + d0;
    }
    inside[q] = 1;
}

// ------------------------------------------------------------------------------------------------------
// Adaptive plot sampling: compute_plot_points_parallel + PlotPoints::new + angular_distance
// (ephemeris_explorer/src/ui/world/plot.rs:93-149,272-374,429-436), one thread per plotted trajectory.
// ------------------------------------------------------------------------------------------------------
struct PlotArgs {
    long long n_plots;
    int n_bodies;
    const BodyEntry *bodies;
    const double *coeffs;
    const int *ncoef;
    const eph_plot_request *req;
    eph_plot_view view;
    const double *knot_t, *knot_pos, *knot_vel;
    long long capacity;
    double *out_t;
    float *out_xyz;
    long long *out_count;
    int *out_status;
    double *out_failed_at;
};
// UniformSpline::state_vector / position of body b  (trajectory.rs:449-470)
__device__ bool plot_body_sv(const PlotArgs &a, int b, double t, V3 &pos, V3 &vel) {
    const BodyEntry be = a.bodies[b];
    long long idx;
    double tau;
    if (!spline_locate(be, t, idx, tau)) return false;
    const double *co = a.coeffs + (be.coeff_off + idx) * kDiv * 3;
    const int nc = a.ncoef[be.coeff_off + idx];
    double rp[3], rv[3];
    for (int c = 0; c < 3; ++c) {                     // Polynomial::eval_and_deriv
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
    pos = {rp[0], rp[1], rp[2]};
    vel = {rv[0], rv[1], rv[2]};
    return true;
}
__device__ bool plot_body_pos(const PlotArgs &a, int b, double t, V3 &out) {
    const BodyEntry be = a.bodies[b];
    long long idx;
    double tau;
    if (!spline_locate(be, t, idx, tau)) return false;
    const double *co = a.coeffs + (be.coeff_off + idx) * kDiv * 3;
    const int nc = a.ncoef[be.coeff_off + idx];
    V3 bp = {0.0, 0.0, 0.0};
    for (int k = nc - 1; k >= 0; --k) {               // Polynomial::eval (Horner)
        bp.x = bp.x * tau + co[k * 3 + 0];
        bp.y = bp.y * tau + co[k * 3 + 1];
        bp.z = bp.z * tau + co[k * 3 + 2];
    }
    out = bp;
    return true;
}
// CubicHermiteSpline::state_vector (trajectory.rs:766-797) on knots [0, nk) of t / pos / vel
__device__ bool plot_hermite_sv(long long nk, const double *t, const double *pos, const double *vel, double x, V3 &p, V3 &v) {
    long long lo = 0, hi = nk, hit = -1;
    while (lo < hi) {                                 // binary_search_by(|(t, _)| t.cmp(&at))
        const long long mid = lo + (hi - lo) / 2;
        const double tm = t[mid];
        if (tm == x) { hit = mid; break; }
        if (tm < x) lo = mid + 1; else hi = mid;
    }
    if (hit >= 0) {
        p = {pos[hit * 3], pos[hit * 3 + 1], pos[hit * 3 + 2]};
        v = {vel[hit * 3], vel[hit * 3 + 1], vel[hit * 3 + 2]};
        return true;
    }
    if (lo == 0 || lo >= nk) return false;            // i.checked_sub(1)? / self.0.get(i + 1)?
    const long long i = lo - 1;
    const double b0 = t[i], dt = t[i + 1] - b0;
    const double dt_recip = 1.0 / dt;
    const double dt_recip_2 = dt_recip * dt_recip;
    const double dt_recip_3 = dt_recip * dt_recip_2;
    const double s = x - b0;
    double op[3], ov[3];
    for (int c = 0; c < 3; ++c) {
        const double v0 = pos[i * 3 + c], v1 = pos[(i + 1) * 3 + c], d0 = vel[i * 3 + c], d1 = vel[(i + 1) * 3 + c];
        const double dt_val = v1 - v0;
        const double a2 = dt_val * dt_recip_2 * 3.0 - (d0 * 2.0 + d1) * dt_recip;
        const double a3 = dt_val * dt_recip_3 * -2.0 + (d0 + d1) * dt_recip_2;
        op[c] = (((a3 * s + a2) * s) + d0) * s + v0;
        ov[c] = ((a3 * s * 3.0 + a2 * 2.0) * s) + d0;
    }
    p = {op[0], op[1], op[2]};
    v = {ov[0], ov[1], ov[2]};
    return true;
}
// glam DMat3::mul_vec3: ((x_axis * v.x) + (y_axis * v.y)) + (z_axis * v.z)   (glam 0.30.10)
__device__ __forceinline__ V3 mat3_mul(const double (&m)[9], V3 v) {
    const V3 x = {m[0], m[1], m[2]}, y = {m[3], m[4], m[5]}, z = {m[6], m[7], m[8]};
    return add(add(scale(x, v.x), scale(y, v.y)), scale(z, v.z));
}
// angular_distance  plot.rs:429-436: DVec3::normalize = self * self.length().recip()
__device__ __forceinline__ double plot_angular_distance(V3 cam, V3 p1, V3 p2) {
    const V3 d1 = sub(p1, cam), d2 = sub(p2, cam);
    const V3 v1 = scale(d1, length_recip(d1)), v2 = scale(d2, length_recip(d2));
    const V3 w = cross(v1, v2);
    const double d = dot(v1, v2);
    return dot(w, w) / (d * d);
}
__device__ __forceinline__ double ord_clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

__global__ void __launch_bounds__(64) k_plot_points(const PlotArgs a) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_plots) return;
    const eph_plot_request rq = a.req[p];
    a.out_count[p] = 0;
    a.out_status[p] = EPH_OK;
    a.out_failed_at[p] = 0.0;
    const bool src_body = rq.source_body >= 0;
    const double *kt = a.knot_t + rq.knot_first, *kp = a.knot_pos + 3 * rq.knot_first, *kv = a.knot_vel + 3 * rq.knot_first;
    const long long nk = rq.knot_count;
    // RelativeTrajectory bounds / segment count  trajectory.rs:277-308
    double start, end;
    long long segs;
    if (src_body) {
        const BodyEntry be = a.bodies[rq.source_body];
        start = be.start; end = be.start + be.span; segs = be.npoly;
    } else {
        start = nk > 0 ? kt[0] : -1.7976931348623157e308;          // Epoch::MIN / MAX of an empty spline :756-763
        end = nk > 0 ? kt[nk - 1] : 1.7976931348623157e308;
        segs = nk > 0 ? nk - 1 : 0;
    }
    double rstart = 0.0, rend = 0.0;
    if (rq.reference_body >= 0) {
        const BodyEntry rb = a.bodies[rq.reference_body];
        rstart = rb.start; rend = rb.start + rb.span;
        start = rstart < start ? start : rstart;                    // Ord::max / Ord::min
        end = rend < end ? rend : end;
        segs = rb.npoly < segs ? rb.npoly : segs;
    }
    if (!rq.enabled || segs == 0 || start > end) return;            // plot.enabled && !relative.is_empty()  :324
    const double current = a.view.current;
    const double current_clamped = ord_clamp(current, start, end);
    double tmin = ord_clamp(rq.start, start, end), tmax = ord_clamp(rq.end, start, end);
    if (rq.bound == 1) tmin = current_clamped < tmin ? tmin : current_clamped;      // min.max(current_clamped)
    else if (rq.bound == 2) tmax = current_clamped < tmax ? current_clamped : tmax; // max.min(current_clamped)
    if (tmin >= tmax) return;
    // translation: reference.position(current.clamp(r.start(), r.end())).unwrap()  :355-361
    V3 tr = {0.0, 0.0, 0.0};
    if (rq.reference_body >= 0) {
        const double tc = ord_clamp(current, rstart, rend);
        if (!plot_body_pos(a, rq.reference_body, tc, tr)) { a.out_status[p] = EPH_EVAL_FAILED; a.out_failed_at[p] = tc; return; }
    }
    const V3 cam = {a.view.camera_position[0], a.view.camera_position[1], a.view.camera_position[2]};
    const V3 cell = {a.view.cell_offset[0], a.view.cell_offset[1], a.view.cell_offset[2]};
    const V3 gt = {a.view.grid_translation[0], a.view.grid_translation[1], a.view.grid_translation[2]};
    // |t| Some(root.to_global_sv(relative.state_vector(t)? + translation))
    auto eval = [&](double t, V3 &gp, V3 &gv) -> bool {
        V3 rp = {0.0, 0.0, 0.0}, rv = {0.0, 0.0, 0.0};              // reference first (trajectory.rs:329-333)
        if (rq.reference_body >= 0 && !plot_body_sv(a, rq.reference_body, t, rp, rv)) return false;
        V3 sp, sv;
        if (src_body ? !plot_body_sv(a, rq.source_body, t, sp, sv) : !plot_hermite_sv(nk, kt, kp, kv, t, sp, sv)) return false;
        const V3 pos = add(sub(sp, rp), tr);
        const V3 vel = add(sub(sv, rv), V3{0.0, 0.0, 0.0});         // + StateVector::from_position(..).velocity
        gp = add(mat3_mul(a.view.grid_matrix3, sub(pos, cell)), gt);   // transform_point3(point - cell_to_float)
        gv = mat3_mul(a.view.grid_matrix3, vel);                    // transform_vector3
        return true;
    };
    if (rq.max_points == 0) return;                                 // :101-103
    const double target = rq.tan2_angular_resolution * rq.tan2_angular_resolution;
    double previous_time = tmin;
    V3 ppos, pvel;
    if (!eval(previous_time, ppos, pvel)) { a.out_status[p] = EPH_EVAL_FAILED; a.out_failed_at[p] = previous_time; return; }
    double delta = tmax - previous_time;
    bool have_est = false;
    double estimated = 0.0;
    double *ot = a.out_t + p * a.capacity;
    float *ox = a.out_xyz + p * a.capacity * 3;
    long long np = 0;
    auto push = [&](double t, V3 q) { ot[np] = t; ox[3 * np] = (float)q.x; ox[3 * np + 1] = (float)q.y; ox[3 * np + 2] = (float)q.z; ++np; };
    push(previous_time, ppos);
    while (previous_time < tmax && np < rq.max_points) {
        double t, next_error;
        V3 cpos, cvel;
        for (unsigned trial = 0;; ++trial) {
            if (have_est && estimated > 0.0) delta = delta * 0.9 * sqrt(sqrt(target / estimated));
            t = previous_time + delta;
            if (t > tmax) t = tmax;
            delta = t - previous_time;
            const V3 extrapolated = add(ppos, scale(pvel, delta));
            if (!eval(t, cpos, cvel)) { a.out_count[p] = np; a.out_status[p] = EPH_EVAL_FAILED; a.out_failed_at[p] = t; return; }
            const double error = plot_angular_distance(cam, extrapolated, cpos) / 16.0;
            if (error <= target) { next_error = error; break; }
            have_est = true;
            estimated = error;
            if (trial >= (1u << 20)) { a.out_count[p] = np; a.out_status[p] = EPH_MAX_ITERATIONS_REACHED; a.out_failed_at[p] = t; return; }
        }
        previous_time = t;
        ppos = cpos;
        pvel = cvel;
        have_est = true;
        estimated = next_error;
        push(t, ppos);
    }
    a.out_count[p] = np;
}

// 16 bytes per lane, consecutive lanes consecutive: between device memory and the pinned, device-mapped staging buffer (mem.cpp), either way
__global__ void __launch_bounds__(256) k_copy16(long long n16, const double2 *__restrict__ src, double2 *__restrict__ dst) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
// few spacecraft: one wave each (k_craft_wave, k_craft_events<true>); many: one thread each. Measured crossover on
// MI355X, Verner87, 32 bodies: see scripts/bench_craft_small.py and profiles/README.md
static bool craft_wave_form(long long n_craft) {
    static const int form = [] {
        const char *e = getenv("EPH_CRAFT_FORM");      // "wave" | "thread" (tuning / tests)
        return !e ? 0 : (e[0] == 'w' ? 1 : 2);
    }();
    return form ? form == 1 : n_craft <= kCraftWaveMax;
}
// Which sweep kernel (craft_sweep.hip holds them, once per evaluation order of the point-mass term). The queue form pays when
// craft need very different numbers of attempts AND there are more craft than the chip holds at once (two waves per SIMD), so that
// a finished lane has something to take; a batch whose craft were dealt to the lanes by orbital time scale (a.perm: every
// thread-per-craft batch by default) has waves of similar craft: the static form. EPH_CRAFT_QUEUE=0|1, EPH_CRAFT_OCC=1|2 override
// (tuning, tests).
static int craft_launch(int pv, hipStream_t s, const CraftArgs &a, bool heterogeneous) {
    CraftLaunch how{};
    how.wave_form = craft_wave_form(a.n_craft);
    if (!how.wave_form) {
        static const long long simds = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return (long long)std::max(cus, 1) * 4;
        }();
        const long long waves = (a.n_craft + 63) / 64;
        static const int forced_q = [] { const char *e = getenv("EPH_CRAFT_QUEUE"); return !e || !*e ? -1 : (e[0] == '0' ? 0 : 1); }();
        how.queue = forced_q >= 0 ? forced_q == 1 : (heterogeneous && !a.perm && waves > 2 * simds);
        if (how.queue) {
            how.resident_waves = std::min(waves, 2 * simds);
            // the queue starts behind the craft the grid's own lanes begin with
            if (hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), s) != hipSuccess ||
                hipMemsetD32Async((hipDeviceptr_t)a.queue, (int)(how.resident_waves * 64), 1, s) != hipSuccess) {
                set_last_error("craft queue reset", hipGetLastError());
                return EPH_ERR_HIP;
            }
        }
        // more waves of craft than SIMDs (256 CUs x 4): hold the kernel to two waves per SIMD
        static const int forced = [] { const char *e = getenv("EPH_CRAFT_OCC"); return e ? atoi(e) : 0; }();
        how.occ2 = forced ? forced == 2 : waves > simds;
    }
    return launch_craft(pv, s, a, how);
}

}  // namespace eph

using namespace eph;

// The device-resident Vec<UniformSpline> of the massive bodies. LIVE, like the reference's: GravitationalBody.trajectory is
// Trajectory(Arc<RwLock<PredictionTrajectory>>) (ephemeris_explorer/src/dynamics/spacecraft.rs:52-74, dynamics/mod.rs:84-85), merged
// N-body snapshots grow it (dynamics/celestial.rs:198-204,220-226 -> UniformSpline::append / prepend / clear_*,
// ephemeris/src/trajectory.rs:515-549) and every spacecraft propagator holding the context sees the new extent at its next
// evaluation. Here: `splines` is the authoritative host copy (the reference's own operations, host.h), the device table follows it
// incrementally -- body b owns rows [base[b], base[b] + cap[b]) of `coeffs` / `ncoef` with its polynomials at coeff_off .. +npoly, so
// an append uploads the new rows only and a clear moves two integers; a region that overflows re-lays the table with headroom
// proportional to its size (amortised O(1) per polynomial). `mu` is the RwLock: sweeps, plots and scans hold it shared for the
// whole (synchronous) call, append / clear exclusively -- a writer never changes rows a kernel is reading.
struct eph_ephemeris {
    int device = 0;
    int n_bodies = 0;
    mutable std::shared_mutex mu;
    uint64_t revision = 0;                     // bumped by every append / clear
    std::vector<UniformSpline> splines;
    std::vector<double> gm;
    DevBuf<BodyEntry> bodies;
    DevBuf<double> coeffs;
    DevBuf<int> ncoef;
    std::vector<BodyEntry> host_bodies;        // what `bodies` holds (rinv: filled on the device only)
    std::vector<long long> base, cap;          // body b's region of rows
    std::vector<char> grows_front;             // body b has been prepended to: keep headroom in front as well
};

struct eph_craft_batch {
    int pv = 0;                               // evaluation order of the point-mass term this batch was created under
    const eph_ephemeris *eph = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    long long n = 0;
    int max_knots = 0;
    ErkCoeffs rk{};
    eph_adaptive_params params{};
    DevBuf<double> time, y, next_h, klast, kfirst, last_knot, knot_t, knot_y;   // klast / kfirst: the FSAL pairs' k[S-1] / k[0] between calls
    DevBuf<unsigned> n_attempts, rk_i, steps;
    DevBuf<int> cur_seg, status, nknots;
    DevBuf<long long> seg_off;
    DevBuf<SegmentDev> segs;
    DevBuf<ErkCoeffs> rk_dev;
    DevBuf<eph_craft_record> summary;         // eph_craft_batch_summary's device-side records (a clone's: on first use)
    DevBuf<unsigned long long> queue;         // k_craft_queue's work queue (one counter)
    bool heterogeneous = false;               // the craft's dynamical time scales differ widely (craft_time_scales): queue form
    DevBuf<BodyEntry> bodies_ordered;         // eph_craft_batch_set_body_order: the ephemeris's table permuted (empty: table order),
    DevBuf<int> body_order_dev;               //   re-gathered from the live table before every sweep
    std::vector<int32_t> body_order;
    bool retry = false;                       // eph_craft_batch_retry_failed: the next sweep steps the craft whose last step failed
    DevBuf<int> perm, slot_of;                // heterogeneous batches: lane / queue position -> craft by dynamical time, and back
    std::vector<int> h_slot;                  //   (craft_sort); the knot slabs' columns are lane positions
    // SpacecraftSolout events (optional)
    bool events = false;
    int max_tr = 0, max_ap = 0;
    DevBuf<double> soi, tr_time, ap_time, ap_dist;
    DevBuf<int> ev_seg, ntr, nap, ev_status, tr_body, ap_body, ap_kind;
    double kernel_ms = 0;
    ~eph_craft_batch() {
        if (stream) {
            (void)hipSetDevice(device);
            (void)hipStreamSynchronize(stream);
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
            (void)hipStreamDestroy(stream);
        }
    }
};

// Scheduling estimate (never part of a result): do the 64 craft that would share a WAVE need very different numbers of
// steps? The step size of an embedded pair follows the local dynamical time sqrt(d^3 / mu) of the nearest massive body,
// so eight waves scattered over the batch (64 consecutive craft each) get tau_i = min over bodies of
// sqrt(|r_i - r_b(t0_i)|^3 / mu_b) from the host copy of the ephemeris (plain Horner; approximate is fine), and the batch
// counts as heterogeneous when inside any of them the largest and smallest tau differ by more than 4x: a low orbit 860 s,
// a heliocentric cruise 5e6 s. Families in contiguous blocks do NOT count (measured: the static kernel is then the
// faster one, 183 against 213 ms -- the hardware's wave dispatch already is a queue of whole waves); craft_launch uses
// the answer to pick k_craft_queue over the static kernel.
// A work estimate per craft for the deal of craft to lanes: the time scale of its ORBIT about its dominant body (the body with
// the smallest local dynamical time sqrt(d^3 / mu)) -- sqrt(a^3 / mu) with the semi-major axis a from the vis-viva energy of the
// relative state when the orbit is bound, 16 x the local value when it is not (a fly-by leaves the body quickly). The LOCAL time
// alone mixes families exactly where the work is: a transfer orbit at perigee and a departing lunar transfer look like a low
// circular orbit (measured: dealing by it, 452 ms against the queue kernel's 347 on the mixed population).
__global__ void __launch_bounds__(256) k_craft_tau(long long n, int n_bodies, const BodyEntry *__restrict__ bodies,
                                                   const double *__restrict__ coeffs, const double *__restrict__ time,
                                                   const double *__restrict__ y, float *__restrict__ tau) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t = time[i], px = y[i], py = y[n + i], pz = y[2 * n + i];
    const double vx = y[3 * n + i], vy = y[4 * n + i], vz = y[5 * n + i];
    double best = INFINITY, orbit = INFINITY;
    for (int b = 0; b < n_bodies; ++b) {
        const BodyEntry be = bodies[b];
        if (!(be.mu > 0.0) || be.npoly <= 0) continue;
        long long idx;
        double tq;
        if (!spline_locate(be, t, idx, tq)) continue;
        const double *c = coeffs + (be.coeff_off + idx) * kDiv * 3;
        double bp[3] = {0.0, 0.0, 0.0}, bd[3] = {0.0, 0.0, 0.0};
        for (int k = kDiv - 1; k >= 0; --k)
            for (int d = 0; d < 3; ++d) {
                bd[d] = bd[d] * tq + bp[d];               // derivative with respect to tau, then / interval
                bp[d] = bp[d] * tq + c[k * 3 + d];
            }
        const double dx = px - bp[0], dy = py - bp[1], dz = pz - bp[2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        const double local = sqrt(d2 * sqrt(d2) / be.mu);
        if (local < best) {
            best = local;
            const double ux = vx - bd[0] / be.interval, uy = vy - bd[1] / be.interval, uz = vz - bd[2] / be.interval;
            const double energy = 0.5 * (ux * ux + uy * uy + uz * uz) - be.mu / sqrt(d2);
            if (energy < 0.0) {
                const double sma = -be.mu / (2.0 * energy);
                orbit = sqrt(sma * sma * sma / be.mu);
            } else {
                orbit = 16.0 * local;
            }
        }
    }
    tau[i] = (float)orbit;
}
__global__ void __launch_bounds__(256) k_knot0_to_lanes(long long n, const int *__restrict__ perm, const double *__restrict__ time,
                                                        const double *__restrict__ y, double *__restrict__ knot_t,
                                                        double *__restrict__ knot_y) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const long long i = perm[q];
    knot_t[q] = time[i];
    for (int d = 0; d < 6; ++d) knot_y[(long long)d * n + q] = y[(long long)d * n + i];
}
// rows of a [rows][n] slab from lane order back to craft order (eph_craft_batch_knot_slabs of a sorted batch)
__global__ void __launch_bounds__(256) k_rows_to_craft_order(long long rows, long long n, const int *__restrict__ slot_of,
                                                             const double *__restrict__ src, double *__restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long col = slot_of[i];
    for (long long r = 0; r < rows; ++r) dst[r * n + i] = src[r * n + col];
}
// one 80-byte record per craft from the SoA state arrays (coalesced reads, one record per thread written as ten 8-byte words)
__global__ void __launch_bounds__(256) k_craft_summary(long long n, const double *__restrict__ time, const double *__restrict__ y,
                                                       const double *__restrict__ next_h, const int *__restrict__ status,
                                                       const int *__restrict__ nknots, const unsigned *__restrict__ attempts,
                                                       const unsigned *__restrict__ steps, eph_craft_record *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    eph_craft_record r;
    r.t = time[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) { r.pos[d] = y[d * n + i]; r.vel[d] = y[(3 + d) * n + i]; }
    r.next_h = next_h[i];
    r.status = status[i];
    r.nknots = nknots[i];
    r.attempts = attempts[i];
    r.steps = steps[i];
    out[i] = r;
}
// the ephemeris's table entries in the order Bodies::acceleration visits them (eph_craft_batch_set_body_order), gathered from the LIVE
// table before every sweep: the entries change when the ephemeris grows
__global__ void __launch_bounds__(64) k_permute_bodies(int n, const int *__restrict__ order, const BodyEntry *__restrict__ table,
                                                       BodyEntry *__restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = table[order[q]];
}
// Which craft a lane integrates is free (craft are independent, every craft's operations are the reference's whoever runs them),
// so every thread-per-craft batch is dealt to the lanes in the order of k_craft_tau's estimate -- a stable radix sort on the host,
// shortest orbit first (most steps first). The lanes of a wave then carry craft of similar step counts: on the mixed population
// 185 against 343 ms with the queue kernel in craft order (profiles/r03_craft_queue.md); on the north star's own sweep (one
// transfer arc +- 100 km) the accepted steps of a craft follow its orbital energy with correlation -0.99, and the deal takes the
// max / mean steps per wave from 1.09 to 1.01 (oracle, 1500 craft). The knot slabs keep LANE columns (coalesced knot writes
// whatever the deal); every reader translates through the inverse. EPH_CRAFT_SORT=0 switches it off (tuning, tests).
// EPH_CRAFT_SORT: 0 = never deal | 1 = heterogeneous batches only (round 3's default) | unset / 2 = every thread-per-craft batch
static int craft_sort_mode() {
    static const int mode = [] { const char *e = getenv("EPH_CRAFT_SORT"); return !e || !*e ? 2 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2)); }();
    return mode;
}
static int craft_sort(eph_craft_batch *b) {
    const long long n = b->n;
    const int mode = craft_sort_mode();
    if (mode == 0 || (mode == 1 && !b->heterogeneous) || n < 128 || n > 0x7fffffffLL) return EPH_OK;
    // Transfers through the process's pinned staging buffer (mem.cpp), moved by kernels: no pinning and unpinning of three
    // short-lived host vectors per creation.
    const size_t n4 = ((size_t)n + 3) & ~(size_t)3;    // 16-byte granules for k_copy16
    DevBuf<float> tau;
    int st = tau.alloc(n4);
    if (st) return st;
    hipLaunchKernelGGL(k_craft_tau, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, n, b->eph->n_bodies, b->eph->bodies.p,
                       b->eph->coeffs.p, b->time.p, b->y.p, tau.p);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { set_last_error("k_craft_tau", he); return EPH_ERR_HIP; }
    if ((st = b->perm.alloc(n4)) || (st = b->slot_of.alloc(n4))) return st;
    PinnedStage stage(2 * n4 * sizeof(int));
    if (stage.status()) return stage.status();
    StreamIdleOnExit idle(b->stream);
    const unsigned cgrid = (unsigned)std::min<size_t>((n4 / 4 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_copy16, dim3(cgrid), dim3(256), 0, b->stream, (long long)(n4 / 4), (const double2 *)tau.p, (double2 *)stage.dev());
    EPH_HIP(hipStreamSynchronize(b->stream));
    const float *h = static_cast<const float *>(stage.host());
    // stable LSD radix sort of the estimates (positive binary32 values order like their bit patterns; anything else goes last)
    std::vector<uint32_t> key((size_t)n);
    for (long long i = 0; i < n; ++i) {
        const float v = h[(size_t)i];
        uint32_t u;
        std::memcpy(&u, &v, sizeof(u));
        key[(size_t)i] = v > 0.0f && std::isfinite(v) ? u : 0xffffffffu;
    }
    std::vector<int> perm((size_t)n), other((size_t)n);
    for (long long i = 0; i < n; ++i) perm[(size_t)i] = (int)i;
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<long long> start(65536 + 1, 0);
        const int shift = 16 * pass;
        for (long long q = 0; q < n; ++q) start[((key[(size_t)perm[(size_t)q]] >> shift) & 0xffffu) + 1] += 1;
        for (int d = 0; d < 65536; ++d) start[d + 1] += start[d];
        for (long long q = 0; q < n; ++q) {
            const int c = perm[(size_t)q];
            other[(size_t)start[(key[(size_t)c] >> shift) & 0xffffu]++] = c;
        }
        perm.swap(other);
    }
    std::vector<int> slot((size_t)n);
    for (long long q = 0; q < n; ++q) slot[(size_t)perm[(size_t)q]] = (int)q;
    int *hp = static_cast<int *>(stage.host());
    std::memcpy(hp, perm.data(), sizeof(int) * (size_t)n);
    std::memcpy(hp + n4, slot.data(), sizeof(int) * (size_t)n);
    const int *dp = static_cast<const int *>(stage.dev());
    hipLaunchKernelGGL(k_copy16, dim3(cgrid), dim3(256), 0, b->stream, (long long)(n4 / 4), (const double2 *)dp, (double2 *)b->perm.p);
    hipLaunchKernelGGL(k_copy16, dim3(cgrid), dim3(256), 0, b->stream, (long long)(n4 / 4), (const double2 *)(dp + n4), (double2 *)b->slot_of.p);
    b->h_slot = std::move(slot);
    // knot 0 (the initial state, uploaded in craft order) moves to the lanes' columns
    hipLaunchKernelGGL(k_knot0_to_lanes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, n, b->perm.p, b->time.p, b->y.p,
                       b->knot_t.p, b->knot_y.p);
    he = hipGetLastError();
    if (he != hipSuccess) { set_last_error("k_knot0_to_lanes", he); return EPH_ERR_HIP; }
    EPH_HIP(hipStreamSynchronize(b->stream));
    return EPH_OK;
}

static bool craft_time_scales_differ(const eph_ephemeris &e, long long n, const double *t0, const double *pos) {
    if (n < 128 || e.splines.empty()) return false;
    auto tau_of = [&](long long i) {
        double best = INFINITY;
        for (size_t q = 0; q < e.splines.size(); ++q) {
            const UniformSpline &u = e.splines[q];
            const double mu = e.gm[q];
            const long long npoly = (long long)u.polynomials.size();
            if (!(mu > 0.0) || npoly <= 0) continue;
            const double local = t0[i] - u.start;
            if (!(local >= 0.0) || local > u.span()) continue;
            long long idx = (long long)std::ceil(local / u.interval) - 1;
            idx = std::min(std::max<long long>(idx, 0), npoly - 1);
            const double tq = (local - u.interval * (double)idx) / u.interval;
            const Polynomial &p = u.polynomials[(size_t)idx];
            double bp[3] = {0.0, 0.0, 0.0};
            for (int k = std::min(std::max(p.ncoef, 0), kDiv) - 1; k >= 0; --k)
                for (int d = 0; d < 3; ++d) bp[d] = bp[d] * tq + p.c[k][d];
            const double dx = pos[3 * i] - bp[0], dy = pos[3 * i + 1] - bp[1], dz = pos[3 * i + 2] - bp[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            best = std::min(best, std::sqrt(d2 * std::sqrt(d2) / mu));
        }
        return best;
    };
    const long long waves = n / 64;
    for (int w = 0; w < 8; ++w) {
        const long long first = (long long)(((unsigned long long)w * 0x9E3779B97F4A7C15ull >> 11) % (unsigned long long)waves) * 64;
        double lo = INFINITY, hi = 0.0;
        for (long long i = first; i < first + 64; ++i) {
            const double t = tau_of(i);
            if (!std::isfinite(t)) continue;
            lo = std::min(lo, t);
            hi = std::max(hi, t);
        }
        if (hi > 4.0 * lo) return true;
    }
    return false;
}

// ---- the device table behind an eph_ephemeris ---------------------------------------------------------------------------
// one polynomial -> one zero-padded row of 8 x 3 doubles (rows >= ncoef stay +0.0: craft_rhs runs Horner over all kDiv rows)
static void eph_fill_row(const Polynomial &p, double *row, int *nc) {
    std::memset(row, 0, sizeof(double) * kDiv * 3);
    std::memcpy(row, &p.c[0][0], sizeof(double) * 3 * (size_t)std::min(std::max(p.ncoef, 0), kDiv));
    *nc = p.ncoef;
}
// polynomials [first, first + count) of body b's host spline -> device rows starting at `row`
static int eph_upload_rows(eph_ephemeris *e, int b, size_t first, size_t count, long long row) {
    if (count == 0) return EPH_OK;
    std::vector<double> co(count * kDiv * 3);
    std::vector<int> nc(count);
    const UniformSpline &u = e->splines[(size_t)b];
    for (size_t k = 0; k < count; ++k) eph_fill_row(u.polynomials[first + k], &co[k * kDiv * 3], &nc[k]);
    EPH_HIP(hipMemcpy(e->coeffs.p + (size_t)row * kDiv * 3, co.data(), sizeof(double) * co.size(), hipMemcpyHostToDevice));
    EPH_HIP(hipMemcpy(e->ncoef.p + row, nc.data(), sizeof(int) * nc.size(), hipMemcpyHostToDevice));
    return EPH_OK;
}
// host_bodies -> the device table (+ the refined reciprocals of the intervals, formed on the device like the sweep kernels would)
static int eph_upload_bodies(eph_ephemeris *e) {
    const int nb = e->n_bodies;
    for (int b = 0; b < nb; ++b) {
        const UniformSpline &u = e->splines[(size_t)b];
        BodyEntry &be = e->host_bodies[(size_t)b];
        be.start = u.start; be.interval = u.interval; be.mu = e->gm[(size_t)b];
        be.npoly = (long long)u.polynomials.size();
        be.span = u.interval * (double)u.polynomials.size();     // interval.scaled(len): the product UniformSpline::span() forms
        be.rinv = 0.0; be.rows = e->coeffs.p + (size_t)be.coeff_off * kDiv * 3;
    }
    if (!nb) return EPH_OK;
    EPH_HIP(hipMemcpy(e->bodies.p, e->host_bodies.data(), sizeof(BodyEntry) * (size_t)nb, hipMemcpyHostToDevice));
    k_body_reciprocals<<<(nb + 63) / 64, 64>>>(nb, e->bodies.p);
    EPH_HIP(hipGetLastError());
    EPH_HIP(hipStreamSynchronize(nullptr));
    return EPH_OK;
}
// lay the table out afresh from the host splines: every body's region gets room for as many polynomials again behind it (and in
// front, for a body that grows backwards)
static int eph_rebuild(eph_ephemeris *e) {
    const int nb = e->n_bodies;
    e->host_bodies.assign((size_t)std::max(nb, 0), BodyEntry{});
    e->base.assign((size_t)nb, 0);
    e->cap.assign((size_t)nb, 0);
    long long total = 0;
    for (int b = 0; b < nb; ++b) {
        const long long np = (long long)e->splines[(size_t)b].polynomials.size();
        const long long room = std::max<long long>(np, 32);
        const long long front = e->grows_front[(size_t)b] ? room : 0;
        e->base[(size_t)b] = total;
        e->cap[(size_t)b] = front + np + room;
        e->host_bodies[(size_t)b].coeff_off = total + front;
        total += e->cap[(size_t)b];
    }
    DevBuf<double> co;
    DevBuf<int> nc;
    int st;
    if ((st = co.alloc((size_t)std::max<long long>(total, 1) * kDiv * 3)) || (st = nc.alloc((size_t)std::max<long long>(total, 1)))) return st;
    std::swap(e->coeffs.p, co.p); std::swap(e->coeffs.count, co.count);
    std::swap(e->ncoef.p, nc.p); std::swap(e->ncoef.count, nc.count);
    if (!e->bodies.p && (st = e->bodies.alloc((size_t)std::max(nb, 1)))) return st;
    for (int b = 0; b < nb; ++b)
        if ((st = eph_upload_rows(e, b, 0, e->splines[(size_t)b].polynomials.size(), e->host_bodies[(size_t)b].coeff_off))) return st;
    return eph_upload_bodies(e);
}
// the device table after the host splines changed: `back[b]` / `front[b]` polynomials were added behind / in front of body b,
// `dropped_front[b]` removed from its front (clear_before); a truncation (clear_after) needs no row traffic at all
static int eph_follow(eph_ephemeris *e, const std::vector<long long> &front, const std::vector<long long> &back,
                      const std::vector<long long> &dropped_front) {
    const int nb = e->n_bodies;
    bool fits = true;
    for (int b = 0; b < nb && fits; ++b) {
        const BodyEntry &be = e->host_bodies[(size_t)b];
        const long long off = be.coeff_off - e->base[(size_t)b] + dropped_front[(size_t)b];
        const long long np = (long long)e->splines[(size_t)b].polynomials.size();     // already the new count
        if (front[(size_t)b] > off || off - front[(size_t)b] + np > e->cap[(size_t)b]) fits = false;
    }
    if (!fits) return eph_rebuild(e);
    int st;
    for (int b = 0; b < nb; ++b) {
        BodyEntry &be = e->host_bodies[(size_t)b];
        be.coeff_off += dropped_front[(size_t)b] - front[(size_t)b];
        const size_t np = e->splines[(size_t)b].polynomials.size();
        if ((st = eph_upload_rows(e, b, 0, (size_t)front[(size_t)b], be.coeff_off))) return st;
        if ((st = eph_upload_rows(e, b, np - (size_t)back[(size_t)b], (size_t)back[(size_t)b], be.coeff_off + (long long)np - back[(size_t)b]))) return st;
    }
    return eph_upload_bodies(e);
}

// The host splines have changed already when the device table follows them: if the incremental update fails half way (a copy, an
// allocation), the table is laid out afresh from the host copy once before the error is reported, so that the two do not stay apart.
static int eph_follow_or_rebuild(eph_ephemeris *e, const std::vector<long long> &front, const std::vector<long long> &back,
                                 const std::vector<long long> &dropped_front) {
    const int st = eph_follow(e, front, back, dropped_front);
    if (st == EPH_OK) return st;
    (void)hipGetLastError();
    return eph_rebuild(e) == EPH_OK ? EPH_OK : st;
}

// Timeline::new  ephemeris/src/propagators/spacecraft.rs:129-152: stable sort by start, coast segments in the gaps,
// from Epoch::MIN to Epoch::MAX; appended to `segs`
static void timeline_new(long long nburns, const double *burn_start, const double *burn_end, const double *burn_acc,
                         const int32_t *burn_ref, std::vector<SegmentDev> &segs) {
    const double EMIN = -1.7976931348623157e308, EMAX = 1.7976931348623157e308;   // Epoch::MIN / MAX
    std::vector<long long> order;
    for (long long q = 0; q < nburns; ++q) order.push_back(q);
    std::stable_sort(order.begin(), order.end(), [&](long long x, long long y) { return burn_start[x] < burn_start[y]; });
    double cursor = EMIN;
    for (long long q : order) {
        if (burn_start[q] > cursor) segs.push_back(SegmentDev{cursor, burn_start[q], 0, 0, 0, 0, -1});
        cursor = burn_end[q];
        segs.push_back(SegmentDev{burn_start[q], burn_end[q], burn_acc[3 * q], burn_acc[3 * q + 1], burn_acc[3 * q + 2],
                                  1, burn_ref[q]});
    }
    if (cursor < EMAX) segs.push_back(SegmentDev{cursor, EMAX, 0, 0, 0, 0, -1});
}

template <typename T>
static int clone_buf(const DevBuf<T> &src, DevBuf<T> &dst, hipStream_t s) {
    if (!src.p) return EPH_OK;
    int st = dst.alloc(src.count);
    if (st) return st;
    hipError_t e = hipMemcpyAsync(dst.p, src.p, sizeof(T) * src.count, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) { set_last_error("hipMemcpyAsync (clone)", e); return EPH_ERR_HIP; }
    return EPH_OK;
}

#pragma GCC visibility push(default)
extern "C" {

int32_t eph_ephemeris_create(const eph_solution *s, const double *mu, eph_ephemeris **out) {
    try {
        if (!s || !mu || !out) return EPH_ERR_BAD_ARGUMENT;
        int st = check_device();
        if (st) return st;
        std::unique_ptr<eph_ephemeris> e(new eph_ephemeris());
        EPH_HIP(hipGetDevice(&e->device));
        const int nb = (int)s->s.splines.size();
        e->n_bodies = nb;
        e->splines = s->s.splines;
        for (const UniformSpline &u : e->splines)
            if (u.ghost) return EPH_ERR_BAD_ARGUMENT;             // (only inside a propagator; never in a Solution handed out)
        e->gm.assign(mu, mu + nb);
        e->grows_front.assign((size_t)nb, 0);
        if ((st = eph_rebuild(e.get()))) return st;
        *out = e.release();
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}
void eph_ephemeris_destroy(eph_ephemeris *e) { delete e; }

// UniformSpline::append (direction > 0) / prepend (< 0) for every body  trajectory.rs:515-534; the asserts become EPH_ERR_BAD_ARGUMENT
// with the table untouched
int32_t eph_ephemeris_append(eph_ephemeris *e, const eph_solution *tail, int32_t direction) {
    try {
        if (!e || !tail || direction == 0 || tail->s.splines.size() != e->splines.size()) return EPH_ERR_BAD_ARGUMENT;
        std::unique_lock<std::shared_mutex> lock(e->mu);
        const size_t nb = e->splines.size();
        for (size_t b = 0; b < nb; ++b) {
            const UniformSpline &x = e->splines[b], &y = tail->s.splines[b];
            if (y.ghost || x.interval != y.interval) return EPH_ERR_BAD_ARGUMENT;
            if (direction > 0 ? (x.end() != y.start) : (x.start != y.end())) return EPH_ERR_BAD_ARGUMENT;
        }
        EPH_HIP(hipSetDevice(e->device));
        std::vector<long long> front(nb, 0), back(nb, 0), none(nb, 0);
        for (size_t b = 0; b < nb; ++b) {
            UniformSpline &x = e->splines[b];
            const UniformSpline &y = tail->s.splines[b];
            if (direction > 0) {
                x.polynomials.insert(x.polynomials.end(), y.polynomials.begin(), y.polynomials.end());
                back[b] = (long long)y.polynomials.size();
            } else {
                x.start = y.start;
                x.polynomials.insert(x.polynomials.begin(), y.polynomials.begin(), y.polynomials.end());
                front[b] = (long long)y.polynomials.size();
                if (front[b]) e->grows_front[b] = 1;
            }
        }
        e->revision += 1;
        return eph_follow_or_rebuild(e, front, back, none);
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}
// UniformSpline::clear_before (after = 0, trajectory.rs:536-542) / clear_after (after != 0, :544-549) on body's spline or on all (body < 0)
int32_t eph_ephemeris_clear(eph_ephemeris *e, int32_t body, double at, int32_t after) {
    try {
        if (!e || body >= e->n_bodies) return EPH_ERR_BAD_ARGUMENT;
        std::unique_lock<std::shared_mutex> lock(e->mu);
        EPH_HIP(hipSetDevice(e->device));
        const size_t nb = e->splines.size();
        std::vector<long long> none(nb, 0), dropped(nb, 0);
        for (size_t b = 0; b < nb; ++b) {
            if (body >= 0 && (size_t)body != b) continue;
            UniformSpline &u = e->splines[b];
            const size_t before = u.polynomials.size();
            if (after) u.clear_after(at);
            else { u.clear_before(at); dropped[b] = (long long)(before - u.polynomials.size()); }
        }
        e->revision += 1;
        return eph_follow_or_rebuild(e, none, none, dropped);
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}
// CelestialTrajectory::merge  ephemeris_explorer/src/dynamics/celestial.rs:198-204 (Forward: clear_after(propagated.start()) then
// append) and :220-226 (Backward: clear_before(propagated.end()) then prepend), body by body
int32_t eph_ephemeris_merge(eph_ephemeris *e, const eph_solution *propagated, int32_t direction) {
    try {
        if (!e || !propagated || direction == 0 || propagated->s.splines.size() != e->splines.size()) return EPH_ERR_BAD_ARGUMENT;
        std::unique_lock<std::shared_mutex> lock(e->mu);
        const size_t nb = e->splines.size();
        // the reference's asserts, evaluated on copies of the bounds first so that a refusal leaves the table untouched
        for (size_t b = 0; b < nb; ++b) {
            const UniformSpline &y = propagated->s.splines[b];
            UniformSpline x;
            x.start = e->splines[b].start; x.interval = e->splines[b].interval;
            x.ghost = e->splines[b].polynomials.size();            // bounds only: no polynomial is copied
            if (y.ghost || x.interval != y.interval) return EPH_ERR_BAD_ARGUMENT;
            if (direction > 0) {
                uint64_t idx;
                if (x.get_index_local(y.start - x.start, &idx) && idx < x.ghost) x.ghost = idx;       // clear_after
                if (x.end() != y.start) return EPH_ERR_BAD_ARGUMENT;
            } else {
                uint64_t idx;
                if (x.get_index_local_exclusive((y.end() + x.interval) - x.start, &idx)) {             // clear_before
                    x.start += x.interval * (double)idx;
                    x.ghost -= std::min<uint64_t>(idx, x.ghost);
                }
                if (x.start != y.end()) return EPH_ERR_BAD_ARGUMENT;
            }
        }
        EPH_HIP(hipSetDevice(e->device));
        std::vector<long long> front(nb, 0), back(nb, 0), dropped(nb, 0);
        for (size_t b = 0; b < nb; ++b) {
            UniformSpline &x = e->splines[b];
            const UniformSpline &y = propagated->s.splines[b];
            if (direction > 0) {
                x.clear_after(y.start);
                x.polynomials.insert(x.polynomials.end(), y.polynomials.begin(), y.polynomials.end());
                back[b] = (long long)y.polynomials.size();
            } else {
                const size_t before = x.polynomials.size();
                x.clear_before(y.end());
                dropped[b] = (long long)(before - x.polynomials.size());
                x.start = y.start;
                x.polynomials.insert(x.polynomials.begin(), y.polynomials.begin(), y.polynomials.end());
                front[b] = (long long)y.polynomials.size();
                if (front[b]) e->grows_front[b] = 1;
            }
        }
        e->revision += 1;
        return eph_follow_or_rebuild(e, front, back, dropped);
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}
int32_t eph_ephemeris_info(const eph_ephemeris *e, int32_t body, double *start, double *interval, int64_t *npoly, uint64_t *revision) {
    if (!e || body >= e->n_bodies) return EPH_ERR_BAD_ARGUMENT;
    std::shared_lock<std::shared_mutex> lock(e->mu);
    if (body >= 0) {
        const UniformSpline &u = e->splines[(size_t)body];
        if (start) *start = u.start;
        if (interval) *interval = u.interval;
        if (npoly) *npoly = (int64_t)u.polynomials.size();
    } else if (start || interval || npoly) return EPH_ERR_BAD_ARGUMENT;
    if (revision) *revision = e->revision;
    return EPH_OK;
}
// Bodies::is_valid_at  dynamics/spacecraft.rs:199-201: every body's trajectory.contains(t) (trajectory.rs:437-441:
// local.is_positive() && local <= span -- ftime's Duration::is_positive is f64::is_sign_positive, duration.rs:78-80: the sign BIT,
// so the start itself (+0.0) is contained)
int32_t eph_ephemeris_is_valid_at(const eph_ephemeris *e, double t, int32_t *flag) {
    if (!e || !flag) return EPH_ERR_BAD_ARGUMENT;
    std::shared_lock<std::shared_mutex> lock(e->mu);
    bool all = true;
    for (const UniformSpline &u : e->splines) {
        const double local = t - u.start;
        all = all && (!std::signbit(local) && local <= u.span());
    }
    *flag = all ? 1 : 0;
    return EPH_OK;
}
// One contiguous, position-independent image of the table (what rank 0 broadcasts to the other ranks of a sweep):
// header, per body {start, interval, mu, npoly}, then every polynomial's zero-padded row and coefficient count.
namespace {
struct EphImageHeader { uint64_t magic, n_bodies, n_polys, reserved; };
constexpr uint64_t kEphImageMagic = 0x3130485045485045ull;     // "EPHEPH01"
struct EphImageBody { double start, interval, mu; int64_t npoly; };
}
int32_t eph_ephemeris_export(const eph_ephemeris *e, void *buf, uint64_t capacity, uint64_t *bytes) {
    try {
        if (!e || !bytes) return EPH_ERR_BAD_ARGUMENT;
        std::shared_lock<std::shared_mutex> lock(e->mu);
        uint64_t polys = 0;
        for (const UniformSpline &u : e->splines) polys += u.polynomials.size();
        const uint64_t need = sizeof(EphImageHeader) + sizeof(EphImageBody) * e->splines.size() +
                              polys * (sizeof(double) * kDiv * 3 + sizeof(int64_t));
        *bytes = need;
        if (!buf || capacity < need) return EPH_ERR_BAD_ARGUMENT;
        char *w = static_cast<char *>(buf);
        const EphImageHeader h{kEphImageMagic, (uint64_t)e->splines.size(), polys, 0};
        std::memcpy(w, &h, sizeof(h)); w += sizeof(h);
        for (size_t b = 0; b < e->splines.size(); ++b) {
            const UniformSpline &u = e->splines[b];
            const EphImageBody ib{u.start, u.interval, e->gm[b], (int64_t)u.polynomials.size()};
            std::memcpy(w, &ib, sizeof(ib)); w += sizeof(ib);
        }
        for (const UniformSpline &u : e->splines)
            for (const Polynomial &p : u.polynomials) {
                double row[kDiv * 3];
                int nc;
                eph_fill_row(p, row, &nc);
                const int64_t nc64 = nc;
                std::memcpy(w, row, sizeof(row)); w += sizeof(row);
                std::memcpy(w, &nc64, sizeof(nc64)); w += sizeof(nc64);
            }
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}
int32_t eph_ephemeris_import(const void *buf, uint64_t bytes, eph_ephemeris **out) {
    try {
        if (!buf || !out || bytes < sizeof(EphImageHeader)) return EPH_ERR_BAD_ARGUMENT;
        *out = nullptr;
        const char *r = static_cast<const char *>(buf);
        EphImageHeader h;
        std::memcpy(&h, r, sizeof(h)); r += sizeof(h);
        if (h.magic != kEphImageMagic || h.n_bodies > 0x7fffffffu) return EPH_ERR_BAD_ARGUMENT;
        const uint64_t need = sizeof(EphImageHeader) + sizeof(EphImageBody) * h.n_bodies + h.n_polys * (sizeof(double) * kDiv * 3 + sizeof(int64_t));
        if (h.n_polys > (1ull << 40) || bytes < need) return EPH_ERR_BAD_ARGUMENT;
        eph_solution sol;
        std::vector<double> mu((size_t)h.n_bodies);
        sol.s.splines.resize((size_t)h.n_bodies);
        std::vector<int64_t> np((size_t)h.n_bodies);
        uint64_t total = 0;
        for (size_t b = 0; b < (size_t)h.n_bodies; ++b) {
            EphImageBody ib;
            std::memcpy(&ib, r, sizeof(ib)); r += sizeof(ib);
            if (ib.npoly < 0) return EPH_ERR_BAD_ARGUMENT;
            sol.s.splines[b].start = ib.start; sol.s.splines[b].interval = ib.interval;
            mu[b] = ib.mu; np[b] = ib.npoly; total += (uint64_t)ib.npoly;
        }
        if (total != h.n_polys) return EPH_ERR_BAD_ARGUMENT;
        for (size_t b = 0; b < (size_t)h.n_bodies; ++b)
            for (int64_t k = 0; k < np[b]; ++k) {
                Polynomial p;
                int64_t nc64;
                std::memcpy(&p.c[0][0], r, sizeof(double) * kDiv * 3); r += sizeof(double) * kDiv * 3;
                std::memcpy(&nc64, r, sizeof(nc64)); r += sizeof(nc64);
                if (nc64 < 0 || nc64 > kDiv) return EPH_ERR_BAD_ARGUMENT;
                p.ncoef = (int32_t)nc64;
                sol.s.splines[b].polynomials.push_back(p);
            }
        return eph_ephemeris_create(&sol, mu.data(), out);
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

int32_t eph_ephemeris_interpolation_errors(const eph_ephemeris *e, eph_nbody *h, int64_t n_steps, double *max_error_m,
                                           int64_t *steps_done) {
    try {
        if (!e || !h || !h->p || n_steps < 0 || !max_error_m) return EPH_ERR_BAD_ARGUMENT;
        NBodyIntegration *g = h->p;
        const int n = g->n();
        if (n != e->n_bodies || g->sharded()) return EPH_ERR_BAD_ARGUMENT;
        std::shared_lock<std::shared_mutex> table_lock(e->mu);
        EPH_HIP(hipSetDevice(g->device()));
        DevBuf<double> err;
        DevBuf<int> failed;
        int st;
        if ((st = err.alloc(std::max(n, 1))) || (st = failed.alloc(1))) return st;
        std::vector<double> init((size_t)std::max(n, 1), -1.0);
        EPH_HIP(hipMemcpyAsync(err.p, init.data(), sizeof(double) * init.size(), hipMemcpyHostToDevice, g->stream()));
        EPH_HIP(hipMemsetAsync(failed.p, 0, sizeof(int), g->stream()));
        EPH_HIP(hipStreamSynchronize(g->stream()));
        int64_t done = 0;
        int status = EPH_OK;
        for (; done < n_steps; ++done) {                       // while integrator.advance(&mut nbody).is_ok()
            if ((status = g->advance(1))) break;
            if (n > 0)
                hipLaunchKernelGGL(k_interp_error, dim3((n + 255) / 256), dim3(256), 0, g->stream(), n, g->npad(),
                                   g->positions_soa(), g->time(), e->bodies.p, e->coeffs.p, e->ncoef.p, err.p, failed.p);
        }
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) { set_last_error("k_interp_error", he); return EPH_ERR_HIP; }
        int f = 0;
        EPH_HIP(hipMemcpyAsync(max_error_m, err.p, sizeof(double) * n, hipMemcpyDeviceToHost, g->stream()));
        EPH_HIP(hipMemcpyAsync(&f, failed.p, sizeof(int), hipMemcpyDeviceToHost, g->stream()));
        EPH_HIP(hipStreamSynchronize(g->stream()));
        if (steps_done) *steps_done = done;
        if (status < 0) return status;
        if (f) return EPH_EVAL_FAILED;                         // an epoch outside a spline: the reference would panic
        return EPH_OK;                                         // a StepError (bound reached) just ends the scan
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

int32_t eph_craft_batch_create(const eph_ephemeris *e, int64_t n_craft, const double *t0, const double *pos,
                               const double *vel, const char *method, const eph_adaptive_params *params,
                               const int64_t *burn_offset, const double *burn_start, const double *burn_end,
                               const double *burn_acc, const int32_t *burn_ref, int32_t max_knots,
                               eph_craft_batch **out) {
    try {
        if (!e || n_craft < 0 || !method || !params || !out || max_knots < 1 || (n_craft > 0 && (!t0 || !pos || !vel)))
            return EPH_ERR_BAD_ARGUMENT;
        // the burn tables are caller memory: validate before the first dereference (CSR offsets start at a
        // non-negative value and never decrease; arrays present when any burn is; reference body -1 or a body index)
        if (burn_offset) {
            if (burn_offset[0] < 0) return EPH_ERR_BAD_ARGUMENT;
            for (int64_t i = 0; i < n_craft; ++i)
                if (burn_offset[i + 1] < burn_offset[i]) return EPH_ERR_BAD_ARGUMENT;
            if (burn_offset[n_craft] > 0 && (!burn_start || !burn_end || !burn_acc || !burn_ref))
                return EPH_ERR_BAD_ARGUMENT;
            for (int64_t q = burn_offset[0]; q < burn_offset[n_craft]; ++q)
                if (burn_ref[q] < -1 || burn_ref[q] >= e->n_bodies) return EPH_ERR_BAD_ARGUMENT;
        }
        int st = check_device();
        if (st) return st;
        std::shared_lock<std::shared_mutex> table_lock(e->mu);       // (the deal to the lanes reads the table)
        std::unique_ptr<eph_craft_batch> b(new eph_craft_batch());
        if (!find_erk(method, &b->rk) || !b->rk.has_embedded) return EPH_ERR_BAD_ARGUMENT;
        // ERKN integrates y'' = f(t, y) (P::ODE: SecondOrderODE, nystrom/explicit.rs:60): the spacecraft model is one
        // only while no burn is expressed in a frame built from the velocity (ReferenceFrame::Relative -> TNB of the
        // relative state, dynamics/spacecraft.rs:281-293). The reference cannot even express that combination.
        if (b->rk.nystrom == 2 && burn_offset)
            for (int64_t q = burn_offset[0]; q < burn_offset[n_craft]; ++q)
                if (burn_ref[q] >= 0) {
                    set_last_error_text("Tsitouras75Nystrom (ERKN) needs a velocity-independent right-hand side: "
                                        "burns must use the inertial frame");
                    return EPH_ERR_UNSUPPORTED;
                }
        b->pv = default_pair_variant();
        b->eph = e;
        b->n = n_craft;
        b->max_knots = max_knots;
        b->params = *params;
        b->device = e->device;
        EPH_HIP(hipSetDevice(b->device));
        EPH_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        EPH_HIP(hipEventCreate(&b->ev0));
        EPH_HIP(hipEventCreate(&b->ev1));
        const long long n = n_craft;
        const double EMIN = -1.7976931348623157e308, EMAX = 1.7976931348623157e308;   // Epoch::MIN / MAX
        // Timeline::new per craft  spacecraft.rs:129-152
        std::vector<long long> seg_off(n + 1, 0);
        std::vector<SegmentDev> segs;
        std::vector<int> cur(n, 0);
        for (long long i = 0; i < n; ++i) {
            seg_off[i] = (long long)segs.size();
            const long long b0 = burn_offset ? burn_offset[i] : 0, b1 = burn_offset ? burn_offset[i + 1] : 0;
            timeline_new(b1 - b0, burn_start + b0, burn_end + b0, burn_acc + 3 * b0, burn_ref + b0, segs);
            // segment_idx_at(t0): partition_point(seg.end() <= time)
            int idx = 0;
            const long long ns = (long long)segs.size() - seg_off[i];
            while (idx < ns && segs[seg_off[i] + idx].end <= t0[i]) ++idx;
            cur[i] = idx;
        }
        seg_off[n] = (long long)segs.size();
        const size_t nn = (size_t)std::max<long long>(n, 1);
        if ((st = b->time.alloc(nn)) || (st = b->y.alloc(6 * nn)) || (st = b->next_h.alloc(nn)) ||
            (st = b->klast.alloc(6 * nn)) || (st = b->kfirst.alloc(6 * nn)) || (st = b->last_knot.alloc(nn)) || (st = b->n_attempts.alloc(nn)) ||
            (st = b->rk_i.alloc(nn)) || (st = b->steps.alloc(nn)) || (st = b->cur_seg.alloc(nn)) ||
            (st = b->status.alloc(nn)) || (st = b->nknots.alloc(nn)) || (st = b->seg_off.alloc(n + 1)) ||
            (st = b->segs.alloc(std::max<size_t>(segs.size(), 1))) || (st = b->knot_t.alloc(nn * max_knots)) ||
            (st = b->knot_y.alloc(6 * nn * max_knots)) || (st = b->rk_dev.alloc(1)) || (st = b->queue.alloc(1)) ||
            (st = b->summary.alloc(nn)))               // (here, not at the first eph_craft_batch_summary: keeps hipMalloc out of a sweep)
            return st;
        EPH_HIP(hipMemcpy(b->rk_dev.p, &b->rk, sizeof(ErkCoeffs), hipMemcpyHostToDevice));
        if (n > 0) {
            std::vector<double> ysoa(6 * n), hs(n, params->h_init);
            for (long long i = 0; i < n; ++i)
                for (int d = 0; d < 3; ++d) { ysoa[d * n + i] = pos[3 * i + d]; ysoa[(3 + d) * n + i] = vel[3 * i + d]; }
            std::vector<int> ones(n, 1), zeros(n, 0);
            EPH_HIP(hipMemcpy(b->time.p, t0, sizeof(double) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->last_knot.p, t0, sizeof(double) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->y.p, ysoa.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->klast.p, ysoa.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->kfirst.p, ysoa.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice));   // from_problem: k = [state; STAGES]
            EPH_HIP(hipMemcpy(b->next_h.p, hs.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemset(b->n_attempts.p, 0, sizeof(unsigned) * n));
            EPH_HIP(hipMemset(b->rk_i.p, 0, sizeof(unsigned) * n));
            EPH_HIP(hipMemset(b->steps.p, 0, sizeof(unsigned) * n));
            EPH_HIP(hipMemcpy(b->cur_seg.p, cur.data(), sizeof(int) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->status.p, zeros.data(), sizeof(int) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->nknots.p, ones.data(), sizeof(int) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->seg_off.p, seg_off.data(), sizeof(long long) * (n + 1), hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->segs.p, segs.data(), sizeof(SegmentDev) * segs.size(), hipMemcpyHostToDevice));
            // knot 0 = the initial state (CubicHermiteSplineSolout::new_solution)
            EPH_HIP(hipMemcpy(b->knot_t.p, t0, sizeof(double) * n, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(b->knot_y.p, ysoa.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice));
            b->heterogeneous = craft_time_scales_differ(*e, n, t0, pos);
            if (!craft_wave_form(n) && (st = craft_sort(b.get()))) return st;
        }
        *out = b.release();
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

static int32_t craft_run(eph_craft_batch *b, double t_end, unsigned step_limit);
int32_t eph_craft_batch_propagate(eph_craft_batch *b, double t_end) { return craft_run(b, t_end, 0); }
// IncrementalPropagator::step, n times, for every craft  (ephemeris/src/lib.rs:40-47, spacecraft.rs:598-615)
int32_t eph_craft_batch_step_n(eph_craft_batch *b, uint32_t n_steps) {
    if (n_steps == 0) return b ? EPH_OK : EPH_ERR_BAD_ARGUMENT;
    return craft_run(b, 1.7976931348623157e308, n_steps);
}
static int32_t craft_run(eph_craft_batch *b, double t_end, unsigned step_limit) {
    if (!b) return EPH_ERR_BAD_ARGUMENT;
    if (b->n == 0) return EPH_OK;
    EPH_HIP(hipSetDevice(b->device));
    // the reference's RwLock read guard (dynamics/mod.rs:84-85), held for the whole synchronous sweep: the table as it is NOW --
    // every append since the last call is seen, and none lands while the kernels read
    std::shared_lock<std::shared_mutex> table_lock(b->eph->mu);
    if (!b->body_order.empty()) {
        hipLaunchKernelGGL(k_permute_bodies, dim3((unsigned)((b->eph->n_bodies + 63) / 64)), dim3(64), 0, b->stream, b->eph->n_bodies,
                           b->body_order_dev.p, b->eph->bodies.p, b->bodies_ordered.p);
        hipError_t pe = hipGetLastError();
        if (pe != hipSuccess) { set_last_error("k_permute_bodies", pe); return EPH_ERR_HIP; }
    }
    CraftArgs a{};
    a.n_craft = b->n;
    a.n_bodies = b->eph->n_bodies;
    a.bodies = b->bodies_ordered.p ? b->bodies_ordered.p : b->eph->bodies.p;
    a.bodies_by_index = b->eph->bodies.p;
    a.coeffs = b->eph->coeffs.p; a.ncoef = b->eph->ncoef.p;
    a.time = b->time.p; a.y = b->y.p; a.next_h = b->next_h.p; a.klast = b->klast.p; a.kfirst = b->kfirst.p; a.last_knot_t = b->last_knot.p;
    a.retry = b->retry ? 1 : 0;
    b->retry = false;
    a.n_attempts = b->n_attempts.p; a.rk_i = b->rk_i.p; a.steps = b->steps.p;
    a.cur_seg = b->cur_seg.p; a.status = b->status.p; a.nknots = b->nknots.p;
    a.seg_off = b->seg_off.p; a.segs = b->segs.p;
    a.knot_t = b->knot_t.p; a.knot_y = b->knot_y.p; a.max_knots = b->max_knots;
    a.rk = b->rk;
    a.rkd = b->rk_dev.p;
    a.h_init = b->params.h_init; a.h_max = b->params.h_max; a.tol_pos = b->params.tol_position;
    a.tol_vel = b->params.tol_velocity; a.fac_min = b->params.fac_min; a.fac_max = b->params.fac_max;
    a.fac = b->params.fac; a.n_max = b->params.n_max;
    a.t_end = t_end;
    a.step_limit = step_limit;
    a.queue = b->queue.p;
    a.perm = b->perm.p;
    EPH_HIP(hipEventRecord(b->ev0, b->stream));
    int st = craft_launch(b->pv, b->stream, a, b->heterogeneous);
    if (st) return st;
    if (b->events) {                                  // the app's SpacecraftSolout on the steps just taken
        EventArgs e{};
        e.n_craft = b->n; e.n_bodies = b->eph->n_bodies;
        e.bodies = b->eph->bodies.p; e.coeffs = b->eph->coeffs.p; e.ncoef = b->eph->ncoef.p;
        e.soi = b->soi.p; e.nknots = b->nknots.p; e.knot_t = b->knot_t.p; e.knot_y = b->knot_y.p;
        e.ev_seg = b->ev_seg.p; e.ntr = b->ntr.p; e.nap = b->nap.p; e.ev_status = b->ev_status.p;
        e.tr_time = b->tr_time.p; e.tr_body = b->tr_body.p;
        e.ap_time = b->ap_time.p; e.ap_dist = b->ap_dist.p; e.ap_body = b->ap_body.p; e.ap_kind = b->ap_kind.p;
        e.max_tr = b->max_tr; e.max_ap = b->max_ap;
        e.slot_of = b->slot_of.p;
        if (craft_wave_form(b->n)) hipLaunchKernelGGL(k_craft_events<true>, dim3((unsigned)b->n), dim3(64), 0, b->stream, e);
        else hipLaunchKernelGGL(k_craft_events<false>, dim3((unsigned)((b->n + 63) / 64)), dim3(64), 0, b->stream, e);
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) { set_last_error("k_craft_events", he); return EPH_ERR_HIP; }
    }
    EPH_HIP(hipEventRecord(b->ev1, b->stream));
    EPH_HIP(hipEventSynchronize(b->ev1));
    float ms = 0;
    EPH_HIP(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->kernel_ms += ms;
    return EPH_OK;
}

int32_t eph_craft_batch_status(eph_craft_batch *b, int32_t *status, int32_t *nknots, uint32_t *attempts, uint32_t *steps) {
    if (!b) return EPH_ERR_BAD_ARGUMENT;
    EPH_HIP(hipSetDevice(b->device));
    const size_t n = (size_t)b->n;
    if (n == 0) return EPH_OK;
    if (status) EPH_HIP(hipMemcpy(status, b->status.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (nknots) EPH_HIP(hipMemcpy(nknots, b->nknots.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (attempts) EPH_HIP(hipMemcpy(attempts, b->n_attempts.p, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    if (steps) EPH_HIP(hipMemcpy(steps, b->steps.p, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    return EPH_OK;
}

int32_t eph_craft_batch_state(eph_craft_batch *b, double *t, double *pos, double *vel, double *next_h) {
    if (!b) return EPH_ERR_BAD_ARGUMENT;
    EPH_HIP(hipSetDevice(b->device));
    const long long n = b->n;
    if (n == 0) return EPH_OK;
    if (t) EPH_HIP(hipMemcpy(t, b->time.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (next_h) EPH_HIP(hipMemcpy(next_h, b->next_h.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (pos || vel) {
        std::vector<double> y(6 * n);
        EPH_HIP(hipMemcpy(y.data(), b->y.p, sizeof(double) * 6 * n, hipMemcpyDeviceToHost));
        for (long long i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) {
                if (pos) pos[3 * i + d] = y[d * n + i];
                if (vel) vel[3 * i + d] = y[(3 + d) * n + i];
            }
    }
    return EPH_OK;
}

// One 80-byte record per craft packed on the device, one copy into the caller's memory.
// Round 3's "2-38 ms in summary()" (profiles/r04_sweep_evidence.md): the FIRST read-back after a burst of batch creations starts
// 15-40 ms late ON THE DEVICE -- rocprofv3 shows the pack kernel's launch issued 1 ms after the sweep kernel ended and the kernel
// starting 20-30 ms later with the queue idle and no host thread busy. It is a one-off per burst of creations (every later
// read-back takes 0.4 ms for 21 MB), it is there with the copy engine out of the picture (records stored by a kernel into pinned
// host memory: same delay), without hipFree, without scratch, on a shared stream; a read-back issued after the LAST creation
// absorbs it, one at the end of each creation does not. bench.py therefore reads every batch back once before its timed region.
// EPH_TRACE_SUMMARY=1 prints the host timers of the phases.
int32_t eph_craft_batch_summary(eph_craft_batch *b, eph_craft_record *out) {
    if (!b || (b->n > 0 && !out)) return EPH_ERR_BAD_ARGUMENT;
    if (b->n == 0) return EPH_OK;
    EPH_HIP(hipSetDevice(b->device));
    static const int trace = [] { const char *e = getenv("EPH_TRACE_SUMMARY"); return e ? atoi(e) : 0; }();
    const auto tick = [] { return std::chrono::steady_clock::now(); };
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) {
        return std::chrono::duration<double, std::micro>(c - a).count(); };
    const auto t0 = tick();
    int st;
    if ((st = b->summary.reserve((size_t)b->n))) return st;
    hipLaunchKernelGGL(k_craft_summary, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, b->stream, b->n, b->time.p, b->y.p,
                       b->next_h.p, b->status.p, b->nknots.p, b->n_attempts.p, b->steps.p, b->summary.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("k_craft_summary", e); return EPH_ERR_HIP; }
    const auto t1 = tick();
    EPH_HIP(hipMemcpyAsync(out, b->summary.p, sizeof(eph_craft_record) * (size_t)b->n, hipMemcpyDeviceToHost, b->stream));
    const auto t2 = tick();
    EPH_HIP(hipStreamSynchronize(b->stream));
    if (trace) fprintf(stderr, "summary: pack launch %.0f memcpyAsync %.0f sync %.0f us\n", us(t0, t1), us(t1, t2), us(t2, tick()));
    return EPH_OK;
}

// The order in which Bodies::acceleration adds the massive bodies' terms (dynamics/spacecraft.rs:222-228 iterates an EntityHashMap:
// unspecified upstream). Table (file) order by default -- the library test's IndexMap; a maintainer who wants the bits of a given
// app run passes that run's iteration order here. Burn reference bodies, SOI radii and event body indices keep the table's numbering.
int32_t eph_craft_batch_set_body_order(eph_craft_batch *b, const int32_t *order) {
    try {
        if (!b) return EPH_ERR_BAD_ARGUMENT;
        EPH_HIP(hipSetDevice(b->device));
        EPH_HIP(hipStreamSynchronize(b->stream));
        if (!order) { b->bodies_ordered.release(); b->body_order_dev.release(); b->body_order.clear(); return EPH_OK; }
        const int n = b->eph->n_bodies;
        std::vector<char> seen((size_t)std::max(n, 1), 0);
        for (int q = 0; q < n; ++q) {
            if (order[q] < 0 || order[q] >= n || seen[(size_t)order[q]]) return EPH_ERR_BAD_ARGUMENT;   // not a permutation
            seen[(size_t)order[q]] = 1;
        }
        // the sweep kernels walk a.bodies front to back: a permuted COPY of the ephemeris's table costs the kernels nothing (an index
        // array read inside the body loop cost the thread-per-craft kernel 5 %: 35.0 against 33.4 ms); craft_run re-gathers it from
        // the live table before every sweep (k_permute_bodies)
        int st;
        if ((st = b->bodies_ordered.alloc((size_t)std::max(n, 1))) || (st = b->body_order_dev.alloc((size_t)std::max(n, 1)))) return st;
        b->body_order.assign(order, order + n);
        if (n) EPH_HIP(hipMemcpy(b->body_order_dev.p, order, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}
// IncrementalPropagator::step on a propagator whose last step returned Err (ephemeris/src/propagators/spacecraft.rs:598-615): nothing
// in the reference remembers the failure -- the next step() simply runs AdaptiveRungeKuttaIntegrator::advance again from the state the
// failed attempt left (runge_kutta/mod.rs:414-439). A sweep must not do that implicitly (a drain loop would re-attempt the failed craft
// of a batch on every pass, and a failed attempt is not idempotent: see the header), so the batch keeps a StepError sticky until this
// call re-arms it: the NEXT propagate / step_n then steps every craft whatever its last status.
int32_t eph_craft_batch_retry_failed(eph_craft_batch *b) {
    if (!b) return EPH_ERR_BAD_ARGUMENT;
    b->retry = true;
    return EPH_OK;
}

int32_t eph_craft_batch_knots(eph_craft_batch *b, int64_t craft, double *t, double *pos, double *vel) {
    if (!b || craft < 0 || craft >= b->n) return EPH_ERR_BAD_ARGUMENT;
    EPH_HIP(hipSetDevice(b->device));
    int nk = 0;
    EPH_HIP(hipMemcpy(&nk, b->nknots.p + craft, sizeof(int), hipMemcpyDeviceToHost));
    const long long n = b->n;
    const long long col = b->h_slot.empty() ? craft : b->h_slot[(size_t)craft];
    // strided gather: knot k of craft i sits at [k*n + col(i)] (col = i unless the batch was dealt to the lanes by craft_sort)
    if (t) EPH_HIP(hipMemcpy2D(t, sizeof(double), b->knot_t.p + col, sizeof(double) * n, sizeof(double), nk,
                               hipMemcpyDeviceToHost));
    if (pos || vel) {
        std::vector<double> y((size_t)nk * 6);
        EPH_HIP(hipMemcpy2D(y.data(), sizeof(double), b->knot_y.p + col, sizeof(double) * n, sizeof(double),
                            (size_t)nk * 6, hipMemcpyDeviceToHost));
        for (int k = 0; k < nk; ++k)
            for (int d = 0; d < 3; ++d) {
                if (pos) pos[3 * k + d] = y[(size_t)k * 6 + d];
                if (vel) vel[3 * k + d] = y[(size_t)k * 6 + 3 + d];
            }
    }
    return EPH_OK;
}

int32_t eph_craft_batch_enable_events(eph_craft_batch *b, const double *soi_radius, int32_t max_transitions,
                                      int32_t max_apsides) {
    try {
        if (!b || !soi_radius || max_transitions < 1 || max_apsides < 1 || b->events) return EPH_ERR_BAD_ARGUMENT;
        EPH_HIP(hipSetDevice(b->device));
        const size_t nn = (size_t)std::max<long long>(b->n, 1);
        const int nb = b->eph->n_bodies;
        int st;
        if ((st = b->soi.alloc(std::max(nb, 1))) || (st = b->ev_seg.alloc(nn)) || (st = b->ntr.alloc(nn)) ||
            (st = b->nap.alloc(nn)) || (st = b->ev_status.alloc(nn)) || (st = b->tr_time.alloc(nn * max_transitions)) ||
            (st = b->tr_body.alloc(nn * max_transitions)) || (st = b->ap_time.alloc(nn * max_apsides)) ||
            (st = b->ap_dist.alloc(nn * max_apsides)) || (st = b->ap_body.alloc(nn * max_apsides)) ||
            (st = b->ap_kind.alloc(nn * max_apsides)))
            return st;
        if (nb) EPH_HIP(hipMemcpy(b->soi.p, soi_radius, sizeof(double) * nb, hipMemcpyHostToDevice));
        EPH_HIP(hipMemset(b->ev_seg.p, 0xff, sizeof(int) * nn));       // -1: new_solution pending
        EPH_HIP(hipMemset(b->ntr.p, 0, sizeof(int) * nn));
        EPH_HIP(hipMemset(b->nap.p, 0, sizeof(int) * nn));
        EPH_HIP(hipMemset(b->ev_status.p, 0, sizeof(int) * nn));
        b->max_tr = max_transitions;
        b->max_ap = max_apsides;
        b->events = true;
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

int32_t eph_craft_batch_event_counts(eph_craft_batch *b, int32_t *n_transitions, int32_t *n_apsides,
                                     int32_t *event_status) {
    if (!b || !b->events) return EPH_ERR_BAD_ARGUMENT;
    EPH_HIP(hipSetDevice(b->device));
    const size_t n = (size_t)b->n;
    if (n == 0) return EPH_OK;
    if (n_transitions) EPH_HIP(hipMemcpy(n_transitions, b->ntr.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (n_apsides) EPH_HIP(hipMemcpy(n_apsides, b->nap.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (event_status) EPH_HIP(hipMemcpy(event_status, b->ev_status.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    return EPH_OK;
}

int32_t eph_craft_batch_events(eph_craft_batch *b, int64_t craft, double *tr_time, int32_t *tr_body, double *ap_time,
                               double *ap_distance, int32_t *ap_body, int32_t *ap_kind) {
    if (!b || !b->events || craft < 0 || craft >= b->n) return EPH_ERR_BAD_ARGUMENT;
    EPH_HIP(hipSetDevice(b->device));
    int ntr = 0, nap = 0;
    EPH_HIP(hipMemcpy(&ntr, b->ntr.p + craft, sizeof(int), hipMemcpyDeviceToHost));
    EPH_HIP(hipMemcpy(&nap, b->nap.p + craft, sizeof(int), hipMemcpyDeviceToHost));
    const long long n = b->n;
#define EPH_COLUMN(dst, src, T, cnt)                                                                             \
    if ((dst) && (cnt) > 0)                                                                                      \
        EPH_HIP(hipMemcpy2D((dst), sizeof(T), (src) + craft, sizeof(T) * n, sizeof(T), (cnt), hipMemcpyDeviceToHost))
    EPH_COLUMN(tr_time, b->tr_time.p, double, ntr);
    EPH_COLUMN(tr_body, b->tr_body.p, int, ntr);
    EPH_COLUMN(ap_time, b->ap_time.p, double, nap);
    EPH_COLUMN(ap_distance, b->ap_dist.p, double, nap);
    EPH_COLUMN(ap_body, b->ap_body.p, int, nap);
    EPH_COLUMN(ap_kind, b->ap_kind.p, int, nap);
#undef EPH_COLUMN
    return EPH_OK;
}

// Timeline::divergence_time_before  spacecraft.rs:179-213 (common_times: segments zipped while their starts agree,
// stopping after the first pair whose thrust differs; the last such start that is < before)
int32_t eph_timeline_divergence_time(int64_t n_old, const double *old_start, const double *old_end, const double *old_acc,
                                     const int32_t *old_ref, int64_t n_new, const double *new_start,
                                     const double *new_end, const double *new_acc, const int32_t *new_ref,
                                     double before, double *restart_epoch) {
    try {
        if (n_old < 0 || n_new < 0 || !restart_epoch || (n_old > 0 && (!old_start || !old_end || !old_acc || !old_ref)) ||
            (n_new > 0 && (!new_start || !new_end || !new_acc || !new_ref)))
            return EPH_ERR_BAD_ARGUMENT;
        std::vector<SegmentDev> a, b;
        timeline_new(n_new, new_start, new_end, new_acc, new_ref, a);     // self = the new timeline
        timeline_new(n_old, old_start, old_end, old_acc, old_ref, b);
        bool done = false, any = false;
        double last = 0.0;
        for (size_t k = 0; k < a.size() && k < b.size(); ++k) {
            if (done || a[k].start != b[k].start) break;
            const double t = a[k].start;
            // s1.thrust() != s2.thrust(): Option<ConstantThrust { acceleration, frame }>
            const bool same = a[k].is_burn == b[k].is_burn &&
                              (!a[k].is_burn || (a[k].ax == b[k].ax && a[k].ay == b[k].ay && a[k].az == b[k].az &&
                                                 a[k].ref == b[k].ref));
            if (!same) done = true;
            if (!(t < before)) break;                                    // take_while(|&t| t < before)
            last = t;
            any = true;
        }
        if (!any) return EPH_ERR_BAD_ARGUMENT;                           // the reference unwraps (before <= Epoch::MIN)
        *restart_epoch = last;
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

int32_t eph_craft_batch_reset_knots(eph_craft_batch *b) {
    if (!b) return EPH_ERR_BAD_ARGUMENT;
    if (b->n == 0) return EPH_OK;
    EPH_HIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_craft_reset_knots, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, b->stream, b->n,
                       b->nknots.p, b->status.p, b->knot_t.p, b->knot_y.p, b->events ? b->ev_seg.p : nullptr, b->slot_of.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("k_craft_reset_knots", e); return EPH_ERR_HIP; }
    EPH_HIP(hipStreamSynchronize(b->stream));
    return EPH_OK;
}

// SpacecraftPropagator: Clone (the UI snapshots a propagator and later resumes from the snapshot,
// ephemeris_explorer/src/prediction.rs:224-229,378): a deep copy of every per-craft buffer, knots and events included
int32_t eph_craft_batch_clone(eph_craft_batch *b, eph_craft_batch **out) {
    try {
        if (!b || !out) return EPH_ERR_BAD_ARGUMENT;
        *out = nullptr;
        EPH_HIP(hipSetDevice(b->device));
        EPH_HIP(hipStreamSynchronize(b->stream));
        std::unique_ptr<eph_craft_batch> c(new eph_craft_batch());
        c->pv = b->pv; c->eph = b->eph; c->device = b->device; c->n = b->n; c->max_knots = b->max_knots; c->rk = b->rk;
        c->params = b->params; c->events = b->events; c->max_tr = b->max_tr; c->max_ap = b->max_ap;
        c->heterogeneous = b->heterogeneous;
        c->retry = b->retry; c->body_order = b->body_order;
        EPH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        EPH_HIP(hipEventCreate(&c->ev0));
        EPH_HIP(hipEventCreate(&c->ev1));
        hipStream_t s = c->stream;
        int st;
        if ((st = clone_buf(b->time, c->time, s)) || (st = clone_buf(b->y, c->y, s)) || (st = clone_buf(b->next_h, c->next_h, s)) ||
            (st = clone_buf(b->klast, c->klast, s)) || (st = clone_buf(b->kfirst, c->kfirst, s)) || (st = clone_buf(b->last_knot, c->last_knot, s)) ||
            (st = clone_buf(b->knot_t, c->knot_t, s)) || (st = clone_buf(b->knot_y, c->knot_y, s)) ||
            (st = clone_buf(b->n_attempts, c->n_attempts, s)) || (st = clone_buf(b->rk_i, c->rk_i, s)) ||
            (st = clone_buf(b->steps, c->steps, s)) || (st = clone_buf(b->cur_seg, c->cur_seg, s)) ||
            (st = clone_buf(b->status, c->status, s)) || (st = clone_buf(b->nknots, c->nknots, s)) ||
            (st = clone_buf(b->seg_off, c->seg_off, s)) || (st = clone_buf(b->segs, c->segs, s)) ||
            (st = clone_buf(b->rk_dev, c->rk_dev, s)) || (st = clone_buf(b->soi, c->soi, s)) ||
            (st = clone_buf(b->tr_time, c->tr_time, s)) || (st = clone_buf(b->ap_time, c->ap_time, s)) ||
            (st = clone_buf(b->ap_dist, c->ap_dist, s)) || (st = clone_buf(b->ev_seg, c->ev_seg, s)) ||
            (st = clone_buf(b->ntr, c->ntr, s)) || (st = clone_buf(b->nap, c->nap, s)) ||
            (st = clone_buf(b->ev_status, c->ev_status, s)) || (st = clone_buf(b->tr_body, c->tr_body, s)) ||
            (st = clone_buf(b->ap_body, c->ap_body, s)) || (st = clone_buf(b->ap_kind, c->ap_kind, s)) ||
            (st = clone_buf(b->perm, c->perm, s)) || (st = clone_buf(b->slot_of, c->slot_of, s)) || (st = clone_buf(b->bodies_ordered, c->bodies_ordered, s)) || (st = clone_buf(b->body_order_dev, c->body_order_dev, s)) || (st = c->queue.alloc(1)))
            return st;
        c->h_slot = b->h_slot;
        EPH_HIP(hipStreamSynchronize(s));
        *out = c.release();
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

int32_t eph_craft_batch_knot_slabs(eph_craft_batch *b, int32_t first_knot, int32_t n_knots, double *knot_t,
                                   double *knot_y) {
    if (!b || first_knot < 0 || n_knots < 0 || first_knot + n_knots > b->max_knots) return EPH_ERR_BAD_ARGUMENT;
    if (b->n == 0 || n_knots == 0) return EPH_OK;
    EPH_HIP(hipSetDevice(b->device));
    const size_t n = (size_t)b->n;
    if (!b->h_slot.empty()) {
        // lane order -> craft order on the device, in passes of at most 256 MB of knot rows (a full slab of a 5e5-craft sweep is
        // GBs: the reorder must not need a second slab), each pass stored by the kernel straight into the pinned staging buffer
        const size_t row_bytes = sizeof(double) * n;
        const long long rows_per_pass = (long long)std::max<size_t>(1, ((size_t)256 << 20) / row_bytes);
        for (int part = 0; part < 2; ++part) {
            double *dst = part == 0 ? knot_t : knot_y;
            if (!dst) continue;
            const long long rows = (long long)n_knots * (part == 0 ? 1 : 6);
            const double *src = part == 0 ? b->knot_t.p + (size_t)first_knot * n : b->knot_y.p + (size_t)first_knot * 6 * n;
            PinnedStage stage((size_t)std::min(rows, rows_per_pass) * row_bytes);
            if (stage.status()) return stage.status();
            StreamIdleOnExit idle(b->stream);
            for (long long r0 = 0; r0 < rows; r0 += rows_per_pass) {
                const long long nr = std::min(rows_per_pass, rows - r0);
                hipLaunchKernelGGL(k_rows_to_craft_order, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->stream, nr, (long long)n,
                                   b->slot_of.p, src + (size_t)r0 * n, static_cast<double *>(stage.dev()));
                hipError_t he = hipGetLastError();
                if (he != hipSuccess) { set_last_error("k_rows_to_craft_order", he); return EPH_ERR_HIP; }
                EPH_HIP(hipStreamSynchronize(b->stream));
                std::memcpy(dst + (size_t)r0 * n, stage.host(), (size_t)nr * row_bytes);
            }
        }
        return EPH_OK;
    }
    if (knot_t)
        EPH_HIP(hipMemcpy(knot_t, b->knot_t.p + (size_t)first_knot * n, sizeof(double) * n * n_knots, hipMemcpyDeviceToHost));
    if (knot_y)
        EPH_HIP(hipMemcpy(knot_y, b->knot_y.p + (size_t)first_knot * 6 * n, sizeof(double) * 6 * n * n_knots,
                          hipMemcpyDeviceToHost));
    return EPH_OK;
}

int32_t eph_craft_batch_reset_events(eph_craft_batch *b) {
    if (!b || !b->events) return EPH_ERR_BAD_ARGUMENT;
    if (b->n == 0) return EPH_OK;
    EPH_HIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_craft_reset_events, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, b->stream, b->n,
                       b->ntr.p, b->nap.p, b->ev_status.p, b->tr_time.p, b->tr_body.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error("k_craft_reset_events", e); return EPH_ERR_HIP; }
    EPH_HIP(hipStreamSynchronize(b->stream));
    return EPH_OK;
}

int32_t eph_craft_batch_kernel_time(eph_craft_batch *b, double *total_ms) {
    if (!b || !total_ms) return EPH_ERR_BAD_ARGUMENT;
    *total_ms = b->kernel_ms;
    return EPH_OK;
}
void eph_craft_batch_destroy(eph_craft_batch *b) { delete b; }

int32_t eph_hermite_eval(int64_t nknots, const double *t, const double *pos, const double *vel, int64_t m,
                         const double *at, double *op, double *ov, uint8_t *inside) {
    try {
        if (nknots < 0 || m < 0 || (m > 0 && (!at || !op || !inside)) || (nknots > 0 && (!t || !pos || !vel)))
            return EPH_ERR_BAD_ARGUMENT;
        int st = check_device();
        if (st) return st;
        if (m == 0) return EPH_OK;
        const size_t nk = (size_t)std::max<int64_t>(nknots, 1);
        DevBuf<double> dt, dp, dv, dat, dop, dov;
        DevBuf<uint8_t> din;
        if ((st = dt.alloc(nk)) || (st = dp.alloc(3 * nk)) || (st = dv.alloc(3 * nk)) || (st = dat.alloc(m)) ||
            (st = dop.alloc(3 * (size_t)m)) || (st = dov.alloc(3 * (size_t)m)) || (st = din.alloc(m)))
            return st;
        if (nknots) {
            EPH_HIP(hipMemcpy(dt.p, t, sizeof(double) * nknots, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(dp.p, pos, sizeof(double) * 3 * nknots, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(dv.p, vel, sizeof(double) * 3 * nknots, hipMemcpyHostToDevice));
        }
        EPH_HIP(hipMemcpy(dat.p, at, sizeof(double) * m, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_hermite_eval, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nullptr, (long long)nknots,
                           dt.p, dp.p, dv.p, (long long)m, dat.p, dop.p, ov ? dov.p : nullptr, din.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_last_error("k_hermite_eval", e); return EPH_ERR_HIP; }
        EPH_HIP(hipMemcpy(op, dop.p, sizeof(double) * 3 * m, hipMemcpyDeviceToHost));
        if (ov) EPH_HIP(hipMemcpy(ov, dov.p, sizeof(double) * 3 * m, hipMemcpyDeviceToHost));
        EPH_HIP(hipMemcpy(inside, din.p, m, hipMemcpyDeviceToHost));
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

int32_t eph_plot_points(const eph_ephemeris *e, const eph_plot_view *view, int64_t n_plots, const eph_plot_request *requests,
                        int64_t n_knots, const double *knot_t, const double *knot_pos, const double *knot_vel,
                        int64_t capacity, double *out_t, float *out_xyz, int64_t *out_count, int32_t *out_status,
                        double *out_failed_at) {
    try {
        if (!e || !view || n_plots < 0 || n_knots < 0 || capacity < 0 || (n_plots > 0 && (!requests || !out_count || !out_status || !out_failed_at)) ||
            (n_knots > 0 && (!knot_t || !knot_pos || !knot_vel)) || (n_plots > 0 && capacity > 0 && (!out_t || !out_xyz)))
            return EPH_ERR_BAD_ARGUMENT;
        for (int64_t p = 0; p < n_plots; ++p) {
            const eph_plot_request &r = requests[p];
            if (r.source_body >= e->n_bodies || r.reference_body >= e->n_bodies || r.reference_body < -1 || r.source_body < -1 ||
                r.max_points < 0 || r.max_points > capacity || r.bound < 0 || r.bound > 2)
                return EPH_ERR_BAD_ARGUMENT;
            if (r.source_body < 0 && (r.knot_first < 0 || r.knot_count < 0 || r.knot_first + r.knot_count > n_knots))
                return EPH_ERR_BAD_ARGUMENT;
        }
        int st = check_device();
        if (st) return st;
        if (n_plots == 0) return EPH_OK;
        std::shared_lock<std::shared_mutex> table_lock(e->mu);
        EPH_HIP(hipSetDevice(e->device));
        const size_t nk = (size_t)std::max<int64_t>(n_knots, 1), np = (size_t)n_plots, cap = (size_t)std::max<int64_t>(capacity, 1);
        DevBuf<eph_plot_request> d_req;
        DevBuf<double> d_kt, d_kp, d_kv, d_t, d_fail;
        DevBuf<float> d_xyz;
        DevBuf<long long> d_cnt;
        DevBuf<int> d_st;
        if ((st = d_req.alloc(np)) || (st = d_kt.alloc(nk)) || (st = d_kp.alloc(3 * nk)) || (st = d_kv.alloc(3 * nk)) ||
            (st = d_t.alloc(np * cap)) || (st = d_xyz.alloc(3 * np * cap)) || (st = d_cnt.alloc(np)) || (st = d_st.alloc(np)) ||
            (st = d_fail.alloc(np)))
            return st;
        EPH_HIP(hipMemcpy(d_req.p, requests, sizeof(eph_plot_request) * np, hipMemcpyHostToDevice));
        if (n_knots) {
            EPH_HIP(hipMemcpy(d_kt.p, knot_t, sizeof(double) * n_knots, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(d_kp.p, knot_pos, sizeof(double) * 3 * n_knots, hipMemcpyHostToDevice));
            EPH_HIP(hipMemcpy(d_kv.p, knot_vel, sizeof(double) * 3 * n_knots, hipMemcpyHostToDevice));
        }
        PlotArgs a{};
        a.n_plots = n_plots; a.n_bodies = e->n_bodies;
        a.bodies = e->bodies.p; a.coeffs = e->coeffs.p; a.ncoef = e->ncoef.p;
        a.req = d_req.p; a.view = *view;
        a.knot_t = d_kt.p; a.knot_pos = d_kp.p; a.knot_vel = d_kv.p;
        a.capacity = capacity; a.out_t = d_t.p; a.out_xyz = d_xyz.p; a.out_count = d_cnt.p; a.out_status = d_st.p;
        a.out_failed_at = d_fail.p;
        hipLaunchKernelGGL(k_plot_points, dim3((unsigned)((n_plots + 63) / 64)), dim3(64), 0, nullptr, a);
        hipError_t he = hipGetLastError();
        if (he != hipSuccess) { set_last_error("k_plot_points", he); return EPH_ERR_HIP; }
        static_assert(sizeof(long long) == sizeof(int64_t), "count type");
        EPH_HIP(hipMemcpy(out_count, d_cnt.p, sizeof(int64_t) * np, hipMemcpyDeviceToHost));
        EPH_HIP(hipMemcpy(out_status, d_st.p, sizeof(int32_t) * np, hipMemcpyDeviceToHost));
        EPH_HIP(hipMemcpy(out_failed_at, d_fail.p, sizeof(double) * np, hipMemcpyDeviceToHost));
        if (capacity > 0) {
            EPH_HIP(hipMemcpy(out_t, d_t.p, sizeof(double) * np * cap, hipMemcpyDeviceToHost));
            EPH_HIP(hipMemcpy(out_xyz, d_xyz.p, sizeof(float) * 3 * np * cap, hipMemcpyDeviceToHost));
        }
        return EPH_OK;
    } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }
}

}  // extern "C"
#pragma GCC visibility pop

// ---- test hooks: the device halves (the extern "C" entry points live in debug_api.cpp, which only the test-hooks library and the
// tuning builds link: csrc/eph_debug.h) ---------------------------------------------------------------------------------------------
namespace eph {
int debug_div_device(int64_t n, const double *a, const double *b, double *fast, double *ieee) {
    try {
        if (n < 0 || (n > 0 && (!a || !b || !fast || !ieee))) return EPH_ERR_BAD_ARGUMENT;
        int st = check_device();
        if (st) return st;
        if (n == 0) return EPH_OK;
        DevBuf<double> da, db, df, di;
        if ((st = da.alloc(n)) || (st = db.alloc(n)) || (st = df.alloc(n)) || (st = di.alloc(n))) return st;
        EPH_HIP(hipMemcpy(da.p, a, sizeof(double) * n, hipMemcpyHostToDevice));
        EPH_HIP(hipMemcpy(db.p, b, sizeof(double) * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_debug_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (long long)n, da.p, db.p,
                           df.p, di.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_last_error("k_debug_div", e); return EPH_ERR_HIP; }
        EPH_HIP(hipMemcpy(fast, df.p, sizeof(double) * n, hipMemcpyDeviceToHost));
        EPH_HIP(hipMemcpy(ieee, di.p, sizeof(double) * n, hipMemcpyDeviceToHost));
        return EPH_OK;
    } catch (...) { return EPH_ERR_HIP; }
}

// raw v_rsq_f64(x) and the h = 0.5 / sqrt(x) that the square root's coupled step leaves (the reciprocal's seed is 8 h^3):
// the two quantities the error-bound note of inv_r3_seeded (pair_term.h) starts from
__global__ void k_debug_rsq(long long n, const double *__restrict__ x, double *__restrict__ y, double *__restrict__ h1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double yy = __builtin_amdgcn_rsq(x[i]);
    const double g = x[i] * yy, h = yy * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    y[i] = yy;
    h1[i] = __builtin_fma(h, r, h);
}
int debug_rsq_device(int64_t n, const double *x, double *rsq, double *h) {
    try {
        if (n < 0 || (n > 0 && (!x || !rsq || !h))) return EPH_ERR_BAD_ARGUMENT;
        int st = check_device();
        if (st) return st;
        if (n == 0) return EPH_OK;
        DevBuf<double> dx, dy, dh;
        if ((st = dx.alloc(n)) || (st = dy.alloc(n)) || (st = dh.alloc(n))) return st;
        EPH_HIP(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_debug_rsq, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (long long)n, dx.p, dy.p, dh.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_last_error("k_debug_rsq", e); return EPH_ERR_HIP; }
        EPH_HIP(hipMemcpy(rsq, dy.p, sizeof(double) * n, hipMemcpyDeviceToHost));
        EPH_HIP(hipMemcpy(h, dh.p, sizeof(double) * n, hipMemcpyDeviceToHost));
        return EPH_OK;
    } catch (...) { return EPH_ERR_HIP; }
}

int debug_pow_device(int64_t n, const double *x, double y, double *out) {
    try {
        if (n < 0 || (n > 0 && (!x || !out))) return EPH_ERR_BAD_ARGUMENT;
        int st = check_device();
        if (st) return st;
        if (n == 0) return EPH_OK;
        DevBuf<double> dx, dout;
        if ((st = dx.alloc(n)) || (st = dout.alloc(n))) return st;
        EPH_HIP(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_debug_pow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (long long)n, dx.p, y, dout.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_last_error("k_debug_pow", e); return EPH_ERR_HIP; }
        EPH_HIP(hipMemcpy(out, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost));
        return EPH_OK;
    } catch (...) { return EPH_ERR_HIP; }
}

}  // namespace eph
