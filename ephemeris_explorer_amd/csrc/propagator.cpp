// propagator.cpp -- NBodyPropagator with the SplineInterpolators solout, on the device.
//
// Mirrors ephemeris/src/propagators/nbody.rs:65-235 (NBodyPropagator), :243-517 (PolyonmialInterpolator,
// SplineInterpolator(s), SplineBound, DirectionalSolout) and the app-side LeastSquaresFit
// (ephemeris_explorer/src/dynamics/celestial.rs:19-186).
//
// Division of labour: the integrator kernels append the sampled positions to per-body device logs; after a
// batch one k_lsq_fit launch turns every complete 9-sample window into a Polynomial. The host keeps exactly
// the counters the reference keeps per SplineInterpolator (last_sample_time, window fill) and derives from
// them -- with the reference's f64 operations -- spline starts, bounds, time() and has_reached().
#include <algorithm>
#include <cmath>
#include <limits>

#include "host.h"

namespace eph {

namespace {
// After how many `last_sample_time += delta` does `last_sample_time == sample_period` hold (nbody.rs:389-391)?
// The test is an exact f64 equality on an accumulated sum; if the sum steps over the period the reference never
// samples that body again. 0 = never.
uint32_t trigger_period(double delta, double sample_period, uint32_t count_hint) {
    double s = 0.0;
    const uint64_t cap = (uint64_t)count_hint * 2 + 64;
    for (uint64_t k = 1; k <= cap; ++k) {
        s += delta;
        if (s == sample_period) return (uint32_t)k;
        if (s > sample_period) return 0;
    }
    return 0;
}
double accumulate(double start, double delta, uint64_t times) {
    double s = start;
    for (uint64_t k = 0; k < times; ++k) s += delta;
    return s;
}
// D::offset / D::distance   propagators/mod.rs:49-56,84-91
inline double dir_offset(int d, double to, double duration) { return d > 0 ? to + duration : to - duration; }
inline double dir_distance(int d, double from, double to) { return d > 0 ? to - from : from - to; }
constexpr uint64_t kSampleBudget = 4u << 20;   // samples in the device logs (96 MB)
}  // namespace

Solution NBodyPropagator::new_solution() const {   // Solout::new_solution  nbody.rs:454-469
    Solution so;
    so.splines.resize(interp_.size());
    const double t = integ_->time();
    for (size_t b = 0; b < interp_.size(); ++b) {
        so.splines[b].start = dir_offset(direction_, t, -interp_[b].time());
        so.splines[b].interval = interp_[b].sample_period * (double)kDiv;
    }
    return so;
}

int NBodyPropagator::create(int n, const double *pos, const double *vel, const double *mu, double t0, double dt,
                            int direction, const char *method, const uint32_t *count, const uint32_t *degree,
                            std::unique_ptr<NBodyPropagator> *out) {
    if (!out || n < 0 || (n > 0 && (!count || !degree)) || !(dt > 0.0) || !std::isfinite(dt) ||
        (direction != EPH_FORWARD && direction != EPH_BACKWARD))
        return EPH_ERR_BAD_ARGUMENT;
    for (int b = 0; b < n; ++b)
        if (degree[b] > (uint32_t)(kDiv - 1) || count[b] == 0) return EPH_ERR_UNSUPPORTED;
    std::unique_ptr<NBodyPropagator> p(new NBodyPropagator());
    p->direction_ = direction;
    p->delta_ = dt;                                   // SplineInterpolators::new(delta, ..)  celestial.rs:183
    const double h = direction > 0 ? std::fabs(dt) : -std::fabs(dt);   // signed_delta  mod.rs:44-46,78-82
    int st = NBodyIntegration::create(n, pos, vel, mu, t0, h, method, &p->integ_);
    if (st) return st;
    p->interp_.resize(n);
    uint64_t total = 0;
    double rate = 0.0;
    for (int b = 0; b < n; ++b) {
        SplineInterpolator &it = p->interp_[b];
        it.sample_period = dt * (double)count[b];     // load/mod.rs:325
        it.degree = degree[b];
        it.period_steps = trigger_period(dt, it.sample_period, count[b]);
        if (it.period_steps) rate += 1.0 / it.period_steps;
    }
    // steps per batch so that the sample logs fit the budget
    int64_t kmax = 1 << 16;
    if (rate * kmax + 10.0 * n > (double)kSampleBudget) kmax = (int64_t)(((double)kSampleBudget - 10.0 * n) / rate);
    if (kmax < 1) return EPH_ERR_OUT_OF_MEMORY;
    p->kmax_ = kmax;
    p->log_off_.resize(n);
    p->log_cap_.resize(n);
    for (int b = 0; b < n; ++b) {
        const uint32_t m = p->interp_[b].period_steps;
        p->log_cap_[b] = 10 + (m ? (uint64_t)(kmax / m) + 1 : 0);
        p->log_off_[b] = total;
        total += p->log_cap_[b];
    }
    if ((st = p->log_.alloc(total * 3))) return st;
    if ((st = p->d_period_.alloc(n)) || (st = p->d_phase_.alloc(n)) || (st = p->d_offset_.alloc(n))) return st;
    if (n > 0) {
        // PolyonmialInterpolator::new(&position): the window starts with the initial position  nbody.rs:251-259
        std::vector<double> first(total * 3, 0.0);
        std::vector<uint32_t> per(n);
        for (int b = 0; b < n; ++b) {
            for (int c = 0; c < 3; ++c) first[p->log_off_[b] * 3 + c] = pos[b * 3 + c];
            per[b] = p->interp_[b].period_steps;
        }
        EPH_HIP(hipMemcpy(p->log_.p, first.data(), sizeof(double) * first.size(), hipMemcpyHostToDevice));
        EPH_HIP(hipMemcpy(p->d_period_.p, per.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    }
    p->solution_ = p->new_solution();                 // with_solout  lib.rs:441-451
    *out = std::move(p);
    return EPH_OK;
}

int NBodyPropagator::clone(std::unique_ptr<NBodyPropagator> *out) {
    if (failed_) return failed_;
    if (deferred_) { const int stf = flush(); if (stf) return stf; }
    std::unique_ptr<NBodyPropagator> p(new NBodyPropagator());
    int st = integ_->clone(&p->integ_);
    if (st) return st;
    p->direction_ = direction_;
    p->delta_ = delta_;
    p->kmax_ = kmax_;
    p->interp_ = interp_;
    p->solution_ = solution_;
    p->log_off_ = log_off_;
    p->log_cap_ = log_cap_;
    const size_t n = interp_.size();
    if ((st = p->log_.alloc(log_.count)) || (st = p->d_period_.alloc(n)) || (st = p->d_phase_.alloc(n)) ||
        (st = p->d_offset_.alloc(n)))
        return st;
    EPH_HIP(hipMemcpy(p->log_.p, log_.p, sizeof(double) * log_.count, hipMemcpyDeviceToDevice));
    EPH_HIP(hipMemcpy(p->d_period_.p, d_period_.p, sizeof(uint32_t) * d_period_.count, hipMemcpyDeviceToDevice));
    if (pend_count_) {                                // device-resident polynomials travel with the clone
        if ((st = p->pend_co_.alloc(pend_count_ * kDiv * 3)) || (st = p->pend_nc_.alloc(pend_count_))) return st;
        EPH_HIP(hipStreamSynchronize(integ_->stream()));
        EPH_HIP(hipMemcpy(p->pend_co_.p, pend_co_.p, sizeof(double) * pend_count_ * kDiv * 3, hipMemcpyDeviceToDevice));
        EPH_HIP(hipMemcpy(p->pend_nc_.p, pend_nc_.p, sizeof(int32_t) * pend_count_, hipMemcpyDeviceToDevice));
        p->pend_count_ = p->pend_cap_ = pend_count_;
        p->pend_batches_ = pend_batches_;
    }
    *out = std::move(p);
    return EPH_OK;
}

// room for `extra` more windows behind the pending ones (grow-only; contents kept). Past kPendMaxWindows, or when
// the device cannot hold a larger list, the pending windows go to the host splines first and the list starts over.
int NBodyPropagator::reserve_pending(size_t extra, hipStream_t s) {
    if (pend_count_ + extra <= pend_cap_) return EPH_OK;
    if (pend_count_ && pend_count_ + extra > kPendMaxWindows) {
        const int st = materialize();
        if (st) return st;
        if (extra <= pend_cap_) return EPH_OK;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        const size_t need = pend_count_ + extra;
        const size_t cap = attempt == 0 ? std::max(need + need / 2, (size_t)4096) : need;
        DevBuf<double> co;
        DevBuf<int32_t> nc;
        int st;
        if ((st = co.alloc(cap * kDiv * 3)) || (st = nc.alloc(cap))) {
            if (st != EPH_ERR_OUT_OF_MEMORY || attempt == 1) return st;
            (void)hipGetLastError();
            if ((st = materialize())) return st;          // frees nothing by itself, but the copy below disappears
            pend_co_.release();
            pend_nc_.release();
            pend_cap_ = 0;
            continue;
        }
        if (pend_count_) {
            EPH_HIP(hipMemcpyAsync(co.p, pend_co_.p, sizeof(double) * pend_count_ * kDiv * 3, hipMemcpyDeviceToDevice, s));
            EPH_HIP(hipMemcpyAsync(nc.p, pend_nc_.p, sizeof(int32_t) * pend_count_, hipMemcpyDeviceToDevice, s));
            EPH_HIP(hipStreamSynchronize(s));
        }
        std::swap(pend_co_.p, co.p); std::swap(pend_co_.count, co.count);
        std::swap(pend_nc_.p, nc.p); std::swap(pend_nc_.count, nc.count);
        pend_cap_ = cap;
        return EPH_OK;
    }
    return EPH_ERR_OUT_OF_MEMORY;
}

// download the device-resident polynomials and perform the pushes the bounds already account for
int NBodyPropagator::materialize() {
    if (pend_count_ == 0) return EPH_OK;
    hipStream_t s = integ_->stream();
    EPH_HIP(hipSetDevice(integ_->device()));
    std::vector<double> co(pend_count_ * kDiv * 3);
    std::vector<int32_t> nc(pend_count_);
    EPH_HIP(hipMemcpyAsync(co.data(), pend_co_.p, sizeof(double) * co.size(), hipMemcpyDeviceToHost, s));
    EPH_HIP(hipMemcpyAsync(nc.data(), pend_nc_.p, sizeof(int32_t) * nc.size(), hipMemcpyDeviceToHost, s));
    EPH_HIP(hipStreamSynchronize(s));
    size_t q = 0;
    for (const std::vector<uint32_t> &nwin : pend_batches_)
        for (size_t b = 0; b < nwin.size(); ++b) {
            UniformSpline &traj = solution_.splines[b];
            for (uint32_t w = 0; w < nwin[b]; ++w, ++q) {
                Polynomial poly;
                poly.ncoef = nc[q];
                std::copy(co.begin() + q * kDiv * 3, co.begin() + (q + 1) * kDiv * 3, &poly.c[0][0]);
                // push_at_bound: the start of a backward spline already moved when the window was fitted
                if (direction_ > 0) traj.polynomials.push_back(poly); else traj.polynomials.push_front(poly);
                traj.ghost -= 1;
            }
        }
    pend_count_ = 0;
    pend_batches_.clear();
    return EPH_OK;
}

// One device batch of k integrator steps + solout   (k x [Integration::advance -> Solout::solout], lib.rs:379-391):
// batch_begin puts the sampling schedule of the batch on the device, the integrator advances, fit_and_push does the rest.
int NBodyPropagator::batch_begin() {
    if (failed_) return failed_;
    const int n = (int)interp_.size();
    NBodyIntegration &ig = *integ_;
    hipStream_t s = ig.stream();
    EPH_HIP(hipSetDevice(ig.device()));
    std::vector<uint32_t> phase(n);
    std::vector<uint64_t> offset(n);
    for (int b = 0; b < n; ++b) {
        phase[b] = interp_[b].phase;
        offset[b] = log_off_[b] + interp_[b].len;     // first free slot of the window log
    }
    if (n > 0) {
        EPH_HIP(hipMemcpyAsync(d_phase_.p, phase.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        EPH_HIP(hipMemcpyAsync(d_offset_.p, offset.data(), sizeof(uint64_t) * n, hipMemcpyHostToDevice, s));
        EPH_HIP(hipStreamSynchronize(s));   // host vectors go out of scope; the copies are small
    }
    SampleArgs sa{d_period_.p, d_phase_.p, d_offset_.p, log_.p};
    ig.set_sampling(sa);
    return EPH_OK;
}
int NBodyPropagator::run_batch(int64_t k) {
    int st = batch_begin();
    if (st) return st;
    NBodyIntegration &ig = *integ_;
    int64_t done = 0;
    const int st_adv = ig.advance(k, &done);
    // From here on the integrator has moved: a failure between now and the push of the fitted polynomials would
    // leave counters, sample log and solution out of step with it, so it is made sticky -- every later step /
    // step_to / take_solution / clone returns the same error instead of a spline with a missing segment.
    const int st_fit = fit_and_push(done, ig.stream());
    if (st_fit) return failed_ = st_fit;
    return st_adv;
}

// step_n(k) on several propagators at once. Start-up steps (and anything else advance_many would not take) run per
// propagator as usual; the steady-state steps of all of them share launches: every propagator still has `rem` steps to
// go, the gang advances min(rem, kmax) of them in ONE k_lm_small launch with a workgroup per system, then each fits and
// pushes its own windows. Results are those of step_n(k) on each.
int NBodyPropagator::step_n_many(NBodyPropagator *const *ps, int count, int64_t k) {
    if (count < 0 || (count > 0 && !ps) || k < 0) return EPH_ERR_BAD_ARGUMENT;
    std::vector<int64_t> rem((size_t)count, k);
    for (int i = 0; i < count; ++i) {
        if (!ps[i]) return EPH_ERR_BAD_ARGUMENT;
        for (int j = 0; j < i; ++j)
            if (ps[j] == ps[i]) return EPH_ERR_BAD_ARGUMENT;
        NBodyPropagator &p = *ps[i];
        if (p.failed_) return p.failed_;
        if (p.deferred_) { const int st = p.flush(); if (st) return st; }
        while (rem[i] > 0 && !p.integ_->started()) {                 // start-up: one macro step at a time, as step_n does
            const int st = p.run_batch(1);
            if (st) return st;
            rem[i] -= 1;
        }
    }
    std::vector<NBodyPropagator *> act;
    std::vector<NBodyIntegration *> igs;
    for (;;) {
        act.clear();
        int64_t chunk = INT64_MAX;
        for (int i = 0; i < count; ++i)
            if (rem[i] > 0) { act.push_back(ps[i]); chunk = std::min(chunk, std::min(rem[i], ps[i]->kmax_)); }
        if (act.empty()) return EPH_OK;
        bool gang = act.size() > 1;
        for (NBodyPropagator *p : act) gang = gang && p->integ_->gang_ready(chunk);
        if (!gang) {                                                 // somebody cannot: everybody finishes alone
            for (int i = 0; i < count; ++i)
                if (rem[i] > 0) { const int st = ps[i]->step_n(rem[i]); if (st) return st; }
            return EPH_OK;
        }
        igs.clear();
        for (NBodyPropagator *p : act) {
            const int st = p->batch_begin();
            if (st) return st;
            igs.push_back(p->integ_.get());
        }
        const int st_adv = NBodyIntegration::advance_many(igs.data(), (int)igs.size(), chunk);
        if (st_adv) {                                                // (gang_ready held for all: not a StepError)
            for (NBodyPropagator *p : act) p->failed_ = st_adv;
            return st_adv;
        }
        for (NBodyPropagator *p : act) {
            p->integ_->set_sampling(SampleArgs{});
            const int st = p->fit_and_push(chunk, p->integ_->stream());
            if (st) return p->failed_ = st;
        }
        for (int i = 0; i < count; ++i)
            if (rem[i] > 0) rem[i] -= chunk;
    }
}

// the per-step bookkeeping of SplineInterpolators::solout_with (nbody.rs:371-400) replayed for `done` steps, the
// least-squares fit of every window they completed, and the push of the polynomials into the solution
int NBodyPropagator::fit_and_push(int64_t done, hipStream_t s) {
    const int n = (int)interp_.size();
    NBodyIntegration &ig = *integ_;
    std::vector<uint64_t> first;
    std::vector<uint8_t> deg;
    std::vector<uint32_t> nwin(n, 0), carry_src(n, 0), carry_cnt(n, 0);
    for (int b = 0; b < n; ++b) {
        SplineInterpolator &it = interp_[b];
        const uint32_t m = it.period_steps;
        if (m == 0) {
            it.last_sample_time = accumulate(it.last_sample_time, delta_, (uint64_t)done);
            continue;
        }
        const uint64_t t = (uint64_t)it.phase + (uint64_t)done;
        const uint64_t fresh = t / m;
        it.phase = (uint32_t)(t % m);
        it.last_sample_time = accumulate(0.0, delta_, it.phase);
        const uint64_t have = it.len + fresh;          // samples now in the log region (>= 1)
        const uint64_t w = (have - 1) / kDiv;          // complete windows: 9 samples, sharing end points
        nwin[b] = (uint32_t)w;
        for (uint64_t q = 0; q < w; ++q) {
            first.push_back(log_off_[b] + q * kDiv);
            deg.push_back((uint8_t)it.degree);
        }
        it.len = (uint32_t)(have - w * kDiv);           // finish(): last sample becomes the first of the next window
        if (w > 0) {
            carry_src[b] = (uint32_t)(w * kDiv);
            carry_cnt[b] = it.len;
        }
    }
    const int64_t W = (int64_t)first.size();
    if (W == 0) return EPH_OK;
    // Sharded system (eph_prop_shard): this rank samples and fits only the bodies it owns -- the windows of
    // bodies [lo, hi) are the contiguous range [qlo, qhi) of the window list -- and the fitted polynomials
    // are all-gathered (equal slices of world x wmax windows, 25 doubles each: 24 coefficients + ncoef) so
    // that every rank holds the whole Vec<UniformSpline>. The window bookkeeping above is identical on
    // every rank, so all ranks agree on the ranges.
    const bool sharded = ig.sharded();
    const int world = ig.shard_world(), rank = ig.shard_rank(), slice = ig.shard_slice();
    std::vector<int64_t> qstart(n + 1, 0);
    for (int b = 0; b < n; ++b) qstart[b + 1] = qstart[b] + nwin[b];
    auto body_lo = [&](int r) { return std::min<int64_t>((int64_t)r * slice, n); };
    const int64_t qlo = sharded ? qstart[body_lo(rank)] : 0, qhi = sharded ? qstart[body_lo(rank + 1)] : W;
    int64_t wmax = 0;
    for (int r = 0; r < world && sharded; ++r) wmax = std::max(wmax, qstart[body_lo(r + 1)] - qstart[body_lo(r)]);
    int st;
    if (!sharded) {
        // Unsharded: fit straight into the device-resident pending list; only the bounds move on the host
        // (UniformSpline::push_front's `start -= interval`, one f64 subtraction per window like the reference).
        if ((st = d_first_.reserve(W)) || (st = d_deg_.reserve(W)) || (st = d_src_.reserve(n)) || (st = d_cnt_.reserve(n)) ||
            (st = d_region_.reserve(n)) || (st = reserve_pending((size_t)W, s)))
            return st;
        EPH_HIP(hipMemcpyAsync(d_first_.p, first.data(), sizeof(uint64_t) * W, hipMemcpyHostToDevice, s));
        EPH_HIP(hipMemcpyAsync(d_deg_.p, deg.data(), sizeof(uint8_t) * W, hipMemcpyHostToDevice, s));
        if ((st = launch_lsq_fit(s, W, d_first_.p, d_deg_.p, direction_ < 0, log_.p, pend_co_.p + pend_count_ * kDiv * 3,
                                 pend_nc_.p + pend_count_)))
            return st;
        EPH_HIP(hipMemcpyAsync(d_src_.p, carry_src.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        EPH_HIP(hipMemcpyAsync(d_cnt_.p, carry_cnt.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        EPH_HIP(hipMemcpyAsync(d_region_.p, log_off_.data(), sizeof(uint64_t) * n, hipMemcpyHostToDevice, s));
        if ((st = launch_carry(s, n, d_region_.p, d_src_.p, d_cnt_.p, log_.p))) return st;
        EPH_HIP(hipStreamSynchronize(s));            // the host vectors above go out of scope
        for (int b = 0; b < n; ++b) {
            UniformSpline &traj = solution_.splines[b];
            traj.ghost += nwin[b];
            if (direction_ < 0)
                for (uint32_t w = 0; w < nwin[b]; ++w) traj.start -= traj.interval;
        }
        pend_count_ += (size_t)W;
        pend_batches_.push_back(std::move(nwin));
        return EPH_OK;
    }
    // scratch is grow-only and lives with the propagator: a steady run allocates nothing per batch
    if ((st = d_first_.reserve(W)) || (st = d_deg_.reserve(W)) || (st = d_co_.reserve((size_t)W * kDiv * 3)) ||
        (st = d_nc_.reserve(W)) || (st = d_src_.reserve(n)) || (st = d_cnt_.reserve(n)) || (st = d_region_.reserve(n)))
        return st;
    EPH_HIP(hipMemcpyAsync(d_first_.p, first.data(), sizeof(uint64_t) * W, hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_deg_.p, deg.data(), sizeof(uint8_t) * W, hipMemcpyHostToDevice, s));
    if (qhi > qlo &&
        (st = launch_lsq_fit(s, qhi - qlo, d_first_.p + qlo, d_deg_.p + qlo, direction_ < 0, log_.p,
                             d_co_.p + qlo * kDiv * 3, d_nc_.p + qlo)))
        return st;
    std::vector<double> co((size_t)W * kDiv * 3), all;
    std::vector<int32_t> nc(W);
    const size_t rec = (size_t)kDiv * 3 + 1;                      // doubles per window in the exchange buffer
    const size_t slice_d = (size_t)std::max<int64_t>(wmax, 1) * rec;
    if (sharded) {
        // device-side: pack this rank's windows into its slice of the exchange buffer, all-gather in place,
        // one copy of the whole buffer to the host (which owns the Vec<UniformSpline>)
        if ((st = d_all_.reserve(slice_d * world))) return st;
        if ((st = launch_pack_records(s, qhi - qlo, d_co_.p + qlo * kDiv * 3, d_nc_.p + qlo,
                                      d_all_.p + (size_t)rank * slice_d)))
            return st;
        if ((st = ig.gather_buffer(d_all_.p, sizeof(double) * slice_d))) return st;
        all.resize(slice_d * world);
        EPH_HIP(hipMemcpyAsync(all.data(), d_all_.p, sizeof(double) * all.size(), hipMemcpyDeviceToHost, s));
    } else {
        EPH_HIP(hipMemcpyAsync(co.data(), d_co_.p, sizeof(double) * co.size(), hipMemcpyDeviceToHost, s));
        EPH_HIP(hipMemcpyAsync(nc.data(), d_nc_.p, sizeof(int32_t) * W, hipMemcpyDeviceToHost, s));
    }
    EPH_HIP(hipMemcpyAsync(d_src_.p, carry_src.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_cnt_.p, carry_cnt.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_region_.p, log_off_.data(), sizeof(uint64_t) * n, hipMemcpyHostToDevice, s));
    if ((st = launch_carry(s, n, d_region_.p, d_src_.p, d_cnt_.p, log_.p))) return st;
    EPH_HIP(hipStreamSynchronize(s));
    if (sharded) {
        // a peer that never delivered its records (peer.hip: bounded wait): nothing gathered may be unpacked
        if ((st = integ_->exchange_error())) return st;
        for (int r = 0; r < world; ++r) {
            const int64_t a0 = qstart[body_lo(r)], a1 = qstart[body_lo(r + 1)];
            for (int64_t q = a0; q < a1; ++q) {
                const double *src = all.data() + (size_t)r * slice_d + (size_t)(q - a0) * rec;
                std::copy(src, src + kDiv * 3, co.begin() + q * kDiv * 3);
                nc[q] = (int32_t)src[kDiv * 3];
                if (!(src[kDiv * 3] >= 0.0 && src[kDiv * 3] <= (double)kDiv)) {       // not a record a rank packed
                    set_last_error_text("sharded propagator: a gathered polynomial record carries ncoef outside [0, 8]");
                    return EPH_ERR_COMM;
                }
            }
        }
    }
    int64_t q = 0;
    for (int b = 0; b < n; ++b) {
        UniformSpline &traj = solution_.splines[b];
        for (uint32_t w = 0; w < nwin[b]; ++w, ++q) {
            Polynomial poly;
            poly.ncoef = nc[q];
            std::copy(co.begin() + q * kDiv * 3, co.begin() + (q + 1) * kDiv * 3, &poly.c[0][0]);
            if (direction_ > 0) traj.push_back(poly); else traj.push_front(poly);   // push_at_bound
        }
    }
    return EPH_OK;
}

// IncrementalPropagator::step, lazily (see host.h). Returns exactly what the step will return.
int NBodyPropagator::step_deferred() {
    if (failed_) return failed_;
    NBodyIntegration &ig = *integ_;
    if (!ig.started() || ig.sharded()) return step_n(1);          // start-up and sharded runs execute at once
    const size_t n = interp_.size();
    if (deferred_ == 0) {                                            // open the queue: shadows = the current bookkeeping
        sh_time_ = ig.time();
        sh_phase_.resize(n); sh_len_.resize(n); sh_start_.resize(n); sh_npoly_.resize(n);
        for (size_t b = 0; b < n; ++b) {
            sh_phase_[b] = interp_[b].phase;
            sh_len_[b] = interp_[b].len;
            sh_start_[b] = solution_.splines[b].start;
            sh_npoly_[b] = solution_.splines[b].len();
        }
    }
    // LinearMultistepIntegrator::advance's checks (multistep/mod.rs:201-207) with the time the queued steps lead to: a
    // step that would fail is not queued -- the queue runs and the step executes, returning the StepError itself
    if (sh_time_ >= ig.bound() || sh_time_ + ig.step_size() == sh_time_) {
        const int st = flush();
        return st ? st : step_n(1);
    }
    sh_time_ = sh_time_ + ig.step_size();
    for (size_t b = 0; b < n; ++b) {                                 // SplineInterpolators::solout_with, one step  nbody.rs:371-400
        const uint32_t m = interp_[b].period_steps;
        if (m == 0) continue;                                        // (last_sample_time is rebuilt when the queue runs)
        if (++sh_phase_[b] < m) continue;
        sh_phase_[b] = 0;                                            // a sample
        if (++sh_len_[b] <= (uint32_t)kDiv) continue;
        sh_len_[b] = 1;                                              // ninth sample: the window is fitted and pushed
        sh_npoly_[b] += 1;
        if (direction_ < 0) sh_start_[b] -= solution_.splines[b].interval;   // push_front
    }
    if (++deferred_ >= kDeferMax) return flush();
    return EPH_OK;
}

int NBodyPropagator::flush() {
    if (failed_) return failed_;
    if (deferred_ == 0) return EPH_OK;
    const int64_t k = deferred_;
    deferred_ = 0;
    const int st = step_n(k);                                        // none of these steps can return a StepError (checked when queued)
    if (st) return failed_ = st;
#ifndef NDEBUG
    for (size_t b = 0; b < interp_.size(); ++b)
        if (interp_[b].period_steps && (interp_[b].phase != sh_phase_[b] || interp_[b].len != sh_len_[b] ||
                                        solution_.splines[b].len() != sh_npoly_[b] || solution_.splines[b].start != sh_start_[b])) {
            set_last_error_text("deferred-step bookkeeping diverged from the executed batch");
            return failed_ = EPH_ERR_HIP;
        }
#endif
    return EPH_OK;
}

int NBodyPropagator::step_n(int64_t k) {
    if (deferred_) { const int st = flush(); if (st) return st; }
    while (k > 0) {
        // start-up steps of the multistep method go one at a time (each is many small launches anyway)
        const int64_t chunk = integ_->started() ? std::min<int64_t>(k, kmax_) : 1;
        const int st = run_batch(chunk);
        if (st) return st;   // NBodyPropagatorError::Integration(e)
        k -= chunk;
    }
    return EPH_OK;
}

double NBodyPropagator::time() const {   // DirectionalSolout::solution_time  nbody.rs:501-508
    const auto &sp = solution_.splines;
    if (sp.empty()) return dir_offset(direction_, 0.0, -std::numeric_limits<double>::max());
    double best = bound_at(0);
    for (size_t b = 1; b < sp.size(); ++b) {
        const double x = bound_at(b);
        if (0.0 < dir_distance(direction_, x, best)) best = x;   // min_by(D::cmp), first minimum wins
    }
    return best;
}

bool NBodyPropagator::has_reached(double t) const {   // nbody.rs:510-516
    for (size_t b = 0; b < solution_.splines.size(); ++b)
        if (0.0 < dir_distance(direction_, bound_at(b), t)) return false;   // !D::cmp(&bound, &time).is_ge()
    return true;
}

// smallest number of further step() calls after which has_reached(t) holds (capped)
int64_t NBodyPropagator::steps_until_reached(double t, int64_t cap) const {
    int64_t need = 0;
    for (size_t b = 0; b < interp_.size(); ++b) {
        const SplineInterpolator &it = interp_[b];
        UniformSpline s;   // bound arithmetic only
        s.start = solution_.splines[b].start;
        s.interval = solution_.splines[b].interval;
        uint64_t npoly = solution_.splines[b].len();
        uint64_t extra = 0;
        auto bound = [&]() {
            return direction_ > 0 ? s.start + s.interval * (double)npoly : s.start;
        };
        while (0.0 < dir_distance(direction_, bound(), t)) {
            if (it.period_steps == 0) return cap;
            ++extra;
            ++npoly;
            if (direction_ < 0) s.start -= s.interval;   // push_front
            if ((int64_t)(extra * kDiv * it.period_steps) > cap + (int64_t)kDiv * it.period_steps) return cap;
        }
        if (extra == 0) continue;
        const uint64_t samples = extra * kDiv - (it.len - 1);
        const int64_t steps = (int64_t)(samples * it.period_steps) - (int64_t)it.phase;
        need = std::max(need, steps);
    }
    return std::min(need, cap);
}

int NBodyPropagator::step_to(double t) {   // IncrementalPropagator::step_to  ephemeris/src/lib.rs:49-60
    if (failed_) return failed_;
    if (deferred_) { const int st = flush(); if (st) return st; }
    for (;;) {
        if (has_reached(t)) return EPH_OK;
        int64_t k = integ_->started() ? steps_until_reached(t, kmax_) : 1;
        if (k < 1) k = 1;
        const int st = step_n(k);
        if (st) return st;
    }
}

int NBodyPropagator::take_solution(std::unique_ptr<Solution> *out) {   // nbody.rs:182-189
    if (failed_) return failed_;
    if (deferred_) { const int st = flush(); if (st) return st; }
    const int stm = materialize();
    if (stm) return failed_ = stm;
    std::unique_ptr<Solution> old(new Solution(std::move(solution_)));
    solution_ = new_solution();
    *out = std::move(old);
    return EPH_OK;
}

// ------------------------------------------------------------------------------------------------------
int spline_eval_device(const UniformSpline &sp, int64_t m, const double *at, double *pos, double *vel,
                       uint8_t *inside) {
    if (m < 0 || (m > 0 && (!at || !pos || !inside))) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    if (m == 0) return EPH_OK;
    const int64_t np = (int64_t)sp.polynomials.size();
    std::vector<double> co((size_t)std::max<int64_t>(np, 1) * kDiv * 3, 0.0);
    std::vector<int32_t> nc(std::max<int64_t>(np, 1), 0);
    for (int64_t p = 0; p < np; ++p) {
        nc[p] = sp.polynomials[p].ncoef;
        std::copy(&sp.polynomials[p].c[0][0], &sp.polynomials[p].c[0][0] + kDiv * 3, co.begin() + p * kDiv * 3);
    }
    DevBuf<double> d_co, d_at, d_pos, d_vel;
    DevBuf<int32_t> d_nc;
    DevBuf<uint8_t> d_in;
    if ((st = d_co.alloc(co.size())) || (st = d_nc.alloc(nc.size())) || (st = d_at.alloc(m)) ||
        (st = d_pos.alloc((size_t)m * 3)) || (st = d_vel.alloc((size_t)m * 3)) || (st = d_in.alloc(m)))
        return st;
    hipStream_t s = nullptr;
    EPH_HIP(hipMemcpyAsync(d_co.p, co.data(), sizeof(double) * co.size(), hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_nc.p, nc.data(), sizeof(int32_t) * nc.size(), hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_at.p, at, sizeof(double) * m, hipMemcpyHostToDevice, s));
    if ((st = launch_spline_eval(s, m, d_at.p, sp.start, sp.interval, np, d_co.p, d_nc.p, d_pos.p,
                                 vel ? d_vel.p : nullptr, d_in.p)))
        return st;
    EPH_HIP(hipMemcpyAsync(pos, d_pos.p, sizeof(double) * m * 3, hipMemcpyDeviceToHost, s));
    if (vel) EPH_HIP(hipMemcpyAsync(vel, d_vel.p, sizeof(double) * m * 3, hipMemcpyDeviceToHost, s));
    EPH_HIP(hipMemcpyAsync(inside, d_in.p, m, hipMemcpyDeviceToHost, s));
    EPH_HIP(hipStreamSynchronize(s));
    return EPH_OK;
}

int least_squares_fit_device(int degree, int backward, int64_t nwin, const double *samples, double *coeffs,
                             int32_t *ncoef) {
    if (nwin < 0 || degree < 0 || degree > kDiv - 1 || (nwin > 0 && (!samples || !coeffs || !ncoef)))
        return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    if (nwin == 0) return EPH_OK;
    std::vector<uint64_t> first(nwin);
    std::vector<uint8_t> deg(nwin, (uint8_t)degree);
    for (int64_t w = 0; w < nwin; ++w) first[w] = (uint64_t)w * (kDiv + 1);
    DevBuf<uint64_t> d_first;
    DevBuf<uint8_t> d_deg;
    DevBuf<double> d_log, d_co;
    DevBuf<int32_t> d_nc;
    if ((st = d_first.alloc(nwin)) || (st = d_deg.alloc(nwin)) || (st = d_log.alloc((size_t)nwin * (kDiv + 1) * 3)) ||
        (st = d_co.alloc((size_t)nwin * kDiv * 3)) || (st = d_nc.alloc(nwin)))
        return st;
    hipStream_t s = nullptr;
    EPH_HIP(hipMemcpyAsync(d_first.p, first.data(), sizeof(uint64_t) * nwin, hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_deg.p, deg.data(), nwin, hipMemcpyHostToDevice, s));
    EPH_HIP(hipMemcpyAsync(d_log.p, samples, sizeof(double) * nwin * (kDiv + 1) * 3, hipMemcpyHostToDevice, s));
    if ((st = launch_lsq_fit(s, nwin, d_first.p, d_deg.p, backward, d_log.p, d_co.p, d_nc.p))) return st;
    EPH_HIP(hipMemcpyAsync(coeffs, d_co.p, sizeof(double) * nwin * kDiv * 3, hipMemcpyDeviceToHost, s));
    EPH_HIP(hipMemcpyAsync(ncoef, d_nc.p, sizeof(int32_t) * nwin, hipMemcpyDeviceToHost, s));
    EPH_HIP(hipStreamSynchronize(s));
    return EPH_OK;
}

}  // namespace eph
