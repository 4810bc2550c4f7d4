// nbody.cpp -- NBodyIntegration: `M::new(FixedMethodParams{h}).integrate(NBodyProblem{..})` on the device.
//
// Mirrors, call for call:
//   LinearMultistepIntegrator::advance          integration/src/multistep/mod.rs:194-224
//   ELM2::{advance, advance_with}               integration/src/multistep/second_order/mod.rs:90-153
//   SubstepperIntegrator::advance               integration/src/multistep/mod.rs:97-108
//   FixedRungeKuttaIntegrator::advance + SRKN   integration/src/runge_kutta/mod.rs:106-126, nystrom/symplectic.rs:69-102
// The host replays the reference's scalar bookkeeping (time, bound, step counters) with the reference's f64
// operations; all vector arithmetic runs in the HIP kernels of step_wg.hip / step_wave.hip / step_small.hip / fast.hip / solout.hip.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <mutex>

#include "host.h"

namespace eph {

NBodyIntegration::~NBodyIntegration() {
    if (stream_) {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(stream_);
        if (ev0_) (void)hipEventDestroy(ev0_);
        if (ev1_) (void)hipEventDestroy(ev1_);
        for (auto *v : {&ev_pending_, &ev_free_})
            for (auto &e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (gang_ev_) (void)hipEventDestroy(gang_ev_);
        if (gang_copy_ev_) { (void)hipEventSynchronize(gang_copy_ev_); (void)hipEventDestroy(gang_copy_ev_); }
        if (gang_host_) (void)hipHostFree(gang_host_);
        (void)hipStreamDestroy(stream_);
    }
}

int NBodyIntegration::alloc_buffers() {
    npad_ = ((n_ + 63) / 64) * 64;
    if (npad_ < 64) npad_ = 64;
    int st;
    for (int k = 0; k < 2; ++k)
        if ((st = P_[k].alloc(npad_))) return st;
    if ((st = Y_.alloc((size_t)L_ * 3 * npad_))) return st;
    if ((st = A_.alloc((size_t)L_ * 3 * npad_))) return st;
    if ((st = V_.alloc((size_t)3 * npad_))) return st;
    if ((st = ASR_.alloc((size_t)3 * npad_))) return st;
    if ((st = mu_.alloc(npad_))) return st;
    if ((st = stage_.alloc((size_t)3 * npad_))) return st;
    EPH_HIP(hipMemsetAsync(P_[0].p, 0, sizeof(Body4) * npad_, stream_));
    EPH_HIP(hipMemsetAsync(P_[1].p, 0, sizeof(Body4) * npad_, stream_));
    EPH_HIP(hipMemsetAsync(Y_.p, 0, sizeof(double) * Y_.count, stream_));
    EPH_HIP(hipMemsetAsync(A_.p, 0, sizeof(double) * A_.count, stream_));
    EPH_HIP(hipMemsetAsync(V_.p, 0, sizeof(double) * V_.count, stream_));
    EPH_HIP(hipMemsetAsync(ASR_.p, 0, sizeof(double) * ASR_.count, stream_));
    EPH_HIP(hipMemsetAsync(mu_.p, 0, sizeof(double) * mu_.count, stream_));
    return EPH_OK;
}

int NBodyIntegration::create(int n, const double *pos, const double *vel, const double *mu, double t0, double h,
                             const char *method, std::unique_ptr<NBodyIntegration> *out) {
    if (n < 0 || !method || !out || (n > 0 && (!pos || !vel || !mu))) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    std::unique_ptr<NBodyIntegration> o(new NBodyIntegration());
    o->n_ = n;
    if (find_elm2(method, &o->lm_)) {
        // LinearMultistep::new + Substepper::<4, BlanesMoan6B>::new   multistep/mod.rs:54-57,120-128; methods.rs:37-40
        o->is_multistep_ = true;
        o->L_ = o->lm_.order;
        o->substeps_ = 4;
        if (!find_srkn("BlanesMoan6B", &o->rk_)) return EPH_ERR_UNSUPPORTED;
        o->h_sub_ = h * (1.0 / 4.0);   // params.h * Ratio::from_recip(SUBSTEPS)
    } else if (find_srkn(method, &o->rk_)) {
        o->is_multistep_ = false;
        o->L_ = 1;
        o->substeps_ = 1;
        o->h_sub_ = h;
    } else {
        return EPH_ERR_BAD_ARGUMENT;
    }
    o->h_ = h;
    o->time_ = t0;
    o->bound_ = INFINITY;   // nbody.rs:111
    o->pv_ = default_pair_variant();
    EPH_HIP(hipGetDevice(&o->device_));
    EPH_HIP(hipStreamCreateWithFlags(&o->stream_, hipStreamNonBlocking));
    EPH_HIP(hipEventCreate(&o->ev0_));
    EPH_HIP(hipEventCreate(&o->ev1_));
    if ((st = o->alloc_buffers())) return st;
    o->lo_ = 0;
    o->hi_ = n;
    if (n > 0) {
        hipStream_t s = o->stream_;
        EPH_HIP(hipMemcpyAsync(o->mu_.p, mu, sizeof(double) * n, hipMemcpyHostToDevice, s));
        EPH_HIP(hipMemcpyAsync(o->stage_.p, pos, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
        if ((st = launch_aos_to_soa(s, n, o->npad_, o->stage_.p, o->Yslot(0)))) return st;
        EPH_HIP(hipStreamSynchronize(s));   // stage_ is reused below
        EPH_HIP(hipMemcpyAsync(o->stage_.p, vel, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
        if ((st = launch_aos_to_soa(s, n, o->npad_, o->stage_.p, o->V_.p))) return st;
        if ((st = launch_pack(s, n, o->npad_, o->Yslot(0), o->mu_.p, o->P_[0].p))) return st;
        if ((st = launch_pack(s, n, o->npad_, o->Yslot(0), o->mu_.p, o->P_[1].p))) return st;
        EPH_HIP(hipStreamSynchronize(s));
    }
    *out = std::move(o);
    return EPH_OK;
}

int NBodyIntegration::clone(std::unique_ptr<NBodyIntegration> *out) {
    EPH_HIP(hipSetDevice(device_));
    EPH_HIP(hipStreamSynchronize(stream_));
    std::unique_ptr<NBodyIntegration> o(new NBodyIntegration());
    o->device_ = device_;
    o->n_ = n_; o->L_ = L_;
    o->is_multistep_ = is_multistep_; o->lm_ = lm_; o->rk_ = rk_; o->substeps_ = substeps_;
    o->h_ = h_; o->h_sub_ = h_sub_; o->time_ = time_; o->bound_ = bound_; o->pv_ = pv_;
    o->starter_i_ = starter_i_; o->lm_i_ = lm_i_; o->evals_ = evals_;
    o->cur_ = cur_; o->pp_ = pp_; o->path_ = path_; o->predicted_ = predicted_;
    o->failed_ = failed_;                                  // (a handle a gang launch left behind its bookkeeping stays marked in its copies)
    o->lo_ = lo_; o->hi_ = hi_; o->slice_ = slice_; o->xch_ = xch_;   // a clone of a sharded handle shares the ranks
    EPH_HIP(hipStreamCreateWithFlags(&o->stream_, hipStreamNonBlocking));
    EPH_HIP(hipEventCreate(&o->ev0_));
    EPH_HIP(hipEventCreate(&o->ev1_));
    int st = o->alloc_buffers();
    if (st) return st;
    hipStream_t s = o->stream_;
    for (int k = 0; k < 2; ++k)
        EPH_HIP(hipMemcpyAsync(o->P_[k].p, P_[k].p, sizeof(Body4) * npad_, hipMemcpyDeviceToDevice, s));
    EPH_HIP(hipMemcpyAsync(o->Y_.p, Y_.p, sizeof(double) * Y_.count, hipMemcpyDeviceToDevice, s));
    EPH_HIP(hipMemcpyAsync(o->A_.p, A_.p, sizeof(double) * A_.count, hipMemcpyDeviceToDevice, s));
    EPH_HIP(hipMemcpyAsync(o->V_.p, V_.p, sizeof(double) * V_.count, hipMemcpyDeviceToDevice, s));
    EPH_HIP(hipMemcpyAsync(o->ASR_.p, ASR_.p, sizeof(double) * ASR_.count, hipMemcpyDeviceToDevice, s));
    EPH_HIP(hipMemcpyAsync(o->mu_.p, mu_.p, sizeof(double) * mu_.count, hipMemcpyDeviceToDevice, s));
    EPH_HIP(hipStreamSynchronize(s));
    *out = std::move(o);
    return EPH_OK;
}

// Target partition: rank r owns bodies [r*slice, (r+1)*slice) with slice = npad/world. From here on
// only the owned bodies' history, velocities and accelerations are kept current on this rank; the packed
// positions of ALL bodies are, through one all-gather after every kernel that publishes positions.
int NBodyIntegration::set_shard(std::shared_ptr<Exchange> x) {
    if (!x || xch_) return EPH_ERR_BAD_ARGUMENT;          // a handle is sharded once, while every body is current
    if (n_ <= kSmallN) return EPH_ERR_UNSUPPORTED;        // one-workgroup systems: replicas only
    if (npad_ % (x->world() * kTile) != 0) {
        set_last_error_text("sharding needs the padded body count to be a multiple of 64 * world");
        return EPH_ERR_BAD_ARGUMENT;
    }
    slice_ = npad_ / x->world();
    lo_ = x->rank() * slice_;
    hi_ = lo_ + slice_ < n_ ? lo_ + slice_ : n_;
    if (hi_ < lo_) hi_ = lo_;
    xch_ = std::move(x);
    return EPH_OK;
}
int NBodyIntegration::gather_packed(Body4 *buf) {
    return xch_ ? xch_->all_gather_inplace(buf, sizeof(Body4) * (size_t)slice_, stream_) : EPH_OK;
}
int NBodyIntegration::gather_stage() {
    return xch_ ? xch_->all_gather_inplace(stage_.p, sizeof(double) * 3 * (size_t)slice_, stream_) : EPH_OK;
}

int NBodyIntegration::sync() {
    EPH_HIP(hipSetDevice(device_));
    EPH_HIP(hipStreamSynchronize(stream_));
    return xch_ ? xch_->poll_error() : EPH_OK;          // a peer that never delivered (peer.hip) surfaces here
}

// FixedRungeKuttaIntegrator::advance (runge_kutta/mod.rs:112-125) with SRKN::advance (symplectic.rs:69-102)
int NBodyIntegration::srkn_step(double h, double *y_slot) {
    if (time_ >= bound_) return EPH_BOUND_REACHED;
    if (time_ + h == time_) return EPH_STEP_SIZE_UNDERFLOW;
    int st;
    for (int s = 0; s < rk_.stages; ++s) {
        // *dy = *dy + *ddy * (h * C::B[s]);  *y = *y + *dy * (h * C::A[s])
        const KickDrift kd{V_.p, y_slot, h * rk_.B[s], h * rk_.A[s], P_[pp_ ^ 1].p};
        if (!rk_.fsal || s > 0 || starter_i_ == 0) {
            // problem.ode.eval(t_stage, &problem.state.y, self.ddy.zero()) and the stage update behind it, one launch
            if ((st = launch_accel(pv_, stream_, n_, npad_, P_[pp_].p, nullptr, ASR_.p, force_kind(), lo_, hi_, &kd))) return st;
            evals_++;
        } else if ((st = launch_kick_drift(stream_, n_, npad_, ASR_.p, V_.p, y_slot, kd.hb, kd.ha, mu_.p, P_[pp_ ^ 1].p))) {
            return st;                                 // FSAL first stage: the acceleration is the previous step's last
        }
        if ((st = gather_packed(P_[pp_ ^ 1].p))) return st;
        pp_ ^= 1;
    }
    time_ = time_ + h;
    starter_i_ += 1;
    return EPH_OK;
}

// LinearMultistepIntegrator::advance while `starter.step_count() < ORDER`   multistep/mod.rs:211-218
//   first call only : lm.advance_with(problem, no-op)   -> current_ddy = f(y0)
//   every call      : lm.advance_with(problem, starter) -> ring.front = (state, current_ddy); 4 sub-steps;
//                                                           current_ddy = f(y)
// Ring convention here: slot cur_ holds the newest level (Y and A); the level under construction is built in
// place in the slot that will become the new front.
int NBodyIntegration::startup_macro_step() {
    if (time_ >= bound_) return EPH_BOUND_REACHED;
    if (time_ + h_ == time_) return EPH_STEP_SIZE_UNDERFLOW;
    int st;
    if (starter_i_ / (uint32_t)substeps_ == 0) {
        if ((st = launch_accel(pv_, stream_, n_, npad_, P_[pp_].p, nullptr, Aslot(cur_), force_kind(), lo_, hi_))) return st;
        evals_++;
    }
    const int nslot = (cur_ + L_ - 1) % L_;
    if ((st = launch_copy3(stream_, n_, npad_, Yslot(cur_), Yslot(nslot)))) return st;
    cur_ = nslot;   // from here on the working state is the new front, as in the reference after the clone_from
    for (int s = 0; s < substeps_; ++s)
        if ((st = srkn_step(h_sub_, Yslot(nslot)))) return st;
    if ((st = launch_accel(pv_, stream_, n_, npad_, P_[pp_].p, nullptr, Aslot(nslot), force_kind(), lo_, hi_))) return st;
    evals_++;
    return EPH_OK;
}

// The k-fold replay of `time = time + h` below reproduces the reference's bits of `problem.time`; the two tests inside it
// almost never fire. When they provably cannot -- no bound, and |h| far above the spacing of doubles at any |t| the k steps
// can reach -- the answer is k without the loop, and the replay that updates time_ can run AFTER the launch, under the kernel
// (lm_batch). It matters for gangs: 256 systems x 100 000 steps is 77 M dependent additions (three passes: gang_ready,
// advance, the update) that the device sat waiting for -- the "gang of 256 steps 1.31 us against 0.78 alone" of round 3
// (profiles/r04_small_kernel_evidence.md: every workgroup of the gang took 0.89 us per step; the rest was this loop).
bool NBodyIntegration::steps_certain(int64_t k) const {
    if (!(bound_ == INFINITY) || !std::isfinite(time_) || !std::isfinite(h_) || h_ == 0.0 || k < 0) return false;
    const double reach = (std::fabs(time_) + (double)k * std::fabs(h_)) * 1.000001;   // |t| stays below this (k <= 2^30 roundings of 2^-53)
    // t + h == t needs |h| <= ulp(t) / 2 <= 2^-53 |t|
    return std::isfinite(reach) && std::fabs(h_) > reach * 4.440892098500626e-16;
}

int64_t NBodyIntegration::steps_available(int64_t k, int *status_after) const {
    if (steps_certain(k)) { *status_after = EPH_OK; return k; }
    double t = time_;
    int64_t s = 0;
    *status_after = EPH_OK;
    for (; s < k; ++s) {
        if (t >= bound_) { *status_after = EPH_BOUND_REACHED; break; }
        if (t + h_ == t) { *status_after = EPH_STEP_SIZE_UNDERFLOW; break; }
        t = t + h_;
    }
    return s;
}

// Every pending pair has BOTH events recorded (lm_batch pushes a pair only behind its closing record). A pair whose query fails
// (a device error in between) is skipped and recycled with the others: the list is always emptied, so one bad interval can neither
// stop the accumulation nor fill the list until every later batch fails (advisor, round 4).
int NBodyIntegration::resolve_timing() {
    int st = EPH_OK;
    for (auto &e : ev_pending_) {
        float ms = 0;
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
            kernel_ms_ += ms;
        } else {
            set_last_error("timing events", hipGetLastError());
            st = EPH_ERR_HIP;
        }
    }
    ev_free_.insert(ev_free_.end(), ev_pending_.begin(), ev_pending_.end());
    ev_pending_.clear();
    return st;
}

// k x ELM2::advance   second_order/mod.rs:90-131
int NBodyIntegration::lm_batch(int64_t k) {
    LmArgs a{};
    a.n = n_; a.npad = npad_; a.L = L_;
    a.lo = lo_; a.hi = hi_;
    a.Y = Y_.p; a.A = A_.p; a.V = V_.p;
    for (int j = 0; j < L_; ++j) { a.wa[j] = lm_.wa[j]; a.wb[j] = lm_.wb[j]; a.cw[j] = lm_.cw[j]; }
    a.h = h_;
    a.hh = h_ * h_ * lm_.inv_beta_d;     // h * h * Ratio::from_recip(C::BETA_D)
    a.hc = h_ * lm_.inv_cowell_d;        // h * Ratio::from_recip(Self::BETA_D)
    a.samp = samp_;
    a.kind = force_kind();
    int st;
    const bool fast = path_ == EPH_PATH_FAST || path_ == EPH_PATH_FAST_RSQ || path_ == EPH_PATH_F32_PAIRS;
    // on a target partition only the binary32 pair arithmetic runs (BASELINE configs[4] as stated); fast / fast-rsq stay single-device
    if (fast && (n_ <= kSmallN || (sharded() && path_ != EPH_PATH_F32_PAIRS))) return EPH_ERR_UNSUPPORTED;
    const bool f32_sharded = sharded() && path_ == EPH_PATH_F32_PAIRS;
    if (fast && !fast_partial_.p) {
        if ((st = fast_partial_.alloc(fast_partial_doubles(npad_)))) return st;
        // the arrival tickets of the one-launch step start at 0 (the last arriver of every block puts its ticket back)
        const size_t toff = fast_ticket_offset_doubles(npad_);
        EPH_HIP(hipMemsetAsync(fast_partial_.p + toff, 0, sizeof(double) * (fast_partial_doubles(npad_) - toff), stream_));
    }
    if (path_ == EPH_PATH_F32_PAIRS && !posf_.p) {
        if ((st = posf_.alloc((size_t)4 * npad_))) return st;
    }
    const bool persistent = n_ <= kSmallN && path_ != 1 && path_ != 3 && !fast;
    if (path_ == 2 && n_ > kSmallN) return EPH_ERR_UNSUPPORTED;
    if (collect_ && !persistent) return EPH_ERR_UNSUPPORTED;           // (advance_many checks gang_ready first)
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (timing_ && !collect_) {
        if (ev_pending_.size() >= 1024) (void)resolve_timing();           // (a failed interval is dropped there, not returned here)
        if (!ev_free_.empty()) { ev = ev_free_.back(); ev_free_.pop_back(); }
        else { EPH_HIP(hipEventCreate(&ev.first)); EPH_HIP(hipEventCreate(&ev.second)); }
        if (hipEventRecord(ev.first, stream_) != hipSuccess) {
            ev_free_.push_back(ev);
            set_last_error("hipEventRecord", hipGetLastError());
            return EPH_ERR_HIP;
        }
    }
    // the pair joins the pending list only once its closing event is recorded (below); every earlier return hands it back
    struct TimingGuard {
        std::pair<hipEvent_t, hipEvent_t> &ev; std::vector<std::pair<hipEvent_t, hipEvent_t>> &free_list; bool armed;
        ~TimingGuard() { if (armed && ev.second) free_list.push_back(ev); }
    } timing_guard{ev, ev_free_, true};
    if (persistent) {
        a.cur = cur_;
        a.pos_cur = P_[pp_].p;
        a.pos_next = P_[pp_ ^ 1].p;
        if (collect_) collect_->push_back(a);                           // launched by advance_many with the gang's others
        else if ((st = launch_lm_persistent(pv_, stream_, a, k))) return st;
        cur_ = (int)(((int64_t)cur_ - k % L_ + L_) % L_);
        if (timing_) kernel_launches_ += 1;
    } else {
        a.cur = cur_;
        a.pos_cur = P_[pp_].p;
        a.pos_next = P_[pp_ ^ 1].p;
        // a handle that can neither take the single-workgroup kernels nor exchange positions leaves the prediction of the step
        // after the batch behind (host.h predicted_): one launch and one launch boundary less per call
        const bool leave_prediction = n_ > kSmallN && !sharded();
        if (!predicted_) {
            if ((st = launch_lm_predict(stream_, a))) return st;
            if ((st = gather_packed(a.pos_next))) return st;
        }
        predicted_ = false;
        for (int64_t s = 1; s <= k; ++s) {
            pp_ ^= 1;
            cur_ = (cur_ + L_ - 1) % L_;
            a.cur = cur_;
            a.pos_cur = P_[pp_].p;
            a.pos_next = P_[pp_ ^ 1].p;
            a.do_predict = s < k || leave_prediction;
            a.step = (uint32_t)s;
            if (f32_sharded) {
                // this rank's rows of the binary32 copy (16 B per body: SURVEY 8(e)'s f32x4), one all-gather of them, then the rank's
                // targets against all sources in the single-device slice order. The f64 packed positions of the OTHER ranks' bodies
                // are not needed between the steps of a batch (the batch's first prediction gathers them once, above).
                if ((st = launch_lm_step_fast(pv_, stream_, a, fast_partial_.p, false, posf_.p, 1, lo_, slice_))) return st;
                if ((st = xch_->all_gather_inplace(posf_.p, sizeof(float) * 4 * (size_t)slice_, stream_))) return st;
                if ((st = launch_lm_step_fast(pv_, stream_, a, fast_partial_.p, false, posf_.p, 2))) return st;
                continue;
            }
            if ((st = fast ? launch_lm_step_fast(pv_, stream_, a, fast_partial_.p, path_ == EPH_PATH_FAST_RSQ,
                                                 path_ == EPH_PATH_F32_PAIRS ? posf_.p : nullptr)
                           : launch_lm_step(pv_, stream_, a)))
                return st;
            if (s < k && (st = gather_packed(a.pos_next))) return st;
        }
        predicted_ = leave_prediction;
        if (timing_) kernel_launches_ += (uint64_t)k;
    }
    if (ev.second) {
        EPH_HIP(hipEventRecord(ev.second, stream_));
        ev_pending_.push_back(ev);
        timing_guard.armed = false;
    }
    if (collect_) deferred_time_steps_ += k;              // advance_many replays them once the gang is launched
    else for (int64_t s = 0; s < k; ++s) time_ = time_ + h_;   // problem.time = problem.time + h, per step (the launch is already queued)
    lm_i_ += (uint32_t)k;
    evals_ += (uint64_t)k;
    return EPH_OK;
}

int NBodyIntegration::advance(int64_t n_steps, int64_t *done_out) {
    if (failed_) { if (done_out) *done_out = 0; return failed_; }     // a gang launch failed after this handle's bookkeeping had moved
    EPH_HIP(hipSetDevice(device_));
    int64_t done = 0;
    int st = EPH_OK;
    const SampleArgs samp = samp_;
    while (done < n_steps) {
        if (!is_multistep_) {
            if ((st = srkn_step(h_, Yslot(0)))) break;
            done++;
            if ((st = launch_sample(stream_, n_, npad_, Yslot(0), samp, (uint32_t)done))) break;
        } else if (!started()) {
            if ((st = startup_macro_step())) break;
            done++;
            if ((st = launch_sample(stream_, n_, npad_, Yslot(cur_), samp, (uint32_t)done))) break;
        } else {
            int after = EPH_OK;
            // the persistent kernel takes a 32-bit step index for sampling; keep batches below 2^31
            int64_t want = n_steps - done;
            if (want > (int64_t)1 << 30) want = (int64_t)1 << 30;
            const int64_t k = steps_available(want, &after);
            if (k > 0) {
                // sampling phases are relative to the start of this advance(): shift them past the steps
                // already taken in the other regimes by running the batch with a step offset of `done`
                if (done > 0 && samp.period) {
                    // rare (start-up and steady state in one call): finish one step at a time
                    SampleArgs none{};
                    samp_ = none;
                    for (int64_t s = 0; s < k && !st; ++s) {
                        st = lm_batch(1);
                        if (!st) {
                            done++;
                            st = launch_sample(stream_, n_, npad_, Yslot(cur_), samp, (uint32_t)done);
                        }
                    }
                    samp_ = samp;
                    if (st) break;
                } else {
                    if ((st = lm_batch(k))) break;
                    done += k;
                }
            }
            if (k < want) { st = after; break; }
        }
    }
    samp_ = SampleArgs{};
    if (done_out) *done_out = done;
    return st;
}

bool NBodyIntegration::gang_ready(int64_t k) const {
    if (!is_multistep_ || !started() || sharded() || n_ > kGangMaxN || n_ <= 0 || (path_ != 0 && path_ != 2)) return false;
    if (k <= 0 || k > ((int64_t)1 << 30)) return false;
    int after = EPH_OK;
    return steps_available(k, &after) == k;
}

int NBodyIntegration::advance_many(NBodyIntegration *const *igs, int count, int64_t k) {
    if (count < 0 || (count > 0 && !igs) || k < 0) return EPH_ERR_BAD_ARGUMENT;
    if (count == 0 || k == 0) return EPH_OK;
    bool gang = count > 1;
    for (int i = 0; i < count && gang; ++i)
        gang = igs[i] && igs[i]->gang_ready(k) && igs[i]->device_ == igs[0]->device_ && igs[i]->L_ == igs[0]->L_ &&
               igs[i]->pv_ == igs[0]->pv_;
    for (int i = 0; i < count; ++i) {
        if (!igs[i]) return EPH_ERR_BAD_ARGUMENT;
        for (int j = 0; j < i; ++j)
            if (igs[j] == igs[i]) return EPH_ERR_BAD_ARGUMENT;          // a system cannot be in the gang twice
    }
    if (!gang) {                                                        // the plain meaning: advance(k) on each
        int first = EPH_OK;
        for (int i = 0; i < count; ++i) {
            const int st = igs[i]->advance(k);
            if (st && !first) first = st;
        }
        return first;
    }
    NBodyIntegration &lead = *igs[0];
    EPH_HIP(hipSetDevice(lead.device_));
    for (int i = 0; i < count; ++i)
        if (igs[i]->failed_) return igs[i]->failed_;
    // everything that can fail for lack of resources happens BEFORE any handle's bookkeeping moves
    int st;
    if ((st = lead.gang_args_.reserve((size_t)count))) return st;
    if (!lead.gang_ev_) EPH_HIP(hipEventCreateWithFlags(&lead.gang_ev_, hipEventDisableTiming));
    if (!lead.gang_copy_ev_) EPH_HIP(hipEventCreateWithFlags(&lead.gang_copy_ev_, hipEventDisableTiming));
    if (lead.gang_host_count_ < (size_t)count) {
        if (lead.gang_host_) { EPH_HIP(hipEventSynchronize(lead.gang_copy_ev_)); (void)hipHostFree(lead.gang_host_); }
        lead.gang_host_ = nullptr;
        lead.gang_host_count_ = 0;
        const size_t want = (size_t)count + (size_t)count / 2;
        EPH_HIP(hipHostMalloc((void **)&lead.gang_host_, sizeof(LmArgs) * want, hipHostMallocDefault));
        lead.gang_host_count_ = want;
    }
    // the previous gang's argument copy out of the pinned array must have completed before it is rewritten
    EPH_HIP(hipEventSynchronize(lead.gang_copy_ev_));
    std::vector<LmArgs> args;
    args.reserve((size_t)count);
    for (int i = 0; i < count; ++i) {
        NBodyIntegration &ig = *igs[i];
        ig.collect_ = &args;
        int64_t done = 0;
        st = ig.advance(k, &done);                                      // bookkeeping as usual, launch arguments into `args`
        ig.collect_ = nullptr;
        if (st || done != k || args.size() != (size_t)i + 1) {
            // (cannot happen after gang_ready: no launch was made, so handles 0..i are ahead of their device state)
            for (int j = 0; j <= i; ++j) { igs[j]->replay_deferred_time(); igs[j]->failed_ = st ? st : EPH_ERR_HIP; }
            return st ? st : EPH_ERR_HIP;
        }
    }
    static const int dbg = [] { const char *e = getenv("EPH_DEBUG_SMALL"); return e ? atoi(e) : 0; }();   // tuning switches of k_lm_small
    for (int i = 0; i < count; ++i) { lead.gang_host_[i] = args[(size_t)i]; lead.gang_host_[i].wg_flags = dbg; }
    // From here on a failure leaves every handle's bookkeeping k steps ahead of its device state: the handles are marked
    // failed (every later call returns the status) instead of being handed back as if nothing had happened.
    auto fail = [&](int status) {
        for (int i = 0; i < count; ++i) { igs[i]->replay_deferred_time(); igs[i]->failed_ = status; }
        return status;
    };
#define EPH_GANG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(#call, e_); return fail(EPH_ERR_HIP); } } while (0)
    // the gang's launch goes on the lead's stream, behind whatever the other handles still have in flight on theirs
    for (int i = 1; i < count; ++i) {
        EPH_GANG_HIP(hipEventRecord(lead.gang_ev_, igs[i]->stream_));
        EPH_GANG_HIP(hipStreamWaitEvent(lead.stream_, lead.gang_ev_, 0));
    }
    EPH_GANG_HIP(hipMemcpyAsync(lead.gang_args_.p, lead.gang_host_, sizeof(LmArgs) * (size_t)count, hipMemcpyHostToDevice, lead.stream_));
    EPH_GANG_HIP(hipEventRecord(lead.gang_copy_ev_, lead.stream_));
    if (lead.timing_) EPH_GANG_HIP(hipEventRecord(lead.ev0_, lead.stream_));
    st = launch_lm_small_many(lead.pv_, lead.stream_, lead.gang_args_.p, count, lead.L_, k);
    if (st) return fail(st);
    for (int i = 0; i < count; ++i) igs[i]->replay_deferred_time();     // host bookkeeping under the kernel
    if (lead.timing_) {
        EPH_GANG_HIP(hipEventRecord(lead.ev1_, lead.stream_));
        EPH_GANG_HIP(hipEventSynchronize(lead.ev1_));
        float ms = 0;
        EPH_GANG_HIP(hipEventElapsedTime(&ms, lead.ev0_, lead.ev1_));
        lead.kernel_ms_ += ms;
    }
    EPH_GANG_HIP(hipEventRecord(lead.gang_ev_, lead.stream_));
    for (int i = 1; i < count; ++i) EPH_GANG_HIP(hipStreamWaitEvent(igs[i]->stream_, lead.gang_ev_, 0));
#undef EPH_GANG_HIP
    return EPH_OK;                                                      // not synchronised: like eph_nbody_advance, the call only queues
}

int NBodyIntegration::get_state(double *pos, double *vel, double *t, uint32_t *sc) {
    EPH_HIP(hipSetDevice(device_));
    int st;
    if (!xch_ && n_ > 0 && (pos || vel)) {
        // one device: the two transposition kernels write straight into the pinned, device-mapped staging buffer -- one
        // synchronisation, no copy-engine submissions (the reference's Integration reads the state after every step: 115-119 -> 63 us
        // per step at N = 4096 with this, scripts/time_boundary.py)
        const size_t nd = (size_t)n_;
        PinnedStage stage(sizeof(double) * 6 * nd);       // this call's own buffer: a second handle's read-back does not wait for ours
        if (stage.status()) return stage.status();
        StreamIdleOnExit idle(stream_);
        double *h = static_cast<double *>(stage.host()), *d = static_cast<double *>(stage.dev());
        if (pos && (st = launch_soa_to_aos(stream_, n_, npad_, Yslot(is_multistep_ ? cur_ : 0), d))) return st;
        if (vel && (st = launch_soa_to_aos(stream_, n_, npad_, V_.p, d + 3 * nd))) return st;
        EPH_HIP(hipStreamSynchronize(stream_));
        idle.disarm();
        if (pos) std::memcpy(pos, h, sizeof(double) * 3 * nd);
        if (vel) std::memcpy(vel, h + 3 * nd, sizeof(double) * 3 * nd);
        if (t) *t = time_;
        if (sc) *sc = step_count();
        return EPH_OK;
    }
    if (pos && n_ > 0) {
        if ((st = launch_soa_to_aos(stream_, n_, npad_, Yslot(is_multistep_ ? cur_ : 0), stage_.p))) return st;
        if ((st = gather_stage())) return st;               // sharded: every rank contributes its bodies
        EPH_HIP(hipMemcpyAsync(pos, stage_.p, sizeof(double) * 3 * n_, hipMemcpyDeviceToHost, stream_));
        EPH_HIP(hipStreamSynchronize(stream_));
        if (xch_ && (st = xch_->poll_error())) return st;    // a peer that never delivered: the gathered rows are not data
    }
    if (vel && n_ > 0) {
        if ((st = launch_soa_to_aos(stream_, n_, npad_, V_.p, stage_.p))) return st;
        if ((st = gather_stage())) return st;
        EPH_HIP(hipMemcpyAsync(vel, stage_.p, sizeof(double) * 3 * n_, hipMemcpyDeviceToHost, stream_));
        EPH_HIP(hipStreamSynchronize(stream_));
        if (xch_ && (st = xch_->poll_error())) return st;
    }
    if (t) *t = time_;
    if (sc) *sc = step_count();
    return EPH_OK;
}

int NBodyIntegration::get_acc(double *acc) {
    EPH_HIP(hipSetDevice(device_));
    if (!acc) return EPH_ERR_BAD_ARGUMENT;
    if (n_ == 0) return EPH_OK;
    int st;
    if (!xch_) {                                          // as get_state: through the pinned, device-mapped staging buffer
        PinnedStage stage(sizeof(double) * 3 * (size_t)n_);
        if (stage.status()) return stage.status();
        StreamIdleOnExit idle(stream_);
        if ((st = launch_soa_to_aos(stream_, n_, npad_, is_multistep_ ? Aslot(cur_) : ASR_.p, static_cast<double *>(stage.dev())))) return st;
        EPH_HIP(hipStreamSynchronize(stream_));
        idle.disarm();
        std::memcpy(acc, stage.host(), sizeof(double) * 3 * (size_t)n_);
        return EPH_OK;
    }
    if ((st = launch_soa_to_aos(stream_, n_, npad_, is_multistep_ ? Aslot(cur_) : ASR_.p, stage_.p))) return st;
    if ((st = gather_stage())) return st;
    EPH_HIP(hipMemcpyAsync(acc, stage_.p, sizeof(double) * 3 * n_, hipMemcpyDeviceToHost, stream_));
    EPH_HIP(hipStreamSynchronize(stream_));
    return xch_ ? xch_->poll_error() : EPH_OK;
}

// seam 1: SecondOrderODE::eval for NewtonianGravity, host buffers in and out.
// The call's device buffers are grow-only scratch kept per device between calls (round 5): six hipMalloc / hipFree pairs per call --
// each hipFree a device synchronisation -- were most of what a call cost at the app's sizes. Round 6: a POOL of scratch sets per
// device, one leased per call with its own stream, so concurrent callers (the reference evaluates the RHS from every integrator
// thread) neither wait for each other nor share a buffer; the mutex covers the free list only. The scratch is ordinary library
// memory (eph_release_cached_memory does not touch it; it is a few MB at N = 65 536, and as many sets exist as calls ever overlapped).
namespace {
// grow-only device block taken from the driver directly: it lives as long as the process, so it must not count as a live allocation
// of the library's block cache (mem.cpp releases the cache with the LAST counted allocation on a device)
template <typename T>
struct RawScratch {
    T *p = nullptr;
    size_t count = 0;
    int reserve(size_t n) {
        if (n <= count) return EPH_OK;
        if (p) (void)hipFree(p);
        p = nullptr; count = 0;
        const size_t want = n + n / 2;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; set_last_error("hipMalloc (seam 1 scratch)", e); return e == hipErrorOutOfMemory ? EPH_ERR_OUT_OF_MEMORY : EPH_ERR_HIP; }
        count = want;
        return EPH_OK;
    }
};
struct AccelScratch {
    RawScratch<Body4> P;
    RawScratch<double> soa, init, out;
    hipStream_t stream = nullptr;
};
constexpr int kAccelDevices = 64;
std::mutex g_accel_mu;
std::vector<AccelScratch *> g_accel_free[kAccelDevices];             // (live as long as the process: destroying device memory at exit is the driver's job)
struct AccelLease {                                                  // one scratch set of `device` for the life of the object
    int device;
    AccelScratch *x = nullptr;
    explicit AccelLease(int dev) : device(dev) {
        std::lock_guard<std::mutex> lk(g_accel_mu);
        auto &fl = g_accel_free[device];
        if (!fl.empty()) { x = fl.back(); fl.pop_back(); }
        else x = new AccelScratch();
    }
    ~AccelLease() {
        std::lock_guard<std::mutex> lk(g_accel_mu);
        g_accel_free[device].push_back(x);
    }
    AccelLease(const AccelLease &) = delete;
    AccelLease &operator=(const AccelLease &) = delete;
};
}  // namespace
int accel_eval_device(int n, const double *pos, const double *mu, double *acc) {
    if (n < 0 || (n > 0 && (!pos || !mu || !acc))) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    if (n == 0) return EPH_OK;
    const int npad = ((n + 63) / 64) * 64;
    int device = 0;
    EPH_HIP(hipGetDevice(&device));
    if (device < 0 || device >= kAccelDevices) return EPH_ERR_BAD_ARGUMENT;
    AccelLease lease(device);
    AccelScratch *x = lease.x;
    if (!x->stream) EPH_HIP(hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking));
    hipStream_t s = x->stream;
    if ((st = x->P.reserve(npad)) || (st = x->soa.reserve((size_t)3 * npad)) || (st = x->init.reserve((size_t)3 * npad)) ||
        (st = x->out.reserve((size_t)3 * npad)))
        return st;
    // Host buffers cross the bus through a pinned, device-mapped staging buffer (mem.cpp): the kernels read the caller's numbers and
    // write the result THROUGH it, so a call is five launches and one synchronisation -- no copy-engine submissions (three pageable
    // hipMemcpy in and one out were ~40 us of a 69 us call at 32 bodies: 29 us now; N = 4096: 177 us with the per-call allocations
    // of round 4, 136 without them, 75-80 through the staging buffer -- of which the force kernel is 40)
    const size_t nd = (size_t)n;
    PinnedStage stage(sizeof(double) * 7 * nd);
    if (stage.status()) return stage.status();
    StreamIdleOnExit idle(s);                         // an error return below leaves only once nothing in flight uses stage / scratch
    double *h = static_cast<double *>(stage.host()), *d = static_cast<double *>(stage.dev());
    std::memcpy(h, mu, sizeof(double) * nd);
    std::memcpy(h + nd, pos, sizeof(double) * 3 * nd);
    std::memcpy(h + 4 * nd, acc, sizeof(double) * 3 * nd);
    if ((st = launch_aos_to_soa(s, n, npad, d + nd, x->soa.p))) return st;
    if ((st = launch_pack(s, n, npad, x->soa.p, d, x->P.p))) return st;
    if ((st = launch_aos_to_soa(s, n, npad, d + 4 * nd, x->init.p))) return st;
    if ((st = launch_accel(default_pair_variant(), s, n, npad, x->P.p, x->init.p, x->out.p))) return st;
    if ((st = launch_soa_to_aos(s, n, npad, x->out.p, d + 4 * nd))) return st;
    EPH_HIP(hipStreamSynchronize(s));
    idle.disarm();
    std::memcpy(acc, h + 4 * nd, sizeof(double) * 3 * nd);
    return EPH_OK;
}

}  // namespace eph
