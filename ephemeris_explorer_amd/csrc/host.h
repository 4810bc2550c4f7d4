// host.h -- host-side mirror of the reference's operator surface for the hot path (C++; the reference's host
// language, Rust, is not available in this image -- see DESIGN.md). Names follow the reference:
//   integration::{Method, Integrator, IntegratorState, Integration, StepError}   integration/src/lib.rs
//   integration::multistep::{LinearMultistepIntegrator, Substepper, ELM2}         integration/src/multistep/
//   integration::runge_kutta::{FixedRungeKuttaIntegrator, SRKN}                    integration/src/runge_kutta/
//   ephemeris::{NBodyPropagator, SplineInterpolators, UniformSpline, Polynomial}   ephemeris/src/
// All numerical state lives on the device; the host keeps the scalar bookkeeping the reference keeps
// (time, bound, step counters, sampling counters) and replays it with the reference's f64 operations.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "eph_internal.h"

namespace eph {

// device blocks through the library's cache of large allocations (mem.cpp): dev_alloc reuses a cached block of exactly `bytes`
// (contents unspecified) or calls hipMalloc; dev_free caches blocks of >= 64 MiB (after a device synchronisation, as hipFree has)
// up to the cache's cap and frees the rest; release_cached_memory returns the cache to the driver (bytes released).
int dev_alloc(size_t bytes, void **out);
void dev_free(void *p, size_t bytes);
size_t release_cached_memory();
size_t cached_memory_bytes();
// A pinned, device-mapped staging buffer taken from the library's pool for the life of the object (mem.cpp; one per concurrent user,
// so handles driven from distinct threads do not wait for each other): a kernel stores through dev(), the host reads host() after
// synchronising the kernel's stream. The stream must be idle with respect to the buffer before the object is destroyed.
class PinnedStage {
public:
    explicit PinnedStage(size_t bytes);
    ~PinnedStage();
    PinnedStage(const PinnedStage &) = delete;
    PinnedStage &operator=(const PinnedStage &) = delete;
    int status() const { return status_; }
    void *host() const { return host_; }
    void *dev() const { return dev_; }

private:
    void *host_ = nullptr, *dev_ = nullptr;
    size_t bytes_ = 0;
    int status_ = EPH_OK;
};

// Declared AFTER a PinnedStage (or a scratch lease) whose buffer kernels on `s` use: an early error return leaves the scope only once
// the stream is idle, so no kernel still in flight writes a buffer that has gone back to the pool. disarm() after the happy path's own
// synchronisation.
struct StreamIdleOnExit {
    hipStream_t s;
    bool armed = true;
    explicit StreamIdleOnExit(hipStream_t stream) : s(stream) {}
    StreamIdleOnExit(const StreamIdleOnExit &) = delete;
    StreamIdleOnExit &operator=(const StreamIdleOnExit &) = delete;
    ~StreamIdleOnExit() { if (armed) (void)hipStreamSynchronize(s); }
    void disarm() { armed = false; }
};

template <typename T>
struct DevBuf {   // owning device allocation
    T *p = nullptr;
    size_t count = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) dev_free(p, count * sizeof(T));
        p = nullptr;
        count = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 1;
        void *q = nullptr;
        const int st = dev_alloc(n * sizeof(T), &q);
        if (st) return st;
        p = static_cast<T *>(q);
        count = n;
        return EPH_OK;
    }
    int reserve(size_t n) { return n <= count ? EPH_OK : alloc(n + n / 2); }   // grow-only scratch; contents are lost
};

// Direct-write transport of the exchange step (peer.hip): every rank owns a mailbox that its peers map through
// hipIpc and write into; no collective library involved.
constexpr int kPeerMaxWorld = 16;
constexpr size_t kPeerHandleBytes = 64;            // sizeof(hipIpcMemHandle_t)
class PeerTransport {
public:
    ~PeerTransport();
    // form: 0 = fine-grained mailbox, falling back to plain device memory when allocation or export fails | 1 = fine-grained
    // or fail | 2 = plain device memory
    static int create(int rank, int world, size_t slot_bytes, int form, std::shared_ptr<PeerTransport> *out);
    int form() const { return form_; }                        // 1 fine-grained | 2 plain device memory: what is live
    const std::string &fallback_reason() const { return fallback_reason_; }
    const void *handle() const { return handle_; }           // kPeerHandleBytes, to be passed to every peer
    int connect(const void *handles);                         // world x kPeerHandleBytes in rank order
    int all_gather_inplace(void *buf, size_t slice_bytes, hipStream_t s);
    int poll_error() const;                                   // EPH_ERR_COMM once a wait has timed out
    int rank() const { return rank_; }
    int world() const { return world_; }
    uint64_t epochs() const { return epoch_; }

private:
    PeerTransport() = default;
    int rank_ = 0, world_ = 1, device_ = 0, form_ = 0;
    std::string fallback_reason_;
    size_t slot_ = 0;
    void *base_[kPeerMaxWorld] = {};
    char handle_[kPeerHandleBytes] = {};
    bool connected_ = false;
    uint64_t epoch_ = 0;
    unsigned *status_ = nullptr, *status_dev_ = nullptr;
    unsigned long long timeout_ticks_ = 0;
    // exchanges of one transport are totally ordered on the device, whatever stream they are issued on (a handle
    // and its clones share the transport but not the stream): each launch waits for the previous one's event
    hipEvent_t order_ = nullptr;
    hipStream_t last_stream_ = nullptr;
    bool have_last_ = false;
};

// The exchange step of a target-partitioned run (shard.cpp): in-place all-gather of equal slices over the ranks.
class Exchange {
public:
    ~Exchange();
    static int create(int rank, int world, const void *unique_id, eph_exchange_fn fn, void *ctx,
                      std::shared_ptr<Exchange> *out);
    static int create_peer(std::shared_ptr<PeerTransport> t, std::shared_ptr<Exchange> *out);
    int all_gather_inplace(void *buf, size_t slice_bytes, hipStream_t s);
    int poll_error() const { return peer_ ? peer_->poll_error() : EPH_OK; }
    int rank() const { return rank_; }
    int world() const { return world_; }
    uint64_t gathers() const { return gathers_; }

private:
    Exchange() = default;
    int rank_ = 0, world_ = 1;
    void *comm_ = nullptr;          // ncclComm_t
    eph_exchange_fn fn_ = nullptr;
    void *ctx_ = nullptr;
    std::shared_ptr<PeerTransport> peer_;
    uint64_t gathers_ = 0;
};
int rccl_unique_id(void *out128);

// Integration<NBodyProblem<DVec3>, M>: problem + integrator   (integration/src/lib.rs:394-503,
// ephemeris/src/propagators/nbody.rs:41,93-121)
class NBodyIntegration {
public:
    ~NBodyIntegration();
    static int create(int n, const double *pos, const double *vel, const double *mu, double t0, double h,
                      const char *method, std::unique_ptr<NBodyIntegration> *out);
    int clone(std::unique_ptr<NBodyIntegration> *out);

    // n_steps x Integrator::advance; *done = steps actually taken
    int advance(int64_t n_steps, int64_t *done = nullptr);
    int get_state(double *pos, double *vel, double *t, uint32_t *step_count);
    int get_acc(double *acc);
    int sync();

    // IntegratorState
    uint32_t step_count() const { return is_multistep_ ? starter_i_ / (uint32_t)substeps_ + lm_i_ : starter_i_; }
    double step_size() const { return h_; }
    // ODEProblem
    double time() const { return time_; }
    double bound() const { return bound_; }
    void set_bound(double b) { bound_ = b; }
    int n() const { return n_; }
    bool started() const { return !is_multistep_ || starter_i_ / (uint32_t)substeps_ >= (uint32_t)lm_.order; }
    uint64_t evals() const { return evals_; }
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }
    int pair_variant() const { return pv_; }
    void set_path(int p) { path_ = p; }
    int path() const { return path_; }
    // path 1 -> per-step launches with the one-wave-per-block force, 3 -> with the workgroup-specialised force
    int force_kind() const { return path_ == 1 ? 1 : (path_ == 3 ? 2 : 0); }
    void set_sampling(const SampleArgs &s) { samp_ = s; }   // consumed by the next advance() batch
    // target-partition the system over the ranks of `x`: this rank keeps bodies [lo, hi) current
    int set_shard(std::shared_ptr<Exchange> x);
    bool sharded() const { return (bool)xch_; }
    int exchange_error() const { return xch_ ? xch_->poll_error() : EPH_OK; }   // EPH_ERR_COMM once a peer wait timed out
    int shard_lo() const { return lo_; }
    int shard_hi() const { return hi_; }
    uint64_t gathers() const { return xch_ ? xch_->gathers() : 0; }
    int shard_rank() const { return xch_ ? xch_->rank() : 0; }
    int shard_world() const { return xch_ ? xch_->world() : 1; }
    int shard_slice() const { return xch_ ? slice_ : n_; }          // bodies per rank (the last rank may own fewer)
    // in-place all-gather of equal slices of a device buffer over the ranks, on this handle's stream
    int gather_buffer(void *buf, size_t slice_bytes) {
        return xch_ ? xch_->all_gather_inplace(buf, slice_bytes, stream_) : EPH_OK;
    }
    // device view of the current positions, SoA [3][npad] (for on-device consumers: the interpolation-error scan)
    const double *positions_soa() { return Yslot(is_multistep_ ? cur_ : 0); }
    int npad() const { return npad_; }
    void enable_timing(bool on) { timing_ = on; }
    double kernel_ms() { (void)resolve_timing(); return kernel_ms_; }
    uint64_t kernel_launches() const { return kernel_launches_; }
    // how many of the next k advance() calls would succeed before BoundReached / StepSizeUnderflow
    int64_t steps_available(int64_t k, int *status_after) const;
    bool steps_certain(int64_t k) const;       // neither the bound nor the underflow test can fire in the next k steps (closed form)
    void replay_deferred_time() { for (int64_t s = 0; s < deferred_time_steps_; ++s) time_ = time_ + h_; deferred_time_steps_ = 0; }
    // `count` independent single-workgroup systems advanced by k steps in ONE launch (k_lm_small, one workgroup each):
    // the same as advance(k) on every one of them. Systems that do not qualify (start-up not finished, more than
    // kGangMaxN bodies, sharded, a forced kernel path, a step that would fail) make the call fall back to that.
    static int advance_many(NBodyIntegration *const *igs, int count, int64_t k);
    bool gang_ready(int64_t k) const;

private:
    NBodyIntegration() = default;
    int alloc_buffers();
    int startup_macro_step();                 // one LinearMultistepIntegrator::advance in the start-up regime
    int srkn_step(double h, double *y_slot);  // FixedRungeKuttaIntegrator::advance (SRKN)
    int lm_batch(int64_t k);                  // k steady-state ELM2::advance
    double *Yslot(int s) { return Y_.p + (size_t)s * 3 * npad_; }
    double *Aslot(int s) { return A_.p + (size_t)s * 3 * npad_; }

    int gather_packed(Body4 *buf);            // all-gather the ranks' slices of a packed position buffer
    int gather_stage();                       // same for the AoS staging buffer (get_state / get_acc)

    int device_ = 0;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    // enable_timing: every batch of steps is bracketed by a pair of events that is read LATER (kernel_ms(), or when 1024 pairs
    // wait): waiting for the closing event inside advance() made the call synchronous and cost a 20-step block 2 % of its time
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending_, ev_free_;
    int resolve_timing();
    // the last fused step of a batch left the NEXT step's predicted positions in P_[pp_ ^ 1] and in the ring slot the next step
    // takes (the oldest level's, dead by then): the next batch starts with its force evaluation instead of a predictor launch
    bool predicted_ = false;
    int n_ = 0, npad_ = 0, L_ = 1;
    int lo_ = 0, hi_ = 0, slice_ = 0;         // owned targets [lo_, hi_); slice_ = npad_ / world when sharded
    std::shared_ptr<Exchange> xch_;
    bool is_multistep_ = false;
    Elm2Coeffs lm_{};
    SrknCoeffs rk_{};
    int substeps_ = 1;
    double h_ = 0, h_sub_ = 0;
    double time_ = 0, bound_ = 0;
    uint32_t starter_i_ = 0, lm_i_ = 0;   // SRKN.i (inner RK steps), ELM2.i
    uint64_t evals_ = 0;
    int cur_ = 0, pp_ = 0;
    int path_ = 0;
    bool timing_ = false;
    double kernel_ms_ = 0;
    uint64_t kernel_launches_ = 0;
    SampleArgs samp_{};
    DevBuf<Body4> P_[2];
    DevBuf<double> Y_, A_, V_, ASR_, mu_, stage_;
    DevBuf<double> fast_partial_;             // EPH_PATH_FAST scratch: [S][3][npad] partial sums
    DevBuf<float> posf_;                      // EPH_PATH_F32_PAIRS scratch: the level's positions and mu as 4 floats per body
    int64_t deferred_time_steps_ = 0;         // advance_many: `time = time + h` replays still owed (run after the gang's launch)
    std::vector<LmArgs> *collect_ = nullptr;  // advance_many: lm_batch hands its launch arguments over instead of launching
    DevBuf<LmArgs> gang_args_;                // advance_many: the argument array of the gang this handle leads
    hipEvent_t gang_ev_ = nullptr, gang_copy_ev_ = nullptr;
    LmArgs *gang_host_ = nullptr;             // pinned source of the gang's argument copy (so advance_many need not wait for it)
    size_t gang_host_count_ = 0;
    int pv_ = 0;                              // evaluation order of the point-mass term, fixed at creation (dispatch.cpp)
    int failed_ = EPH_OK;                     // sticky: a gang launch failed with this handle's bookkeeping already advanced
};

// Polynomial<DVec3> (SmallVec<[DVec3; 8]>)   ephemeris/src/trajectory.rs:337-396
struct Polynomial {
    int32_t ncoef = 0;
    double c[kDiv][3] = {};
};

// UniformSpline<DVec3>   ephemeris/src/trajectory.rs:412-633
struct UniformSpline {
    double start = 0, interval = 0;
    std::deque<Polynomial> polynomials;
    // polynomials already pushed as far as the bounds are concerned but still resident on the device (only inside an
    // NBodyPropagator between fits and take_solution; 0 in every Solution handed out)
    uint64_t ghost = 0;
    uint64_t len() const { return (uint64_t)polynomials.size() + ghost; }
    double span() const { return interval * (double)len(); }   // interval.scaled(len)
    double end() const { return start + span(); }
    void push_back(const Polynomial &p) { polynomials.push_back(p); }
    void push_front(const Polynomial &p) {
        polynomials.push_front(p);
        start -= interval;
    }
    // get_index_local / get_index_local_exclusive  trajectory.rs:591-617 (`as usize` saturates; NaN -> 0). false = None
    static uint64_t to_usize(double x) {
        if (!(x > 0.0)) return 0;
        return x >= 18446744073709551616.0 ? ~0ull : (uint64_t)x;
    }
    bool get_index_local(double time, uint64_t *idx) const {
        if (std::signbit(time) || time >= span()) return false;
        *idx = to_usize(time / interval);
        return true;
    }
    bool get_index_local_exclusive(double time, uint64_t *idx) const {
        if (std::signbit(time) || time > span()) return false;
        const uint64_t c = to_usize(std::ceil(time / interval));
        *idx = c == 0 ? 0 : c - 1;                        // saturating_sub(1)
        return true;
    }
    void clear_before(double at) {                        // trajectory.rs:536-542
        uint64_t idx;
        if (!get_index_local_exclusive((at + interval) - start, &idx)) return;
        start += interval * (double)idx;
        const uint64_t k = std::min<uint64_t>(idx, polynomials.size());   // drain(0..idx) (idx <= len here)
        polynomials.erase(polynomials.begin(), polynomials.begin() + (std::ptrdiff_t)k);
    }
    void clear_after(double at) {                         // trajectory.rs:544-549
        uint64_t idx;
        if (!get_index_local(at - start, &idx)) return;
        if (idx < polynomials.size()) polynomials.resize((size_t)idx);       // truncate(idx)
    }
    bool between(double from, double to, UniformSpline *out) const {        // trajectory.rs:484-502
        if (polynomials.empty()) return false;
        uint64_t a, b;
        if (!get_index_local_exclusive(from - start, &a) || !get_index_local_exclusive(to - start, &b)) return false;
        out->start = start + interval * (double)a;
        out->interval = interval;
        out->polynomials.clear();
        for (uint64_t i = a; i < b + 1 && b + 1 != 0; ++i)                   // (start..end + 1).filter_map(get)
            if (i < polynomials.size()) out->polynomials.push_back(polynomials[(size_t)i]);
        return true;
    }
};

struct Solution {   // Vec<UniformSpline<DVec3>>
    std::vector<UniformSpline> splines;
};

// SplineInterpolator   ephemeris/src/propagators/nbody.rs:309-323
struct SplineInterpolator {
    double last_sample_time = 0, sample_period = 0;   // Durations
    uint32_t len = 1;       // PolyonmialInterpolator.index
    uint32_t degree = 0;    // LeastSquaresFit.degree
    // derived: after how many `last_sample_time += delta` the `== sample_period` test fires (0 = never),
    // and how many additions have been made since the last reset
    uint32_t period_steps = 0, phase = 0;
    double time() const { return last_sample_time + sample_period * (double)(len > 0 ? len - 1 : 0); }
};

// NBodyPropagator<D, DVec3, M, SplineInterpolators<D, DVec3, LeastSquaresFit>>
class NBodyPropagator {
public:
    static int create(int n, const double *pos, const double *vel, const double *mu, double t0, double dt,
                      int direction, const char *method, const uint32_t *count, const uint32_t *degree,
                      std::unique_ptr<NBodyPropagator> *out);
    int clone(std::unique_ptr<NBodyPropagator> *out);
    int step_n(int64_t k);                    // k x IncrementalPropagator::step
    // step_n(k) on every propagator of `ps`, the steady-state steps of all of them in shared launches (advance_many)
    static int step_n_many(NBodyPropagator *const *ps, int count, int64_t k);
    int step_to(double t);
    // ONE IncrementalPropagator::step, executed lazily: the reference's callers step in a loop and look at time() /
    // has_reached() after every step (ephemeris_explorer/src/prediction.rs:422-443). Everything those two return is a
    // function of the number of steps taken (sampling counters and spline bounds), not of the numerical state, so a
    // steady-state step only advances that bookkeeping on the host and is queued; the queue runs as one device batch when
    // somebody needs data (take_solution, clone, state, step_n / step_to) or it reaches kDeferMax steps.
    int step_deferred();
    int flush();
    // run the queued steps and bring the device-resident polynomials into the host splines: afterwards the host
    // Solution is complete (eph_prop_shard needs that before the sharded branch starts pushing newer windows)
    int settle() { const int st = flush(); return st ? st : materialize(); }
    double time() const;                      // DirectionalSolout::solution_time
    bool has_reached(double t) const;
    int take_solution(std::unique_ptr<Solution> *out);
    NBodyIntegration *integration() { return integ_.get(); }
    int64_t deferred() const { return deferred_; }

private:
    NBodyPropagator() = default;
    Solution new_solution() const;
    int run_batch(int64_t k);
    int batch_begin();                         // run_batch up to the integrator's advance: sampling schedule on the device
    int64_t steps_until_reached(double t, int64_t cap) const;
    double bound_of(const UniformSpline &s) const { return direction_ > 0 ? s.end() : s.start; }
    int ensure_log_capacity(const std::vector<uint64_t> &need);

    std::unique_ptr<NBodyIntegration> integ_;
    int direction_ = 1;
    double delta_ = 0;
    int64_t kmax_ = 1;          // steps per device batch (bounded by the sample-log budget)
    std::vector<SplineInterpolator> interp_;
    Solution solution_;
    // device-side sample logs: body b owns log[log_off_[b] .. log_off_[b] + log_cap_[b]) samples
    DevBuf<double> log_;
    std::vector<uint64_t> log_off_, log_cap_;
    DevBuf<uint32_t> d_period_, d_phase_;
    DevBuf<uint64_t> d_offset_;
    // run_batch scratch (grow-only, reused by every batch) and the sticky status of a batch that failed half-way
    DevBuf<uint64_t> d_first_, d_region_;
    DevBuf<uint8_t> d_deg_;
    DevBuf<double> d_co_, d_all_;
    DevBuf<int32_t> d_nc_;
    DevBuf<uint32_t> d_src_, d_cnt_;
    int failed_ = EPH_OK;
    int fit_and_push(int64_t done, hipStream_t s);
    // Fitted polynomials stay on the device until somebody takes the solution: a batch appends its windows (body-major)
    // to pend_co_ / pend_nc_ and only the bounds (UniformSpline::start, ::ghost) move on the host, with the reference's
    // own f64 operations. materialize() downloads them and performs the push_back / push_front of every window.
    // deferred steps (step_deferred): how many, and the bookkeeping as it will be once they have run
    static constexpr int64_t kDeferMax = 8192;
    int64_t deferred_ = 0;
    double sh_time_ = 0;                                  // integrator time
    std::vector<uint32_t> sh_phase_, sh_len_;             // SplineInterpolator counters
    std::vector<double> sh_start_;                        // spline starts
    std::vector<uint64_t> sh_npoly_;                      // spline lengths
    double bound_at(size_t b) const {                     // bound of spline b, deferred steps included
        const UniformSpline &s = solution_.splines[b];
        if (deferred_ == 0) return bound_of(s);
        return direction_ > 0 ? sh_start_[b] + s.interval * (double)sh_npoly_[b] : sh_start_[b];
    }
    DevBuf<double> pend_co_;                  // [pend_cap_][kDiv][3]
    DevBuf<int32_t> pend_nc_;
    size_t pend_count_ = 0, pend_cap_ = 0;
    // device-side budget of the pending list (196 B per window): beyond it the list is spilled to the host splines
    // (materialize) instead of growing; a failed device allocation does the same instead of failing the run
    static constexpr size_t kPendMaxWindows = (size_t)1 << 23;   // 1.6 GB
    std::vector<std::vector<uint32_t>> pend_batches_;   // per batch: windows per body
    int reserve_pending(size_t extra, hipStream_t s);
    int materialize();
};

}  // namespace eph
struct eph_nbody {      // the C ABI's opaque Integration handle (a view when it belongs to a propagator)
    std::unique_ptr<eph::NBodyIntegration> own;
    eph::NBodyIntegration *p = nullptr;
    // a view's propagator: eph_prop_step queues steps on the host (NBodyPropagator::step_deferred), so every eph_nbody_*
    // call through the view runs that queue first -- the view never shows a state the propagator's callers have moved past
    eph::NBodyPropagator *owner = nullptr;
    int settle() const { return owner ? owner->flush() : 0; }
};
struct eph_solution {   // the C ABI's opaque Vec<UniformSpline<DVec3>>
    eph::Solution s;
};
namespace eph {

// device evaluation of one UniformSpline at many epochs
int spline_eval_device(const UniformSpline &s, int64_t m, const double *at, double *pos, double *vel, uint8_t *inside);
int least_squares_fit_device(int degree, int backward, int64_t nwin, const double *samples, double *coeffs,
                             int32_t *ncoef);
int accel_eval_device(int n, const double *pos, const double *mu, double *acc);

}  // namespace eph
