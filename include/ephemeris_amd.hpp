// ephemeris_amd.hpp -- the reference's operator surface in C++, header-only, over the C ABI of ephemeris_amd.h.
//
// The reference (Canleskis/ephemeris-explorer) is Rust and the image has no Rust toolchain, so the host side above the C boundary
// that CAN be compiled here is this one: the same type and method names, argument meaning and error behaviour as the reference's
// traits, for C++ callers and as the compiled counterpart of INTEGRATION.md's shims (paths relative to the reference root):
//
//   integration::StepError, Integrator::advance            integration/src/lib.rs:139-169,315-331       -> StepError, NBodyIntegration::advance
//   Propagator / IncrementalPropagator / DirectionalPropagator / BoundedPropagator
//                                                           ephemeris/src/lib.rs:9-79                     -> NBodyPropagator::{step, step_n, step_to, time, has_reached, take_solution, propagate}
//   Vec<UniformSpline<DVec3>>, EvaluateTrajectory           ephemeris/src/trajectory.rs:337-633           -> Solution::{start, interval, len, state_vector, position, append}
//   NewtonianGravity::eval (SecondOrderODE)                 ephemeris/src/propagators/nbody.rs:16-39      -> newtonian_gravity_eval
//   SpacecraftPropagator + CubicHermiteSplineSolout         ephemeris/src/propagators/spacecraft.rs:415-695 -> SpacecraftBatch (n propagators at once)
//   AdaptiveMethodParams, INITIAL_ADAPTIVE_PARAMS           ephemeris_explorer/src/load/mod.rs:472-486    -> AdaptiveParams
//
// Errors: what the reference returns as `Err(StepError::..)` comes back as a StepError VALUE (Result-like: `if (auto e = p.step())`);
// what would be a panic or has no counterpart (bad arguments, no device, HIP failures) throws ephemeris_amd::Error. There is no CPU
// fallback: without a gfx950 device every compute call throws Error{EPH_ERR_NO_DEVICE}.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ephemeris_amd.h"

namespace ephemeris_amd {

using DVec3 = std::array<double, 3>;

// a library / device failure (negative status of the C ABI)
struct Error : std::runtime_error {
    int32_t status;
    Error(int32_t st, const char *where)
        : std::runtime_error(std::string(where) + ": " + eph_status_string(st) + " [" + eph_last_error() + "]"), status(st) {}
};

// integration::StepError (lib.rs:315-331). `None` = Ok(()).
enum class StepError : int32_t {
    None = EPH_OK,
    StepSizeUnderflow = 1,
    MaxIterationsReached = 2,
    BoundReached = 3,
    EvalFailed = 4,
    Solout = 5,   // NBodyPropagatorError::Solout / SpacecraftPropagatorError::Solout
};
inline const char *to_string(StepError e) { return eph_status_string(static_cast<int32_t>(e)); }

namespace detail {
inline StepError step_result(int32_t st, const char *where) {
    if (st < 0) throw Error(st, where);
    return static_cast<StepError>(st);
}
inline void check(int32_t st, const char *where) {
    if (st != EPH_OK) throw Error(st, where);
}
inline const double *flat(const std::vector<DVec3> &v) { return v.empty() ? nullptr : v.front().data(); }
inline double *flat(std::vector<DVec3> &v) { return v.empty() ? nullptr : v.front().data(); }
}  // namespace detail

enum class Direction : int32_t { Forward = EPH_FORWARD, Backward = EPH_BACKWARD };

inline int32_t device_count() {
    int32_t n = 0;
    detail::check(eph_device_count(&n), "eph_device_count");
    return n;
}
// the evaluation order of the `particular` point-mass term new handles take (0..6)
inline void set_pair_variant(int32_t k) { detail::check(eph_set_pair_variant(k), "eph_set_pair_variant"); }

// SecondOrderODE::eval for NewtonianGravity: ddy[i] += sum over the other bodies, reference summation order (nbody.rs:22-38)
inline void newtonian_gravity_eval(const std::vector<DVec3> &y, const std::vector<double> &gravitational_parameters, std::vector<DVec3> &ddy) {
    if (y.size() != gravitational_parameters.size() || ddy.size() != y.size()) throw std::invalid_argument("newtonian_gravity_eval: sizes differ");
    detail::check(eph_accel_eval(static_cast<int32_t>(y.size()), detail::flat(y), gravitational_parameters.data(), detail::flat(ddy)), "eph_accel_eval");
}

// StateVector<DVec3> of one body at one epoch
struct StateVector {
    DVec3 position, velocity;
};

// Vec<UniformSpline<DVec3>>: what NBodyPropagator::take_solution hands over
class Solution {
public:
    explicit Solution(eph_solution *h) : h_(h) {}
    Solution(Solution &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Solution &operator=(Solution &&o) noexcept {
        if (this != &o) { reset(); h_ = std::exchange(o.h_, nullptr); }
        return *this;
    }
    Solution(const Solution &) = delete;
    Solution &operator=(const Solution &) = delete;
    ~Solution() { reset(); }

    int32_t bodies() const {
        int32_t n = 0;
        detail::check(eph_solution_bodies(h_, &n), "eph_solution_bodies");
        return n;
    }
    // UniformSpline::{start, interval, len}
    double start(int32_t body) const { return info(body).start; }
    double interval(int32_t body) const { return info(body).interval; }
    int64_t len(int32_t body) const { return info(body).npoly; }
    double end(int32_t body) const {          // start + interval * len, the f64 expression of trajectory.rs:489-491
        const Info i = info(body);
        return i.start + i.interval * static_cast<double>(i.npoly);
    }
    // EvaluateTrajectory::state_vector: nullopt-like -- `inside == false` where the reference returns None
    bool state_vector(int32_t body, double at, StateVector &out) const {
        uint8_t inside = 0;
        detail::check(eph_solution_eval(h_, body, 1, &at, out.position.data(), out.velocity.data(), &inside), "eph_solution_eval");
        return inside != 0;
    }
    bool position(int32_t body, double at, DVec3 &out) const {
        uint8_t inside = 0;
        detail::check(eph_solution_eval(h_, body, 1, &at, out.data(), nullptr, &inside), "eph_solution_eval");
        return inside != 0;
    }
    // many epochs of one body in one launch (what the app's plotting does): returns the `inside` flags
    std::vector<uint8_t> positions(int32_t body, const std::vector<double> &at, std::vector<DVec3> &out) const {
        out.resize(at.size());
        std::vector<uint8_t> inside(at.size());
        detail::check(eph_solution_eval(h_, body, static_cast<int64_t>(at.size()), at.data(), detail::flat(out), nullptr, inside.data()), "eph_solution_eval");
        return inside;
    }
    // UniformSpline::append / prepend for every body (asserts contiguity in the reference: bad argument here)
    void append(const Solution &tail, Direction d = Direction::Forward) {
        detail::check(eph_solution_append(h_, tail.h_, static_cast<int32_t>(d)), "eph_solution_append");
    }
    // UniformSpline::clear_before / clear_after at `at`, on every body's spline (trajectory.rs:536-549)
    void clear_before(double at) { detail::check(eph_solution_clear(h_, -1, at, 0), "eph_solution_clear"); }
    void clear_after(double at) { detail::check(eph_solution_clear(h_, -1, at, 1), "eph_solution_clear"); }
    // UniformSpline::between for every body; `ok == false` where the reference returns None for any of them
    Solution between(double from, double to, bool &ok) const {
        eph_solution *o = nullptr;
        detail::check(eph_solution_between(h_, from, to, &o), "eph_solution_between");
        ok = o != nullptr;
        return Solution(o);
    }
    // the Polynomials of one body: coefficient k of polynomial p at coeffs[(p*8 + k)*3 + c], lengths after trim() in ncoef
    void polynomials(int32_t body, std::vector<double> &coeffs, std::vector<int32_t> &ncoef) const {
        const int64_t np = len(body);
        coeffs.assign(static_cast<size_t>(np) * 24, 0.0);
        ncoef.assign(static_cast<size_t>(np), 0);
        if (np == 0) return;                              // (a snapshot taken before the body's first polynomial was complete)
        detail::check(eph_solution_coeffs(h_, body, coeffs.data(), ncoef.data()), "eph_solution_coeffs");
    }
    explicit operator bool() const { return h_ != nullptr; }
    eph_solution *raw() const { return h_; }

private:
    struct Info { double start, interval; int64_t npoly; };
    Info info(int32_t body) const {
        Info i{};
        detail::check(eph_solution_info(h_, body, &i.start, &i.interval, &i.npoly), "eph_solution_info");
        return i;
    }
    void reset() {
        if (h_) eph_solution_destroy(h_);
        h_ = nullptr;
    }
    eph_solution *h_;
};

// Integration<NBodyProblem<DVec3>, M> without a solout: M::new(FixedMethodParams::new(h)).integrate(problem)
class NBodyIntegration {
public:
    NBodyIntegration(const std::vector<DVec3> &y, const std::vector<DVec3> &dy, const std::vector<double> &gravitational_parameters, double time,
                     double h, const char *method = "QuinlanTremaine12") : n_(static_cast<int32_t>(y.size())) {
        if (dy.size() != y.size() || gravitational_parameters.size() != y.size()) throw std::invalid_argument("NBodyIntegration: sizes differ");
        detail::check(eph_nbody_create(n_, detail::flat(y), detail::flat(dy), gravitational_parameters.data(), time, h, method, &h_), "eph_nbody_create");
    }
    NBodyIntegration(NBodyIntegration &&o) noexcept : h_(std::exchange(o.h_, nullptr)), n_(o.n_) {}
    NBodyIntegration &operator=(NBodyIntegration &&o) noexcept {
        if (this != &o) { if (h_) eph_nbody_destroy(h_); h_ = std::exchange(o.h_, nullptr); n_ = o.n_; }
        return *this;
    }
    NBodyIntegration(const NBodyIntegration &) = delete;
    NBodyIntegration &operator=(const NBodyIntegration &) = delete;
    ~NBodyIntegration() { if (h_) eph_nbody_destroy(h_); }

    StepError advance(int64_t n_steps = 1) { return detail::step_result(eph_nbody_advance(h_, n_steps), "eph_nbody_advance"); }   // Integrator::advance
    void set_bound(double bound) { detail::check(eph_nbody_set_bound(h_, bound), "eph_nbody_set_bound"); }
    NBodyIntegration clone() const {
        eph_nbody *c = nullptr;
        detail::check(eph_nbody_clone(h_, &c), "eph_nbody_clone");
        return NBodyIntegration(c, n_);
    }
    // problem.{time, state.y, state.dy}, IntegratorState::step_count
    double state(std::vector<DVec3> &y, std::vector<DVec3> &dy, uint32_t *step_count = nullptr) const {
        y.resize(static_cast<size_t>(n_));
        dy.resize(static_cast<size_t>(n_));
        double t = 0.0;
        detail::check(eph_nbody_get_state(h_, detail::flat(y), detail::flat(dy), &t, step_count), "eph_nbody_get_state");
        return t;
    }
    int32_t bodies() const { return n_; }

private:
    NBodyIntegration(eph_nbody *h, int32_t n) : h_(h), n_(n) {}
    eph_nbody *h_ = nullptr;
    int32_t n_ = 0;
};

// NBodyPropagator<D, DVec3, M, SplineInterpolators<D, DVec3, LeastSquaresFit>>  (nbody.rs:65-235; celestial.rs:156-186)
class NBodyPropagator {
public:
    // count[b] / degree[b]: ephemeris.json's samples-per-polynomial divisor and fit degree of body b (load/mod.rs:313-330)
    NBodyPropagator(const std::vector<DVec3> &y, const std::vector<DVec3> &dy, const std::vector<double> &gravitational_parameters, double time,
                    double delta, Direction direction, const std::vector<uint32_t> &count, const std::vector<uint32_t> &degree,
                    const char *method = "QuinlanTremaine12") : n_(static_cast<int32_t>(y.size())) {
        if (dy.size() != y.size() || gravitational_parameters.size() != y.size() || count.size() != y.size() || degree.size() != y.size())
            throw std::invalid_argument("NBodyPropagator: sizes differ");
        detail::check(eph_prop_create(n_, detail::flat(y), detail::flat(dy), gravitational_parameters.data(), time, delta,
                                      static_cast<int32_t>(direction), method, count.data(), degree.data(), &h_), "eph_prop_create");
    }
    NBodyPropagator(NBodyPropagator &&o) noexcept : h_(std::exchange(o.h_, nullptr)), n_(o.n_) {}
    NBodyPropagator &operator=(NBodyPropagator &&o) noexcept {             // `propagator = snapshot` (prediction.rs:378)
        if (this != &o) { if (h_) eph_prop_destroy(h_); h_ = std::exchange(o.h_, nullptr); n_ = o.n_; }
        return *this;
    }
    NBodyPropagator(const NBodyPropagator &) = delete;
    NBodyPropagator &operator=(const NBodyPropagator &) = delete;
    ~NBodyPropagator() { if (h_) eph_prop_destroy(h_); }

    StepError step() { return detail::step_result(eph_prop_step(h_), "eph_prop_step"); }                          // IncrementalPropagator::step
    StepError step_n(int64_t n) { return detail::step_result(eph_prop_step_n(h_, n), "eph_prop_step_n"); }
    StepError step_to(double time) { return detail::step_result(eph_prop_step_to(h_, time), "eph_prop_step_to"); }   // lib.rs:49-60
    double time() const {                                                                                         // DirectionalPropagator::time
        double t = 0.0;
        detail::check(eph_prop_time(h_, &t), "eph_prop_time");
        return t;
    }
    bool has_reached(double time) const {
        int32_t f = 0;
        detail::check(eph_prop_has_reached(h_, time, &f), "eph_prop_has_reached");
        return f != 0;
    }
    Solution take_solution() {                                                                                    // Propagator::take_solution
        eph_solution *s = nullptr;
        detail::check(eph_prop_take_solution(h_, &s), "eph_prop_take_solution");
        return Solution(s);
    }
    // BoundedPropagator::propagate: step_to(to), then take_solution; the StepError (if any) in `error`
    Solution propagate(double to, StepError *error = nullptr) {
        eph_solution *s = nullptr;
        const StepError e = detail::step_result(eph_prop_propagate(h_, to, &s), "eph_prop_propagate");
        if (error) *error = e;
        return Solution(s);
    }
    NBodyPropagator clone() const {
        eph_prop *c = nullptr;
        detail::check(eph_prop_clone(h_, &c), "eph_prop_clone");
        return NBodyPropagator(c, n_);
    }
    int32_t bodies() const { return n_; }

private:
    NBodyPropagator(eph_prop *h, int32_t n) : h_(h), n_(n) {}
    eph_prop *h_ = nullptr;
    int32_t n_ = 0;
};

// integration::AdaptiveMethodParams with the app's INITIAL_ADAPTIVE_PARAMS as defaults (load/mod.rs:472-486)
struct AdaptiveParams : eph_adaptive_params {
    explicit AdaptiveParams(double tolerance = 1e-3) {
        h_init = 60.0; h_max = 1.7976931348623157e308; tol_position = tolerance; tol_velocity = tolerance;
        fac_min = 1.0 / 5.0; fac_max = 5.0 / 1.0; fac = 9.0 / 10.0; n_max = 1000000u;
    }
};

// Timeline segment with a burn: constant acceleration in the inertial frame (reference = -1) or in the TNB frame relative to a body
struct Burn {
    double start, end;
    DVec3 acceleration;
    int32_t reference = -1;
};

// CubicHermiteSpline<DVec3> (trajectory.rs:698-855): a craft's trajectory as its knots
struct CubicHermiteSpline {
    std::vector<double> t;
    std::vector<DVec3> position, velocity;

    size_t len() const { return t.size(); }
    bool is_empty() const { return t.empty(); }
    double start() const { return t.front(); }
    double end() const { return t.back(); }
    // EvaluateTrajectory::state_vector (trajectory.rs:766-797), evaluated on the device; false where the reference returns None
    bool state_vector(double at, StateVector &out) const {
        uint8_t inside = 0;
        detail::check(eph_hermite_eval(static_cast<int64_t>(t.size()), t.data(), detail::flat(position), detail::flat(velocity), 1, &at,
                                       out.position.data(), out.velocity.data(), &inside), "eph_hermite_eval");
        return inside != 0;
    }
    // SpacecraftPropagator::join(lhs = *this, rhs): clear_after(rhs.start()) then extend(rhs)  (spacecraft.rs:558-561)
    void join(const CubicHermiteSpline &rhs) {
        const int64_t cap = static_cast<int64_t>(t.size() + rhs.t.size());
        std::vector<double> ot(static_cast<size_t>(cap));
        std::vector<DVec3> op(static_cast<size_t>(cap)), ov(static_cast<size_t>(cap));
        int64_t n = 0;
        detail::check(eph_hermite_join(static_cast<int64_t>(t.size()), t.data(), detail::flat(position), detail::flat(velocity),
                                       static_cast<int64_t>(rhs.t.size()), rhs.t.data(), detail::flat(rhs.position), detail::flat(rhs.velocity), cap,
                                       ot.data(), detail::flat(op), detail::flat(ov), &n), "eph_hermite_join");
        ot.resize(static_cast<size_t>(n)); op.resize(static_cast<size_t>(n)); ov.resize(static_cast<size_t>(n));
        t.swap(ot); position.swap(op); velocity.swap(ov);
    }
};

// SoiTransitions / Apsides of the app's SpacecraftSolution (dynamics/spacecraft.rs:303-451)
struct SoiTransitions {
    std::vector<double> time;
    std::vector<int32_t> body;
};
enum class ApsisKind : int32_t { Periapsis = 0, Apoapsis = 1 };
struct Apsides {
    std::vector<double> time, distance;
    std::vector<int32_t> body, kind;
};

// Timeline::divergence_time_before (spacecraft.rs:179-213): the epoch a flight plan edited from `old_burns` to `new_burns` restarts from
inline double divergence_time_before(const std::vector<Burn> &old_burns, const std::vector<Burn> &new_burns, double before);

// `Bodies`: the massive bodies' splines resident on the device (dynamics/spacecraft.rs:164-228). LIVE like the reference's context --
// GravitationalBody.trajectory is Trajectory(Arc<RwLock<PredictionTrajectory>>) (dynamics/spacecraft.rs:52-74, dynamics/mod.rs:84-85):
// merged N-body snapshots grow it and every SpacecraftBatch (and clone) bound to it sees the new extent at its next call. The one
// wrapper that may be shared between threads (it carries the RwLock); hold it in a std::shared_ptr where the app holds the Arc.
class Ephemeris {
public:
    Ephemeris(const Solution &splines, const std::vector<double> &gravitational_parameters) {
        if (gravitational_parameters.size() != static_cast<size_t>(splines.bodies()))
            throw std::invalid_argument("Ephemeris: one gravitational parameter per body");
        detail::check(eph_ephemeris_create(splines.raw(), gravitational_parameters.data(), &h_), "eph_ephemeris_create");
    }
    Ephemeris(const Ephemeris &) = delete;
    Ephemeris &operator=(const Ephemeris &) = delete;
    Ephemeris(Ephemeris &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Ephemeris &operator=(Ephemeris &&o) noexcept {
        if (this != &o) { if (h_) eph_ephemeris_destroy(h_); h_ = std::exchange(o.h_, nullptr); }
        return *this;
    }
    ~Ephemeris() { if (h_) eph_ephemeris_destroy(h_); }
    eph_ephemeris *raw() const { return h_; }

    // UniformSpline::append / prepend for every body (trajectory.rs:515-534); the reference's assert_eq! becomes std::invalid_argument,
    // the table untouched
    void append(const Solution &tail, Direction d = Direction::Forward) { grow(eph_ephemeris_append(h_, tail.raw(), static_cast<int32_t>(d)), "eph_ephemeris_append"); }
    // PredictionTarget::merge for the bodies: clear_after(propagated.start()) + append (dynamics/celestial.rs:198-204), Backward:
    // clear_before(propagated.end()) + prepend (:220-226) -- what the app calls with every snapshot the N-body task sends
    void merge(const Solution &propagated, Direction d = Direction::Forward) { grow(eph_ephemeris_merge(h_, propagated.raw(), static_cast<int32_t>(d)), "eph_ephemeris_merge"); }
    void clear_before(double at, int32_t body = -1) { detail::check(eph_ephemeris_clear(h_, body, at, 0), "eph_ephemeris_clear"); }   // trajectory.rs:536-542
    void clear_after(double at, int32_t body = -1) { detail::check(eph_ephemeris_clear(h_, body, at, 1), "eph_ephemeris_clear"); }    // :544-549
    // Bodies::is_valid_at (dynamics/spacecraft.rs:199-201): what flight_plan.rs:363-395 tests before it restarts a ship's prediction
    bool is_valid_at(double t) const {
        int32_t f = 0;
        detail::check(eph_ephemeris_is_valid_at(h_, t, &f), "eph_ephemeris_is_valid_at");
        return f != 0;
    }
    struct Info { double start, interval; int64_t npoly; };           // UniformSpline{start, interval, polynomials.len()} as it is now
    Info info(int32_t body) const {
        Info i{};
        detail::check(eph_ephemeris_info(h_, body, &i.start, &i.interval, &i.npoly, nullptr), "eph_ephemeris_info");
        return i;
    }
    uint64_t revision() const {
        uint64_t r = 0;
        detail::check(eph_ephemeris_info(h_, -1, nullptr, nullptr, nullptr, &r), "eph_ephemeris_info");
        return r;
    }
    // one contiguous image of the table (what rank 0 of a multi-GPU sweep broadcasts) and a table built from one
    std::vector<unsigned char> export_image() const {
        uint64_t need = 0;
        (void)eph_ephemeris_export(h_, nullptr, 0, &need);
        std::vector<unsigned char> buf(static_cast<size_t>(need));
        detail::check(eph_ephemeris_export(h_, buf.data(), need, &need), "eph_ephemeris_export");
        return buf;
    }
    static Ephemeris from_image(const std::vector<unsigned char> &image) {
        eph_ephemeris *h = nullptr;
        detail::check(eph_ephemeris_import(image.data(), image.size(), &h), "eph_ephemeris_import");
        return Ephemeris(h);
    }

private:
    explicit Ephemeris(eph_ephemeris *h) : h_(h) {}
    static void grow(int32_t st, const char *what) {
        if (st == EPH_ERR_BAD_ARGUMENT) throw std::invalid_argument(std::string(what) + ": not contiguous (trajectory.rs:517-518,530-531)");
        detail::check(st, what);
    }
    eph_ephemeris *h_ = nullptr;
};

// n x SpacecraftPropagator<[StateVector; 1], ReferenceFrame, Bodies, <adaptive ERK pair>, CubicHermiteSplineSolout>: the batch form of
// INTEGRATION.md 4b (a batch of one is the drop-in for the app's single propagator). Per-craft outcomes are StepError values.
class SpacecraftBatch {
public:
    SpacecraftBatch(const Ephemeris &bodies, double initial_time, const std::vector<StateVector> &initial_states, const char *method = "Verner87",
                    const AdaptiveParams &params = AdaptiveParams(), const std::vector<std::vector<Burn>> &timelines = {}, int32_t max_knots = 4096)
        : n_(static_cast<int64_t>(initial_states.size())) {
        std::vector<double> t0(initial_states.size(), initial_time), pos, vel;
        for (const StateVector &sv : initial_states) {
            pos.insert(pos.end(), sv.position.begin(), sv.position.end());
            vel.insert(vel.end(), sv.velocity.begin(), sv.velocity.end());
        }
        std::vector<int64_t> off(initial_states.size() + 1, 0);
        std::vector<double> bs{0.0}, be{0.0}, ba{0.0, 0.0, 0.0};
        std::vector<int32_t> br{0};
        if (!timelines.empty()) {
            if (timelines.size() != initial_states.size()) throw std::invalid_argument("SpacecraftBatch: one timeline per craft");
            bs.clear(); be.clear(); ba.clear(); br.clear();
            for (size_t i = 0; i < timelines.size(); ++i) {
                for (const Burn &b : timelines[i]) {
                    bs.push_back(b.start); be.push_back(b.end); br.push_back(b.reference);
                    ba.insert(ba.end(), b.acceleration.begin(), b.acceleration.end());
                }
                off[i + 1] = static_cast<int64_t>(bs.size());
            }
            if (bs.empty()) { bs = {0.0}; be = {0.0}; ba = {0.0, 0.0, 0.0}; br = {0}; }
        }
        detail::check(eph_craft_batch_create(bodies.raw(), n_, t0.data(), pos.data(), vel.data(), method, &params, off.data(), bs.data(), be.data(),
                                             ba.data(), br.data(), max_knots, &h_), "eph_craft_batch_create");
    }
    SpacecraftBatch(const SpacecraftBatch &) = delete;
    SpacecraftBatch &operator=(const SpacecraftBatch &) = delete;
    ~SpacecraftBatch() { if (h_) eph_craft_batch_destroy(h_); }

    void step_to(double time) { detail::check(eph_craft_batch_propagate(h_, time), "eph_craft_batch_propagate"); }   // every craft: IncrementalPropagator::step_to
    void step(uint32_t n_steps = 1) { detail::check(eph_craft_batch_step_n(h_, n_steps), "eph_craft_batch_step_n"); }
    // A craft whose last step returned a StepError stays put until the batch is re-armed: the next step_to / step then steps it again,
    // exactly as the reference's next step() on that propagator would (runge_kutta/mod.rs:414-439) -- how a stored ship propagator
    // resumes once the bodies' ephemeris has grown (prediction.rs:378)
    void retry_failed() { detail::check(eph_craft_batch_retry_failed(h_), "eph_craft_batch_retry_failed"); }
    // per craft: Ok / the StepError its propagator returned (EPH_KNOTS_FULL = 6: drain the knots and resume)
    std::vector<int32_t> status(std::vector<int32_t> *nknots = nullptr) const {
        std::vector<int32_t> st(static_cast<size_t>(n_)), nk(static_cast<size_t>(n_));
        std::vector<uint32_t> at(static_cast<size_t>(n_)), sp(static_cast<size_t>(n_));
        detail::check(eph_craft_batch_status(h_, st.data(), nk.data(), at.data(), sp.data()), "eph_craft_batch_status");
        if (nknots) *nknots = nk;
        return st;
    }
    // CubicHermiteSpline of one craft: knot times, positions, velocities (trajectory.rs:698-855)
    void knots(int64_t craft, int32_t nknots, std::vector<double> &t, std::vector<DVec3> &position, std::vector<DVec3> &velocity) const {
        t.resize(static_cast<size_t>(nknots));
        position.resize(static_cast<size_t>(nknots));
        velocity.resize(static_cast<size_t>(nknots));
        detail::check(eph_craft_batch_knots(h_, craft, t.data(), detail::flat(position), detail::flat(velocity)), "eph_craft_batch_knots");
    }
    CubicHermiteSpline trajectory(int64_t craft) const {
        if (craft < 0 || craft >= n_) throw std::invalid_argument("SpacecraftBatch::trajectory: craft index");
        std::vector<int32_t> nk;
        (void)status(&nk);
        CubicHermiteSpline sp;
        knots(craft, nk[static_cast<size_t>(craft)], sp.t, sp.position, sp.velocity);
        return sp;
    }
    // problem.{time, state} and the controller's next step size of every craft
    void state(std::vector<double> &t, std::vector<DVec3> &position, std::vector<DVec3> &velocity, std::vector<double> *next_h = nullptr) const {
        const size_t n = static_cast<size_t>(n_);
        t.resize(n); position.resize(n); velocity.resize(n);
        if (next_h) next_h->resize(n);
        detail::check(eph_craft_batch_state(h_, t.data(), detail::flat(position), detail::flat(velocity), next_h ? next_h->data() : nullptr), "eph_craft_batch_state");
    }
    // the app's SpacecraftSolout (SOI transitions + apsides) instead of the library's CubicHermiteSplineSolout; before the first step
    void enable_events(const std::vector<double> &soi_radius, int32_t max_transitions = 64, int32_t max_apsides = 256) {
        detail::check(eph_craft_batch_enable_events(h_, soi_radius.data(), max_transitions, max_apsides), "eph_craft_batch_enable_events");
    }
    // one craft's event lists; returns EPH_OK or EPH_EVENTS_FULL (read, reset_events(), resume)
    int32_t events(int64_t craft, SoiTransitions &transitions, Apsides &apsides) const {
        if (craft < 0 || craft >= n_) throw std::invalid_argument("SpacecraftBatch::events: craft index");
        const size_t n = static_cast<size_t>(n_), c = static_cast<size_t>(craft);
        std::vector<int32_t> ntr(n), nap(n), st(n);
        detail::check(eph_craft_batch_event_counts(h_, ntr.data(), nap.data(), st.data()), "eph_craft_batch_event_counts");
        transitions.time.assign(static_cast<size_t>(ntr[c]), 0.0);
        transitions.body.assign(static_cast<size_t>(ntr[c]), 0);
        apsides.time.assign(static_cast<size_t>(nap[c]), 0.0);
        apsides.distance.assign(static_cast<size_t>(nap[c]), 0.0);
        apsides.body.assign(static_cast<size_t>(nap[c]), 0);
        apsides.kind.assign(static_cast<size_t>(nap[c]), 0);
        detail::check(eph_craft_batch_events(h_, craft, transitions.time.data(), transitions.body.data(), apsides.time.data(), apsides.distance.data(),
                                             apsides.body.data(), apsides.kind.data()), "eph_craft_batch_events");
        return st[c];
    }
    void reset_knots() { detail::check(eph_craft_batch_reset_knots(h_), "eph_craft_batch_reset_knots"); }      // drain point of a long propagation
    void reset_events() { detail::check(eph_craft_batch_reset_events(h_), "eph_craft_batch_reset_events"); }
    // the order Bodies::acceleration adds the massive bodies' terms in (an EntityHashMap upstream: unspecified); default = table order
    void set_body_order(const std::vector<int32_t> &order) { detail::check(eph_craft_batch_set_body_order(h_, order.data()), "eph_craft_batch_set_body_order"); }
    SpacecraftBatch clone() const {                                                                            // #[derive(Clone)]: the UI's snapshots
        eph_craft_batch *c = nullptr;
        detail::check(eph_craft_batch_clone(h_, &c), "eph_craft_batch_clone");
        return SpacecraftBatch(c, n_);
    }
    SpacecraftBatch(SpacecraftBatch &&o) noexcept : h_(std::exchange(o.h_, nullptr)), n_(o.n_) {}
    SpacecraftBatch &operator=(SpacecraftBatch &&o) noexcept {
        if (this != &o) { if (h_) eph_craft_batch_destroy(h_); h_ = std::exchange(o.h_, nullptr); n_ = o.n_; }
        return *this;
    }
    int64_t len() const { return n_; }

private:
    SpacecraftBatch(eph_craft_batch *h, int64_t n) : h_(h), n_(n) {}
    eph_craft_batch *h_ = nullptr;
    int64_t n_ = 0;
};

inline double divergence_time_before(const std::vector<Burn> &old_burns, const std::vector<Burn> &new_burns, double before) {
    struct Csr { std::vector<double> s, e, a; std::vector<int32_t> r; };
    auto pack = [](const std::vector<Burn> &b) {
        Csr c;
        for (const Burn &x : b) { c.s.push_back(x.start); c.e.push_back(x.end); c.r.push_back(x.reference); c.a.insert(c.a.end(), x.acceleration.begin(), x.acceleration.end()); }
        return c;
    };
    const Csr o = pack(old_burns), n = pack(new_burns);
    double at = 0.0;
    detail::check(eph_timeline_divergence_time(static_cast<int64_t>(old_burns.size()), o.s.data(), o.e.data(), o.a.data(), o.r.data(),
                                               static_cast<int64_t>(new_burns.size()), n.s.data(), n.e.data(), n.a.data(), n.r.data(), before, &at),
                  "eph_timeline_divergence_time");
    return at;
}

}  // namespace ephemeris_amd
