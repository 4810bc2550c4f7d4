// api.cpp -- the extern "C" boundary (include/ephemeris_amd.h). Thin: argument checks, handle ownership,
// status codes. No exceptions cross the boundary.
#include <cstdio>
#include <cstring>
#include <new>

#include "host.h"

namespace eph {
namespace {
thread_local std::string g_last_error;
}
void set_last_error(const char *what, hipError_t e) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}
void set_last_error_text(const std::string &s) { g_last_error = s; }

// The product path has no CPU fallback: without a HIP device every compute entry point fails loudly.
int check_device() {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_last_error = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e)
                                       : std::string("no HIP device visible");
        (void)hipGetLastError();
        return EPH_ERR_NO_DEVICE;
    }
    return EPH_OK;
}
}  // namespace eph

using namespace eph;

struct eph_prop {
    std::unique_ptr<NBodyPropagator> p;
    eph_nbody view;
};

#define EPH_GUARD_BEGIN try {
#define EPH_GUARD_END                                   \
    }                                                   \
    catch (const std::bad_alloc &) {                    \
        return EPH_ERR_OUT_OF_MEMORY;                   \
    }                                                   \
    catch (...) {                                       \
        set_last_error_text("unexpected C++ exception"); \
        return EPH_ERR_HIP;                             \
    }

// the library is compiled with -fvisibility=hidden: the boundary below (and craft.hip's half of it) is ALL it exports
#pragma GCC visibility push(default)
extern "C" {

int32_t eph_abi_version(void) { return EPH_ABI_VERSION; }
int32_t eph_pair_variant(void) { return eph::default_pair_variant(); }
int32_t eph_set_pair_variant(int32_t k) { return eph::set_default_pair_variant(k); }
int32_t eph_release_cached_memory(uint64_t *bytes) {
    const size_t b = eph::release_cached_memory();
    if (bytes) *bytes = (uint64_t)b;
    return EPH_OK;
}

const char *eph_status_string(int32_t st) {
    switch (st) {
        case EPH_OK: return "ok";
        case EPH_STEP_SIZE_UNDERFLOW: return "step size underflow";          // lib.rs:323-331
        case EPH_MAX_ITERATIONS_REACHED: return "max iterations reached";
        case EPH_BOUND_REACHED: return "integration bound reached";
        case EPH_EVAL_FAILED: return "failed to evaluate ODE";
        case EPH_SOLOUT_EXIT: return "solout exit";                          // nbody.rs:52
        case EPH_ERR_BAD_ARGUMENT: return "bad argument";
        case EPH_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        case EPH_ERR_HIP: return "HIP runtime error";
        case EPH_ERR_UNSUPPORTED: return "unsupported configuration";
        case EPH_ERR_OUT_OF_MEMORY: return "out of memory";
        case EPH_ERR_COMM: return "exchange (RCCL / callback) failed";
        default: return "unknown status";
    }
}
const char *eph_last_error(void) { return g_last_error.c_str(); }

int32_t eph_device_count(int32_t *count) {
    if (!count) return EPH_ERR_BAD_ARGUMENT;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count = c;
    return EPH_OK;
}
int32_t eph_set_device(int32_t device) {
    int st = check_device();
    if (st) return st;
    EPH_HIP(hipSetDevice(device));
    return EPH_OK;
}
int32_t eph_device_name(char *buf, int32_t buflen) {
    if (!buf || buflen <= 0) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    int dev = 0;
    EPH_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EPH_HIP(hipGetDeviceProperties(&prop, dev));
    std::snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return EPH_OK;
}

int32_t eph_srkn_coeffs(const char *name, int32_t *stages, int32_t *fsal, double *A, double *B) {
    SrknCoeffs c;
    if (!name || !stages || !fsal || !A || !B || !find_srkn(name, &c)) return EPH_ERR_BAD_ARGUMENT;
    *stages = c.stages;
    *fsal = c.fsal;
    for (int s = 0; s < c.stages; ++s) { A[s] = c.A[s]; B[s] = c.B[s]; }
    return EPH_OK;
}
int32_t eph_elm2_coeffs(const char *name, int32_t *order, double *wa, double *wb, double *inv_beta_d, double *cw,
                        double *inv_cowell_d) {
    Elm2Coeffs c;
    if (!name || !order || !wa || !wb || !inv_beta_d || !cw || !inv_cowell_d || !find_elm2(name, &c))
        return EPH_ERR_BAD_ARGUMENT;
    *order = c.order;
    for (int j = 0; j < c.order; ++j) { wa[j] = c.wa[j]; wb[j] = c.wb[j]; cw[j] = c.cw[j]; }
    *inv_beta_d = c.inv_beta_d;
    *inv_cowell_d = c.inv_cowell_d;
    return EPH_OK;
}

int32_t eph_accel_eval(int32_t n, const double *pos, const double *mu, double *acc) {
    EPH_GUARD_BEGIN
    return accel_eval_device(n, pos, mu, acc);
    EPH_GUARD_END
}

// ---- eph_nbody --------------------------------------------------------------------------------------
int32_t eph_nbody_create(int32_t n, const double *pos, const double *vel, const double *mu, double t0, double h,
                         const char *method, eph_nbody **out) {
    EPH_GUARD_BEGIN
    if (!out) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<eph_nbody> hnd(new eph_nbody());
    int st = NBodyIntegration::create(n, pos, vel, mu, t0, h, method, &hnd->own);
    if (st) return st;
    hnd->p = hnd->own.get();
    *out = hnd.release();
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_nbody_advance(eph_nbody *h, int64_t n_steps) {
    EPH_GUARD_BEGIN
    if (!h || !h->p || n_steps < 0) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;
    return h->p->advance(n_steps);
    EPH_GUARD_END
}
int32_t eph_nbody_advance_many(eph_nbody *const *handles, int32_t count, int64_t n_steps) {
    EPH_GUARD_BEGIN
    if (count < 0 || n_steps < 0 || (count > 0 && !handles)) return EPH_ERR_BAD_ARGUMENT;
    std::vector<NBodyIntegration *> igs;
    for (int32_t i = 0; i < count; ++i) {
        if (!handles[i] || !handles[i]->p) return EPH_ERR_BAD_ARGUMENT;
        if (const int st = handles[i]->settle()) return st;
        igs.push_back(handles[i]->p);
    }
    return NBodyIntegration::advance_many(igs.data(), count, n_steps);
    EPH_GUARD_END
}
int32_t eph_nbody_get_state(eph_nbody *h, double *pos, double *vel, double *t, uint32_t *sc) {
    EPH_GUARD_BEGIN
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;
    return h->p->get_state(pos, vel, t, sc);
    EPH_GUARD_END
}
int32_t eph_nbody_get_acc(eph_nbody *h, double *acc) {
    EPH_GUARD_BEGIN
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;
    return h->p->get_acc(acc);
    EPH_GUARD_END
}
int32_t eph_nbody_set_bound(eph_nbody *h, double bound) {
    EPH_GUARD_BEGIN
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;         // queued steps were accepted against the old bound: run them first
    h->p->set_bound(bound);
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_nbody_clone(eph_nbody *h, eph_nbody **out) {
    EPH_GUARD_BEGIN
    if (!h || !h->p || !out) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<eph_nbody> c(new eph_nbody());
    int st = h->settle();
    if (st) return st;
    if ((st = h->p->clone(&c->own))) return st;
    c->p = c->own.get();
    *out = c.release();
    return EPH_OK;
    EPH_GUARD_END
}
void eph_nbody_destroy(eph_nbody *h) {
    if (h && h->own) delete h;   // borrowed views (eph_prop_integrator) are not owned
}
int32_t eph_nbody_eval_count(eph_nbody *h, uint64_t *count) {
    EPH_GUARD_BEGIN
    if (!h || !h->p || !count) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;         // (settle runs queued steps, fits and allocations: it can fail)
    *count = h->p->evals();
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_nbody_set_path(eph_nbody *h, int32_t path) {
    EPH_GUARD_BEGIN
    if (!h || !h->p || path < 0 || path > EPH_PATH_F32_PAIRS) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;         // steps queued through a propagator view ran on the path they were queued for
    h->p->set_path(path);
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_nbody_kernel_time(eph_nbody *h, double *total_ms, uint64_t *launches) {
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (total_ms) *total_ms = h->p->kernel_ms();
    if (launches) *launches = h->p->kernel_launches();
    return EPH_OK;
}
int32_t eph_nbody_enable_timing(eph_nbody *h, int32_t on) {
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    h->p->enable_timing(on != 0);
    return EPH_OK;
}

int32_t eph_nbody_sync(eph_nbody *h) {
    EPH_GUARD_BEGIN
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (const int st = h->settle()) return st;
    return h->p->sync();
    EPH_GUARD_END
}

// ---- target-partitioned multi-GPU run (shard.cpp) -----------------------------------------------------
int32_t eph_rccl_unique_id(void *out128) {
    EPH_GUARD_BEGIN
    if (!out128) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    return rccl_unique_id(out128);
    EPH_GUARD_END
}
int32_t eph_nbody_shard(eph_nbody *h, int32_t rank, int32_t world, const void *rccl_unique_id, eph_exchange_fn fn,
                        void *ctx) {
    EPH_GUARD_BEGIN
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (hipSetDevice(h->p->device()) != hipSuccess) return EPH_ERR_HIP;
    std::shared_ptr<Exchange> x;
    int st = Exchange::create(rank, world, rccl_unique_id, fn, ctx, &x);
    if (st) return st;
    return h->p->set_shard(std::move(x));
    EPH_GUARD_END
}
int32_t eph_prop_shard(eph_prop *p, int32_t rank, int32_t world, const void *rccl_unique_id, eph_exchange_fn fn,
                       void *ctx) {
    if (!p || !p->p) return EPH_ERR_BAD_ARGUMENT;
    // Steps taken before sharding left their polynomials on the device (and maybe a queue of deferred steps); the
    // sharded branch pushes its windows straight into the host splines, so everything older must be there first.
    {
        EPH_GUARD_BEGIN
        const int st = p->p->settle();
        if (st) return st;
        EPH_GUARD_END
    }
    return eph_nbody_shard(&p->view, rank, world, rccl_unique_id, fn, ctx);
}
// ---- direct-write transport (peer.hip) ----------------------------------------------------------------
struct eph_peer {
    std::shared_ptr<PeerTransport> t;
};
int32_t eph_peer_create_ex(int32_t rank, int32_t world, uint64_t slot_bytes, int32_t memory_form, eph_peer **out) {
    EPH_GUARD_BEGIN
    if (!out) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    int st = check_device();
    if (st) return st;
    std::unique_ptr<eph_peer> p(new eph_peer());
    if ((st = PeerTransport::create(rank, world, (size_t)slot_bytes, memory_form, &p->t))) return st;
    *out = p.release();
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_peer_create(int32_t rank, int32_t world, uint64_t slot_bytes, eph_peer **out) {
    return eph_peer_create_ex(rank, world, slot_bytes, EPH_PEER_MEMORY_AUTO, out);
}
int32_t eph_peer_memory_form(eph_peer *p, int32_t *form) {
    if (!p || !p->t || !form) return EPH_ERR_BAD_ARGUMENT;
    *form = p->t->form();
    if (!p->t->fallback_reason().empty()) set_last_error_text("eph_peer: plain device memory because " + p->t->fallback_reason());
    return EPH_OK;
}
int32_t eph_peer_handle(eph_peer *p, void *out64) {
    if (!p || !p->t || !out64) return EPH_ERR_BAD_ARGUMENT;
    std::memcpy(out64, p->t->handle(), kPeerHandleBytes);
    return EPH_OK;
}
int32_t eph_peer_connect(eph_peer *p, const void *handles) {
    EPH_GUARD_BEGIN
    if (!p || !p->t) return EPH_ERR_BAD_ARGUMENT;
    return p->t->connect(handles);
    EPH_GUARD_END
}
int32_t eph_peer_destroy(eph_peer *p) {
    delete p;
    return EPH_OK;
}
int32_t eph_nbody_shard_peer(eph_nbody *h, eph_peer *p) {
    EPH_GUARD_BEGIN
    if (!h || !h->p || !p || !p->t) return EPH_ERR_BAD_ARGUMENT;
    if (hipSetDevice(h->p->device()) != hipSuccess) return EPH_ERR_HIP;
    std::shared_ptr<Exchange> x;
    int st = Exchange::create_peer(p->t, &x);
    if (st) return st;
    return h->p->set_shard(std::move(x));
    EPH_GUARD_END
}
int32_t eph_prop_shard_peer(eph_prop *p, eph_peer *peer) {
    if (!p || !p->p) return EPH_ERR_BAD_ARGUMENT;
    {
        EPH_GUARD_BEGIN
        const int st = p->p->settle();
        if (st) return st;
        EPH_GUARD_END
    }
    return eph_nbody_shard_peer(&p->view, peer);
}
int32_t eph_nbody_shard_info(eph_nbody *h, int32_t *lo, int32_t *hi, uint64_t *gathers) {
    if (!h || !h->p) return EPH_ERR_BAD_ARGUMENT;
    if (lo) *lo = h->p->shard_lo();
    if (hi) *hi = h->p->shard_hi();
    if (gathers) *gathers = h->p->gathers();
    return EPH_OK;
}

// ---- eph_prop ---------------------------------------------------------------------------------------
int32_t eph_prop_create(int32_t n, const double *pos, const double *vel, const double *mu, double t0, double dt,
                        int32_t direction, const char *method, const uint32_t *count, const uint32_t *degree,
                        eph_prop **out) {
    EPH_GUARD_BEGIN
    if (!out) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<eph_prop> hnd(new eph_prop());
    int st = NBodyPropagator::create(n, pos, vel, mu, t0, dt, direction, method, count, degree, &hnd->p);
    if (st) return st;
    hnd->view.p = hnd->p->integration();
    hnd->view.owner = hnd->p.get();
    *out = hnd.release();
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_prop_step(eph_prop *p) {      // lazily: queued, run as one device batch when data is needed (host.h)
    EPH_GUARD_BEGIN
    if (!p) return EPH_ERR_BAD_ARGUMENT;
    return p->p->step_deferred();
    EPH_GUARD_END
}
int32_t eph_prop_step_n(eph_prop *p, int64_t n) {
    EPH_GUARD_BEGIN
    if (!p || n < 0) return EPH_ERR_BAD_ARGUMENT;
    return p->p->step_n(n);
    EPH_GUARD_END
}
int32_t eph_prop_step_n_many(eph_prop *const *props, int32_t count, int64_t n) {
    EPH_GUARD_BEGIN
    if (count < 0 || n < 0 || (count > 0 && !props)) return EPH_ERR_BAD_ARGUMENT;
    std::vector<NBodyPropagator *> ps;
    for (int32_t i = 0; i < count; ++i) {
        if (!props[i] || !props[i]->p) return EPH_ERR_BAD_ARGUMENT;
        ps.push_back(props[i]->p.get());
    }
    return NBodyPropagator::step_n_many(ps.data(), count, n);
    EPH_GUARD_END
}
int32_t eph_prop_step_to(eph_prop *p, double t) {
    EPH_GUARD_BEGIN
    if (!p) return EPH_ERR_BAD_ARGUMENT;
    return p->p->step_to(t);
    EPH_GUARD_END
}
int32_t eph_prop_time(eph_prop *p, double *t) {
    if (!p || !t) return EPH_ERR_BAD_ARGUMENT;
    *t = p->p->time();
    return EPH_OK;
}
int32_t eph_prop_has_reached(eph_prop *p, double t, int32_t *flag) {
    if (!p || !flag) return EPH_ERR_BAD_ARGUMENT;
    *flag = p->p->has_reached(t) ? 1 : 0;
    return EPH_OK;
}
int32_t eph_prop_integrator_time(eph_prop *p, double *t) {
    if (!p || !t) return EPH_ERR_BAD_ARGUMENT;
    const int st = p->p->flush();             // queued steps run first
    if (st) return st;
    *t = p->p->integration()->time();
    return EPH_OK;
}
int32_t eph_prop_get_state(eph_prop *p, double *pos, double *vel, double *t, uint32_t *sc) {
    EPH_GUARD_BEGIN
    if (!p) return EPH_ERR_BAD_ARGUMENT;
    const int st = p->p->flush();             // queued steps run first
    if (st) return st;
    return p->p->integration()->get_state(pos, vel, t, sc);
    EPH_GUARD_END
}
int32_t eph_prop_take_solution(eph_prop *p, eph_solution **out) {
    EPH_GUARD_BEGIN
    if (!p || !out) return EPH_ERR_BAD_ARGUMENT;
    std::unique_ptr<Solution> s;
    int st = p->p->take_solution(&s);
    if (st) return st;
    eph_solution *o = new eph_solution();
    o->s = std::move(*s);
    *out = o;
    return EPH_OK;
    EPH_GUARD_END
}
int32_t eph_prop_propagate(eph_prop *p, double to, eph_solution **out) {
    int32_t st = eph_prop_step_to(p, to);
    if (st) return st;
    return eph_prop_take_solution(p, out);
}
int32_t eph_prop_clone(eph_prop *p, eph_prop **out) {
    EPH_GUARD_BEGIN
    if (!p || !out) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<eph_prop> c(new eph_prop());
    int st = p->p->clone(&c->p);
    if (st) return st;
    c->view.p = c->p->integration();
    c->view.owner = c->p.get();
    *out = c.release();
    return EPH_OK;
    EPH_GUARD_END
}
void eph_prop_destroy(eph_prop *p) { delete p; }
eph_nbody *eph_prop_integrator(eph_prop *p) {
    if (!p) return nullptr;
    (void)p->p->flush();                      // the view reads the integration directly: queued steps run first
    return &p->view;
}

// ---- eph_solution -----------------------------------------------------------------------------------
int32_t eph_solution_bodies(const eph_solution *s, int32_t *n) {
    if (!s || !n) return EPH_ERR_BAD_ARGUMENT;
    *n = (int32_t)s->s.splines.size();
    return EPH_OK;
}
int32_t eph_solution_info(const eph_solution *s, int32_t body, double *start, double *interval, int64_t *npoly) {
    if (!s || body < 0 || (size_t)body >= s->s.splines.size()) return EPH_ERR_BAD_ARGUMENT;
    const UniformSpline &u = s->s.splines[body];
    if (start) *start = u.start;
    if (interval) *interval = u.interval;
    if (npoly) *npoly = (int64_t)u.polynomials.size();
    return EPH_OK;
}
int32_t eph_solution_coeffs(const eph_solution *s, int32_t body, double *coeffs, int32_t *ncoef) {
    if (!s || body < 0 || (size_t)body >= s->s.splines.size() || !coeffs || !ncoef) return EPH_ERR_BAD_ARGUMENT;
    const UniformSpline &u = s->s.splines[body];
    size_t p = 0;
    for (const Polynomial &q : u.polynomials) {
        ncoef[p] = q.ncoef;
        std::memcpy(coeffs + p * kDiv * 3, &q.c[0][0], sizeof(double) * kDiv * 3);
        ++p;
    }
    return EPH_OK;
}
int32_t eph_solution_eval(const eph_solution *s, int32_t body, int64_t m, const double *at, double *pos, double *vel,
                          uint8_t *inside) {
    EPH_GUARD_BEGIN
    if (!s || body < 0 || (size_t)body >= s->s.splines.size()) return EPH_ERR_BAD_ARGUMENT;
    return spline_eval_device(s->s.splines[body], m, at, pos, vel, inside);
    EPH_GUARD_END
}
int32_t eph_solution_append(eph_solution *s, const eph_solution *tail, int32_t direction) {
    EPH_GUARD_BEGIN
    // s == tail would insert a deque's own iterator range into itself (undefined behaviour); the reference's
    // append takes `other` by value, so aliasing cannot be expressed there
    if (!s || !tail || s == tail || s->s.splines.size() != tail->s.splines.size()) return EPH_ERR_BAD_ARGUMENT;
    // check every spline first so a failure leaves `s` untouched (the reference would have panicked)
    for (size_t b = 0; b < s->s.splines.size(); ++b) {
        const UniformSpline &x = s->s.splines[b], &y = tail->s.splines[b];
        if (x.interval != y.interval) return EPH_ERR_BAD_ARGUMENT;
        if (direction > 0 ? (x.end() != y.start) : (x.start != y.end())) return EPH_ERR_BAD_ARGUMENT;
    }
    for (size_t b = 0; b < s->s.splines.size(); ++b) {
        UniformSpline &x = s->s.splines[b];
        const UniformSpline &y = tail->s.splines[b];
        if (direction > 0) {   // append  trajectory.rs:528-534
            x.polynomials.insert(x.polynomials.end(), y.polynomials.begin(), y.polynomials.end());
        } else {               // prepend trajectory.rs:515-526
            x.start = y.start;
            x.polynomials.insert(x.polynomials.begin(), y.polynomials.begin(), y.polynomials.end());
        }
    }
    return EPH_OK;
    EPH_GUARD_END
}
// Vec<UniformSpline> from its parts (host only): what a shim needs to hand an ephemeris built elsewhere -- or one it
// stored -- to eph_solution_eval / eph_ephemeris_create
int32_t eph_solution_create(int32_t n_bodies, const double *start, const double *interval, const int64_t *npoly,
                            const double *coeffs, const int32_t *ncoef, eph_solution **out) {
    EPH_GUARD_BEGIN
    if (n_bodies < 0 || !out || (n_bodies > 0 && (!start || !interval || !npoly))) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<eph_solution> s(new eph_solution());
    s->s.splines.resize((size_t)n_bodies);
    int64_t q = 0;
    for (int b = 0; b < n_bodies; ++b) {
        if (npoly[b] < 0 || (npoly[b] > 0 && (!coeffs || !ncoef))) return EPH_ERR_BAD_ARGUMENT;
        UniformSpline &u = s->s.splines[b];
        u.start = start[b];
        u.interval = interval[b];
        for (int64_t k = 0; k < npoly[b]; ++k, ++q) {
            if (ncoef[q] < 0 || ncoef[q] > kDiv) return EPH_ERR_BAD_ARGUMENT;
            Polynomial p;
            p.ncoef = ncoef[q];
            // rows at or beyond ncoef stay +0.0 (the invariant k_lsq_fit produces and eph_solution_coeffs returns)
            std::copy(coeffs + q * kDiv * 3, coeffs + q * kDiv * 3 + (size_t)ncoef[q] * 3, &p.c[0][0]);
            u.polynomials.push_back(p);
        }
    }
    *out = s.release();
    return EPH_OK;
    EPH_GUARD_END
}
// UniformSpline::clear_before / clear_after on one body's spline, or on all of them (body < 0)
int32_t eph_solution_clear(eph_solution *s, int32_t body, double at, int32_t after) {
    EPH_GUARD_BEGIN
    if (!s || body >= (int32_t)s->s.splines.size()) return EPH_ERR_BAD_ARGUMENT;
    for (size_t b = 0; b < s->s.splines.size(); ++b) {
        if (body >= 0 && (size_t)body != b) continue;
        if (after) s->s.splines[b].clear_after(at); else s->s.splines[b].clear_before(at);
    }
    return EPH_OK;
    EPH_GUARD_END
}
// UniformSpline::between for every body: *out = the sub-splines (NULL if any body's is None)
int32_t eph_solution_between(const eph_solution *s, double from, double to, eph_solution **out) {
    EPH_GUARD_BEGIN
    if (!s || !out) return EPH_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<eph_solution> r(new eph_solution());
    r->s.splines.resize(s->s.splines.size());
    for (size_t b = 0; b < s->s.splines.size(); ++b)
        if (!s->s.splines[b].between(from, to, &r->s.splines[b])) return EPH_OK;   // None
    *out = r.release();
    return EPH_OK;
    EPH_GUARD_END
}
void eph_solution_destroy(eph_solution *s) { delete s; }

int32_t eph_least_squares_fit(int32_t degree, int32_t backward, int64_t nwin, const double *samples, double *coeffs,
                              int32_t *ncoef) {
    EPH_GUARD_BEGIN
    return least_squares_fit_device(degree, backward, nwin, samples, coeffs, ncoef);
    EPH_GUARD_END
}

// SpacecraftPropagator::join  spacecraft.rs:558-561 = CubicHermiteSpline::clear_after (trajectory.rs:842-845) + extend
// (:847-849). Host only.
int32_t eph_hermite_join(int64_t n_lhs, const double *t_lhs, const double *pos_lhs, const double *vel_lhs,
                         int64_t n_rhs, const double *t_rhs, const double *pos_rhs, const double *vel_rhs,
                         int64_t capacity, double *t_out, double *pos_out, double *vel_out, int64_t *n_out) {
    if (n_lhs < 0 || n_rhs < 0 || capacity < 0 || !n_out || (n_lhs > 0 && (!t_lhs || !pos_lhs || !vel_lhs)) ||
        (n_rhs > 0 && (!t_rhs || !pos_rhs || !vel_rhs)))
        return EPH_ERR_BAD_ARGUMENT;
    // rhs.start(): self.0.first().map(|i| i.0).unwrap_or(Epoch::MIN)   trajectory.rs:756-758
    const double at = n_rhs > 0 ? t_rhs[0] : -1.7976931348623157e308;
    int64_t keep = 0;
    for (int64_t k = 0; k < n_lhs; ++k) keep += at > t_lhs[k] ? 1 : 0;        // retain(|(k, _)| &at > k)
    *n_out = keep + n_rhs;
    if (*n_out > capacity || (*n_out > 0 && (!t_out || !pos_out || !vel_out))) return EPH_ERR_BAD_ARGUMENT;
    int64_t w = 0;
    for (int64_t k = 0; k < n_lhs; ++k) {                                     // w <= k: in place on the lhs arrays is fine
        if (!(at > t_lhs[k])) continue;
        t_out[w] = t_lhs[k];
        for (int c = 0; c < 3; ++c) { pos_out[3 * w + c] = pos_lhs[3 * k + c]; vel_out[3 * w + c] = vel_lhs[3 * k + c]; }
        ++w;
    }
    for (int64_t k = 0; k < n_rhs; ++k, ++w) {                                // self.0.extend(rhs.0)
        t_out[w] = t_rhs[k];
        for (int c = 0; c < 3; ++c) { pos_out[3 * w + c] = pos_rhs[3 * k + c]; vel_out[3 * w + c] = vel_rhs[3 * k + c]; }
    }
    return EPH_OK;
}

// SoiTransitions::clear_after(at) then ::extend(rhs)   (app dynamics/spacecraft.rs:341-346, 356-361, 331-337): the
// event half of PredictionTarget::merge (:836-839). Times inside one list are unique (insert replaces), so the
// binary search is a partition point.
int32_t eph_transitions_join(int64_t n_lhs, const double *t_lhs, const int32_t *body_lhs, int64_t n_rhs, const double *t_rhs,
                             const int32_t *body_rhs, double at, int64_t capacity, double *t_out, int32_t *body_out,
                             int64_t *n_out) {
    if (n_lhs < 0 || n_rhs < 0 || capacity < 0 || !n_out || (n_lhs > 0 && (!t_lhs || !body_lhs)) ||
        (n_rhs > 0 && (!t_rhs || !body_rhs)) || capacity < n_lhs + n_rhs || (capacity > 0 && (!t_out || !body_out)))
        return EPH_ERR_BAD_ARGUMENT;
    int64_t n = 0;
    for (int64_t k = 0; k < n_lhs && t_lhs[k] <= at; ++k, ++n) {              // Ok(i) => truncate(i + 1), Err(i) => truncate(i)
        t_out[n] = t_lhs[k];
        body_out[n] = body_lhs[k];
    }
    for (int64_t r = 0; r < n_rhs; ++r) {                                     // for (time, entity) in other.0 { self.insert(..) }
        int64_t i = 0;
        while (i < n && t_out[i] < t_rhs[r]) ++i;
        if (i < n && t_out[i] == t_rhs[r]) { body_out[i] = body_rhs[r]; continue; }          // Ok(i) => self.0[i] = ..
        if (i > 0 && body_out[i - 1] == body_rhs[r]) continue;                               // same sphere as before: no entry
        for (int64_t k = n; k > i; --k) { t_out[k] = t_out[k - 1]; body_out[k] = body_out[k - 1]; }
        t_out[i] = t_rhs[r];
        body_out[i] = body_rhs[r];
        ++n;
    }
    *n_out = n;
    return EPH_OK;
}

// Apsides::clear_after(at) then ::extend(rhs)   (app dynamics/spacecraft.rs:431-436, 426-428): extend is a plain
// append here (no sorted insert, unlike the transitions).
int32_t eph_apsides_join(int64_t n_lhs, const double *t_lhs, const double *distance_lhs, const int32_t *kind_lhs,
                         const int32_t *body_lhs, int64_t n_rhs, const double *t_rhs, const double *distance_rhs,
                         const int32_t *kind_rhs, const int32_t *body_rhs, double at, int64_t capacity, double *t_out,
                         double *distance_out, int32_t *kind_out, int32_t *body_out, int64_t *n_out) {
    if (n_lhs < 0 || n_rhs < 0 || capacity < 0 || !n_out || (n_lhs > 0 && (!t_lhs || !distance_lhs || !kind_lhs || !body_lhs)) ||
        (n_rhs > 0 && (!t_rhs || !distance_rhs || !kind_rhs || !body_rhs)) || capacity < n_lhs + n_rhs ||
        (capacity > 0 && (!t_out || !distance_out || !kind_out || !body_out)))
        return EPH_ERR_BAD_ARGUMENT;
    int64_t n = 0;
    for (int64_t k = 0; k < n_lhs && t_lhs[k] <= at; ++k, ++n) {
        t_out[n] = t_lhs[k]; distance_out[n] = distance_lhs[k]; kind_out[n] = kind_lhs[k]; body_out[n] = body_lhs[k];
    }
    for (int64_t r = 0; r < n_rhs; ++r, ++n) {
        t_out[n] = t_rhs[r]; distance_out[n] = distance_rhs[r]; kind_out[n] = kind_rhs[r]; body_out[n] = body_rhs[r];
    }
    *n_out = n;
    return EPH_OK;
}

}  // extern "C"
#pragma GCC visibility pop
