/* ephemeris_amd.h -- C ABI of the MI355X-native ephemeris propagator (libephemeris_amd.so).
 *
 * Drop-in boundary for ONE hot path of Canleskis/ephemeris-explorer: the all-pairs Newtonian acceleration,
 * the fixed-step high-order integrator that advances the massive bodies, the solout that turns integrator
 * samples into a piecewise-polynomial ephemeris, and the evaluator / massless-body propagation that sample
 * it. The reference has no FFI; its seams are Rust traits. Each entry point below names
 * the trait method(s) it stands in for (paths relative to the reference repository root); INTEGRATION.md
 * shows the Rust shim (`extern "C"` block + trait impls) a maintainer would add.
 *
 * Conventions
 *  - plain C types only; caller owns every host buffer; the library owns all device memory.
 *  - vectors are AoS xyz f64 (the layout of Vec<glam::DVec3>), units km, km/s, km^3/s^2, seconds.
 *  - time is f64 seconds since 1958-01-01 TAI (ftime::Epoch::as_offset_seconds, ftime/src/epoch.rs).
 *  - every function returns an eph_status; errors are values, nothing throws or aborts.
 *  - a handle is not thread-safe; distinct handles may be used from distinct threads (the reference runs its forward and
 *    backward propagators and one task per ship concurrently, ephemeris_explorer/src/prediction.rs:385-391). Each handle
 *    owns one HIP stream on the device that was current when it was created. The one handle meant to be SHARED between
 *    threads is eph_ephemeris: it carries the reference's RwLock (see there). A handle passed as `const` is only read: an
 *    eph_solution that no thread modifies may be read by several at once (Rust's &T of a Sync type). Handles may be created on
 *    one thread, used on another and destroyed on a third (the reference's propagators are Send: the task pool moves them). The
 *    library holds no process-wide lock across a device synchronisation (examples/threads.cpp, tests/test_gpu_threads.py).
 *  - results are bit-identical to the reference algorithm's f64 arithmetic (same operation order, no FMA
 *    contraction); see DESIGN.md for the one unpinned formula (the `particular` pair interaction).
 */
#ifndef EPHEMERIS_AMD_H
#define EPHEMERIS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): the eph_debug_* hooks left the boundary (csrc/eph_debug.h); eph_pair_variant reports the process-wide default
 * order new handles take (round 4: the order is a run-time choice), eph_nbody_advance_many / eph_prop_step_n_many only queue. */
/* 3 (round 6): the device ephemeris is LIVE (eph_ephemeris_append / _merge / _clear / _info / _is_valid_at, _export / _import);
 * eph_craft_batch_retry_failed; an FSAL pair's first stage is part of a batch's state. */
#define EPH_ABI_VERSION 3

/* integration::StepError (integration/src/lib.rs:312-318), NBodyPropagatorError::Solout
 * (ephemeris/src/propagators/nbody.rs:43-47); negative values are library / HIP failures. */
typedef enum eph_status {
    EPH_OK = 0,
    EPH_STEP_SIZE_UNDERFLOW = 1,
    EPH_MAX_ITERATIONS_REACHED = 2,
    EPH_BOUND_REACHED = 3,
    EPH_EVAL_FAILED = 4,
    EPH_SOLOUT_EXIT = 5,
    EPH_ERR_BAD_ARGUMENT = -1,
    EPH_ERR_NO_DEVICE = -2,   /* no usable gfx950 device / HIP runtime: the library never falls back to CPU */
    EPH_ERR_HIP = -3,
    EPH_ERR_UNSUPPORTED = -4,
    EPH_ERR_OUT_OF_MEMORY = -5,
    EPH_ERR_COMM = -6         /* RCCL / exchange callback failure (eph_last_error has the text) */
} eph_status;

/* PropagationDirection: Forward / Backward (ephemeris/src/propagators/mod.rs:23-93) */
#define EPH_FORWARD 1
#define EPH_BACKWARD (-1)

int32_t eph_abi_version(void);
/* The evaluation order of the point-mass term -- `particular::gravity::newtonian` acceleration_paired / acceleration_at::<false>
 * (crate `particular` 0.8.0-dev @ d490707a, Cargo.lock:4277-4285; call sites ephemeris/src/propagators/nbody.rs:29,
 * ephemeris_explorer/src/dynamics/spacecraft.rs:73), whose source is not in the reference tree. The library carries seven orders
 * (csrc/pair_term.h: 0 = `d * (mu * (1 / (n2 * sqrt n2)))`, the published crate's form and the default; 1-3 other one-reciprocal
 * orders; 4 = `(d * mu) / p`, 5 = `d * (mu / p)`, 6 = `(d / p) * mu` with p = n2 * sqrt n2 and three true divisions), each
 * bit-identical to the CPU restatement in the same order. eph_set_pair_variant(k) chooses the order for every handle CREATED
 * afterwards (eph_nbody_create, eph_prop_create, eph_craft_batch_create; clones inherit; eph_accel_eval uses the current value);
 * the initial value is the environment's EPH_PAIR_VARIANT, else 0. tools/identify_pair_variant.py names the right k from a
 * print-out of the real crate (tools/particular_probe.rs, tests/golden/pair_probe.json). */
int32_t eph_pair_variant(void);
int32_t eph_set_pair_variant(int32_t k);   /* EPH_ERR_BAD_ARGUMENT outside 0..6 */
const char *eph_status_string(int32_t status);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char *eph_last_error(void);
int32_t eph_device_count(int32_t *count);
int32_t eph_set_device(int32_t device);
int32_t eph_device_name(char *buf, int32_t buflen);
/* The library keeps device blocks of >= 64 MiB that a destroyed handle owned (the knot slabs of a spacecraft batch are GBs) for
 * the next handle that asks for exactly that size: a sweep loop creates identical batches, and taking GB-sized blocks from the
 * driver and handing them back costs ~100 ms per batch. The cache is per device, at most an eighth of that device's memory
 * (EPH_POOL_MAX_MB overrides; 0 disables), a reused block is cleared, and the cache does not outlive the library's use of a device:
 * when the last handle that holds device memory on it is destroyed, its cached blocks return to the driver. While handles are alive,
 * a process that shares the GPU with another allocator calls this to return the cache; *bytes (optional) = what was released. */
int32_t eph_release_cached_memory(uint64_t *bytes);

/* ---- coefficient tables (integration/src/methods.rs, ratio.rs:221-228) ---------------------------
 * The f64 value of every coefficient exactly as the reference multiplies with it. */
int32_t eph_srkn_coeffs(const char *name, int32_t *stages, int32_t *fsal, double *A /*[32]*/, double *B /*[32]*/);
int32_t eph_elm2_coeffs(const char *name, int32_t *order, double *w_alpha /*[16]*/, double *w_beta /*[16]*/,
                        double *inv_beta_d, double *cowell /*[16]*/, double *inv_cowell_d);

/* ---- seam 1: the ODE right-hand side -----------------------------------------------------------
 * SecondOrderODE::eval for NewtonianGravity (ephemeris/src/propagators/nbody.rs:16-39; trait
 * integration/src/problem.rs:122-125). `acc_xyz` is ACCUMULATED into, like the reference's `ddy` (callers
 * pass it zeroed); the per-body summation order is the reference's. Host buffers, n >= 0. */
int32_t eph_accel_eval(int32_t n, const double *pos_xyz, const double *mu, double *acc_xyz);

/* ---- seam 2: Method / Integrator ------------------------------------------------------------------
 * `M::new(FixedMethodParams::new(h)).integrate(NBodyProblem{time:t0, bound:+inf, ..})`
 * (integration/src/lib.rs:139-169, ephemeris/src/propagators/nbody.rs:93-121) with
 * method = "QuinlanTremaine12" | "Stormer13"  (LinearMultistep<_, f64, Substepper<4, BlanesMoan6B>>,
 *                                              integration/src/methods.rs:37-40), or any SRKN table name
 *          "BlanesMoan6B" | "BlanesMoan11B" | "BlanesMoan14A" | "ForestRuth" | "McLachlanO4" |
 *          "McLachlanSS17" | "Pefrl" | "Ruth"  (FixedRungeKutta<SRKN>, methods.rs:22-29).
 * h is signed (Backward propagation = negative h). */
typedef struct eph_nbody eph_nbody;
int32_t eph_nbody_create(int32_t n, const double *pos_xyz, const double *vel_xyz, const double *mu, double t0,
                         double h, const char *method, eph_nbody **out);
/* n_steps x Integrator::advance (integration/src/lib.rs:359-360; multistep/mod.rs:201-224;
 * runge_kutta/mod.rs:112-125). Stops early with the reference's error (BoundReached, StepSizeUnderflow). */
int32_t eph_nbody_advance(eph_nbody *h, int64_t n_steps);
/* problem.state.y / .dy, problem.time, IntegratorState::step_count (lib.rs:276-289). Any pointer may be NULL. */
int32_t eph_nbody_get_state(eph_nbody *h, double *pos_xyz, double *vel_xyz, double *t, uint32_t *step_count);
/* the acceleration the integrator holds (ELM2.current_ddy / SRKN.ddy) */
int32_t eph_nbody_get_acc(eph_nbody *h, double *acc_xyz);
int32_t eph_nbody_set_bound(eph_nbody *h, double bound); /* ODEProblem.bound (problem.rs:2-7) */
int32_t eph_nbody_clone(eph_nbody *h, eph_nbody **out);   /* #[derive(Clone)] on Integration (lib.rs:409-425) */
void eph_nbody_destroy(eph_nbody *h);
/* number of right-hand-side evaluations performed so far */
int32_t eph_nbody_eval_count(eph_nbody *h, uint64_t *count);
/* Which device path `advance` uses: 0 = auto, 1 = one launch per step with the one-wave-per-block force kernel,
 * 2 = persistent single workgroup (n <= 64 only), 3 = one launch per step with the workgroup-specialised force
 * kernel. Paths 0-3 produce identical bits (the reference's summation order); for tests and tuning.
 * 4 = EPH_PATH_FAST, OPT-IN: steady multistep steps with slice-parallel partial sums combined in slice order -- the
 * same pair arithmetic but NOT the summation order of NewtonianGravity::eval (nbody.rs:22-38), so results differ from
 * the reference in the last bits of every acceleration (deterministic run to run). Unsharded systems of more than
 * 64 bodies; start-up steps and SRKN methods keep the ordered kernels. DESIGN.md "what bit-exactness costs". */
#define EPH_PATH_FAST 4
/* 5 = EPH_PATH_FAST_RSQ, OPT-IN: the fast path with 1/r^3 from v_rsq_f64 + two Newton steps instead of the IEEE square
 * root and division (SURVEY 7 stage 3 "fast mode"): the pair terms themselves differ from the reference's in the last bit. */
#define EPH_PATH_FAST_RSQ 5
/* 6 = EPH_PATH_F32_PAIRS, OPT-IN, for large systems (BASELINE.json configs[4], "65 536-body f32"): the fast path with the
 * PAIR arithmetic in binary32 (packed f32 instructions, v_rsq_f32 + one Newton step), contributions accumulated in f64 in
 * slice order, integrator state and formulae f64. The reference has no f32 path (nbody.rs:13,19): accelerations agree
 * with the exact path to ~1e-6 relative, trajectories diverge accordingly. Never the default. */
#define EPH_PATH_F32_PAIRS 6
int32_t eph_nbody_set_path(eph_nbody *h, int32_t path);
/* device time of the steady-state kernels launched by this handle so far, measured with HIP events on the
 * handle's stream: total milliseconds and launch count (used by bench.py for the roofline figure) */
int32_t eph_nbody_kernel_time(eph_nbody *h, double *total_ms, uint64_t *launches);
int32_t eph_nbody_enable_timing(eph_nbody *h, int32_t on);
/* block until every launch queued on the handle's stream has finished */
int32_t eph_nbody_sync(eph_nbody *h);
/* eph_nbody_advance(h, n_steps) on `count` independent systems at once. Systems of at most 32 bodies advance in ONE
 * workgroup each (a step is a chain of dependent operations, not work: 32 bodies use one of 256 compute units), so the
 * steady-state steps of all of them go into one launch with a workgroup per system -- the reference runs its forward and
 * backward propagators concurrently (ephemeris_explorer/src/load/mod.rs:673-687), and ensembles are independent too.
 * Same results as the separate calls; systems that do not qualify (start-up steps, more bodies, a sharded handle, a
 * step that would return a StepError) simply make the call do those. Handles must be distinct and on one device.
 * ASYNCHRONOUS, like eph_nbody_advance: the call queues the launch and returns; completion is eph_nbody_sync / get_state of a
 * member. A device failure AFTER the members' bookkeeping has moved (the shared launch itself) marks every member failed: all their
 * later calls -- and those of their clones -- return that status; kernel time of a timed gang accrues to the FIRST handle. */
int32_t eph_nbody_advance_many(eph_nbody *const *handles, int32_t count, int64_t n_steps);

/* ---- multi-GPU: the massive-body system partitioned by TARGET body over the ranks of one node -----------------
 * (the reference runs one propagator per task, ephemeris_explorer/src/prediction.rs:422-443, and has
 * no counterpart.) One process per GPU, each creating the SAME eph_nbody (same bodies, same order) and then calling
 * eph_nbody_shard with its rank. Rank r owns bodies [r*S, (r+1)*S), S = padded_n / world (padded_n = n rounded up
 * to 64; it must be a multiple of 64*world): it keeps their history and evaluates their all-pairs sums against ALL
 * sources in the order of NewtonianGravity::eval (nbody.rs:22-38), so every rank's results are bit-identical to
 * the single-device run. The only exchange is an in-place all-gather of the packed positions (32 B/body) after
 * each kernel that publishes positions -- one per force evaluation -- and of the AoS staging buffer in
 * eph_nbody_get_state / get_acc (which therefore are collective calls, as are advance and clone).
 * Transport: rccl_unique_id != NULL -> ncclAllGather (RCCL over xGMI) on the handle's stream, the 128-byte id
 * coming from eph_rccl_unique_id on rank 0 and distributed by the caller; else `fn`, called as
 * fn(ctx, device_buffer, slice_bytes, rank, world, hipStream_t): it must gather world slices of slice_bytes in
 * place (own slice at rank*slice_bytes), ordered after the work already enqueued on the stream, and either
 * enqueue itself on that stream or complete before returning; non-zero return = failure (EPH_ERR_COMM).
 * Systems of <= 64 bodies are not sharded (EPH_ERR_UNSUPPORTED): run replicas. */
typedef int32_t (*eph_exchange_fn)(void *ctx, void *device_buffer, uint64_t slice_bytes, int32_t rank,
                                   int32_t world, void *hip_stream);
int32_t eph_rccl_unique_id(void *out128);
int32_t eph_nbody_shard(eph_nbody *h, int32_t rank, int32_t world, const void *rccl_unique_id,
                        eph_exchange_fn fn, void *ctx);
/* Third transport: direct peer writes (csrc/peer.hip; SURVEY 5 "fully-connected direct write"). Each rank creates an
 * eph_peer -- a mailbox in its own device memory, world slots of slot_bytes (rounded up to 256) per exchange parity --
 * hands its 64-byte hipIpc handle to every other rank through any channel the caller has (MPI, a pipe,
 * torch.distributed ...), and connects with the table of all world handles in rank order. An exchange is then ONE
 * small launch on the handle's stream: write my slice into every peer's mailbox, raise a flag there, wait for the
 * peers' flags, copy their slices into place -- no collective library, no host synchronisation; slices larger than a
 * slot move in several rounds. One eph_peer serves any number of handles of the process (clones share it), as long as
 * every rank issues its exchanges in the same order. A peer that does not deliver within EPH_PEER_TIMEOUT_MS
 * (default 20000) ends the wait; the failure is reported as EPH_ERR_COMM by the next exchange or eph_nbody_sync /
 * get_state. Works between processes on one device as well as across devices with peer access (xGMI). world <= 16. */
typedef struct eph_peer eph_peer;
int32_t eph_peer_create(int32_t rank, int32_t world, uint64_t slot_bytes, eph_peer **out);
/* The mailbox lives in fine-grained device memory when that can be allocated and its hipIpc handle exported, in plain
 * hipMalloc memory otherwise (every mailbox access in the kernel is system-scope either way). eph_peer_create is
 * eph_peer_create_ex(.., EPH_PEER_MEMORY_AUTO, ..); eph_peer_memory_form reports which form is live (and, after a fallback,
 * leaves the reason in eph_last_error). If eph_peer_connect fails (EPH_ERR_COMM: a peer's mailbox could not be mapped even
 * after enabling peer access explicitly), every rank destroys its eph_peer, re-creates it with EPH_PEER_MEMORY_COARSE and
 * exchanges handles again -- ephemeris_explorer_amd/parallel.py peer_transport() does exactly that. */
#define EPH_PEER_MEMORY_AUTO 0
#define EPH_PEER_MEMORY_FINE 1
#define EPH_PEER_MEMORY_COARSE 2
int32_t eph_peer_create_ex(int32_t rank, int32_t world, uint64_t slot_bytes, int32_t memory_form, eph_peer **out);
int32_t eph_peer_memory_form(eph_peer *p, int32_t *form);
int32_t eph_peer_handle(eph_peer *p, void *out64);
int32_t eph_peer_connect(eph_peer *p, const void *handles_world_x_64);
int32_t eph_peer_destroy(eph_peer *p);
int32_t eph_nbody_shard_peer(eph_nbody *h, eph_peer *p);
/* owned bodies [lo, hi) and the number of all-gathers issued so far (any output may be NULL) */
int32_t eph_nbody_shard_info(eph_nbody *h, int32_t *lo, int32_t *hi, uint64_t *gathers);

/* ---- seam 3: Propagator / IncrementalPropagator / DirectionalPropagator / BoundedPropagator -----------
 * ephemeris::NBodyPropagator<D, DVec3, M, SplineInterpolators<D, DVec3, LeastSquaresFit>>
 * (ephemeris/src/lib.rs:9-79, propagators/nbody.rs:65-235,309-517; ephemeris_explorer/src/dynamics/
 * celestial.rs:19-186). dt > 0; direction = EPH_FORWARD | EPH_BACKWARD; count[b] = sample_period_b / dt
 * (ephemeris_explorer/src/load/mod.rs:325); degree[b] = LeastSquaresFit.degree (<= 7). */
typedef struct eph_prop eph_prop;
typedef struct eph_solution eph_solution; /* Vec<UniformSpline<DVec3>>, ephemeris/src/trajectory.rs:412-633 */
int32_t eph_prop_create(int32_t n, const double *pos_xyz, const double *vel_xyz, const double *mu, double t0,
                        double dt, int32_t direction, const char *method, const uint32_t *count,
                        const uint32_t *degree, eph_prop **out);
/* The same partition for a whole propagator (call right after eph_prop_create on every rank): each rank integrates,
 * samples and fits the bodies it owns; after every batch the new polynomials are all-gathered, so eph_prop_time /
 * has_reached / take_solution see the complete Vec<UniformSpline> on every rank, bit-identical to the single-device
 * propagator. eph_prop_step* / step_to / propagate / clone become collective calls. */
int32_t eph_prop_shard(eph_prop *p, int32_t rank, int32_t world, const void *rccl_unique_id, eph_exchange_fn fn,
                       void *ctx);
int32_t eph_prop_shard_peer(eph_prop *p, eph_peer *peer);
/* eph_prop_step_n(p, n) on `count` propagators at once, their steady-state steps in shared launches
 * (eph_nbody_advance_many); each propagator samples, fits and pushes its own windows. Queues and returns like it: the fits are
 * settled by the next call that needs them (eph_prop_time, has_reached, take_solution, get_state). */
int32_t eph_prop_step_n_many(eph_prop *const *props, int32_t count, int64_t n);
/* IncrementalPropagator::step  nbody.rs:200-207. Executed lazily: the reference's callers step in a loop and read
 * time() / has_reached() after every step (ephemeris_explorer/src/prediction.rs:422-443); both are functions of the number
 * of steps taken, so a steady-state step only advances that bookkeeping on the host (and returns the StepError the step
 * would return) and the queued steps run as ONE device batch when data is needed -- take_solution, clone, get_state,
 * step_n / step_to -- or after 8192 of them. Results are identical to executing every step at once. */
int32_t eph_prop_step(eph_prop *p);
int32_t eph_prop_step_n(eph_prop *p, int64_t n);     /* n x step(), batched on the device */
int32_t eph_prop_step_to(eph_prop *p, double t);     /* IncrementalPropagator::step_to  lib.rs:49-60 */
int32_t eph_prop_time(eph_prop *p, double *t);       /* DirectionalPropagator::time  nbody.rs:225-227,502-508 */
int32_t eph_prop_has_reached(eph_prop *p, double t, int32_t *flag); /* nbody.rs:229-232,510-516 */
int32_t eph_prop_integrator_time(eph_prop *p, double *t);           /* NBodyPropagator::time nbody.rs:150-152 */
int32_t eph_prop_get_state(eph_prop *p, double *pos_xyz, double *vel_xyz, double *t, uint32_t *step_count);
int32_t eph_prop_take_solution(eph_prop *p, eph_solution **out);    /* Propagator::take_solution nbody.rs:182-189 */
int32_t eph_prop_propagate(eph_prop *p, double to, eph_solution **out); /* BoundedPropagator::propagate lib.rs:71-78 */
int32_t eph_prop_clone(eph_prop *p, eph_prop **out);
void eph_prop_destroy(eph_prop *p);
/* the eph_nbody inside (borrowed; do not destroy) -- for timing / path selection */
eph_nbody *eph_prop_integrator(eph_prop *p);

/* ---- the solution: Vec<UniformSpline<DVec3>> -----------------------------------------------------*/
int32_t eph_solution_bodies(const eph_solution *s, int32_t *n);
/* UniformSpline{start, interval, polynomials.len()} */
int32_t eph_solution_info(const eph_solution *s, int32_t body, double *start, double *interval, int64_t *npoly);
/* coeffs: npoly*8*3 doubles, coefficient k of polynomial p at [(p*8+k)*3 + c], zero padded;
 * ncoef[p] = Polynomial length after trim() (trajectory.rs:387-395) */
int32_t eph_solution_coeffs(const eph_solution *s, int32_t body, double *coeffs, int32_t *ncoef);
/* EvaluateTrajectory::state_vector / position (trajectory.rs:459-470) for m query times of one body, on the
 * device. inside[k] = 0 where the reference returns None. vel_xyz may be NULL (position only). */
int32_t eph_solution_eval(const eph_solution *s, int32_t body, int64_t m, const double *at, double *pos_xyz,
                          double *vel_xyz, uint8_t *inside);
/* UniformSpline::append (direction = EPH_FORWARD) / prepend (EPH_BACKWARD), trajectory.rs:515-539; the
 * reference's assert_eq! contiguity checks become EPH_ERR_BAD_ARGUMENT. `tail` is left unchanged. */
int32_t eph_solution_append(eph_solution *s, const eph_solution *tail, int32_t direction);
/* Vec<UniformSpline<DVec3>> from its parts (host only; an ephemeris built or stored elsewhere): per body start,
 * interval and polynomial count; polynomials concatenated in body order as coeffs[poly][8][3] and ncoef[poly]. */
int32_t eph_solution_create(int32_t n_bodies, const double *start, const double *interval, const int64_t *npoly,
                            const double *coeffs, const int32_t *ncoef, eph_solution **out);
/* UniformSpline::clear_before (after = 0, trajectory.rs:536-542) / clear_after (after = 1, :544-549) at epoch `at` on
 * body's spline, or on every spline (body < 0). Host only. */
int32_t eph_solution_clear(eph_solution *s, int32_t body, double at, int32_t after);
/* UniformSpline::between(from, to) (trajectory.rs:484-502) for every body: *out = the sub-splines, or NULL where the
 * reference returns None for any body. Host only. */
int32_t eph_solution_between(const eph_solution *s, double from, double to, eph_solution **out);
void eph_solution_destroy(eph_solution *s);

/* LeastSquaresFit::interpolate (ephemeris_explorer/src/dynamics/celestial.rs:24-135) for `nwin` windows of 9
 * samples each, on the device: samples[(w*9+k)*3+c], backward != 0 selects tau_k = 1-k/8
 * (nbody.rs:422-442). coeffs[(w*8+k)*3+c] zero padded, ncoef[w] after trim. */
int32_t eph_least_squares_fit(int32_t degree, int32_t backward, int64_t nwin, const double *samples,
                              double *coeffs, int32_t *ncoef);

/* ---- massless bodies: a batch of independent spacecraft propagated against the (live) ephemeris ----------------
 * ephemeris::SpacecraftPropagator<[StateVector<DVec3>;1], ReferenceFrame, Bodies, AdaptiveRungeKutta<ERK pair>,
 * CubicHermiteSplineSolout> (ephemeris/src/propagators/spacecraft.rs:224-695) with the app's acceleration model
 * and burn frames (ephemeris_explorer/src/dynamics/spacecraft.rs:70-74,218-293,609-641), one device thread per
 * craft. Bodies are visited in index order. */
typedef struct eph_ephemeris eph_ephemeris;   /* device-resident table of the massive bodies' UniformSplines */
typedef struct eph_craft_batch eph_craft_batch;
#define EPH_KNOTS_FULL 6                       /* per-craft status: the knot slab is full (library limit, not a StepError) */
#define EPH_EVENTS_FULL 7                      /* per-craft event status: a transition / apsis slab is full */

/* uploads the splines of `s` (Vec<UniformSpline>) with the bodies' mu; `s` may be destroyed afterwards */
int32_t eph_ephemeris_create(const eph_solution *s, const double *mu, eph_ephemeris **out);
void eph_ephemeris_destroy(eph_ephemeris *e);
/* The table is LIVE, like the reference's context: GravitationalBody.trajectory is Trajectory(Arc<RwLock<PredictionTrajectory>>)
 * (ephemeris_explorer/src/dynamics/spacecraft.rs:52-74, dynamics/mod.rs:84-85); merged N-body snapshots grow it
 * (dynamics/celestial.rs:198-204,220-226), auto_extend asks for more every frame (auto_extend.rs:182-202), and every spacecraft
 * propagator that holds the context -- stored ones and clones included (prediction.rs:378) -- evaluates against the new extent from
 * then on. So here: every eph_craft_batch (and clone) bound to `e`, eph_plot_points and the error scan see the table as it is when
 * THEIR call starts. eph_ephemeris is the one handle that may be shared between threads: readers hold its lock shared for their
 * whole (synchronous) call, the calls below take it exclusively -- the reference's RwLock. Only the new polynomials travel to the
 * device (amortised O(1) per polynomial).
 *   eph_ephemeris_append   UniformSpline::append (direction = EPH_FORWARD) / prepend (EPH_BACKWARD) for every body,
 *                          ephemeris/src/trajectory.rs:515-534; the reference's assert_eq! contiguity / interval checks become
 *                          EPH_ERR_BAD_ARGUMENT with the table untouched. `tail` is left unchanged.
 *   eph_ephemeris_merge    the app's PredictionTarget::merge for the bodies: Forward clear_after(propagated.start()) then append
 *                          (dynamics/celestial.rs:198-204), Backward clear_before(propagated.end()) then prepend (:220-226).
 *   eph_ephemeris_clear    UniformSpline::clear_before (after = 0, trajectory.rs:536-542) / clear_after (after = 1, :544-549) at
 *                          epoch `at` on body's spline, or on every spline (body < 0).
 *   eph_ephemeris_info     UniformSpline{start, interval, polynomials.len()} of one body (body >= 0) and / or the table's revision
 *                          (a counter that every call above increments; body < 0: revision only). Any pointer may be NULL.
 *   eph_ephemeris_is_valid_at   Bodies::is_valid_at (dynamics/spacecraft.rs:199-201): every body's spline contains(t)
 *                          (trajectory.rs:437-441: sign bit of t - start clear and t - start <= span); what
 *                          flight_plan.rs:363-395 tests before it restarts a ship's prediction.
 * A craft that ran off the table's end holds EPH_EVAL_FAILED (spacecraft.rs:264-281); after the table has grown,
 * eph_craft_batch_retry_failed + the next propagate continue it exactly as the reference's next step() would. */
int32_t eph_ephemeris_append(eph_ephemeris *e, const eph_solution *tail, int32_t direction);
int32_t eph_ephemeris_merge(eph_ephemeris *e, const eph_solution *propagated, int32_t direction);
int32_t eph_ephemeris_clear(eph_ephemeris *e, int32_t body, double at, int32_t after);
int32_t eph_ephemeris_info(const eph_ephemeris *e, int32_t body, double *start, double *interval, int64_t *npoly,
                           uint64_t *revision);
int32_t eph_ephemeris_is_valid_at(const eph_ephemeris *e, double t, int32_t *flag);
/* One contiguous, position-independent image of the table: what rank 0 of a multi-GPU sweep builds once and broadcasts (SURVEY
 * 8(e); ephemeris_explorer_amd/parallel.py broadcast_ephemeris) instead of every rank integrating the bodies again. *bytes = the
 * image's size; with buf == NULL or capacity too small nothing is written and the status is EPH_ERR_BAD_ARGUMENT. An imported table
 * is bit-identical to the exported one (bounds, mu, every coefficient), lives on the current device and is independent of it. */
int32_t eph_ephemeris_export(const eph_ephemeris *e, void *buf, uint64_t capacity, uint64_t *bytes);
int32_t eph_ephemeris_import(const void *buf, uint64_t bytes, eph_ephemeris **out);
/* The debug window's interpolation-error scan (ephemeris_explorer/src/ui/windows/debug.rs:182-238): advance `h` (an
 * eph_nbody over the same bodies, e.g. QuinlanTremaine12 with the ephemeris dt and a bound) step by step and, after
 * every step, compare each body's position with its spline in `e` at that epoch; max_error_m[b] = the maximum of
 * position.distance(traj_position) * 1e3 (-1 if no step was taken). Stops at n_steps or at the first StepError
 * (e.g. the bound); an epoch outside a spline gives EPH_EVAL_FAILED (the reference unwraps). */
int32_t eph_ephemeris_interpolation_errors(const eph_ephemeris *e, eph_nbody *h, int64_t n_steps, double *max_error_m,
                                           int64_t *steps_done);

/* AdaptiveMethodParams (integration/src/lib.rs:171-274); the app's values: h_init 60, h_max f64::MAX, tol 1e-3,
 * fac_min 1/5, fac_max 5, fac 9/10, n_max 1e6 (ephemeris_explorer/src/load/mod.rs:472-486) */
typedef struct eph_adaptive_params {
    double h_init, h_max, tol_position, tol_velocity, fac_min, fac_max, fac;
    uint32_t n_max;
} eph_adaptive_params;

/* n_craft spacecraft: t0[i], pos_xyz[3i..], vel_xyz[3i..]; method = an embedded ERK pair name ("Verner87",
 * "DormandPrince54", "DormandPrince87", "CashKarp45", "Fehlberg45", "Tsitouras75", "Verner98") or the embedded
 * ERKNG pair "Fine45" (runge_kutta/nystrom/explicit_generalized.rs; dynamics/spacecraft.rs:797) -- the app's eight.
 * Timelines (Timeline::new, spacecraft.rs:129-152) in CSR form: craft i owns burns burn_offset[i] ..
 * burn_offset[i+1]-1: [burn_start, burn_end), burn_acc_xyz in the burn frame, burn_ref = body index whose TNB frame
 * the burn is given in, or -1 for the inertial frame. burn_offset may be NULL (no burns). max_knots = slab depth
 * per craft (CubicHermiteSpline points, including the initial one). */
int32_t eph_craft_batch_create(const eph_ephemeris *e, int64_t n_craft, const double *t0, const double *pos_xyz,
                               const double *vel_xyz, const char *method, const eph_adaptive_params *params,
                               const int64_t *burn_offset, const double *burn_start, const double *burn_end,
                               const double *burn_acc_xyz, const int32_t *burn_ref, int32_t max_knots,
                               eph_craft_batch **out);
/* The order in which the massive bodies' terms are added in a craft's acceleration (Bodies::acceleration iterates an EntityHashMap,
 * ephemeris_explorer/src/dynamics/spacecraft.rs:164-165,222-228: unspecified upstream). Default: the ephemeris table's (file) order,
 * which is what the library test's IndexMap gives (ephemeris/tests/spacecraft_propagation.rs:226-240). `order` = a permutation of
 * 0 .. n_bodies-1 (position in the sum -> body), NULL = back to table order; takes effect for the steps that follow. */
int32_t eph_craft_batch_set_body_order(eph_craft_batch *b, const int32_t *order);
/* IncrementalPropagator::step_to for every craft: step() until solution.end() >= t_end (spacecraft.rs:598-615,
 * 691-693) or an error; per-craft outcomes via eph_craft_batch_status. Returns EPH_OK if the sweep ran. */
int32_t eph_craft_batch_propagate(eph_craft_batch *b, double t_end);
/* IncrementalPropagator::step n_steps times for every craft (ephemeris/src/lib.rs:40-47, spacecraft.rs:598-615): each
 * craft takes exactly n_steps accepted steps (one knot each) unless it fails or its knot slab fills. */
int32_t eph_craft_batch_step_n(eph_craft_batch *b, uint32_t n_steps);
/* A craft whose last step returned a StepError keeps that status and is NOT stepped by later propagate / step_n calls (a sweep's
 * drain loop must not re-attempt the failed craft of a batch on every pass). The reference's propagator remembers nothing: calling
 * step() again runs AdaptiveRungeKuttaIntegrator::advance once more from what the failed attempt left
 * (integration/src/runge_kutta/mod.rs:414-439) -- that is how a stored ship propagator resumes after the bodies' ephemeris has been
 * extended (prediction.rs:378). This call re-arms the batch: the NEXT propagate / step_n steps every craft whatever its status (a
 * craft that has reached t_end already returns EPH_OK, like step_to). What a failed attempt leaves, and the library keeps bit for bit:
 * time, state and the step counters of the last accepted step (`n` counts completed attempts only: mod.rs:427-428 returns before
 * n += 1); next_h after the rejections that preceded the failure and the clamp to the bound (:422-424); and the stage registers --
 * for an FSAL pair k[0] and k[S-1] were ALREADY swapped (explicit.rs:76-79), the failing stage is zeroed (:92), so the retry's swap
 * brings back the stage-0 derivative of the step BEFORE as its k[0]. That is a quirk of the reference (the retried step starts from a
 * stale first stage; the embedded error estimate usually rejects it once); it is restated, not repaired. Retrying
 * StepSizeUnderflow / MaxIterationsReached / BoundReached changes nothing and returns the same error. */
int32_t eph_craft_batch_retry_failed(eph_craft_batch *b);
/* per craft: status (eph_status or EPH_KNOTS_FULL), knots in the slab, attempts of the current integrator (n),
 * accepted steps since creation. Any pointer may be NULL. */
int32_t eph_craft_batch_status(eph_craft_batch *b, int32_t *status, int32_t *nknots, uint32_t *attempts,
                               uint32_t *steps);
/* current problem state of every craft: time, position, velocity, next step size */
int32_t eph_craft_batch_state(eph_craft_batch *b, double *t, double *pos_xyz, double *vel_xyz, double *next_h);
/* Everything eph_craft_batch_status and eph_craft_batch_state return, as ONE record per craft packed on the device and
 * brought over in one copy (a sweep of 1e5..1e6 craft reads its outcome this way: eight strided copies and their host
 * transposes were a third of a sweep's wall time). out: n_craft records. */
typedef struct eph_craft_record {
    double t, pos[3], vel[3], next_h;
    int32_t status, nknots;
    uint32_t attempts, steps;
} eph_craft_record;
int32_t eph_craft_batch_summary(eph_craft_batch *b, eph_craft_record *out);
/* the CubicHermiteSpline of one craft: nknots[craft] x (t, pos, vel) */
int32_t eph_craft_batch_knots(eph_craft_batch *b, int64_t craft, double *t, double *pos_xyz, double *vel_xyz);
/* The app's solout, SpacecraftSolout (ephemeris_explorer/src/dynamics/spacecraft.rs:514-587): besides the knots,
 * every accepted step is searched for sphere-of-influence crossings of every body and for apsides relative to the
 * current sphere's body (find_zero_crossing :112-162: sign test + bisection, <= 100 halvings, 1e-3 s), giving the
 * SoiTransitions (:303-375) and Apsides (:409-451) of SpacecraftSolution. soi_radius[b] per body (infinity for
 * the root; load/mod.rs:283-307). Call once, before the first propagate; from then on eph_craft_batch_propagate also
 * runs the event search on the steps it took. Bodies are visited in body order (the reference iterates an
 * EntityHashMap, whose order is unspecified). apsis kind: 0 = Periapsis, 1 = Apoapsis. */
int32_t eph_craft_batch_enable_events(eph_craft_batch *b, const double *soi_radius, int32_t max_transitions,
                                      int32_t max_apsides);
/* per craft: number of transitions / apsides, and EPH_OK or EPH_EVENTS_FULL (raised at a step boundary as soon as
 * fewer than two entries are free in either slab, so max_transitions and max_apsides must be >= 3). Any pointer may
 * be NULL. */
int32_t eph_craft_batch_event_counts(eph_craft_batch *b, int32_t *n_transitions, int32_t *n_apsides,
                                     int32_t *event_status);
/* one craft's sorted lists (arrays sized by eph_craft_batch_event_counts; any may be NULL) */
int32_t eph_craft_batch_events(eph_craft_batch *b, int64_t craft, double *tr_time, int32_t *tr_body, double *ap_time,
                               double *ap_distance, int32_t *ap_body, int32_t *ap_kind);
/* SpacecraftPropagator: Clone -- the UI snapshots a propagator and later resumes from the snapshot
 * (ephemeris_explorer/src/prediction.rs:224-229,378). A deep copy: state, knots and events; the clone refers to the
 * same (live) eph_ephemeris, which must outlive both: a clone taken before an eph_ephemeris_append resumes against the
 * extended table. */
int32_t eph_craft_batch_clone(eph_craft_batch *b, eph_craft_batch **out);
/* Bulk read of the knot slabs for sweeps (one copy instead of one strided gather per craft): knots first_knot ..
 * first_knot + n_knots - 1 of EVERY craft in the device layout, knot_t[k][craft] and knot_y[k][d][craft] (d = x, y, z,
 * vx, vy, vz); entries at or beyond a craft's nknots are unspecified. Either pointer may be NULL. (A heterogeneous batch
 * keeps its slabs in the order its craft were dealt to the lanes at creation; this call returns craft order all the same,
 * through a temporary of the requested size on the device.) */
int32_t eph_craft_batch_knot_slabs(eph_craft_batch *b, int32_t first_knot, int32_t n_knots, double *knot_t,
                                   double *knot_y);
/* Flight-plan restart (ephemeris_explorer/src/flight_plan.rs:263-303): Timeline::divergence_time_before
 * (ephemeris/src/propagators/spacecraft.rs:179-213) of the NEW burn list against the OLD one -- the start of the last
 * segment, earlier than `before`, up to which both timelines agree; the caller restarts a craft from the knot at
 * that epoch (max'ed with the trajectory start). Host-only logic (no device needed). Burns as in
 * eph_craft_batch_create. The reference panics if no common start precedes `before`: EPH_ERR_BAD_ARGUMENT. */
int32_t eph_timeline_divergence_time(int64_t n_old, const double *old_start, const double *old_end, const double *old_acc,
                                     const int32_t *old_ref, int64_t n_new, const double *new_start,
                                     const double *new_end, const double *new_acc, const int32_t *new_ref,
                                     double before, double *restart_epoch);
/* Drain point for long propagations: after the caller has read the knots it wants, the newest knot of every craft
 * becomes knot 0 of an otherwise empty slab (so consecutive pieces of the CubicHermiteSpline share their end point,
 * what CubicHermiteSpline::extend, ephemeris/src/trajectory.rs:842-844, needs to stitch them, minus the duplicate), EPH_KNOTS_FULL is cleared and the next
 * eph_craft_batch_propagate continues. Events already found are kept. */
int32_t eph_craft_batch_reset_knots(eph_craft_batch *b);
/* Drain point for the event slabs: after the caller has read them, only the newest transition of every craft (the
 * sphere it is in) is kept, apsides are emptied and EPH_EVENTS_FULL is cleared. A craft whose slab filled up in the
 * middle of a propagate call resumes its event search from the step where it stopped at the next propagate. */
int32_t eph_craft_batch_reset_events(eph_craft_batch *b);
int32_t eph_craft_batch_kernel_time(eph_craft_batch *b, double *total_ms);
void eph_craft_batch_destroy(eph_craft_batch *b);
/* CubicHermiteSpline::state_vector (ephemeris/src/trajectory.rs:766-797) at m epochs, on the device */
int32_t eph_hermite_eval(int64_t nknots, const double *t, const double *pos_xyz, const double *vel_xyz, int64_t m,
                         const double *at, double *out_pos_xyz, double *out_vel_xyz, uint8_t *inside);

/* ---- adaptive plot sampling (ephemeris_explorer/src/ui/world/plot.rs) -------------------------------
 * compute_plot_points_parallel (:272-374) + PlotPoints::new (:93-149) + angular_distance (:429-436) for a batch of
 * plotted trajectories, one device thread per plot (the sampler is a sequential adaptive loop per trajectory; the app
 * runs one per body and ship every frame). A plot's source is a body of the ephemeris (PredictionTrajectory::
 * UniformSpline) or a ship's CubicHermiteSpline given as knots; its optional reference is a body (PlotSource.reference).
 * What the UI contributes is passed in as numbers: the camera position and the floating-origin grid's affine map
 * (GridExt::to_global_sv, floating_origin.rs:28-50: M * (p - cell_offset) + T for points, M * v for vectors). */
typedef struct eph_plot_view {
    double camera_position[3];   /* camera_transform.translation().as_dvec3()   plot.rs:430 */
    double grid_matrix3[9];      /* local_floating_origin().grid_transform().matrix3, column major (x_axis, y_axis, z_axis) */
    double grid_translation[3];  /* .grid_transform().translation */
    double cell_offset[3];       /* grid.cell_to_float(&origin.cell()) */
    double current;              /* sim_time.current(), seconds since 1958-01-01 TAI */
} eph_plot_view;
typedef struct eph_plot_request {      /* PlotConfig + PlotSource  plot.rs:15-83 */
    int32_t source_body;         /* >= 0: body of the ephemeris; -1: the hermite knots [knot_first, knot_first + knot_count) */
    int32_t reference_body;      /* body index or -1 (None) */
    int64_t knot_first, knot_count;
    double start, end;           /* PlotConfig.start / .end */
    int32_t bound;               /* PlotBound: 0 None, 1 Start, 2 End */
    int32_t enabled;
    double tan2_angular_resolution;   /* (plot.threshold * ARC_MINUTE * perspective.fov) as f64   :326-327 */
    int64_t max_points;
} eph_plot_request;
/* out_t[p][k], out_xyz[p][k][3] (f32: `as_vec3()`), k < out_count[p] <= capacity (capacity >= every max_points).
 * out_status[p]: EPH_OK; EPH_EVAL_FAILED with out_failed_at[p] = the epoch where the reference's closure returns None
 * (the app then panics, :369); EPH_MAX_ITERATIONS_REACHED when the step-size search did not end within 2^20 trials (a
 * NaN error estimate makes the reference spin forever). Plots that draw nothing (disabled, empty relative trajectory,
 * min >= max) have count 0 and EPH_OK. */
int32_t eph_plot_points(const eph_ephemeris *e, const eph_plot_view *view, int64_t n_plots, const eph_plot_request *requests,
                        int64_t n_knots, const double *knot_t, const double *knot_pos_xyz, const double *knot_vel_xyz,
                        int64_t capacity, double *out_t, float *out_xyz, int64_t *out_count, int32_t *out_status,
                        double *out_failed_at);

/* SpacecraftPropagator::join(lhs, rhs) (ephemeris/src/propagators/spacecraft.rs:558-561; the app's
 * PredictionTarget::merge, ephemeris_explorer/src/dynamics/spacecraft.rs:830-841): lhs.clear_after(rhs.start())
 * -- keep the knots with t < rhs.start() (trajectory.rs:842-845; rhs.start() of an empty spline is Epoch::MIN,
 * :756-758) -- then lhs.extend(rhs) (:847-849). Host only (knot lists are host data). Writes the joined spline
 * to the *_out arrays (capacity knots; t_out / pos_out / vel_out may be the lhs arrays themselves) and its length
 * to *n_out; when capacity is too small nothing is written, *n_out is the needed length and the status is
 * EPH_ERR_BAD_ARGUMENT. This is what stitches the slab drained by eph_craft_batch_knots onto the trajectory the
 * caller already holds. */
int32_t eph_hermite_join(int64_t n_lhs, const double *t_lhs, const double *pos_lhs, const double *vel_lhs,
                         int64_t n_rhs, const double *t_rhs, const double *pos_rhs, const double *vel_rhs,
                         int64_t capacity, double *t_out, double *pos_out, double *vel_out, int64_t *n_out);

/* The event half of PredictionTarget::merge (ephemeris_explorer/src/dynamics/spacecraft.rs:836-839), host only:
 *   item.transitions.clear_after(solution.trajectory.start()); item.transitions.extend(solution.transitions);
 *   item.apsides.clear_after(solution.trajectory.start());     item.apsides.extend(solution.apsides);
 * `at` is solution.trajectory.start() (the first knot of the slab being merged). clear_after keeps the entries with
 * t <= at (:341-346, :431-436: Ok(i) => truncate(i + 1), Err(i) => truncate(i); with several apsides at exactly `at`
 * Rust's binary search may stop at any of them -- this keeps them all). SoiTransitions::extend inserts every entry
 * in time order, replaces an entry with the same time and drops one whose predecessor is the same body (:331-337,
 * :356-361); Apsides::extend appends (:426-428). lhs and rhs are the parallel arrays eph_craft_batch_events fills
 * (kind: 0 = Periapsis, 1 = Apoapsis); the outputs hold capacity >= n_lhs + n_rhs entries and may be the lhs
 * arrays themselves (never the rhs ones: the sorted insert moves entries); *n_out is the joined length. */
int32_t eph_transitions_join(int64_t n_lhs, const double *t_lhs, const int32_t *body_lhs, int64_t n_rhs, const double *t_rhs,
                             const int32_t *body_rhs, double at, int64_t capacity, double *t_out, int32_t *body_out,
                             int64_t *n_out);
int32_t eph_apsides_join(int64_t n_lhs, const double *t_lhs, const double *distance_lhs, const int32_t *kind_lhs,
                         const int32_t *body_lhs, int64_t n_rhs, const double *t_rhs, const double *distance_rhs,
                         const int32_t *kind_rhs, const int32_t *body_rhs, double at, int64_t capacity, double *t_out,
                         double *distance_out, int32_t *kind_out, int32_t *body_out, int64_t *n_out);

/* (The test and tuning hooks -- eph_debug_* -- are not part of this boundary: csrc/eph_debug.h, exported by the test-hooks
 * library libephemeris_amd_testhooks.so and by tuning builds only.) */

#ifdef __cplusplus
}
#endif
#endif /* EPHEMERIS_AMD_H */
