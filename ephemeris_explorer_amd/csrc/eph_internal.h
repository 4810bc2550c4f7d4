// eph_internal.h -- shared declarations of libephemeris_amd (host C++ + HIP kernels for gfx950).
//
// Device data layout (all f64, resident in HBM for the life of a handle):
//   P[2]      : packed bodies {x, y, z, mu} (32 B each, one 2x dwordx4 load per body) -- what the pair kernel
//               streams; ping-pong so a launch can publish the next positions while its peers still read
//               the current ones.
//   Y, A      : history rings [L][3][npad] (level, component, body): positions and accelerations of the last
//               L = ORDER integrator levels (12 for QuinlanTremaine12). Slot `cur` = newest level,
//               (cur + j) % L = j levels back.
//   V         : [3][npad] current velocity.  ASR: [3][npad] SRKN stage acceleration (FSAL carry).
//   samples   : per-body append log of sampled positions (AoS xyz) for the solout windows.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ephemeris_amd.h"

namespace eph {

constexpr int kMaxOrder = 16;   // ELM2 orders: 12 (QT12), 13 (Stormer13)
constexpr int kTile = 64;       // one wave64 = one tile of source bodies
constexpr int kRow = 66;        // LDS row stride in doubles of the contribution tile (16-B aligned, conflict-free)
constexpr int kDiv = 8;         // ephemeris/src/trajectory.rs:335  (DIV)
constexpr int kSmallN = 64;     // persistent single-workgroup path handles n <= one tile

struct Body4 { double x, y, z, mu; };
static_assert(sizeof(Body4) == 32, "Body4 must be 32 bytes");

// thread-local HIP error text
void set_last_error(const char *what, hipError_t e);
void set_last_error_text(const std::string &s);
int check_device();  // EPH_OK or EPH_ERR_NO_DEVICE

#define EPH_HIP(call)                                              \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) {                                    \
            ::eph::set_last_error(#call, e_);                      \
            return EPH_ERR_HIP;                                    \
        }                                                          \
    } while (0)

// ---- coefficient tables -> the f64 values the reference multiplies with (coeffs.cpp) ------------
struct SrknCoeffs { int stages = 0; bool fsal = false; double A[32], B[32]; };
struct Elm2Coeffs {
    int order = 0;
    double wa[kMaxOrder], wb[kMaxOrder], cw[kMaxOrder];
    double inv_beta_d = 0, inv_cowell_d = 0;
};
struct ErkCoeffs {             // ERK + embedded pair; A[s][j] for j < s   (integration/src/runge_kutta/explicit.rs:14-38)
    int stages = 0, order = 0, order_embedded = 0;
    int fsal = 0, has_embedded = 0;
    double A[16][16], B[16], C[16], E[16];
    // nystrom = 1: ERKNG pair (runge_kutta/nystrom/explicit_generalized.rs:14-41), at most 8 stages:
    // A = AP, A2 = AV, B = BP, B2 = BV, E = EP, E2 = EV. nystrom = 2: ERKN (nystrom/explicit.rs), the same with A2 = 0
    int nystrom = 0;
    double A2[8][8], B2[8], E2[8];
};
bool find_erk(const char *name, ErkCoeffs *out);
bool find_srkn(const char *name, SrknCoeffs *out);
bool find_elm2(const char *name, Elm2Coeffs *out);

// ---- kernel argument blocks ---------------------------------------------------------------------
struct SampleArgs {            // solout sampling schedule for one batch (device arrays, may be null)
    const uint32_t *period;    // m_b: steps between samples (0 = never)
    const uint32_t *phase;     // r_b: steps since the last sample at batch start
    const uint64_t *offset;    // first free slot of body b in `log` (in samples)
    double *log;               // AoS xyz
};

struct LmArgs {                // fused linear-multistep step (step_wg.hip, step_wave.hip, step_small.hip: k_lm_step_wg / k_lm_step / k_lm_small / k_lm_persistent)
    int n, npad, L, cur;       // cur = ring slot of the level whose acceleration is evaluated
    int lo, hi;                // target bodies [lo, hi) of this launch (0, n unless the system is sharded over ranks)
    const Body4 *pos_cur;      // packed positions of that level
    Body4 *pos_next;           // packed positions of the next level (written when do_predict)
    double *Y, *A, *V;
    double wa[kMaxOrder], wb[kMaxOrder], cw[kMaxOrder];
    double h, hh, hc;          // h, h*h*(1/BETA_D), h*(1/COWELL_D)
    int do_predict;
    uint32_t step;             // 1-based index of this step inside the batch (for sampling)
    int kind;                  // force kernel: 0 auto, 1 wave, 2 workgroup
    int wg_flags;              // workgroup kernel tuning bits (tuning builds: EPH_DEBUG_SMALL, step_small.hip)
    SampleArgs samp;
};

// ---- launchers -------------------------------------------------------------------------------------------------------------------
// The kernels that contain the point-mass term exist once per evaluation order of that term (pair_term.h, pair_ns.h): `pv` (0..6)
// says whose. A handle fixes it at creation (eph_set_pair_variant / EPH_PAIR_VARIANT); dispatch.cpp chooses the kernel form
// (wave / workgroup / single workgroup, bodies per wave or workgroup) and routes to that order's table.
constexpr int kPairVariants = 7;
int default_pair_variant();                 // what new handles take: eph_set_pair_variant, else EPH_PAIR_VARIANT, else 0
int set_default_pair_variant(int pv);       // EPH_ERR_BAD_ARGUMENT outside 0..6
// a[b] = acc_init[b] (or 0) + sum over the other bodies in the reference order; SoA [3][npad] output
// kind: 0 = auto by the number of targets, 1 = one wave per block (wave_force), 2 = workgroup-specialised (wg_force)
// lo, hi: target bodies [lo, hi) (hi < 0: n); lo must be a multiple of 16
// kd (optional): the SRKN stage update fused behind the evaluation, per (body, component) of the launch's targets:
//   v += a * hb ; y += v * ha ; pos_out = y      (symplectic.rs:90-97) -- one launch per stage instead of two
struct KickDrift { double *v, *y; double hb, ha; Body4 *pos_out; };
int launch_accel(int pv, hipStream_t s, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out,
                 int kind = 0, int lo = 0, int hi = -1, const KickDrift *kd = nullptr);
int launch_lm_step(int pv, hipStream_t s, const LmArgs &a);                 // one fused step, all CUs
int launch_lm_persistent(int pv, hipStream_t s, const LmArgs &a, int64_t nsteps);
// `count` single-workgroup systems (n <= kGangMaxN each, same L), one workgroup each, argv in device memory
constexpr int kGangMaxN = 32;
int launch_lm_small_many(int pv, hipStream_t s, const LmArgs *argv_dev, int count, int L, int64_t nsteps);
// opt-in fast path: slice-parallel partial sums combined in slice order (NOT the reference's summation order)
int fast_slices(int npad, bool approx = false);                     // S (the rsq form takes twice the waves)
// posf: EPH_PATH_F32_PAIRS scratch, 4 floats per padded body. f32_stage 0: the whole step; 1: only the binary32 copy of rows
// [conv_lo, conv_lo + conv_cnt) (conv_cnt < 0: all); 2: the step on a copy that is complete already (a sharded handle gathers between)
size_t fast_partial_doubles(int npad);           // scratch of the fast paths incl. the arrival tickets behind the partial sums
size_t fast_ticket_offset_doubles(int npad);
int launch_lm_step_fast(int pv, hipStream_t s, const LmArgs &a, double *partial, bool approx, float *posf = nullptr, int f32_stage = 0,
                        int conv_lo = 0, int conv_cnt = -1);
// the massless sweep (craft_sweep.hip); CraftArgs: craft_device.h
struct CraftArgs;
struct CraftLaunch { bool wave_form, queue, occ2; long long resident_waves; };
int launch_craft(int pv, hipStream_t s, const CraftArgs &a, const CraftLaunch &how);
// test hooks: 1/(x*sqrt(x)) and a/(x*sqrt(x)) through the in-range sequences and through the compiler's IEEE expansions
int launch_debug_inv_r3(int pv, hipStream_t s, int64_t n, const double *n2, double *fast, double *ieee);
int launch_debug_inv_r3_sweep(int pv, hipStream_t s, uint64_t seed, int64_t n, unsigned long long *out2);   // rounds n up to 2^20
int launch_debug_quot(int pv, hipStream_t s, int64_t n, const double *x, const double *a, double *fast, double *ieee);
int debug_wg_cycles(int pv, long long *out);   // tuning builds (-DEPH_EXPERIMENTS): cycle accounting of the workgroup / small kernels
struct PairKernels {                        // one evaluation order's launchers (pair_launchers.h), filled in step_wave.hip
    int (*accel_wave)(hipStream_t, int, int, int, const Body4 *, const double *, double *, int, int, const KickDrift &);
    int (*accel_wg)(hipStream_t, int, int, int, const Body4 *, const double *, double *, int, int, const KickDrift &);
    int (*lm_step_wave)(hipStream_t, int, const LmArgs &);
    int (*lm_step_wg)(hipStream_t, int, const LmArgs &);
    int (*lm_persistent)(hipStream_t, const LmArgs &, int64_t);
    int (*lm_small)(hipStream_t, const LmArgs &, int64_t);
    int (*lm_small_many)(hipStream_t, const LmArgs *, int, int, int64_t);
    int (*lm_step_fast)(hipStream_t, const LmArgs &, double *, int, int, bool, float *, int, int, int, unsigned *);
    int (*craft_launch)(hipStream_t, const CraftArgs &, const CraftLaunch &);
    int (*debug_inv_r3)(hipStream_t, int64_t, const double *, double *, double *);
    int (*debug_inv_r3_sweep)(hipStream_t, uint64_t, int64_t, unsigned long long *);
    int (*debug_quot)(hipStream_t, int64_t, const double *, const double *, double *, double *);
    int (*debug_wg_cycles)(long long *);
    int variant;
};
// ---- launchers of the kernels without a point-mass term (solout.hip) --------------------------------------------------------------
int launch_pack(hipStream_t s, int n, int npad, const double *Yslot, const double *mu, Body4 *pos);
int launch_copy3(hipStream_t s, int n, int npad, const double *src, double *dst);
// SRKN stage update: v += a*hb ; y += v*ha ; also publishes packed positions   (symplectic.rs:90-97)
int launch_kick_drift(hipStream_t s, int n, int npad, const double *a, double *v, double *y, double hb, double ha,
                      const double *mu, Body4 *pos_out);
int launch_lm_predict(hipStream_t s, const LmArgs &a);              // y_{m+1} from the ring (no force)
int lm_bodies_per_wave(int n);
int launch_sample(hipStream_t s, int n, int npad, const double *Yslot, const SampleArgs &sa, uint32_t step);
// carry: samples [src[b], src[b]+cnt[b]) of body b's region move to its front (src[b] == 0: nothing to do)
int launch_carry(hipStream_t s, int n, const uint64_t *region, const uint32_t *src, const uint32_t *cnt, double *log);
// AoS <-> SoA staging
int launch_aos_to_soa(hipStream_t s, int n, int npad, const double *aos, double *soa);
int launch_soa_to_aos(hipStream_t s, int n, int npad, const double *soa, double *aos);
// LeastSquaresFit over windows of 9 samples; window w of the launch reads log[(first[w]) .. +8]
int launch_lsq_fit(hipStream_t s, int64_t nwin, const uint64_t *first_sample, const uint8_t *degree, int backward,
                   const double *log, double *coeffs, int32_t *ncoef);
// records of kDiv*3 + 1 doubles per window (coefficients, then ncoef) for the sharded propagator's all-gather
int launch_pack_records(hipStream_t s, int64_t nwin, const double *coeffs, const int32_t *ncoef, double *rec);
int launch_spline_eval(hipStream_t s, int64_t m, const double *at, double start, double interval, int64_t npoly,
                       const double *coeffs, const int32_t *ncoef, double *pos, double *vel, uint8_t *inside);

}  // namespace eph
