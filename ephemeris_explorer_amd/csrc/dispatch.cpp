// dispatch.cpp -- which kernel form runs a launch, and whose: the kernels that contain the point-mass term exist once per
// evaluation order of that term (namespaces eph::pv0 .. pv6, pair_ns.h); every handle carries the order it was created under,
// and the launch_* functions below route to that order's table after choosing the form (one wave per block / workgroup-
// specialised / single workgroup; bodies per wave or workgroup) from the number of TARGET bodies of the launch.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "eph_internal.h"

namespace eph {

namespace pv0 { const PairKernels *pair_table(); }
namespace pv1 { const PairKernels *pair_table(); }
namespace pv2 { const PairKernels *pair_table(); }
namespace pv3 { const PairKernels *pair_table(); }
namespace pv4 { const PairKernels *pair_table(); }
namespace pv5 { const PairKernels *pair_table(); }
namespace pv6 { const PairKernels *pair_table(); }

static const PairKernels *table(int pv) {
    static const PairKernels *const t[kPairVariants] = {pv0::pair_table(), pv1::pair_table(), pv2::pair_table(), pv3::pair_table(),
                                                        pv4::pair_table(), pv5::pair_table(), pv6::pair_table()};
    return pv >= 0 && pv < kPairVariants ? t[pv] : nullptr;
}

static std::atomic<int> g_default_pv{-1};
int default_pair_variant() {
    int v = g_default_pv.load(std::memory_order_relaxed);
    if (v < 0) {                                       // first use: the environment's choice, else order 0
        const char *e = getenv("EPH_PAIR_VARIANT");
        v = e && *e >= '0' && *e <= '6' && !e[1] ? *e - '0' : 0;
        g_default_pv.store(v, std::memory_order_relaxed);
    }
    return v;
}
int set_default_pair_variant(int pv) {
    if (pv < 0 || pv >= kPairVariants) return EPH_ERR_BAD_ARGUMENT;
    g_default_pv.store(pv, std::memory_order_relaxed);
    return EPH_OK;
}

// ---- the choice of form (measured on MI355X; us per step, QT12, order 0: workgroup form with 4 / 8 / 16 bodies | wave form) ------
//   n = 512 9.0 | 9.05, 640 9.7 | 10.7, 1024 11.9 | 14.2, 1536 15.4 (14.8 six-wave) | 22.2, 2048 18.3 (17.7) | 25.1, 4096 36.9 | 54, 8192 134 | 143,
//   16384 511 | 552, 65536 7864 | 8823 (profiles/r03_time_sizes.txt)
static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
// bodies per wave of the wave form: enough waves to cover the 1024 SIMDs of the chip, as many bodies per wave as that allows
int lm_bodies_per_wave(int n) {
    static const int forced = env_int("EPH_BPW", 0);   // tuning override (1, 2, 4, 8)
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    if (n >= 8 * 1024) return 8;
    if (n >= 4 * 1024) return 4;
    if (n >= 2 * 1024) return 2;
    return 1;
}
// bodies per workgroup: 16, or 8 / 4 when 16-body workgroups would leave CUs idle (one workgroup per CU either way, and the
// chain wave's time per tile does not depend on how many chains it carries)
static int wg_bodies(int nt) {
    static const int forced = env_int("EPH_WG_BODIES", 0);
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    if (forced == 9) return 9;
    // 9 = the six-wave 8-body workgroup (step_wg.hip): 13.2 / 14.8 / 16.3 / 17.7 us at 1280 / 1536 / 1792 / 2048 bodies against the
    // twelve-wave 8-body form's 14.0 / 15.6 / 17.1 / 18.6 (round 5)
    return nt <= 1024 ? 4 : nt <= 2048 ? 9 : 16;
}
// 1 = one wave per block (step_wave.hip), 2 = workgroup-specialised (step_wg.hip)
static int force_kernel_kind(int nt, int requested) {
    if (requested == 1 || requested == 2) return requested;
    static const int forced = [] {
        const char *e = getenv("EPH_FORCE");           // tuning override: "wave" | "wg"
        return !e || !*e ? 0 : (e[1] == 'a' ? 1 : 2);
    }();
    if (forced) return forced;
    return nt > 512 ? 2 : 1;
}

int launch_accel(int pv, hipStream_t s, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out, int kind,
                 int lo, int hi, const KickDrift *kdp) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    if (hi < 0) hi = n;
    const KickDrift kd = kdp ? *kdp : KickDrift{nullptr, nullptr, 0.0, 0.0, nullptr};
    const int nt = hi - lo;                            // targets of this launch; the kernel choice follows them
    if (n <= 0 || nt <= 0) return EPH_OK;
    // (the six-wave 8-body form exists for the step kernel only: the start-up's evaluations take the twelve-wave one)
    if (force_kernel_kind(nt, kind) == 2) return t->accel_wg(s, wg_bodies(nt) == 9 ? 8 : wg_bodies(nt), n, npad, pos, acc_init, acc_out, lo, hi, kd);
    return t->accel_wave(s, lm_bodies_per_wave(nt), n, npad, pos, acc_init, acc_out, lo, hi, kd);
}
int launch_lm_step(int pv, hipStream_t s, const LmArgs &a) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    if (a.n <= 0 || a.hi <= a.lo) return EPH_OK;
    const int nt = a.hi - a.lo;
    if (force_kernel_kind(nt, a.kind) == 2) return t->lm_step_wg(s, wg_bodies(nt), a);
    return t->lm_step_wave(s, lm_bodies_per_wave(nt), a);
}
int launch_lm_persistent(int pv, hipStream_t s, const LmArgs &a, int64_t nsteps) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    if (a.n <= 0 || nsteps <= 0) return EPH_OK;
    if (a.n > kSmallN) return EPH_ERR_UNSUPPORTED;
    static const int first_design = env_int("EPH_SMALL", 0) == 1;      // tuning: k_lm_persistent for every n <= 64
    if (!first_design && a.n <= kGangMaxN) {
        LmArgs b = a;
        b.wg_flags = env_int("EPH_DEBUG_SMALL", 0);    // tuning builds: 1 no pair stage, 2 no chain stage, 4 tick accounting
        return t->lm_small(s, b, nsteps);
    }
    return t->lm_persistent(s, a, nsteps);
}
int launch_lm_small_many(int pv, hipStream_t s, const LmArgs *argv_dev, int count, int L, int64_t nsteps) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    if (count <= 0 || nsteps <= 0) return EPH_OK;
    return t->lm_small_many(s, argv_dev, count, L, nsteps);
}
// slices of the fast path: enough waves for two per SIMD (2048; FOUR for the rsq form, whose shorter dependent chains want them), a
// multiple of the workgroup's 4, at least 8, at most 64
int fast_slices(int npad, bool approx) {
    static const int forced = env_int("EPH_FAST_SLICES", 0);
    int S = forced > 0 ? forced : (approx ? 4096 : 2048) / (npad / 64);
    S = std::max(8, std::min(64, S));            // (8 rather than 4 slices at 65 536 bodies, eight waves per SIMD: 1.17 -> 1.13 ms on the f32 path)
    return (S + 3) / 4 * 4;
}
// the fast paths' scratch: [S][3][npad] partial sums (the larger of the two slice counts) + one arrival ticket per block of 64 targets
size_t fast_partial_doubles(int npad) { return (size_t)fast_slices(npad, true) * 3 * npad + (size_t)(npad / 64 + 2) / 2 + 1; }
size_t fast_ticket_offset_doubles(int npad) { return (size_t)fast_slices(npad, true) * 3 * npad; }
int launch_lm_step_fast(int pv, hipStream_t s, const LmArgs &a, double *partial, bool approx, float *posf, int f32_stage, int conv_lo,
                        int conv_cnt) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    if (a.n <= 0) return EPH_OK;
    // only the binary32 pair arithmetic runs on a target partition (BASELINE configs[4]); the f64 reordered paths stay single-device
    if (!posf && (a.lo != 0 || a.hi != a.n)) return EPH_ERR_UNSUPPORTED;
    if (conv_cnt < 0) { conv_lo = 0; conv_cnt = a.npad; }
    if (f32_stage != 1 && a.hi <= a.lo) return EPH_OK;                  // a rank whose slice is all padding
    static const int unroll = env_int("EPH_FAST_UNROLL", 4) == 8 ? 8 : 4;
    // One launch per step (k_fast_step: the last workgroup of a block finishes it) for the IEEE and the binary32 forms, two launches
    // for the rsq form; EPH_FAST_FUSED=0|1 forces one or the other. Measured on one box, us per step, one launch / two launches
    // (scripts/time_fast.py, profiles/r06_fast_fused.md): fast 31.4 / 32.8, f32 pairs 14.3 / 15.1 at N = 4096 and 1114 / 1125 at
    // 65 536, fast-rsq 22.1 / 20.1-21.6 -- the finish is latency-bound either way (64 dependent loads-then-adds per body), fusing
    // removes a launch gap and pays a barrier, a ticket and the epilogue's registers. The arrival tickets live behind the partial
    // sums (fast_partial_doubles).
    static const int forced = env_int("EPH_FAST_FUSED", -1);
    const bool fused = forced >= 0 ? forced != 0 : !approx;
    unsigned *tickets = fused ? reinterpret_cast<unsigned *>(partial + fast_ticket_offset_doubles(a.npad)) : nullptr;
    return t->lm_step_fast(s, a, partial, fast_slices(a.npad, approx), unroll, approx, posf, f32_stage, conv_lo, conv_cnt, tickets);
}
int launch_craft(int pv, hipStream_t s, const CraftArgs &a, const CraftLaunch &how) {
    const PairKernels *t = table(pv);
    return t ? t->craft_launch(s, a, how) : EPH_ERR_BAD_ARGUMENT;
}
int launch_debug_inv_r3(int pv, hipStream_t s, int64_t n, const double *n2, double *fast, double *ieee) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    return n <= 0 ? EPH_OK : t->debug_inv_r3(s, n, n2, fast, ieee);
}
int launch_debug_inv_r3_sweep(int pv, hipStream_t s, uint64_t seed, int64_t n, unsigned long long *out2) {
    const PairKernels *t = table(pv);
    return t ? t->debug_inv_r3_sweep(s, seed, n, out2) : EPH_ERR_BAD_ARGUMENT;
}
int launch_debug_quot(int pv, hipStream_t s, int64_t n, const double *x, const double *a, double *fast, double *ieee) {
    const PairKernels *t = table(pv);
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    return n <= 0 ? EPH_OK : t->debug_quot(s, n, x, a, fast, ieee);
}
int debug_wg_cycles(int pv, long long *out) {
    const PairKernels *t = table(pv);
    return t ? t->debug_wg_cycles(out) : EPH_ERR_BAD_ARGUMENT;
}

}  // namespace eph
