// step_small.hip -- k_lm_small: a system of at most 32 bodies (the reference's shipped full_solar_system has exactly 32) in ONE
// workgroup, `nsteps` integrator steps per launch; and the gang form, one workgroup per SYSTEM (eph_nbody_advance_many).
// Compiled once per evaluation order of the point-mass term (pair_ns.h).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <type_traits>
#include <utility>

#include "pair_ns.h"

namespace eph {
namespace EPH_PV_NS {

#if EPH_EXPERIMENTS
__device__ long long g_small_ticks[8];               // (tuning builds) shader-clock accounting of workgroup 0
__device__ long long g_small_place[4][1024];         // (tuning builds) HW_ID, XCC_ID + 1, start and end (100 MHz) of every workgroup
#endif

// ------------------------------------------------------------------------------------------------------
// k_lm_small: n <= 64, the whole system in ONE workgroup, `nsteps` integrator steps per launch (second design
// of the persistent kernel; k_lm_persistent above is the first and stays selectable for comparison).
// Inside one workgroup the pair symmetry the reference uses CAN be shared: thread p owns the unordered pair
// (i, j), i < j, computes d, n2, 1/(n2*sqrt(n2)) once and writes both directed contributions
//     c(i<-j) =  d * (mu_j * inv)  -> U[i][j]   ("sources after the body")
//     c(j<-i) = -d * (mu_i * inv)  -> Lw[j][i]  ("sources before the body")
// exactly the reference's acceleration_paired halves. Rows of U / Lw are zero outside those ranges (written once
// at kernel start), so the two ordered chains of a body are plain in-order sums over a row (adding +0.0 is exact;
// the accumulators start at +0.0 and never become -0.0), one thread per (body, component, half):
//     ddy[i] = (0 + c(0,i) + ... + c(i-1,i)) + (0 + c(i,i+1) + ... + c(i,n-1)).
// Two barriers per step; history ring, velocity, predictor and Cowell formula live in the (body, component) thread.
// ------------------------------------------------------------------------------------------------------
// the step loop of k_lm_small unrolled over the L ring rotations: copy K runs at rotation (L - K) % L
template <typename Step, int... Ks>
__device__ __forceinline__ void small_steps(Step &step, long long &s, long long nsteps, int &rot, bool &more,
                                            std::integer_sequence<int, Ks...>) {
    constexpr int L = sizeof...(Ks);
    (void)std::initializer_list<int>{(more ? (step(std::integral_constant<int, (L - Ks) % L>{}, s),
                                              rot = (L - Ks + L - 1) % L, more = ++s <= nsteps, 0)
                                           : 0)...};
}
constexpr int kSmallThreads = 512;        // eight waves: one unordered pair per thread (496 at 32 bodies), the 3 n chains x 2 halves in waves 0-2
constexpr int kSmallThreads2 = 256;       // the TWO form (gangs of more systems than CUs): four waves, two unordered pairs per thread
constexpr int kSmallMaxN = 32;           // bodies (the reference's shipped system has exactly 32); 33..64 -> k_lm_persistent
constexpr int kSmallRow = 32 + 2;        // doubles per row: 16-byte aligned rows an odd number of 16-byte units apart
constexpr int kSmallRows = 3 * kSmallMaxN;
// -DEPH_EXPERIMENTS=1 -DEPH_SMALL_ACCOUNT=1 (tuning build, scripts/build_exp.sh): thread 0 of k_lm_small accumulates shader-clock ticks per phase
// of a step into g_small_ticks[3..7] (wait at barrier A | sum1 + pair | wait at barrier B | row sums | sum2 + hand-over)
#if !EPH_EXPERIMENTS || !defined(EPH_SMALL_ACCOUNT)
#undef EPH_SMALL_ACCOUNT
#define EPH_SMALL_ACCOUNT 0
#endif
// (ONE rolled copy of the step, the history shifted through the registers every step -- 24 v_mov_b64 -- instead of twelve copies,
// one per ring rotation (45 KB of code): measured slower at every gang size, profiles/r03_small_kernel_evidence.md.)
#define SMALL_TICK(k) do { if constexpr (EPH_SMALL_ACCOUNT) { const long long now_ = (long long)__builtin_readcyclecounter(); acct[k] += now_ - acct_t; acct_t = now_; } } while (0)

// k_lm_small, round 3. The per-phase tick accounting of round 2's kernel (thread 0, 32 bodies, 2125 ticks per step:
// wait 124 | sum1 + pair 802 | wait 183 | row sums 643 | sum2 + hand-over 373) and its ISA showed:
//   * the partner exchange of the two half sums was a ds_bpermute -- an LDS round trip between two dependent chains;
//   * the position half of the predictor (sum1) and the pair arithmetic sat in separate exec-masked regions, executed one
//     after the other instead of interleaved -- and once that was fixed in the source the compiler SANK sum1 back behind
//     the force (its only user is there), so it is pinned where it is computed;
//   * Cowell's velocity was formed at every step although nothing reads it before the launch ends;
//   * __syncthreads() waited for the solout's global stores as well.
// Built and measured on the way (gpurun_out r03, scripts/clock_small.py): contributions of both directions stored in one
// orientation ([i][c][j], rows of 49 doubles: conflict-free writes), the "before" chains walking columns through 32
// per-lane LDS addresses with everything outside a chain's range redirected to one shared zero -- SLOWER, 0.87 vs 0.80
// us per step: 32 ds_read_b64 per chain thread cost the workgroup's one LDS pipe more than 16 ds_read_b128 of half
// padding, and the transposed writes it removed were not what the pair phase waits for.
// MULTI: one workgroup per SYSTEM, its arguments argv[blockIdx.x] (eph_nbody_advance_many / eph_prop_step_n_many: the
// app runs a forward and a backward propagator concurrently, ephemeris_explorer/src/load/mod.rs:673-687, and ensembles
// are independent too): K latency-bound single-workgroup systems advance in the time of one.
template <int L, bool MULTI, bool TWO = false>
__global__ void __launch_bounds__(TWO ? kSmallThreads2 : kSmallThreads) k_lm_small(const LmArgs a0, const LmArgs *__restrict__ argv, long long nsteps) {
    __shared__ __attribute__((aligned(16))) double U[kSmallRows][kSmallRow];    // [body*3 + comp][source]: sources after the body
    __shared__ __attribute__((aligned(16))) double Lw[kSmallRows][kSmallRow];   //                        sources before the body
    __shared__ __attribute__((aligned(32))) Body4 sP[kTile];
    const LmArgs &a = MULTI ? argv[blockIdx.x] : a0;

    const int tid = threadIdx.x, n = a.n;
    // chain threads: tid = (body*3 + comp)*2 + half   (half 0 = sources before, 1 = sources after)
    const int chain = tid >> 1, half = tid & 1;
    const bool chain_thread = chain < 3 * n;
    const bool owner = chain_thread && half == 0;     // the (body, comp) thread: history, velocity, predictor
    // waves that hold chain threads (tid < 6 n): the only ones that carry the predictor's arithmetic (wave-uniform by construction)
    const bool owner_wave = __builtin_amdgcn_readfirstlane(tid >> 6) <= ((6 * n - 1) >> 6);
    const int my_i = chain_thread ? chain / 3 : 0, cc = chain_thread ? chain % 3 : 0;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + my_i;
    const int npairs = n * (n - 1) / 2;
    const int wg_flags = a.wg_flags;

    const int nrow = (n + 15) & ~15;   // row length the chains walk (the padding holds zeros)
    for (int k = tid; k < kSmallRows * kSmallRow; k += blockDim.x) { (&U[0][0])[k] = 0.0; (&Lw[0][0])[k] = 0.0; }
    if (tid < kTile) sP[tid] = a.pos_cur[tid < n ? tid : n - 1];
    // history and coefficients live in VGPRs for the whole launch: yv[j] / av[j] = level (newest - j).
    // (Kernel arguments would otherwise be re-fetched through the scalar cache every step.)
    double yv[L], av[L], wa[L], wb[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = a.Y[slot * lvl + off];
        av[j] = a.A[slot * lvl + off];
        wa[j] = a.wa[j]; wb[j] = a.wb[j];
        asm volatile("" : "+v"(wa[j]), "+v"(wb[j]));
    }
    double hh = a.hh;
    asm volatile("" : "+v"(hh));
    double v = owner ? a.V[off] : 0.0;
    // solout sampling schedule of this thread's body, read once (maybe_sample would fetch it every step)
    // (a countdown instead of `(phase + s) % period` every step: samples fall on the steps where phase + s is a
    // multiple of the period, the q-th of them into slot offset + q)
    uint32_t samp_m = 0, samp_left = 0;
    uint64_t samp_slot = 0;
    double *samp_log = a.samp.log;
    if (owner && a.samp.period) {
        samp_m = a.samp.period[my_i];
        samp_left = samp_m ? samp_m - a.samp.phase[my_i] % samp_m : 0;
        samp_slot = a.samp.offset[my_i];
    }
    // this thread's unordered pair (i < j), row-major over the strict upper triangle (n <= 32: at most 496 pairs for 512 threads),
    // decoded once. A slot without a pair runs pair (0, 1) again and stores nothing: straight-line code.
    // TWO (round 5): pairs `tid` and `tid + 256` in FOUR waves instead of one pair per thread in eight. Measured for one system it is
    // SLOWER (0.854 against 0.784 us per step at 32 bodies: one wave's two interleaved interactions cover less latency than two
    // waves' one each) -- but two such workgroups fit a CU, and a gang of more systems than the chip has CUs gains a third
    // (1024 systems: 2.63 against 3.51 us per step of the gang, 1.25e10 body-steps/s): lm_small_many takes it above 256 systems.
    auto decode = [&](int q, int &pi, int &pj) {
        pi = 0; pj = 1;
        if (q < npairs) {
            int i = 0;
            while ((i + 1) * (2 * n - i - 2) / 2 <= q) ++i;
            pi = i;
            pj = i + 1 + (q - i * (2 * n - i - 1) / 2);
        }
    };
    int pi0, pj0, pi1 = 0, pj1 = 1;
    decode(tid, pi0, pj0);
    if constexpr (TWO) decode(tid + kSmallThreads2, pi1, pj1);
    const bool live0 = tid < npairs && !(wg_flags & 1);    // (wg_flags: tuning switches, EPH_DEBUG_SMALL: 1 no pair stores, 4 accounting; 0 in normal runs)
    const bool live1 = TWO && tid + kSmallThreads2 < npairs && !(wg_flags & 1);
    __syncthreads();

    // Every thread runs the pair arithmetic: straight-line code, so the scheduler interleaves it with the predictor's
    // position chain (sum1) instead of executing one exec-masked region after the other. The wrapper-free sqrt /
    // reciprocal sequences run first and unconditionally; the range test that validates them (pair_term.h) is decided
    // behind them, where the branch no longer stalls the wave, and an out-of-range operand anywhere in the wave redoes
    // the term in the full IEEE form.
    // `before_branch()` runs between the wrapper-free results and the (rare, wave-uniform) branch that redoes them in the full
    // IEEE form: the owner waves pin the predictor's position chain there, so that it is emitted in the SAME block as the
    // sequences and fills their issue gaps; without it the compiler places the chain behind the stores (round 5, read off the ISA).
    auto pair = [&](const double4 &vi, const double4 &vj, const double4 &wi, const double4 &wj, auto &&before_branch) {
        const double dx = vj.x - vi.x, dy = vj.y - vi.y, dz = vj.z - vi.z;
        const double n2 = dx * dx + dy * dy + dz * dz;
        double ex = 0.0, ey = 0.0, ez = 0.0, m2 = 1.0;
        if constexpr (TWO) { ex = wj.x - wi.x; ey = wj.y - wi.y; ez = wj.z - wi.z; m2 = ex * ex + ey * ey + ez * ez; }
        double ax, ay, az, bx, by, bz, cx = 0.0, cy = 0.0, cz = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
        {
            // (TWO: the compiler emits the two reciprocal-cube sequences one after the other; writing them stage by stage with the
            // order pinned, as pair_finish_staged does for the workgroup kernel, was measured: no change, 0.87 us either way)
            const PairDen den = pair_den<true>(n2);
            pair_apply<true>(den, dx, dy, dz, vj.w, ax, ay, az);
            pair_apply<true>(den, -dx, -dy, -dz, vi.w, bx, by, bz);
            if constexpr (TWO) {
                const PairDen den1 = pair_den<true>(m2);
                pair_apply<true>(den1, ex, ey, ez, wj.w, cx, cy, cz);
                pair_apply<true>(den1, -ex, -ey, -ez, wi.w, gx, gy, gz);
            }
        }
        // the wrapper-free results exist HERE (otherwise the compiler sinks them into the else-side of the branch below)
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz));
        if constexpr (TWO) asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz), "+v"(gx), "+v"(gy), "+v"(gz));
        before_branch();
        if (__builtin_amdgcn_ballot_w64(!(in_range(n2) && (!TWO || in_range(m2)))) != 0) {
            const PairDen den = pair_den<false>(n2);
            pair_apply<false>(den, dx, dy, dz, vj.w, ax, ay, az);
            pair_apply<false>(den, -dx, -dy, -dz, vi.w, bx, by, bz);
            if constexpr (TWO) {
                const PairDen den1 = pair_den<false>(m2);
                pair_apply<false>(den1, ex, ey, ez, wj.w, cx, cy, cz);
                pair_apply<false>(den1, -ex, -ey, -ez, wi.w, gx, gy, gz);
            }
        }
        if (live0) {
            U[pi0 * 3 + 0][pj0] = ax;
            U[pi0 * 3 + 1][pj0] = ay;
            U[pi0 * 3 + 2][pj0] = az;
            Lw[pj0 * 3 + 0][pi0] = bx;
            Lw[pj0 * 3 + 1][pi0] = by;
            Lw[pj0 * 3 + 2][pi0] = bz;
        }
        if (live1) {
            U[pi1 * 3 + 0][pj1] = cx;
            U[pi1 * 3 + 1][pj1] = cy;
            U[pi1 * 3 + 2][pj1] = cz;
            Lw[pj1 * 3 + 0][pi1] = gx;
            Lw[pj1 * 3 + 1][pi1] = gy;
            Lw[pj1 * 3 + 2][pi1] = gz;
        }
    };

    // ---- predictor (ELM2::advance) of the first step, in the (body, comp) threads
    double ynew = 0.0;
    if (owner) {
        ynew = lm_predict<L>(yv, av, wa, wb, hh);
        reinterpret_cast<double *>(&sP[my_i])[cc] = ynew;
    }
    // One step with the history ring at rotation R: level (newest - j) lives in yv[(R + j) % L]. The new level
    // overwrites the oldest in place, so the ring never moves through registers; the step loop is unrolled over
    // the L rotations (R is a compile-time constant in each copy).
    // What is NOT done every step: Cowell's velocity (cowell.rs:17-53). The recurrence never reads it -- only
    // get_state / a clone / the next launch do -- so it is formed once, for the last level of the launch, from the same
    // twelve accelerations and two positions the reference would have used at that step: the same bits, eleven
    // multiply-adds per component and step less in the threads every barrier waits for.
    long long acct[5] = {0, 0, 0, 0, 0}, acct_t = EPH_SMALL_ACCOUNT ? (long long)__builtin_readcyclecounter() : 0;
    auto step = [&](auto rc, long long s) {
        constexpr int R = decltype(rc)::value;
        constexpr int Rn = (R + L - 1) % L;            // slot of the oldest level = slot of the level being built
        lds_barrier();     // positions of the new level visible
        SMALL_TICK(0);
        // ---- the position half of the NEXT level's predictor (sum1 of ELM2::advance): it needs this level's position,
        // not its acceleration, so its chain of dependent adds runs here, interleaved with the pair arithmetic below
        // instead of behind the force where everybody waits for it (every thread: non-owners carry junk, unused)
        // (all the products first, then the dependent adds: a v_mul_f64 issued right in front of the v_add_f64 that needs it
        // costs the chain its full latency every term -- measured ~25 cycles per term instead of ~8.4, 330 cycles for the
        // twelve terms behind the force. The eleven products of the acceleration half that do not involve the new
        // acceleration are formed here too.)
        // ONLY the waves that hold owners run it (round 5): every wave executed these 35 f64 instructions for the sake of three,
        // and the phase is issue-bound with two waves per SIMD (accounting build: the first wave of a SIMD left at 581 ticks, the
        // second at 962, without the stores). A wave-uniform branch; both sides hold the pair arithmetic as straight-line code.
        double q2[L], s1;                              // (set and read in the owner waves only)
        // the pair's two bodies: the reads are in flight under the products below
        const double4 vi = *reinterpret_cast<const double4 *>(&sP[pi0]), vj = *reinterpret_cast<const double4 *>(&sP[pj0]);
        double4 wi = vi, wj = vj;
        if constexpr (TWO) { wi = *reinterpret_cast<const double4 *>(&sP[pi1]); wj = *reinterpret_cast<const double4 *>(&sP[pj1]); }
        if (owner_wave) {
            double p1[L];
            p1[0] = ynew * wa[0];
#pragma unroll
            for (int j = 1; j < L; ++j) { p1[j] = yv[(R + j - 1) % L] * wa[j]; q2[j] = av[(R + j - 1) % L] * wb[j]; }
            __builtin_amdgcn_sched_barrier(0);
            s1 = 0.0;
#pragma unroll
            for (int j = 0; j < L; ++j) s1 = s1 + p1[j];
            // ---- pairs (i < j): one reciprocal cube per unordered pair, both directed contributions
            pair(vi, vj, wi, wj, [&]() {
                asm volatile("" : "+v"(s1));           // computed HERE, among the sequences
#pragma unroll
                for (int j = 1; j < L; ++j) asm volatile("" : "+v"(q2[j]));
            });
        } else {
            pair(vi, vj, wi, wj, []() {});
        }
        if constexpr (EPH_SMALL_ACCOUNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SMALL_TICK(1);
        lds_barrier();     // contributions visible
        SMALL_TICK(2);
        // ---- ordered chains: plain in-order sums over the rows (zeros outside each chain's range; adding +0.0 is exact and
        // the sums never are -0.0), every read in flight before the first add
        double acc = 0.0;
        if (chain_thread && !(wg_flags & 2)) {
            const double *row = half ? &U[chain][0] : &Lw[chain][0];
            auto sum_blocks = [&](auto nb) {
                constexpr int NB = decltype(nb)::value;
                // (the optimiser hoists the common first eight reads and two adds of the two row lengths in front of the branch and
                // issues the long row's other eight reads behind sixteen adds. Forcing all sixteen reads in front of the first add
                // was SLOWER -- row sums 610 -> 677 ticks: a lone wave's ds_read_b128 costs it ~12 issue cycles each and overlaps
                // nothing of its own, profiles/r02_chain2_ubench.txt -- so the accident stays; round 5)
                double2 r[8 * NB];
#pragma unroll
                for (int k = 0; k < 8 * NB; ++k) r[k] = *reinterpret_cast<const double2 *>(row + 2 * k);
#pragma unroll
                for (int k = 0; k < 8 * NB; ++k) {
                    acc = acc + r[k].x;
                    acc = acc + r[k].y;
                }
            };
            if (nrow == 16) sum_blocks(std::integral_constant<int, 1>{});
            else sum_blocks(std::integral_constant<int, 2>{});
        }
        // the partner half sits in the adjacent lane: a DPP quad permutation, not an LDS round trip (ds_bpermute)
        const double other = dpp_xor1(acc);
        if constexpr (EPH_SMALL_ACCOUNT) asm volatile("" :: "v"(other));
        SMALL_TICK(3);
        if (owner) {
            const double anew = acc + other;               // ddy[i] (lower sum) += output_i (upper sum)
            // the acceleration half of the predictor (sum2), the only chain behind the force
            double s2 = 0.0;
            s2 = s2 + anew * wb[0];
#pragma unroll
            for (int j = 1; j < L; ++j) s2 = s2 + q2[j];
            const double ynext = s1 + s2 * hh;             // *y = *sum1 + *sum2 * (h * h * Ratio::from_recip(BETA_D))
            if (s < nsteps) reinterpret_cast<double *>(&sP[my_i])[cc] = ynext;
            if (samp_m && --samp_left == 0) {              // SplineInterpolators::solout_with  nbody.rs:389-397
                samp_log[samp_slot * 3 + cc] = ynew;
                samp_slot += 1;
                samp_left = samp_m;
            }
            if (s == nsteps) {                             // Cowell::update_velocity of the launch's last level
                double al[L], cw[L];
#pragma unroll
                for (int j = 0; j < L; ++j) { al[j] = av[(R + j) % L]; cw[j] = a.cw[j]; }
                v = lm_cowell<L>(anew, al, ynew, yv[R], cw, a.h, a.hc);
            }
            yv[Rn] = ynew;
            av[Rn] = anew;
            ynew = ynext;
        }
        if constexpr (EPH_SMALL_ACCOUNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SMALL_TICK(4);
        // the predictor writes sP after every pair thread of this step passed the barrier above; U / Lw are
        // rewritten only after the next "positions visible" barrier
    };
    int rot = 0;                                       // rotation after the steps taken so far
#if EPH_EXPERIMENTS
    if ((wg_flags & 4) && tid == 0 && blockIdx.x < 1024) {   // tuning: where the dispatcher put this workgroup (HW_ID, XCC_ID)
        g_small_place[0][blockIdx.x] = (long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        g_small_place[1][blockIdx.x] = (long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11)) + 1;
        g_small_place[2][blockIdx.x] = (long long)wall_clock64();
    }
#endif
#if EPH_EXPERIMENTS
    const long long dbg_c0 = (wg_flags & 4) ? (long long)__builtin_readcyclecounter() : 0;
    const long long dbg_w0 = (wg_flags & 4) ? (long long)wall_clock64() : 0;
#endif
    {
        long long s = 1;
        bool more = nsteps >= 1;
        while (more) small_steps(step, s, nsteps, rot, more, std::make_integer_sequence<int, L>{});
    }
#if EPH_EXPERIMENTS
    if ((wg_flags & 4) && tid == 0 && blockIdx.x < 1024) g_small_place[3][blockIdx.x] = (long long)wall_clock64();
    if ((wg_flags & 4) && tid == 0 && blockIdx.x == 0) {   // tuning (EPH_DEBUG_SMALL=4): shader-clock ticks, 100 MHz ticks, steps
        g_small_ticks[0] = (long long)__builtin_readcyclecounter() - dbg_c0;
        g_small_ticks[1] = (long long)wall_clock64() - dbg_w0;
        g_small_ticks[2] = nsteps;
        if constexpr (EPH_SMALL_ACCOUNT)
            for (int q = 0; q < 5; ++q) g_small_ticks[3 + q] = acct[q];
    }
#else
    (void)acct;
#endif

    if (owner) {
        const int cur = (int)(((long long)a.cur - nsteps % L + L) % L);   // slot of the newest level after nsteps
#pragma unroll
        for (int p = 0; p < L; ++p) {                     // register p holds level (newest - j), j = (p - rot) mod L
            const int j = (p - rot + L) % L;
            const int slot = (cur + j) % L;
            a.Y[slot * lvl + off] = yv[p];
            a.A[slot * lvl + off] = av[p];
        }
        a.V[off] = v;
        const double ynewest = a.Y[(size_t)cur * lvl + off];
        reinterpret_cast<double *>(a.pos_next + my_i)[cc] = ynewest;
        reinterpret_cast<double *>(const_cast<Body4 *>(a.pos_cur) + my_i)[cc] = ynewest;
    }
}

// tuning builds: out[0..7] = k_lm_small's tick accounting; EPH_DEBUG_PLACEMENT=1|2 prints where the dispatcher put the workgroups
// of the last gang launch and how long each ran (stderr)
int debug_wg_cycles(long long *out) {
    for (int k = 0; k < 8; ++k) out[k] = 0;
#if EPH_EXPERIMENTS
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_ticks), sizeof(long long) * 8);
    if (e != hipSuccess) { set_last_error("hipMemcpyFromSymbol", e); return EPH_ERR_HIP; }
    const char *pl = getenv("EPH_DEBUG_PLACEMENT");
    if (!pl) return EPH_OK;
    static long long span[4][1024];
    e = hipMemcpyFromSymbol(span, HIP_SYMBOL(g_small_place), sizeof(span));
    if (e != hipSuccess) { set_last_error("hipMemcpyFromSymbol", e); return EPH_ERR_HIP; }
    static int per_cu[8][128];
    std::memset(per_cu, 0, sizeof(per_cu));
    int wgs = 0, cus = 0, worst = 0;
    long long first = 0;
    for (int b = 0; b < 1024; ++b) {
        if (span[1][b] == 0) continue;
        const unsigned hw = (unsigned)span[0][b], xcc = (unsigned)(span[1][b] - 1) & 7u;
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 3u;   // gfx9 HW_ID fields
        int &d = per_cu[xcc][(se << 5) | (sh << 4) | cu];
        if (d++ == 0) ++cus;
        worst = std::max(worst, d);
        ++wgs;
        if (first == 0 || span[2][b] < first) first = span[2][b];
    }
    fprintf(stderr, "placement: %d workgroups on %d distinct (xcc, se, cu), at most %d on one\n", wgs, cus, worst);
    for (int x = 0; x < 8; ++x) {
        double lo_d = 1e30, hi_d = 0, sum_d = 0, hi_s = 0;
        int cnt = 0;
        for (int b = 0; b < 1024; ++b) {
            if (span[1][b] == 0 || (unsigned)((span[1][b] - 1) & 7) != (unsigned)x) continue;
            const double dur = (double)(span[3][b] - span[2][b]) / 100.0, st = (double)(span[2][b] - first) / 100.0;
            lo_d = std::min(lo_d, dur); hi_d = std::max(hi_d, dur); sum_d += dur; hi_s = std::max(hi_s, st); ++cnt;
        }
        if (cnt) fprintf(stderr, "  xcc %d: %3d workgroups, duration us min %.0f mean %.0f max %.0f, latest start +%.0f us\n", x, cnt, lo_d, sum_d / cnt, hi_d, hi_s);
    }
    if (pl[0] == '2')
        for (int b = 0; b < 1024; ++b)
            if (span[1][b]) fprintf(stderr, "  wg %4d xcc %d hw %05x start +%.0f us dur %.0f us\n", b, (int)((span[1][b] - 1) & 7), (unsigned)span[0][b] & 0xfffff,
                                    (double)(span[2][b] - first) / 100.0, (double)(span[3][b] - span[2][b]) / 100.0);
    std::memset(span, 0, sizeof(span));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_small_place), span, sizeof(span));
#endif
    return EPH_OK;
}
// EPH_SMALL_FORM=4|8 forces the four- / eight-wave form (tuning, tests); default: eight for one system, lm_small_many decides by count
static int small_form() {
    static const int f = [] { const char *e = getenv("EPH_SMALL_FORM"); return e ? atoi(e) : 0; }();
    return f;
}
int lm_small(hipStream_t s, const LmArgs &a, int64_t nsteps) {
    const bool two = small_form() == 4;
    if (a.L == 12 && two) hipLaunchKernelGGL((k_lm_small<12, false, true>), dim3(1), dim3(kSmallThreads2), 0, s, a, (const LmArgs *)nullptr, (long long)nsteps);
    else if (a.L == 13 && two) hipLaunchKernelGGL((k_lm_small<13, false, true>), dim3(1), dim3(kSmallThreads2), 0, s, a, (const LmArgs *)nullptr, (long long)nsteps);
    else if (a.L == 12) hipLaunchKernelGGL((k_lm_small<12, false>), dim3(1), dim3(kSmallThreads), 0, s, a, (const LmArgs *)nullptr, (long long)nsteps);
    else if (a.L == 13) hipLaunchKernelGGL((k_lm_small<13, false>), dim3(1), dim3(kSmallThreads), 0, s, a, (const LmArgs *)nullptr, (long long)nsteps);
    else return EPH_ERR_UNSUPPORTED;
    return launched("k_lm_small");
}
int lm_small_many(hipStream_t s, const LmArgs *argv_dev, int count, int L, int64_t nsteps) {
    const LmArgs none{};
    // more systems than CUs: the four-wave form, two workgroups per CU (1024 systems 2.63 against 3.51 us per step of the gang)
    static const int cus = [] { int dev = 0, c = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev); return c; }();
    const bool two = small_form() ? small_form() == 4 : count > cus;
    if (L == 12 && two) hipLaunchKernelGGL((k_lm_small<12, true, true>), dim3((unsigned)count), dim3(kSmallThreads2), 0, s, none, argv_dev, (long long)nsteps);
    else if (L == 13 && two) hipLaunchKernelGGL((k_lm_small<13, true, true>), dim3((unsigned)count), dim3(kSmallThreads2), 0, s, none, argv_dev, (long long)nsteps);
    else if (L == 12) hipLaunchKernelGGL((k_lm_small<12, true>), dim3((unsigned)count), dim3(kSmallThreads), 0, s, none, argv_dev, (long long)nsteps);
    else if (L == 13) hipLaunchKernelGGL((k_lm_small<13, true>), dim3((unsigned)count), dim3(kSmallThreads), 0, s, none, argv_dev, (long long)nsteps);
    else return EPH_ERR_UNSUPPORTED;
    return launched("k_lm_small (gang)");
}

}  // namespace EPH_PV_NS
}  // namespace eph
