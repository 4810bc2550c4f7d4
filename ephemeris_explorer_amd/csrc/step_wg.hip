// step_wg.hip -- the workgroup-specialised force (wg_force) and the kernels built on it: k_accel_wg and k_lm_step_wg, the
// dominant kernel of the massive-body path (one launch per integrator step for N > 512 targets; dispatch.cpp chooses).
// Compiled once per evaluation order of the point-mass term (pair_ns.h), -ffp-contract=off (parity is bit for bit).
//
// Measured on MI355X (scripts/ubench/lat.hip): a dependent v_add_f64 issues every 8.4 cycles from one wave, v_rsq_f64 /
// v_rcp_f64 cost ~17 issue cycles each, other f64 ops ~4.1-5. In the one-wave-per-block form (step_wave.hip) every wave pays
// the 64 ordered adds and 32 ds_read_b128 of a tile's ordered sums for only 3 * BPW useful lanes. Here ONE wave of a workgroup
// (the chain wave) carries the ordered sums of all the workgroup's bodies -- 3 * WB of its 64 lanes -- while pair waves (lane =
// source) produce the contribution tiles through LDS. The shipped arrangement ("layout 5", round 2; history and the six retired
// layouts: profiles/r02_step_kernel_evidence.md, profiles/r03_step_kernel_evidence.md; the retired code: step_wg_retired.inc,
// compiled only with -DEPH_EXPERIMENTS):
//   * 16 bodies per workgroup, TWELVE waves, three per SIMD (waves go to the four SIMDs round-robin: wave k -> SIMD k % 4),
//     <= 168 VGPRs each: SIMDs 1-3 carry pair waves of 2 / 2 / 1 bodies, SIMD 0 a one-body pair wave, the chain wave (at raised
//     issue priority) and the tail wave;
//   * ONE s_barrier per 128 sources (two 64-source tiles; the first two tiles go singly so the chain wave starts early), six
//     25 KB LDS tile buffers: pair waves run one 128-source "big tile" ahead of the chain wave;
//   * the tail wave (k_lm_step_wg) loads the 2 L history values while the others work, receives the new acceleration through
//     LDS and does Cowell's velocity, the solout sample and the predictor;
//   * 8- and 4-body workgroups for target counts that would leave CUs without a 16-body workgroup (<= 2048 / <= 1024 targets): the
//     8-body one with SIX waves (four pair waves of two bodies, chain, tail; round 5 -- the twelve-wave form with one body per pair
//     wave stays selectable, EPH_WG_BODIES=8), the 4-body form with one body per pair wave and TWO chain waves on alternate tiles.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "pair_ns.h"

#ifndef EPH_EXPERIMENTS
#define EPH_EXPERIMENTS 0
#endif
// tuning builds only (scripts/build_exp.sh NAME -DEPH_EXPERIMENTS=1 -DEPH_WG_SIDE=k): 1 = the chain wave skips its sums (pair side
// alone), 2 = the pair waves skip their tiles (chain side alone); results are then meaningless. Compile-time on purpose: the same
// two tests as run-time flags cost the default path 3.6 us per step.
#if !EPH_EXPERIMENTS || !defined(EPH_WG_SIDE)
#undef EPH_WG_SIDE
#define EPH_WG_SIDE 0
#endif
// ablations for the same tuning builds (results meaningless): 1 = pair waves keep their contributions in registers (no ds_write),
// 2 = no barrier inside the tile loops -- what the floor model of profiles/r04_step_kernel_evidence.md is calibrated with
#if !EPH_EXPERIMENTS || !defined(EPH_WG_ABLATE)
#undef EPH_WG_ABLATE
#define EPH_WG_ABLATE 0
#endif
#define WG_LOOP_BARRIER() do { if constexpr (!(EPH_WG_ABLATE & 2)) __syncthreads(); } while (0)
// PREPARED, NOT MEASURED (-DEPH_EXPERIMENTS=1 -DEPH_WG_DIAG_PATCH=1): the interval that holds the workgroup's own tile takes the
// IEEE form for BOTH of its tiles on every pair wave (n2 = 0 on the self lanes fails the range test of the whole wave): about 1.8 x
// the instructions for one interval of 33, up to 0.5 us per step at N = 4096 if the pair side is the longer one there. With the
// switch the self lanes get in-range operands instead (n2 = 1; their contribution is read and discarded by chain_masked) and the
// interval stays on the seeded sequences. Round 2 measured the same idea on the barrier-per-tile kernel (40.1 vs 40.0 us, no
// gain): to be re-measured on this one (scripts/experiment.sh TAG -l product -l diagpatch sizes 2048 4096, and
// EPH_AMD_LIBRARY=...exp_diagpatch.so pytest tests/test_gpu_parity.py tests/test_gpu_horizon.py for the bits).
#if EPH_EXPERIMENTS && defined(EPH_WG_DIAG_PATCH)
#define EPH_WG_DIAG_PATCH_ON 1
#else
#define EPH_WG_DIAG_PATCH_ON 0
#endif

// The SIX-WAVE 8-body workgroup (round 5; `DUO` in the templates below, wb code 9 in the launcher): four pair waves x two bodies, a
// chain wave, a tail wave, 76 KB of LDS. Built to put TWO workgroups on a CU at N = 4096 so that one's barrier drain is covered by
// the other's arithmetic -- that lost badly (56.9 against 36.1 us: two chain waves and sixteen bodies of pair work share the four
// SIMDs with no say in the placement) -- but ONE such workgroup per CU beats the twelve-wave 8-body form (one body per pair wave: a
// bare dependent chain per wave) wherever that one was used: 1280 / 1536 / 1792 / 2048 bodies 13.2 / 14.8 / 16.3 / 17.7 against
// 14.0 / 15.6 / 17.1 / 18.6 us per step, bit-identical. dispatch.cpp takes it for 1024 < targets <= 2048.
// profiles/r05_step_kernel_evidence.md.

namespace eph {
namespace EPH_PV_NS {

constexpr int kWgBodies = 16;
constexpr int kDuoThreads = 64 * 6;                  // the duo form: waves 0-3 pair (two bodies each), 4 chain, 5 tail
constexpr int kDuoTailWave = 5;
constexpr int kWgChainWave = 4;                      // the chain wave (wave 4: lands on SIMD 0)
constexpr int kWgTailWave = 8;                       // the step kernel's tail wave (SIMD 0 too)
constexpr int kWgThreads = 64 * 12;
constexpr int kWgTileBufs = 6;                       // LDS tile buffers of the barrier-per-128-sources schedule
constexpr int kWgSplitBufs = 3;                      // ... of the barrier-per-64 schedule (4-body workgroups, two chain waves)

// pair wave: NB bodies (local indices b0..) against the 64 sources in pj -> rows of `tile`
template <int NB>
__device__ __forceinline__ void wg_pair_tile(const double (&xi)[NB], const double (&yi)[NB], const double (&zi)[NB],
                                             const Body4 &pj, bool ieee, double *tile, int b0, int lane) {
    // (the workgroup's own tile goes to the IEEE form as a whole -- n2 = 0 on the self lanes; giving those lanes a
    // harmless in-range operand instead, since the chain wave never reads them, was measured: no gain, 40.1 vs 40.0 us)
    PairPre pre[NB];
    unsigned worst = ieee ? kRangeSpan : mu_key(pj.mu), low = ~0u;   // max of the range keys: one add + one max per body
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        pre[b] = pair_pre(xi[b], yi[b], zi[b], pj);
        worst = max(worst, range_key(pre[b].n2));
        low = min(low, pre[b].lo);
    }
    worst = max(worst, low_key(low));
    double c[3 * NB];
    if (__builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) pair_finish<true>(pre[b], pj.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
    } else {   // the tile holding the workgroup's own bodies (n2 = 0 on the self lane) or an operand outside the guarded ranges
#pragma unroll
        for (int b = 0; b < NB; ++b) pair_finish<false>(pre[b], pj.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
    }
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) {
        if constexpr (EPH_WG_ABLATE & 1) asm volatile("" ::"v"(c[q]));
        else tile[(3 * b0 + q) * kRow + lane] = c[q];
    }
}

// Barrier-per-64-sources schedule (every wave executes tiles + 1 barriers): B_0 after tiles 0 and 1 are in LDS; iteration t:
// pair waves produce tile t + 2 into buffer (t + 2) % 3 while the chain waves sum tile t from buffer t % 3; barrier.
template <int NB, typename PosPtr>
__device__ __forceinline__ void wg_pair_wave(PosPtr pos, int n, int i0, int b0, double *C, int lane, int tiles, int tdiag, int wbuf) {
    double xi[NB], yi[NB], zi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ii = min(i0 + b0 + b, n - 1);
        xi[b] = pos[ii].x;
        yi[b] = pos[ii].y;
        zi[b] = pos[ii].z;
    }
    auto load_src = [&](int t) -> Body4 {
        const int j = min(t, tiles - 1) * kTile + lane;
        return pos[j < n ? j : n - 1];
    };
    Body4 pj = load_src(0);
    Body4 pjn = load_src(1);
    wg_pair_tile<NB>(xi, yi, zi, pj, tdiag == 0, C, b0, lane);
    pj = pjn;
    pjn = load_src(2);
    if (tiles > 1) wg_pair_tile<NB>(xi, yi, zi, pj, tdiag == 1, C + wbuf, b0, lane);
    __syncthreads();
    for (int t = 0; t < tiles; ++t) {
        if (t + 2 < tiles) {
            pj = pjn;
            pjn = load_src(t + 3);
            wg_pair_tile<NB>(xi, yi, zi, pj, tdiag == t + 2, C + ((t + 2) % kWgSplitBufs) * wbuf, b0, lane);
        }
        __syncthreads();
    }
}

// two 64-source tiles at once: 2 x NB independent interactions, written stage by stage (pair_finish_staged, pair_term.h)
template <int NB>
__device__ __forceinline__ void wg_pair_tile2(const double (&xi)[NB], const double (&yi)[NB], const double (&zi)[NB],
                                              const Body4 &pa, const Body4 &pb, bool ieee, double *tile_a, double *tile_b,
                                              int b0, int lane, int self_a = -64, int self_b = -64) {
    PairPre pre[2 * NB];
    unsigned worst = ieee ? kRangeSpan : max(mu_key(pa.mu), mu_key(pb.mu)), low = ~0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        pre[b] = pair_pre(xi[b], yi[b], zi[b], pa);
        pre[NB + b] = pair_pre(xi[b], yi[b], zi[b], pb);
        if constexpr (EPH_WG_DIAG_PATCH_ON) {          // self_x: the lane that holds body b0 as a source of tile x (far off: none)
            if (lane == self_a + b) { pre[b].n2 = 1.0; pre[b].lo = 0x3ff00000u; }
            if (lane == self_b + b) { pre[NB + b].n2 = 1.0; pre[NB + b].lo = 0x3ff00000u; }
        }
        worst = max(worst, max(range_key(pre[b].n2), range_key(pre[NB + b].n2)));
        low = min(low, min(pre[b].lo, pre[NB + b].lo));
    }
    worst = max(worst, low_key(low));
    double c[6 * NB];
    if (__builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0) {
        // (round 2 wrote the stages out WITHOUT pinning the order and the scheduler put them back: 37.38 vs 37.04 us)
        if constexpr (kPairStaged) {
            double mus[2 * NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { mus[b] = pa.mu; mus[NB + b] = pb.mu; }
            pair_finish_staged<2 * NB>(pre, mus, c);
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                pair_finish<true>(pre[b], pa.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
                pair_finish<true>(pre[NB + b], pb.mu, c[3 * (NB + b)], c[3 * (NB + b) + 1], c[3 * (NB + b) + 2]);
            }
        }
        // (writing each interaction's three values as soon as they exist, instead of the burst below, was measured too:
        // 36.9 vs 37.0 us, although SQ_LDS_DATA_FIFO_FULL is raised 13 % of the time -- profiles/r02_pmc2.json)
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            pair_finish<false>(pre[b], pa.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
            pair_finish<false>(pre[NB + b], pb.mu, c[3 * (NB + b)], c[3 * (NB + b) + 1], c[3 * (NB + b) + 2]);
        }
    }
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) {
        if constexpr (EPH_WG_ABLATE & 1) {
            asm volatile("" ::"v"(c[q]), "v"(c[3 * NB + q]));
        } else {
            tile_a[(3 * b0 + q) * kRow + lane] = c[q];
            tile_b[(3 * b0 + q) * kRow + lane] = c[3 * NB + q];
        }
    }
}

// Layout 3 barrier schedule. The 64-source tiles are grouped into "big" tiles, one barrier each: big tiles 0 and 1 are
// single tiles (so the chain wave starts after two tiles, not four: its wait for the first barrier was 2.7 us of a
// 42 us launch), every later one is two tiles. Every wave executes TB + 1 barriers: B_0 after big tiles 0 and 1 are
// in LDS; iteration K: pair waves produce big tile K + 2 while the chain wave sums big tile K (and prefetches the head
// of K + 1, complete since the previous barrier); barrier. Tile t lives in LDS buffer t % 6; the tiles alive at any time
// span at most six consecutive indices.
// (Single tiles at the END as well -- the pair waves run two big tiles ahead, so the chain wave sums the last two alone --
// were measured: 37.6-37.7 vs 37.0 us at N = 4096, 18.6 vs 18.3 at 2048, 12.05 vs 11.9 at 1024. Not kept.)
__device__ __forceinline__ int big_start(int K) { return K < 2 ? K : 2 * K - 2; }
__device__ __forceinline__ int big_count(int tiles) { return tiles <= 2 ? tiles : 2 + (tiles - 2 + 1) / 2; }
// (Round 4 measured what the loop's barriers cost -- pair side alone 34.4 us, 33.4 without its LDS writes, 28.6 without the
// barriers either -- and replaced them by LDS counters: pair waves free-running up to two big tiles ahead, each adding 1 to a counter
// behind its ds_writes, the chain wave polling that counter and publishing how far it has consumed. Bit-identical and SLOWER: 37.9
// against 36.3 us at N = 4096, 19.6 against 18.6 at 2048 (39.1 with memory-model fences, which also wait for the prefetched global
// loads). The barrier keeps the write bursts of ten waves and the chain wave's reads in separate phases of the one LDS pipe.
// profiles/r04_step_kernel_evidence.md section 4; the patch: scripts/experiments/wg_counter_sync.patch.)
template <int NB, typename PosPtr>
__device__ __forceinline__ void wg_pair_wave_big(PosPtr pos, int n, int i0, int b0, double *C, int lane, int tiles, int tdiag, int wbuf) {
    double xi[NB], yi[NB], zi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ii = min(i0 + b0 + b, n - 1);
        xi[b] = pos[ii].x;
        yi[b] = pos[ii].y;
        zi[b] = pos[ii].z;
    }
    // (Measured and NOT kept, rounds 3 and 4, each bit-identical: source rows addressed as uniform tile base + loop-invariant lane
    // offset -- ten VALU fewer per interval -- 36.7 against 36.2 us; two intervals per trip with the prefetched register sets
    // trading places instead of being copied 38.7; own bodies' positions through scalar loads 36.4; the first source tiles requested
    // before the own positions 36.3; nontemporal stores of the tail wave 36.1 (nothing): profiles/r04_step_kernel_evidence.md.)
    auto load_src = [&](int t) -> Body4 {
        const int j = min(t, tiles - 1) * kTile + lane;
        return pos[j < n ? j : n - 1];
    };
    auto produce = [&](int K, const Body4 &pa, const Body4 &pb) {          // big tile K
        const int t = big_start(K);
        if (t >= tiles || EPH_WG_SIDE == 2) return;
        double *ta = C + (t % kWgTileBufs) * wbuf, *tb = C + ((t + 1) % kWgTileBufs) * wbuf;
        if constexpr (EPH_WG_DIAG_PATCH_ON) {
            if (K >= 2 && t + 1 < tiles) {
                const int own = i0 + b0;                // (bodies beyond n are clamped copies of body n - 1: never diagonal lanes)
                wg_pair_tile2<NB>(xi, yi, zi, pa, pb, false, ta, tb, b0, lane, tdiag == t ? own - t * kTile : -64,
                                  tdiag == t + 1 ? own - (t + 1) * kTile : -64);
                return;
            }
        }
        if (K >= 2 && t + 1 < tiles) wg_pair_tile2<NB>(xi, yi, zi, pa, pb, tdiag == t || tdiag == t + 1, ta, tb, b0, lane);
        else wg_pair_tile<NB>(xi, yi, zi, pa, tdiag == t, ta, b0, lane);
    };
    const int TB = big_count(tiles);
    Body4 pa = load_src(0), pb = load_src(1), na = load_src(2), nb = load_src(3);
    produce(0, pa, pa);
    produce(1, pb, pb);
    __syncthreads();
    for (int K = 0; K < TB; ++K) {
        pa = na; pb = nb;
        na = load_src(big_start(K + 3)); nb = load_src(big_start(K + 3) + 1);
        produce(K + 2, pa, pb);
        WG_LOOP_BARRIER();
    }
}
// a wave with no tile work: TB + 1 barriers like everybody
__device__ __forceinline__ void wg_idle_wave(int tiles) {
    __syncthreads();
    for (int T = 0; T < big_count(tiles); ++T) WG_LOOP_BARRIER();
}

// chain over a full tile whose first two chunks are already in q[0], q[1]; leaves the first two chunks of the
// NEXT tile (row_next, complete since the previous barrier) in q[0], q[1]
__device__ __forceinline__ double chain_full_pf(const double *row, const double *row_next, double2 (&q)[4][8],
                                                double acc) {
    load_chunk(row, 2, q[2]);
    acc = add_chunk(q[0], acc);
    load_chunk(row, 3, q[3]);
    acc = add_chunk(q[1], acc);
    load_chunk(row_next, 0, q[0]);
    acc = add_chunk(q[2], acc);
    load_chunk(row_next, 1, q[1]);
    return add_chunk(q[3], acc);
}

// The 4-body workgroups (512 < N <= 1024 targets, where the step IS the chain wave's time) with TWO chain
// waves taking ALTERNATE tiles: while one adds the 64 sources of tile t out of its registers, the other reads tile t + 1
// into its own (a wave's ds_read_b128 and its dependent adds do not overlap; two waves' do), and the 3 * WB partial sums
// change hands through LDS at every tile's barrier. Same number of LDS reads as one chain wave (the split by LANE doubled
// them). One barrier per 64-source tile (three tile buffers), pair waves two tiles ahead.
// MEASURED (bit-identical; us per step, split | default): N = 640 9.29 | 9.83, 1024 11.74 | 12.01, 1536 16.10 | 15.37,
// 2048 19.58 | 18.34 -- a gain only where four bodies per workgroup leave the pair side idle anyway; with eight one-body
// pair waves a barrier per tile makes the pair side (one interaction per wave and tile: a bare dependent chain) the
// slower one. Producing the tiles in pairs in every other interval to get two interleaved interactions back: 10.4 / 13.0 /
// 18.7 / 23.0, worse still (the reader waits out the double intervals). Default: the 4-body workgroups only.
// (profiles/r03_step_kernel_evidence.md section 1b; -DEPH_WG_TILE_SPLIT=1 applies it to the 8-body workgroups too, 0 switches it off)
#ifndef EPH_WG_TILE_SPLIT
#define EPH_WG_TILE_SPLIT 2
#endif
constexpr bool wg_tile_split(int wb) { return EPH_WG_TILE_SPLIT == 1 ? wb < 16 : (EPH_WG_TILE_SPLIT == 2 && wb == 4); }
constexpr int wg_split_wave_b(int wb) { return wb == 8 ? 11 : 6; }   // an idle wave of another SIMD than the chain wave's
// body of the one-body pair wave `wave` in the 8- / 4-body workgroups (-1: not a pair wave)
template <int WB>
__device__ __forceinline__ int wg_small_body(int wave) {
    if constexpr (WB == 8) {
        switch (wave) { case 1: return 0; case 2: return 1; case 3: return 2; case 5: return 3; case 6: return 4; case 7: return 5;
                        case 9: return 6; case 10: return 7; default: return -1; }
    } else {
        switch (wave) { case 1: return 0; case 2: return 1; case 3: return 2; case 5: return 3; default: return -1; }
    }
}
template <int WB, typename PosPtr>
__device__ __forceinline__ double wg_force_split(PosPtr pos, int n, int i0, double init, double *C, int tid) {
    constexpr int kRows = 3 * WB, kBuf = kRows * kRow;
    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = (n + kTile - 1) / kTile;
    const int tdiag = i0 / kTile;
    double *H = C + kWgSplitBufs * kBuf;                // hand-over: H[lane] = running sum, H[64 + lane] = closed lower sum
    const int body = wg_small_body<WB>(wave);
    if (body >= 0) { wg_pair_wave<1>(pos, n, i0, body, C, lane, tiles, tdiag, kBuf); return 0.0; }
    const bool isA = wave == kWgChainWave, isB = wave == wg_split_wave_b(WB);
    if (!isA && !isB) { for (int t = 0; t <= tiles; ++t) __syncthreads(); return 0.0; }
    const int ch = lane < kRows ? lane : kRows - 1;
    const double *row = C + ch * kRow;
    const int gself = (i0 % kTile) / WB;
    const int li = (i0 % kTile) + ch / 3;
    double acc = init, accL = 0.0;
    double2 q[4][8];
    auto plain = [&](int t) { return t < tiles && t != tdiag && min(kTile, n - t * kTile) == kTile; };
    auto preload = [&](int t) {
        const double *r = row + (t % kWgSplitBufs) * kBuf;
        load_chunk(r, 0, q[0]); load_chunk(r, 1, q[1]); load_chunk(r, 2, q[2]); load_chunk(r, 3, q[3]);
    };
    __syncthreads();                                    // B_0: tiles 0 and 1 ready
    if (isA && plain(0)) preload(0);
    for (int t = 0; t < tiles; ++t) {
        const bool mine = ((t & 1) != 0) == isB;
        if (mine) {
            if (t > 0) { acc = H[lane]; accL = H[64 + lane]; }
            if (plain(t)) {
                acc = add_chunk(q[0], acc); acc = add_chunk(q[1], acc); acc = add_chunk(q[2], acc); acc = add_chunk(q[3], acc);
            } else {
                chain_masked<WB>(row + (t % kWgSplitBufs) * kBuf, min(kTile, n - t * kTile), t == tdiag ? gself : -1, li, acc, accL);
            }
            H[lane] = acc;
            H[64 + lane] = accL;
        } else if (plain(t + 1)) {
            preload(t + 1);                             // complete since the previous barrier
        }
        __syncthreads();
    }
    return isA ? H[64 + lane] + H[lane] : 0.0;
}

// Returns on chain-wave lane ch < 3 * WB: component ch % 3 of body i0 + ch / 3 (init + the reference-order sum over the other
// bodies). Every thread of the workgroup must call it.
template <int WB = kWgBodies, bool DUO = false, typename PosPtr>
__device__ __forceinline__ double wg_force(PosPtr pos, int n, int i0, double init, double *C, int tid) {
    constexpr int kRows = 3 * WB, kBuf = kRows * kRow;   // chains of the chain wave, doubles per LDS tile buffer
    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = (n + kTile - 1) / kTile;
    const int tdiag = i0 / kTile;
    if constexpr (DUO) {
        static_assert(WB == 8, "the duo form is the 8-body workgroup with two bodies per pair wave");
        const int w = __builtin_amdgcn_readfirstlane(wave);
        if (w < 4) { wg_pair_wave_big<2>(pos, n, i0, 2 * w, C, lane, tiles, tdiag, kBuf); return 0.0; }
        if (w == kDuoTailWave) { wg_idle_wave(tiles); return 0.0; }
    } else
    if constexpr (wg_tile_split(WB)) return wg_force_split<WB>(pos, n, i0, init, C, tid);
    else if constexpr (WB != kWgBodies) {
        // 8 / 4 bodies: one body per pair wave, SIMD 0 left to the chain wave (its cost per tile does not depend on how many of its
        // lanes carry a chain, so with one workgroup per CU the step takes the chain wave's time)
        const int body = wg_small_body<WB>(wave);
        if (body >= 0) { wg_pair_wave_big<1>(pos, n, i0, body, C, lane, tiles, tdiag, kBuf); return 0.0; }
        if (wave != kWgChainWave) { wg_idle_wave(tiles); return 0.0; }
    } else {
        // Roles of the twelve waves (wave k runs on SIMD k % 4): pair waves of 2 / 2 / 1 bodies on SIMDs 1-3, a one-body pair wave
        // on SIMD 0 beside the chain wave (4) and the tail wave (8). ONE copy of the pair loop per body count, the role's first
        // body a run-time scalar: inlining the loop once per role (ten copies, round 2-3) measured 0.15 us slower at N = 4096
        // (36.20 against 36.04 us, profiles/r04_step_kernel_evidence.md) and is 40 KB more code per evaluation order.
        const int w = __builtin_amdgcn_readfirstlane(wave);
        const int nb = (w == kWgChainWave || w == kWgTailWave) ? 0 : ((w == 0 || w >= 9) ? 1 : 2);
        const int b0 = w == 0 ? 0 : w == 1 ? 1 : w == 5 ? 3 : w == 9 ? 5 : w == 2 ? 6 : w == 6 ? 8 : w == 10 ? 10 : w == 3 ? 11 : w == 7 ? 13 : 15;
        if (nb == 2) { wg_pair_wave_big<2>(pos, n, i0, b0, C, lane, tiles, tdiag, kBuf); return 0.0; }
        if (nb == 1) { wg_pair_wave_big<1>(pos, n, i0, b0, C, lane, tiles, tdiag, kBuf); return 0.0; }
        if (w == kWgTailWave) { wg_idle_wave(tiles); return 0.0; }   // (k_lm_step_wg gives this wave the integrator's work instead)
    }
    // chain wave. Its dependent adds issue ahead of the one-body pair wave of its SIMD (s_setprio; the same library with and
    // without, alternating on one box: 36.3 against 36.8 us per step at N = 4096 on two boxes of the pool, 36.2 either way on a
    // third; nothing at the chain-bound sizes)
    const int ch = lane < kRows ? lane : kRows - 1;
    const double *row = C + ch * kRow;
    const int gself = (i0 % kTile) / WB;
    const int li = (i0 % kTile) + ch / 3;
    double acc = init, accL = 0.0;
    double2 q[4][8];
    const int TB = big_count(tiles);
    if constexpr (WB == kWgBodies || DUO) __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                  // B_0: tiles 0 and 1 ready
    load_chunk(row, 0, q[0]);
    load_chunk(row, 1, q[1]);
    for (int T = 0; T < TB; ++T) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int t = big_start(T) + hf;
            if (t >= tiles || (T < 2 && hf)) break;
            const double *r = row + (t % kWgTileBufs) * kBuf;
            const double *rn = row + ((t + 1) % kWgTileBufs) * kBuf;   // complete since the previous barrier
            const int cnt = min(kTile, n - t * kTile);
            if constexpr (EPH_WG_SIDE == 1) {
            } else if (t != tdiag && cnt == kTile) {
                acc = chain_full_pf(r, rn, q, acc);
            } else {
                chain_masked<WB>(r, cnt, t == tdiag ? gself : -1, li, acc, accL);
                load_chunk(rn, 0, q[0]);
                load_chunk(rn, 1, q[1]);
            }
        }
        WG_LOOP_BARRIER();                            // big tile T consumed, big tile T + 2 ready
    }
    return accL + acc;
}
constexpr int wg_lds_doubles(int wb, bool duo = false) { return (wg_tile_split(wb) && !duo ? kWgSplitBufs * 3 * wb * kRow + 128 : kWgTileBufs * 3 * wb * kRow); }

template <int WB = kWgBodies>
__global__ void __launch_bounds__(kWgThreads) k_accel_wg(int n, int npad, const Body4 *__restrict__ pos,
                                                         const double *__restrict__ acc_init, double *__restrict__ acc_out,
                                                         int lo, int hi, KickDrift kd) {
    __shared__ __attribute__((aligned(16))) double C[wg_lds_doubles(WB)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int i0 = lo + blockIdx.x * WB;
    const int my_i = i0 + lane / 3, cc = lane % 3;
    const bool owner = (tid >> 6) == kWgChainWave && lane < 3 * WB && my_i < hi;
    const double init = (owner && acc_init) ? acc_init[cc * npad + my_i] : 0.0;
    const double a = wg_force<WB>(pos, n, i0, init, C, tid);
    if (owner) {
        acc_out[cc * npad + my_i] = a;
        if (kd.v) kick_drift_one(kd, (size_t)cc * npad + my_i, my_i, cc, a);
    }
}

// ONE launch per integrator step: slot `cur` of the ring holds the already predicted positions of the level being completed;
// this launch evaluates its acceleration (reference-order all-pairs sum), recovers its velocity (Cowell), stores the solout sample
// if one is due, predicts the positions of the NEXT level and publishes them (ring + packed ping-pong buffer) -- the kernel
// boundary is the only grid-wide synchronisation a step needs.
// (amdgpu_waves_per_eu(3, 3): for the twelve-wave forms it restates what __launch_bounds__(768) already implies -- three waves per
// SIMD, 168 VGPRs -- and changes nothing in their code; for the six-wave DUO form it is a choice (its 1.5 waves per SIMD would
// allow 256 VGPRs): that form was measured WITH it (17.7 against 18.6 us at 2048 bodies, round 5) and ships as measured.)
template <int L, int WB = kWgBodies, bool DUO = false>
__global__ void __launch_bounds__(DUO ? kDuoThreads : kWgThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_lm_step_wg(const LmArgs a) {
    constexpr int kRows = 3 * WB;
    __shared__ __attribute__((aligned(16))) double C[wg_lds_doubles(WB, DUO)];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool chain_wave = (tid >> 6) == kWgChainWave;
    // the wave that does the integrator's work around the force; wave-uniform by construction, and told so (a scalar
    // branch keeps the history registers out of the other roles' live ranges)
    const bool tail_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == (DUO ? kDuoTailWave : kWgTailWave);
    const int i0 = a.lo + blockIdx.x * WB;
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = tail_wave && lane < kRows && my_i < a.hi;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + (owner ? my_i : 0);
    // the tail wave's whole life is this branch, so its history registers are live across this code only (through
    // wg_force's role switch the allocator would keep them alive in every role and spill)
    if (tail_wave) {
        double yv[L], av[L];   // yv[j] / av[j]: level (new - j); av[0] is filled after the force
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int slot = (a.cur + j) % L;
            yv[j] = a.Y[slot * lvl + off];
            av[j] = j > 0 ? a.A[slot * lvl + off] : 0.0;
        }
        // (Forming everything that does not need the new acceleration here, ahead of the barriers, was built and measured:
        // 37.55 vs 37.2 us per step at N = 4096 -- the early arithmetic takes issue slots from the pair wave and the chain wave
        // of this SIMD when they are the critical path, and the tail's work was not on it.)
        const int tiles = (a.n + kTile - 1) / kTile;
        if constexpr (wg_tile_split(WB) && !DUO) {
            for (int t = 0; t <= tiles; ++t) __syncthreads();                             // one barrier per tile there
        } else {
            wg_idle_wave(tiles);
        }
        __syncthreads();                              // the chain wave's result is in LDS
        if (owner) {
            const double anew = C[lane];
            a.A[(size_t)a.cur * lvl + off] = anew;
            {
                double prev[L];
#pragma unroll
                for (int j = 0; j < L - 1; ++j) prev[j] = av[j + 1];
                prev[L - 1] = 0.0;
                a.V[off] = lm_cowell<L>(anew, prev, yv[0], yv[1], a.cw, a.h, a.hc);
            }
            maybe_sample(a.samp, my_i, cc, a.step, yv[0]);
            if (a.do_predict) {
                av[0] = anew;
                const double ynext = lm_predict<L>(yv, av, a.wa, a.wb, a.hh);
                const int nslot = (a.cur + L - 1) % L;
                a.Y[(size_t)nslot * lvl + off] = ynext;
                reinterpret_cast<double *>(a.pos_next + my_i)[cc] = ynext;
            }
        }
    } else {
        const double anew = wg_force<WB, DUO>(a.pos_cur, a.n, i0, 0.0, C, tid);
        if (chain_wave && lane < kRows) C[lane] = anew;    // every tile buffer is dead after the loop's last barrier
        __syncthreads();
    }
}

// ---- launchers (wb = bodies per workgroup: 16, 8 or 4; dispatch.cpp chooses) --------------------------------------------------
int accel_wg(hipStream_t s, int wb, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out, int lo, int hi,
             const KickDrift &kd) {
    const dim3 grid((hi - lo + wb - 1) / wb), block(kWgThreads);
    if (wb == 8) hipLaunchKernelGGL(k_accel_wg<8>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd);
    else if (wb == 4) hipLaunchKernelGGL(k_accel_wg<4>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd);
    else hipLaunchKernelGGL(k_accel_wg<16>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd);
    return launched("k_accel_wg");
}
int lm_step_wg(hipStream_t s, int wb, const LmArgs &a) {
    if (wb == 9) {                                          // the six-wave 8-body form (dispatch.cpp: 1024 < targets <= 2048; EPH_WG_BODIES=9)
        const dim3 grid((a.hi - a.lo + 7) / 8), block(kDuoThreads);
        if (a.L == 12) hipLaunchKernelGGL((k_lm_step_wg<12, 8, true>), grid, block, 0, s, a);
        else if (a.L == 13) hipLaunchKernelGGL((k_lm_step_wg<13, 8, true>), grid, block, 0, s, a);
        else return EPH_ERR_UNSUPPORTED;
        return launched("k_lm_step_wg (six waves)");
    }
    const dim3 grid((a.hi - a.lo + wb - 1) / wb), block(kWgThreads);
    if (a.L == 12 && wb == 8) hipLaunchKernelGGL((k_lm_step_wg<12, 8>), grid, block, 0, s, a);
    else if (a.L == 13 && wb == 8) hipLaunchKernelGGL((k_lm_step_wg<13, 8>), grid, block, 0, s, a);
    else if (a.L == 12 && wb == 4) hipLaunchKernelGGL((k_lm_step_wg<12, 4>), grid, block, 0, s, a);
    else if (a.L == 13 && wb == 4) hipLaunchKernelGGL((k_lm_step_wg<13, 4>), grid, block, 0, s, a);
    else if (a.L == 12) hipLaunchKernelGGL((k_lm_step_wg<12, 16>), grid, block, 0, s, a);
    else if (a.L == 13) hipLaunchKernelGGL((k_lm_step_wg<13, 16>), grid, block, 0, s, a);
    else return EPH_ERR_UNSUPPORTED;
    return launched("k_lm_step_wg");
}

}  // namespace EPH_PV_NS
}  // namespace eph
