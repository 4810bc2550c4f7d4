// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64). No MFMA: the path is f64
// VALU (sqrt + divide) bound, see DESIGN.md. Compiled with -ffp-contract=off: the reference (Rust) never
// fuses a*b+c, and parity is defined bit-for-bit, so every sum below is written in the reference's
// operation order and must stay un-fused.
//
// Reference citations are relative to the reference repository root.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <type_traits>
#include <utility>

#include "eph_internal.h"
#include "device_math.h"

namespace eph {

// ------------------------------------------------------------------------------------------------------
// Pair interaction: acceleration on a body at (xi,yi,zi) from source body pj = {x,y,z,mu}.
// `particular::gravity::newtonian` acceleration_paired / acceleration_at with softening 0
// (call sites ephemeris/src/propagators/nbody.rs:29, ephemeris_explorer/src/dynamics/spacecraft.rs:73).
// For the pair (k, i), k < i, the reference computes -(p_i - p_k) * (mu_k * inv); (p_k - p_i) * (mu_k * inv)
// is the same f64 (negation is exact), so one directed formula serves both triangles.
//
//   d = p_j - p_i ; n2 = d.x*d.x + d.y*d.y + d.z*d.z ; inv = 1 / (n2 * sqrt(n2)) ; a = d * (mu_j * inv)
//
// sqrt and the reciprocal must be the IEEE correctly rounded results (the CPU's sqrtsd / divsd). The compiler's
// f64 expansions are: v_rsq_f64 / v_rcp_f64 seed + fma refinement, wrapped in range scaling (v_ldexp,
// v_div_scale, v_div_fmas, v_div_fixup) that only acts for operands near the ends of the exponent range.
// `*_inrange` below are exactly those refinement sequences without the scaling wrappers: bit-identical whenever
// the scaling would have been a no-op, which in_range() guarantees (n2 in [2^-300, 2^300], so
// n2*sqrt(n2) in [2^-450, 2^450]). Out-of-range tiles take the full IEEE form. 9 fewer VALU ops per pair.
// ------------------------------------------------------------------------------------------------------
struct PairPre { double dx, dy, dz, n2; };

__device__ __forceinline__ PairPre pair_pre(double xi, double yi, double zi, const Body4 &pj) {
    PairPre p;
    p.dx = pj.x - xi;
    p.dy = pj.y - yi;
    p.dz = pj.z - zi;
    p.n2 = p.dx * p.dx + p.dy * p.dy + p.dz * p.dz;   // glam DVec3::length_squared, left to right
    return p;
}
template <bool FAST>
__device__ __forceinline__ void pair_finish(const PairPre &p, double mu, double &cx, double &cy, double &cz) {
    // IEEE correctly rounded f64 sqrt and divide in the build's evaluation order (device_math.h)
    if constexpr (kPairVariant <= 3) {
        // (written out, not through pair_den / pair_apply: the same operations, but that route costs the workgroup kernel
        // 0.3 us per step at N = 4096 -- 37.2 vs 36.9 -- through a different instruction order; gpurun_out r03 A/B)
        double inv;
        if (FAST) inv = inv_r3_inrange(p.n2);
        else inv = inv_r3_ieee(p.n2);
        const double s = mu * inv;
        cx = p.dx * s;
        cy = p.dy * s;
        cz = p.dz * s;
    } else {
        pair_apply<FAST>(pair_den<FAST>(p.n2), p.dx, p.dy, p.dz, mu, cx, cy, cz);
    }
}
__global__ void k_debug_inv_r3(long long n, const double *__restrict__ n2, double *__restrict__ fast,
                               double *__restrict__ ieee) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = n2[i];
    fast[i] = in_range(x) ? inv_r3_inrange(x) : __builtin_nan("");
    ieee[i] = inv_r3_ieee(x);
}

// Sweep of the in-range sequence against the compiler's IEEE expansion over counter-generated operands (splitmix64 of
// seed + index: 52 random mantissa bits, biased exponent uniform over the guarded range [723, 1323)). out[0] = number of
// operands whose two results differ in any bit, out[1] = the bits of one such operand.
__global__ void k_debug_inv_r3_sweep(unsigned long long seed, int per_thread, unsigned long long *out) {
    unsigned long long idx = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * (unsigned long long)per_thread;
    unsigned bad = 0;
    unsigned long long bad_x = 0;
    for (int k = 0; k < per_thread; ++k, ++idx) {
        unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const unsigned long long expo = 723ull + (z >> 52) % 600ull;
        const unsigned long long bits = (expo << 52) | (z & 0xFFFFFFFFFFFFFull);
        const double x = __longlong_as_double((long long)bits);
        const double f = inv_r3_inrange(x), g = inv_r3_ieee(x);
        if (__double_as_longlong(f) != __double_as_longlong(g)) { ++bad; bad_x = bits; }
    }
    if (bad) {
        atomicAdd(&out[0], (unsigned long long)bad);
        atomicExch(&out[1], bad_x);
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // same-wave LDS hand-off (lane-per-source writes -> lane-per-chain reads): DS ops of one wave execute in
    // order; this only stops the compiler from moving them across.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------
// Ordered accumulation of one 64-wide row of the contribution tile (phase B of wave_force).
// The row is read with ds_read_b128 in four 16-element chunks, the next chunk's reads in flight while the
// current one is added, so the dependent v_add_f64 chain never waits on LDS latency.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_chunk(const double *row, int c, double2 (&r)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const double2 *>(row + c * 16 + 2 * k);   // ds_read_b128
}
__device__ __forceinline__ double add_chunk(const double2 (&r)[8], double acc) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc = acc + r[k].x;
        acc = acc + r[k].y;
    }
    return acc;
}

// full tile, none of the wave's bodies inside it: 64 plain ordered adds
__device__ __forceinline__ double chain_full(const double *row, double acc) {
    double2 ra[8], rb[8];
    load_chunk(row, 0, ra);
    load_chunk(row, 1, rb);
    acc = add_chunk(ra, acc);
    load_chunk(row, 2, ra);
    acc = add_chunk(rb, acc);
    load_chunk(row, 3, rb);
    acc = add_chunk(ra, acc);
    return add_chunk(rb, acc);
}

// The tile that holds the wave's own bodies (and/or the ragged last tile). The wave's BPW bodies are consecutive
// and BPW-aligned, so they occupy exactly one BPW-wide group `gself` of the tile: groups before it are
// "sources before the body" for every chain, groups after it "sources after the body"; only inside that one group
// does a chain skip its own body, close the lower sum and restart from V::default(). cnt = valid sources.
// li = index of this lane's own body inside the tile.
template <int BPW>
__device__ __forceinline__ void chain_masked(const double *row, int cnt, int gself, int li, double &acc,
                                             double &accL) {
    double2 ra[8], rb[8];
    load_chunk(row, 0, ra);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double2(&cur)[8] = (c & 1) ? rb : ra;
        double2(&nxt)[8] = (c & 1) ? ra : rb;
        if (c * 16 >= cnt) break;                       // wave-uniform
        if (c < 3 && (c + 1) * 16 < cnt) load_chunk(row, c + 1, nxt);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int jl = c * 16 + e;
            const double v = (e & 1) ? cur[e >> 1].y : cur[e >> 1].x;
            if (jl >= cnt) continue;                    // wave-uniform
            if (jl / BPW != gself) {                    // wave-uniform
                acc = acc + v;
            } else {
                const bool self = (jl == li);
                const double t = acc + v;               // NaN on the self lane, discarded
                accL = self ? acc : accL;               // lower chain complete
                acc = self ? 0.0 : t;                   // upper chain starts from V::default()
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Software-pipelined tile step. While the ordered sum of tile t (phase B: 64 dependent v_add_f64 fed from LDS)
// advances, the same wave finishes the pair arithmetic of tile t+1 (22 VALU ops per body: in-range sqrt,
// reciprocal, scaling) and starts tile t+2 (8 ops per body: differences and squared distance). With one wave per
// SIMD nothing else can fill the issue slots a dependent add leaves empty, so the two instruction streams are
// interleaved explicitly, one chain add after every few independent pair ops, and pinned with
// sched_barrier (VALU may not cross; SALU / VMEM / DS may) -- the compiler's own schedule clusters the chain.
// ------------------------------------------------------------------------------------------------------
constexpr int kSchedMask = 0x4 | 0x10 | 0x80;   // SALU, VMEM, DS may cross a sched_barrier; VALU stays pinned

template <int BPW>
struct TileCtx {                 // registers of the pipelined tile steps
    PairPre pre[2][BPW];         // ping-pong: differences of the tile being finished / of the one after it
    double mu[2];                // source mu belonging to pre[k]
    Body4 src[2];                // prefetched sources, two steps ahead
    double c[3 * BPW];           // contributions being produced
    double y[BPW], g[BPW], h[BPW], r[BPW], d[BPW], p[BPW], tmp[BPW];
    double2 q[4][8];             // the LDS row being summed, four 16-element chunks
    double acc;
};

// stage S (0..21) of pair_finish<true> for body B on w.pre[PH]; same operations, same order as sqrt_inrange /
// rcp_inrange. Returns the value written (for the scheduling anchor).
template <int BPW, int PH, int S, int B>
__device__ __forceinline__ double &pair_stage(TileCtx<BPW> &w) {
    const PairPre &in = w.pre[PH][B];
    const double x = in.n2;
    if constexpr (S == 0) { w.y[B] = __builtin_amdgcn_rsq(x); return w.y[B]; }
    else if constexpr (S == 1) { w.g[B] = x * w.y[B]; return w.g[B]; }
    else if constexpr (S == 2) { w.h[B] = w.y[B] * 0.5; return w.h[B]; }
    else if constexpr (S == 3) { w.r[B] = __builtin_fma(-w.h[B], w.g[B], 0.5); return w.r[B]; }
    else if constexpr (S == 4) { w.g[B] = __builtin_fma(w.g[B], w.r[B], w.g[B]); return w.g[B]; }
    else if constexpr (S == 5) { w.h[B] = __builtin_fma(w.h[B], w.r[B], w.h[B]); return w.h[B]; }
    else if constexpr (S == 6) { w.d[B] = __builtin_fma(-w.g[B], w.g[B], x); return w.d[B]; }
    else if constexpr (S == 7) { w.g[B] = __builtin_fma(w.d[B], w.h[B], w.g[B]); return w.g[B]; }
    else if constexpr (S == 8) { w.d[B] = __builtin_fma(-w.g[B], w.g[B], x); return w.d[B]; }
    else if constexpr (S == 9) { w.g[B] = __builtin_fma(w.d[B], w.h[B], w.g[B]); return w.g[B]; }       // sqrt(n2)
    else if constexpr (S == 10) { w.p[B] = x * w.g[B]; return w.p[B]; }                                 // n2*sqrt(n2)
    // S11-13: the reciprocal's seed, 8 h^3 from the square root's refined h (inv_r3_seeded, device_math.h) or the
    // hardware seed plus one Newton step (rcp_inrange); the remaining steps are common
    else if constexpr (S == 11 && EPH_RCP_SEED_FROM_RSQ) { w.r[B] = w.h[B] * w.h[B]; return w.r[B]; }
    else if constexpr (S == 12 && EPH_RCP_SEED_FROM_RSQ) { w.r[B] = w.r[B] * w.h[B]; return w.r[B]; }
    else if constexpr (S == 13 && EPH_RCP_SEED_FROM_RSQ) { w.r[B] = w.r[B] * 8.0; return w.r[B]; }
    else if constexpr (S == 11) { w.r[B] = __builtin_amdgcn_rcp(w.p[B]); return w.r[B]; }
    else if constexpr (S == 12) { w.d[B] = __builtin_fma(-w.p[B], w.r[B], 1.0); return w.d[B]; }
    else if constexpr (S == 13) { w.r[B] = __builtin_fma(w.r[B], w.d[B], w.r[B]); return w.r[B]; }
    else if constexpr (S == 14) { w.d[B] = __builtin_fma(-w.p[B], w.r[B], 1.0); return w.d[B]; }
    else if constexpr (S == 15) { w.r[B] = __builtin_fma(w.r[B], w.d[B], w.r[B]); return w.r[B]; }
    else if constexpr (S == 16) { w.d[B] = __builtin_fma(-w.p[B], w.r[B], 1.0); return w.d[B]; }
    else if constexpr (S == 17) { w.r[B] = __builtin_fma(w.d[B], w.r[B], w.r[B]); return w.r[B]; }      // 1/(n2*sqrt(n2))
    else if constexpr (S == 18) { w.h[B] = w.mu[PH] * w.r[B]; return w.h[B]; }
    else if constexpr (S == 19) { w.c[3 * B + 0] = in.dx * w.h[B]; return w.c[3 * B + 0]; }
    else if constexpr (S == 20) { w.c[3 * B + 1] = in.dy * w.h[B]; return w.c[3 * B + 1]; }
    else { w.c[3 * B + 2] = in.dz * w.h[B]; return w.c[3 * B + 2]; }
}
// stage S (0..7) of pair_pre for body B: w.src[PH] -> w.pre[PH ^ 1]
template <int BPW, int PH, int S, int B>
__device__ __forceinline__ double &pre_stage(TileCtx<BPW> &w, const double (&xi)[BPW], const double (&yi)[BPW],
                                             const double (&zi)[BPW]) {
    PairPre &p = w.pre[PH ^ 1][B];
    const Body4 &pj = w.src[PH];
    if constexpr (S == 0) { p.dx = pj.x - xi[B]; return p.dx; }
    else if constexpr (S == 1) { p.dy = pj.y - yi[B]; return p.dy; }
    else if constexpr (S == 2) { p.dz = pj.z - zi[B]; return p.dz; }
    else if constexpr (S == 3) { p.n2 = p.dx * p.dx; return p.n2; }
    else if constexpr (S == 4) { w.tmp[B] = p.dy * p.dy; return w.tmp[B]; }
    else if constexpr (S == 5) { p.n2 = p.n2 + w.tmp[B]; return p.n2; }
    else if constexpr (S == 6) { w.tmp[B] = p.dz * p.dz; return w.tmp[B]; }
    else { p.n2 = p.n2 + w.tmp[B]; return p.n2; }
}
// chain adds M0 .. M1-1 of the 64 of this tile, each anchored so no pass can sink it past the next barrier
template <int BPW, int M0, int M1>
__device__ __forceinline__ void chain_adds(TileCtx<BPW> &w, const double *row) {
    if constexpr (M0 < M1) {
        if constexpr (M0 == 8) load_chunk(row, 2, w.q[2]);
        if constexpr (M0 == 24) load_chunk(row, 3, w.q[3]);
        const double2 &e = w.q[M0 / 16][(M0 % 16) / 2];
        w.acc = w.acc + ((M0 & 1) ? e.y : e.x);
        asm volatile("" : "+v"(w.acc));
        chain_adds<BPW, M0 + 1, M1>(w, row);
    }
}
// One slot of the fused instruction stream: op K of the 30*BPW pair ops, then its share of the 64 chain adds.
// The chain starts a quarter of the way in (the LDS reads issued at the top need ~130 cycles to land, and an
// in-order wave would otherwise sit on the first add with independent work queued behind it).
template <int BPW, int PH, int K>
__device__ __forceinline__ void fused_op(TileCtx<BPW> &w, const double (&xi)[BPW], const double (&yi)[BPW],
                                         const double (&zi)[BPW], const double *row_cur, double *tile_nxt, int lane) {
    constexpr int kOps = 30 * BPW;            // 22 finish + 8 pre per body
    constexpr int kLead = kOps / 4;
    constexpr int S = K / BPW, B = K % BPW;
    // every op is anchored with an empty asm: IR passes may otherwise sink pure arithmetic past the barriers
    // (towards its use in the next loop iteration) and undo the interleave
    if constexpr (S < 22) {
        double &v = pair_stage<BPW, PH, S, B>(w);
        asm volatile("" : "+v"(v));
        if constexpr (S == 21) {              // body B finished: publish its contributions to the OTHER LDS buffer
            tile_nxt[(3 * B + 0) * kRow + lane] = w.c[3 * B + 0];
            tile_nxt[(3 * B + 1) * kRow + lane] = w.c[3 * B + 1];
            tile_nxt[(3 * B + 2) * kRow + lane] = w.c[3 * B + 2];
        }
    } else {
        double &v = pre_stage<BPW, PH, S - 22, B>(w, xi, yi, zi);
        asm volatile("" : "+v"(v));
    }
    if constexpr (K >= kLead) {
        constexpr int J = K - kLead, kSpan = kOps - kLead;
        chain_adds<BPW, J * 64 / kSpan, (J + 1) * 64 / kSpan>(w, row_cur);
    }
    __builtin_amdgcn_sched_barrier(kSchedMask);
}
template <int BPW, int PH, int... K>
__device__ __forceinline__ void fused_ops(TileCtx<BPW> &w, const double (&xi)[BPW], const double (&yi)[BPW],
                                          const double (&zi)[BPW], const double *row_cur, double *tile_nxt, int lane,
                                          std::integer_sequence<int, K...>) {
    (fused_op<BPW, PH, K>(w, xi, yi, zi, row_cur, tile_nxt, lane), ...);
}

// Pipelined step PH (0/1 = ping-pong phase), processing tile t:
//   in : w.pre[PH], w.mu[PH] = tile t+1 (guarded in range); w.src[PH] = sources of tile t+2;
//        LDS buffer PH holds the contributions of tile t
//   out: contributions of tile t+1 in LDS buffer PH^1; w.pre[PH^1], w.mu[PH^1] = tile t+2; w.acc advanced over tile t
template <int BPW, int PH>
__device__ __forceinline__ void tile_step_fast(TileCtx<BPW> &w, const double (&xi)[BPW], const double (&yi)[BPW],
                                               const double (&zi)[BPW], const double *row_cur, double *tile_nxt,
                                               int lane) {
    load_chunk(row_cur, 0, w.q[0]);
    load_chunk(row_cur, 1, w.q[1]);
    fused_ops<BPW, PH>(w, xi, yi, zi, row_cur, tile_nxt, lane, std::make_integer_sequence<int, 30 * BPW>{});
    w.mu[PH ^ 1] = w.src[PH].mu;
}

// ------------------------------------------------------------------------------------------------------
// wave_force<BPW>: one wave64 computes the accelerations of BPW consecutive bodies i0..i0+BPW-1 in EXACTLY the
// reference's summation order (NewtonianGravity::eval, nbody.rs:22-38):
//     ddy[i] = ((init + c(0,i)) + ... + c(i-1,i))  +  ((0 + c(i,i+1)) + ... + c(i,n-1))
// Phase A: lane = source body j of the current 64-body tile; BPW independent interactions per lane go to a
//          wave-private LDS tile C[chain][j]  (chain = body*3 + component, row stride kRow doubles).
// Phase B: lane = chain (< 3*BPW); walks its row in j order with one dependent v_add_f64 per source.
// The sqrt/divide-heavy phase A is fully parallel; only the 3 adds per interaction are ordered. Full tiles away
// from the wave's own bodies are software-pipelined (A of tile t+1 overlaps B of tile t, sources two tiles ahead
// in flight); the tile holding the wave's bodies and a ragged last tile take the masked, un-pipelined form.
// Returns, on lane `ch` < 3*BPW, component ch%3 of body i0 + ch/3.
// ------------------------------------------------------------------------------------------------------
template <int BPW, bool SINGLE_TILE = false, typename PosPtr>
__device__ __forceinline__ double wave_force(PosPtr pos, int n, int i0, double init, double *C, int lane) {
    static_assert(kTile % BPW == 0, "BPW must divide the tile");
    double xi[BPW], yi[BPW], zi[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
        const int ii = min(i0 + b, n - 1);
        xi[b] = pos[ii].x;
        yi[b] = pos[ii].y;
        zi[b] = pos[ii].z;
    }
    const bool chain_lane = lane < 3 * BPW;
    const int ch = chain_lane ? lane : 3 * BPW - 1;
    const double *row = C + ch * kRow;
    const int tiles = (n + kTile - 1) / kTile;
    const int tfull = n / kTile;
    const int tdiag = i0 / kTile;                     // wave-uniform: i0 % BPW == 0 and BPW divides 64
    const int gself = (i0 % kTile) / BPW;
    const int li = (i0 % kTile) + ch / 3;
    double acc = init;   // lower chain (sources before the body), continues from the caller's value
    double accL = 0.0;

    auto load_src = [&](int t) -> Body4 {
        const int j = t * kTile + lane;
        return pos[j < n ? j : n - 1];
    };
    auto store_tile = [&](const double(&c)[3 * BPW]) {
#pragma unroll
        for (int q = 0; q < 3 * BPW; ++q) C[q * kRow + lane] = c[q];
    };
    // un-pipelined tile (holds the wave's own bodies and/or is the ragged last one): IEEE arithmetic throughout
    // (n2 = 0 on the self lane), masked chain
    auto special = [&](int t) {
        const Body4 pj = load_src(t);
        double c[3 * BPW];
#pragma unroll
        for (int b = 0; b < BPW; ++b)
            pair_finish<false>(pair_pre(xi[b], yi[b], zi[b], pj), pj.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
        store_tile(c);
        wave_lds_fence();
        if (chain_lane) chain_masked<BPW>(row, min(kTile, n - t * kTile), t == tdiag ? gself : -1, li, acc, accL);
        wave_lds_fence();
    };
    // pipelined run over the full tiles [tb, te), none of which holds the wave's bodies.
    // LDS is double buffered: step t sums buffer t&1 while the contributions of tile t+1 go to the other one.
    constexpr int kBuf = 3 * BPW * kRow;
    auto run = [&](int tb, int te) {
        if (tb >= te) return;
        TileCtx<BPW> w;
        {
            const Body4 pj = load_src(tb);
            double c[3 * BPW];
#pragma unroll
            for (int b = 0; b < BPW; ++b)
                pair_finish<false>(pair_pre(xi[b], yi[b], zi[b], pj), pj.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
            store_tile(c);                                  // tile tb -> buffer 0
        }
        {
            const Body4 pj = load_src(min(tb + 1, te - 1));
#pragma unroll
            for (int b = 0; b < BPW; ++b) w.pre[0][b] = pair_pre(xi[b], yi[b], zi[b], pj);
            w.mu[0] = pj.mu;
        }
        w.src[0] = load_src(min(tb + 2, te - 1));
        w.src[1] = load_src(min(tb + 3, te - 1));
        w.acc = acc;
        wave_lds_fence();
        // one step: PH = (t - tb) & 1
        auto step = [&](auto ph, int t) {
            constexpr int PH = decltype(ph)::value;
            const double *row_cur = row + PH * kBuf;
            double *tile_nxt = C + (PH ^ 1) * kBuf;
            unsigned worst = 0u;
#pragma unroll
            for (int b = 0; b < BPW; ++b) worst = max(worst, range_key(w.pre[PH][b].n2));
            const bool all_in_range = __builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0;
            if (kPairVariant == 0 && all_in_range) {        // the hand-interleaved stream restates variant 0 only
                tile_step_fast<BPW, PH>(w, xi, yi, zi, row_cur, tile_nxt, lane);
            } else {                                        // an operand near the end of the exponent range
#pragma unroll
                for (int b = 0; b < BPW; ++b) {
                    if (kPairVariant != 0 && all_in_range)
                        pair_finish<true>(w.pre[PH][b], w.mu[PH], w.c[3 * b], w.c[3 * b + 1], w.c[3 * b + 2]);
                    else
                        pair_finish<false>(w.pre[PH][b], w.mu[PH], w.c[3 * b], w.c[3 * b + 1], w.c[3 * b + 2]);
                    w.pre[PH ^ 1][b] = pair_pre(xi[b], yi[b], zi[b], w.src[PH]);
                }
#pragma unroll
                for (int q = 0; q < 3 * BPW; ++q) tile_nxt[q * kRow + lane] = w.c[q];
                w.mu[PH ^ 1] = w.src[PH].mu;
                w.acc = chain_full(row_cur, w.acc);
            }
            w.src[PH] = load_src(min(t + 4, te - 1));       // two steps ahead
            wave_lds_fence();
        };
        int t = tb;
        for (; t + 1 < te - 1; t += 2) {
            step(std::integral_constant<int, 0>{}, t);
            step(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < te - 1) {                                   // odd number of steps: last tile sits in buffer 1
            step(std::integral_constant<int, 0>{}, t);
            w.acc = chain_full(row + kBuf, w.acc);
        } else {
            w.acc = chain_full(row, w.acc);
        }
        acc = w.acc;
        wave_lds_fence();
    };

    if (SINGLE_TILE) {          // n <= 64 (persistent kernel): one masked tile, no pipeline code at all
        special(0);
        return accL + acc;
    }
    run(0, min(tdiag, tfull));
    special(tdiag);
    if (tdiag < tfull) {
        run(tdiag + 1, tfull);
        if (tfull < tiles) special(tfull);
    }
    return accL + acc;   // ddy[i] += output_i
}

// SRKN stage update of one (body, component) behind its force evaluation   symplectic.rs:90-97
__device__ __forceinline__ void kick_drift_one(const KickDrift &kd, size_t o, int body, int comp, double a) {
    const double vn = kd.v[o] + a * kd.hb;            // *dy = *dy + *ddy * (h * C::B[s])
    kd.v[o] = vn;
    const double yn = kd.y[o] + vn * kd.ha;           // *y = *y + *dy * (h * C::A[s])
    kd.y[o] = yn;
    reinterpret_cast<double *>(kd.pos_out + body)[comp] = yn;   // mu is already in both packed buffers
}
// ------------------------------------------------------------------------------------------------------
// k_accel: a = init + sum, SoA [3][npad] output. One wave per block, BPW bodies per wave.
// ------------------------------------------------------------------------------------------------------
template <int BPW>
__global__ void __launch_bounds__(64) k_accel(int n, int npad, const Body4 *__restrict__ pos,
                                              const double *__restrict__ acc_init, double *__restrict__ acc_out,
                                              int lo, int hi, KickDrift kd) {
    __shared__ __attribute__((aligned(16))) double C[2 * 3 * BPW * kRow];   // double buffered contribution tile
    const int lane = threadIdx.x;
    const int i0 = lo + blockIdx.x * BPW;          // targets [lo, hi): the whole system, or this rank's shard
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = lane < 3 * BPW && my_i < hi;
    const double init = (owner && acc_init) ? acc_init[cc * npad + my_i] : 0.0;
    const double a = wave_force<BPW>(pos, n, i0, init, C, lane);
    if (owner) {
        acc_out[cc * npad + my_i] = a;
        if (kd.v) kick_drift_one(kd, (size_t)cc * npad + my_i, my_i, cc, a);
    }
}

// ------------------------------------------------------------------------------------------------------
// Linear multistep pieces shared by the per-step and the persistent kernels.
//   predictor  ELM2::advance           integration/src/multistep/second_order/mod.rs:93-121
//   velocity   Cowell::update_velocity integration/src/multistep/second_order/cowell.rs:19-53
// yv[j], av[j] = position / acceleration component of level (newest - j).
// ------------------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ double lm_predict(const double (&yv)[L], const double (&av)[L], const double *wa,
                                             const double *wb, double hh) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        s1 = s1 + yv[j] * wa[j];   // *sum1 = *sum1 + *y * (1.0 * Ratio::from_int(-ALPHA[j+1]))
        s2 = s2 + av[j] * wb[j];   // *sum2 = *sum2 + *ddy * (1.0 * Ratio::from_int(BETA_N[j+1]))
    }
    return s1 + s2 * hh;           // *y = *sum1 + *sum2 * (h * h * Ratio::from_recip(BETA_D))
}

// a_new = acceleration of the new level; av[0..L-2] = the L-1 levels before it
template <int L>
__device__ __forceinline__ double lm_cowell(double a_new, const double (&av)[L], double y_new, double y_prev,
                                            const double *cw, double h, double hc) {
    double s = 0.0;
    s = s + a_new * cw[0];
#pragma unroll
    for (int j = 1; j < L; ++j) s = s + av[j - 1] * cw[j];
    return (y_new - y_prev) / h + s * hc;   // *dy = (*y - *ym1) / h + *work * (h * Ratio::from_recip(BETA_D))
}

__device__ __forceinline__ void maybe_sample(const SampleArgs &sa, int body, int comp, uint32_t step, double y) {
    if (!sa.period) return;
    const uint32_t m = sa.period[body];
    if (m == 0) return;
    const uint32_t t = sa.phase[body] + step;
    if (t % m == 0) sa.log[(sa.offset[body] + (uint64_t)(t / m - 1)) * 3 + comp] = y;
}

// ------------------------------------------------------------------------------------------------------
// k_lm_step: ONE launch per integrator step (all CUs). Slot `cur` of the ring holds the already predicted
// positions of the level being completed; this launch
//   1. evaluates its acceleration (reference-order all-pairs sum),
//   2. recovers its velocity (Cowell),
//   3. stores the solout sample if one is due,
//   4. predicts the positions of the NEXT level and publishes them (ring + packed ping-pong buffer),
// so the kernel boundary is the only grid-wide synchronisation a step needs.
// History reads are issued before the pair loop so their latency hides under it.
// ------------------------------------------------------------------------------------------------------
template <int BPW, int L>
__global__ void __launch_bounds__(64) k_lm_step(const LmArgs a) {
    __shared__ __attribute__((aligned(16))) double C[2 * 3 * BPW * kRow];   // double buffered contribution tile
    const int lane = threadIdx.x;
    const int i0 = a.lo + blockIdx.x * BPW;
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = lane < 3 * BPW && my_i < a.hi;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + (owner ? my_i : 0);

    double yv[L], av[L];   // yv[j]/av[j]: level (new - j); av[0] is filled after the force
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = a.Y[slot * lvl + off];           // non-owner lanes read body 0 (unused): no exec-masked loads
        av[j] = j > 0 ? a.A[slot * lvl + off] : 0.0;
    }

    const double anew = wave_force<BPW>(a.pos_cur, a.n, i0, 0.0, C, lane);
    if (!owner) return;

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

// ------------------------------------------------------------------------------------------------------
// wg_force: the all-pairs sum for kWgBodies consecutive bodies by one WORKGROUP of 5 waves with fixed roles.
//
// Measured on MI355X (scripts/ubench/lat.hip): a dependent v_add_f64 issues every 8.4 cycles from one wave (two
// issue slots), v_rsq_f64 / v_rcp_f64 cost ~17 issue cycles each, other f64 ops ~4.1. In the one-wave-per-block
// form (wave_force) every wave pays the 64 ordered adds and 32 ds_read_b128 of phase B per tile for only 3*BPW
// useful lanes, and at 4096 bodies there are just enough waves for one per SIMD, so nothing fills the bubbles.
// Here ONE wave (the chain wave) carries the ordered sums of all 16 bodies of the workgroup -- 48 of its 64 lanes
// -- while four pair waves produce the contribution tiles through a double-buffered LDS tile, one s_barrier per
// 64-source tile. The chain wave shares its SIMD with pair wave 0 (waves of a workgroup go to SIMDs round-robin),
// which therefore gets fewer bodies; the hardware interleaves the two and the adds' bubbles get filled.
// ------------------------------------------------------------------------------------------------------
#ifndef EPH_PAIR_LOOP
#define EPH_PAIR_LOOP 0
#endif
// tuning builds of the barrier-per-128-sources layouts (scripts/build_exp.sh): 1 = the chain wave skips its sums (pair
// side alone), 2 = the pair waves skip their tiles (chain side alone); results are then meaningless. Compile-time on
// purpose: the same two tests as RUN-time flags cost the default path 3.6 us per step (36.9 -> 40.5, gpurun_out r02v).
#ifndef EPH_WG_ACCOUNT
#define EPH_WG_ACCOUNT 0
#endif
#ifndef EPH_WG_SIDE
#define EPH_WG_SIDE 0
#endif
// ablations for the same tuning builds (results meaningless): 1 = pair waves keep their contributions in registers (no
// ds_write), 2 = no barrier inside the tile loops, 4 = pair waves reuse their first source tiles (no loads in the loop)
#ifndef EPH_WG_ABLATE
#define EPH_WG_ABLATE 0
#endif
#define WG_LOOP_BARRIER() do { if constexpr (!(EPH_WG_ABLATE & 2)) __syncthreads(); } while (0)
constexpr int kWgBodies = 16;
constexpr int kWgPairWaves = 4;                      // index of the chain wave (wave 4: lands on the SIMD of wave 0)
// Role layouts (waves of a workgroup go to the four SIMDs round-robin: wave k -> SIMD k % 4):
//   0: 5 waves  -- pair waves of 2/5/5/4 bodies + the chain wave (one wave per SIMD, the chain shares SIMD 0)
//   1: 8 waves  -- pair waves of 1/3/3/3 bodies, the chain wave, pair waves of 2/2/2 bodies: SIMDs 1-3 carry TWO pair
//                  waves (5 bodies together) that fill each other's dependency stalls; SIMD 0 the chain + one body
//   2: 8 waves  -- wave 0 only keeps the barrier count, pair waves of 3/3/3, the chain wave, pair waves of 3/2/2: the
//                  chain wave has SIMD 0 to itself; SIMDs 1-3 carry 6 / 5 / 5 bodies in two waves each
//   3: layout 1's roles with ONE barrier per 128 sources (two 64-source tiles, six LDS buffers): half the barriers,
//                  and every pair wave carries twice as many independent interactions between them
//   4: layout 3's barrier schedule with layout 2's roles (chain wave alone on SIMD 0, pair waves 3/3/3 + 3/2/2)
//   5: TWELVE waves (three per SIMD, <= 168 VGPRs each), layout 3's barrier schedule: SIMD 0 = one-body pair wave + chain
//                  wave + an idle wave, SIMDs 1-3 = pair waves of 2/2/1 bodies -- a third wave per SIMD to fill the f64
//                  issue gaps two leave (the fast path's pair arithmetic: 253 cycles per interaction at two waves per
//                  SIMD, 196 at four). The step kernel loads its history AFTER the force here (registers).
//   6: layout 5 with the idle wave given a body (SIMD 0: 1 + chain + 1; SIMDs 1-3: 2/2/1, 2/2/1, 2/1/1)
// In layout 5 the step kernel's history / Cowell / predictor work moves from the chain wave to the idle wave 8 (the "tail
// wave"): it loads the 2 L history values while the others work and receives the new acceleration through LDS. (Giving
// that work to a pair wave stalls the whole workgroup at the first barrier: a wave's loads return in order, so its first
// source tile queues behind 24 history loads -- 40.3-40.7 us. Layout 6 has no idle wave: its chain wave loads the history
// AFTER the force, the 168-register budget having no room to carry it through the loop.)
// Measured and dropped: sixteen waves (four per SIMD, <= 128 VGPRs, chain read ring of two chunks) 41.0-42.9 us; layout 5
// with the chain wave alone on SIMD 0 (pair waves 2/2/2, 2/2/1, 2/2/1) 41.2 us -- five bodies per pair SIMD is the balance.
constexpr int wg_threads(int layout) { return layout >= 5 ? 64 * 12 : layout >= 1 ? 64 * 8 : 64 * 5; }
constexpr int wg_tail_wave(int layout) { return layout == 5 ? 8 : 4; }   // 4 = the chain wave itself
constexpr bool wg_big(int layout) { return layout >= 3; }
constexpr int wg_bufs(int layout) { return wg_big(layout) ? 6 : 3; }
// measured at N = 4096 (us per step): layout 0 47.8, 1 44.6, 2 48.9, 3 41.6 (gpurun_out r02c, r02d); with the seeded
// reciprocal 3: 40.2, 5: 37.0, 6: 39.0 (r02q)
constexpr int kWgDefaultLayout = 5;
constexpr int kWgRows = 3 * kWgBodies;
constexpr int kWgBuf = kWgRows * kRow;               // doubles per LDS buffer
constexpr int kWgBufs = 3;                           // pair waves run two tiles ahead of the chain wave
__device__ long long g_wg_cycles[8];                 // debug (EPH_DEBUG_WG=4): cycle accounting of workgroup 7
__device__ long long g_wg_span[4][1024];             // debug: per workgroup entry / force-done ticks of the step kernel

// pair wave: NB bodies (local indices b0..) against the 64 sources in pj -> rows of `tile`
template <int NB>
__device__ __forceinline__ void wg_pair_tile(const double (&xi)[NB], const double (&yi)[NB], const double (&zi)[NB],
                                             const Body4 &pj, bool ieee, double *tile, int b0, int lane) {
    // (forcing a stage-major interleave of the NB interactions with scheduling anchors was measured: no gain
    // over the compiler's own schedule here, 1380 vs 1400 cycles per 5-body tile)
    // (the workgroup's own tile goes to the IEEE form as a whole -- n2 = 0 on the self lanes; giving those lanes a
    // harmless in-range operand instead, since the chain wave never reads them, was measured: no gain, 40.1 vs 40.0 us)
    PairPre pre[NB];
    unsigned worst = ieee ? kRangeSpan : 0u;           // max of the range keys: one add + one max per body
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        pre[b] = pair_pre(xi[b], yi[b], zi[b], pj);
        worst = max(worst, range_key(pre[b].n2));
    }
    double c[3 * NB];
    if (__builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) pair_finish<true>(pre[b], pj.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
    } else {   // the tile holding the workgroup's own bodies (n2 = 0 on the self lane) or an out-of-range operand
#pragma unroll
        for (int b = 0; b < NB; ++b) pair_finish<false>(pre[b], pj.mu, c[3 * b], c[3 * b + 1], c[3 * b + 2]);
    }
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) {
        if constexpr (EPH_WG_ABLATE & 1) asm volatile("" ::"v"(c[q]));
        else tile[(3 * b0 + q) * kRow + lane] = c[q];
    }
}

// Barrier schedule (every wave executes tiles+1 barriers): B_0 after tiles 0 and 1 are in LDS; iteration t: pair
// waves produce tile t+2 into buffer (t+2)%3 while the chain wave sums tile t from buffer t%3; barrier.
template <int NB, typename PosPtr>
__device__ __forceinline__ void wg_pair_wave(PosPtr pos, int n, int i0, int b0, double *C, int lane, int tiles,
                                             int tdiag, int dbg, int wbuf = kWgBuf) {   // wbuf: doubles per LDS tile buffer
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
    long long t_work = 0, t_bar = 0;
    for (int t = 0; t < tiles; ++t) {
        const long long c0 = __builtin_readcyclecounter();
        if (t + 2 < tiles) {
            pj = pjn;
            pjn = load_src(t + 3);
            if (!(dbg & 2))
                wg_pair_tile<NB>(xi, yi, zi, pj, tdiag == t + 2, C + ((t + 2) % kWgBufs) * wbuf, b0, lane);
        }
        const long long c1 = __builtin_readcyclecounter();
        __syncthreads();
        const long long c2 = __builtin_readcyclecounter();
        t_work += c1 - c0;
        t_bar += c2 - c1;
    }
    if ((dbg & 4) && blockIdx.x == 7 && lane == 0 && b0 == 7) { g_wg_cycles[0] = t_work; g_wg_cycles[1] = t_bar; }
    if ((dbg & 4) && blockIdx.x == 7 && lane == 0 && b0 == 0) { g_wg_cycles[2] = t_work; g_wg_cycles[3] = t_bar; }
}

// two 64-source tiles at once (layout 3): 2 x NB independent interactions for the scheduler to interleave
// The in-range term of variant 0 for M interactions at once, STAGE by stage with the VALU order pinned (sched_barrier between
// stages): the M operations of a stage are independent, so a wave covers part of the dependent latency on its own instead of
// leaving all of it to the other waves of its SIMD. Same operations as pair_finish<true>, same bits. Measured at N = 4096 on one
// box, two runs each: 36.91 / 36.93 us against 37.32 / 37.31 without (-DEPH_PAIR_STAGED=0); nothing at N <= 2048, where the
// ordered sums are the critical path (profiles/r03_step_kernel_evidence.md section 6).
#ifndef EPH_PAIR_STAGED
#define EPH_PAIR_STAGED 1
#endif
template <int M>
__device__ __forceinline__ void pair_finish_staged(const PairPre (&pre)[M], const double (&mu)[M], double (&c)[3 * M]) {
    double x[M], g[M], h[M], r[M], d[M], p[M], q[M], e[M];
#define EPH_STAGE(body) _Pragma("unroll") for (int k = 0; k < M; ++k) { body; } __builtin_amdgcn_sched_barrier(kSchedMask)
    EPH_STAGE(x[k] = pre[k].n2; q[k] = __builtin_amdgcn_rsq(x[k]));
    EPH_STAGE(g[k] = x[k] * q[k]; h[k] = q[k] * 0.5);
    EPH_STAGE(r[k] = __builtin_fma(-h[k], g[k], 0.5));
    EPH_STAGE(g[k] = __builtin_fma(g[k], r[k], g[k]); h[k] = __builtin_fma(h[k], r[k], h[k]));
    EPH_STAGE(d[k] = __builtin_fma(-g[k], g[k], x[k]));
    EPH_STAGE(g[k] = __builtin_fma(d[k], h[k], g[k]));
    EPH_STAGE(d[k] = __builtin_fma(-g[k], g[k], x[k]); q[k] = h[k] * h[k]);
    EPH_STAGE(g[k] = __builtin_fma(d[k], h[k], g[k]); q[k] = q[k] * h[k]);
    EPH_STAGE(p[k] = x[k] * g[k]; q[k] = q[k] * 8.0);
    EPH_STAGE(e[k] = __builtin_fma(-p[k], q[k], 1.0));
    EPH_STAGE(q[k] = __builtin_fma(q[k], e[k], q[k]));
    EPH_STAGE(e[k] = __builtin_fma(-p[k], q[k], 1.0));
    EPH_STAGE(q[k] = __builtin_fma(e[k], q[k], q[k]));
    EPH_STAGE(q[k] = mu[k] * q[k]);
    EPH_STAGE(c[3 * k] = pre[k].dx * q[k]; c[3 * k + 1] = pre[k].dy * q[k]; c[3 * k + 2] = pre[k].dz * q[k]);
#undef EPH_STAGE
}
template <int NB>
__device__ __forceinline__ void wg_pair_tile2(const double (&xi)[NB], const double (&yi)[NB], const double (&zi)[NB],
                                              const Body4 &pa, const Body4 &pb, bool ieee, double *tile_a, double *tile_b,
                                              int b0, int lane) {
    PairPre pre[2 * NB];
    unsigned worst = ieee ? kRangeSpan : 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        pre[b] = pair_pre(xi[b], yi[b], zi[b], pa);
        pre[NB + b] = pair_pre(xi[b], yi[b], zi[b], pb);
        worst = max(worst, max(range_key(pre[b].n2), range_key(pre[NB + b].n2)));
    }
    double c[6 * NB];
    if (__builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0) {
        // (round 2 wrote the stages out WITHOUT pinning the order and the scheduler put them back: 37.38 vs 37.04 us)
        if constexpr (EPH_PAIR_STAGED && kPairVariant == 0 && EPH_RCP_SEED_FROM_RSQ) {
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
template <int NB, typename PosPtr>
__device__ __forceinline__ void wg_pair_wave_big(PosPtr pos, int n, int i0, int b0, double *C, int lane, int tiles, int tdiag,
                                                 int dbg = 0, int wbuf = kWgBuf) {   // wbuf: doubles per LDS tile buffer
    double xi[NB], yi[NB], zi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ii = min(i0 + b0 + b, n - 1);
        xi[b] = pos[ii].x;
        yi[b] = pos[ii].y;
        zi[b] = pos[ii].z;
    }
    // EPH_PAIR_LOOP (tuning; -DEPH_PAIR_LOOP=k, scripts/build_exp.sh): bit 0 = source rows addressed as SGPR tile base +
    // loop-invariant VGPR byte offset (only the last tile of a ragged n clamps it); bit 1 = two iterations per trip with
    // the (a, b) and (na, nb) register sets trading places instead of being copied. Both remove VALU bookkeeping (about 20
    // of the ~260 instructions of an iteration) and both measured SLOWER at N = 4096: 0: 39.78, 1: 40.32, 2: 40.05,
    // 3: 40.27 us per step (gpurun_out r02m) -- the pair waves are bound by f64 issue, not by their integer overhead.
    const unsigned off_full = (unsigned)lane * (unsigned)sizeof(Body4);
    const unsigned off_last = (unsigned)min(lane, n - 1 - (tiles - 1) * kTile) * (unsigned)sizeof(Body4);
    auto load_src = [&](int t) -> Body4 {
        if constexpr (EPH_PAIR_LOOP & 1) {
            const int tt = min(t, tiles - 1);
            const char *tp = (const char *)&pos[(size_t)tt * kTile];
            return *(const Body4 *)(tp + (tt == tiles - 1 ? off_last : off_full));
        } else {
            const int j = min(t, tiles - 1) * kTile + lane;
            return pos[j < n ? j : n - 1];
        }
    };
    auto produce = [&](int K, const Body4 &pa, const Body4 &pb) {          // big tile K
        const int t = big_start(K);
        if (t >= tiles || EPH_WG_SIDE == 2) return;    // -DEPH_WG_SIDE=2 (tuning): pair waves idle, the chain side alone
        double *ta = C + (t % 6) * wbuf, *tb = C + ((t + 1) % 6) * wbuf;
        if (K >= 2 && t + 1 < tiles) wg_pair_tile2<NB>(xi, yi, zi, pa, pb, tdiag == t || tdiag == t + 1, ta, tb, b0, lane);
        else wg_pair_tile<NB>(xi, yi, zi, pa, tdiag == t, ta, b0, lane);
    };
    const int TB = big_count(tiles);
    Body4 pa = load_src(0), pb = load_src(1), na = load_src(2), nb = load_src(3);
    produce(0, pa, pa);
    produce(1, pb, pb);
    __syncthreads();
    if constexpr (EPH_PAIR_LOOP & 2) {
        for (int K = 0; K < TB; K += 2) {
            pa = load_src(big_start(K + 3)); pb = load_src(big_start(K + 3) + 1);
            produce(K + 2, na, nb);
            __syncthreads();
            if (K + 1 >= TB) break;
            na = load_src(big_start(K + 4)); nb = load_src(big_start(K + 4) + 1);
            produce(K + 3, pa, pb);
            __syncthreads();
        }
    } else {
        long long t_work = 0, t_bar = 0;               // -DEPH_WG_ACCOUNT=1 (tuning): s_memtime ticks producing / at the barrier
        for (int K = 0; K < TB; ++K) {
            const long long c0 = EPH_WG_ACCOUNT ? __builtin_readcyclecounter() : 0;
            pa = na; pb = nb;
            if constexpr (!(EPH_WG_ABLATE & 4)) { na = load_src(big_start(K + 3)); nb = load_src(big_start(K + 3) + 1); }
            else asm volatile("" : "+v"(na.x), "+v"(nb.x));   // opaque: nothing may be hoisted out of the loop
            produce(K + 2, pa, pb);
            const long long c1 = EPH_WG_ACCOUNT ? __builtin_readcyclecounter() : 0;
            WG_LOOP_BARRIER();
            if constexpr (EPH_WG_ACCOUNT) { t_work += c1 - c0; t_bar += __builtin_readcyclecounter() - c1; }
        }
        if constexpr (EPH_WG_ACCOUNT) {                // layout 5: b0 = 1 is a two-body wave of SIMD 1, b0 = 5 its one-body wave
            if (blockIdx.x == 7 && lane == 0 && b0 == 1) { g_wg_cycles[0] = t_work; g_wg_cycles[1] = t_bar; }
            if (blockIdx.x == 7 && lane == 0 && b0 == 5) { g_wg_cycles[2] = t_work; g_wg_cycles[3] = t_bar; }
        }
    }
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

// The same ordered sum of a full tile with the instruction order pinned (tools/gen_chain_tile.py): 16 reads kept in
// flight ahead of the dependent adds. Self-contained (no registers carried between tiles). EXPERIMENT (-DEPH_CHAIN_ASM=1):
// alone on a CU it beats the compiler's chunked schedule (724 vs 930 ticks per tile, scripts/ubench/chain2.hip).
#ifndef EPH_CHAIN_ASM
#define EPH_CHAIN_ASM 0
#endif
__device__ __forceinline__ double chain_full_asm(const double *row, double acc) {
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)row;   // LDS byte address
#include "chain_tile.inc"
    return acc;
}

// -DEPH_WG_TILE_SPLIT=1: the 8- and 4-body workgroups (N <= 2048, where the step IS the chain wave's time) with TWO chain
// waves taking ALTERNATE tiles: while one adds the 64 sources of tile t out of its registers, the other reads tile t + 1
// into its own (a wave's ds_read_b128 and its dependent adds do not overlap; two waves' do), and the 3 * WB partial sums
// change hands through LDS at every tile's barrier. Same number of LDS reads as one chain wave (the split by LANE doubled
// them). One barrier per 64-source tile (three tile buffers), pair waves two tiles ahead.
// MEASURED (bit-identical; us per step, split | default): N = 640 9.29 | 9.83, 1024 11.74 | 12.01, 1536 16.10 | 15.37,
// 2048 19.58 | 18.34 -- a gain only where four bodies per workgroup leave the pair side idle anyway; with eight one-body
// pair waves a barrier per tile makes the pair side (one interaction per wave and tile: a bare dependent chain) the
// slower one. Producing the tiles in pairs in every other interval to get two interleaved interactions back: 10.4 / 13.0 /
// 18.7 / 23.0, worse still (the reader waits out the double intervals). Default: the 4-body workgroups only.
#ifndef EPH_WG_TILE_SPLIT
#define EPH_WG_TILE_SPLIT 2      // 0 off | 1 the 8- and 4-body workgroups | 2 (default) the 4-body workgroups only, where it wins
#endif
constexpr bool wg_tile_split(int wb) { return EPH_WG_TILE_SPLIT == 1 ? wb < 16 : (EPH_WG_TILE_SPLIT == 2 && wb == 4); }
constexpr int wg_split_wave_b(int wb) { return wb == 8 ? 11 : 6; }   // an idle wave of another SIMD than the chain wave's
template <int WB, typename PosPtr>
__device__ __forceinline__ double wg_force_split(PosPtr pos, int n, int i0, double init, double *C, int tid, int dbg) {
    constexpr int kRows = 3 * WB, kBuf = kRows * kRow;
    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = (n + kTile - 1) / kTile;
    const int tdiag = i0 / kTile;
    double *H = C + 3 * kBuf;                           // hand-over: H[lane] = running sum, H[64 + lane] = closed lower sum
    int body = -1;                                      // pair waves: one body each
    if constexpr (WB == 8) {
        switch (wave) { case 1: body = 0; break; case 2: body = 1; break; case 3: body = 2; break; case 5: body = 3; break;
                        case 6: body = 4; break; case 7: body = 5; break; case 9: body = 6; break; case 10: body = 7; break; default: break; }
    } else {
        switch (wave) { case 1: body = 0; break; case 2: body = 1; break; case 3: body = 2; break; case 5: body = 3; break; default: break; }
    }
    if (body >= 0) { wg_pair_wave<1>(pos, n, i0, body, C, lane, tiles, tdiag, dbg & ~7, kBuf); return 0.0; }
    const bool isA = wave == kWgPairWaves, isB = wave == wg_split_wave_b(WB);
    if (!isA && !isB) { for (int t = 0; t <= tiles; ++t) __syncthreads(); return 0.0; }
    const int ch = lane < kRows ? lane : kRows - 1;
    const double *row = C + ch * kRow;
    const int gself = (i0 % kTile) / WB;
    const int li = (i0 % kTile) + ch / 3;
    double acc = init, accL = 0.0;
    double2 q[4][8];
    auto plain = [&](int t) { return t < tiles && t != tdiag && min(kTile, n - t * kTile) == kTile; };
    auto preload = [&](int t) {
        const double *r = row + (t % 3) * kBuf;
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
                chain_masked<WB>(row + (t % 3) * kBuf, min(kTile, n - t * kTile), t == tdiag ? gself : -1, li, acc, accL);
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

// (Round 3, for the 8- and 4-body workgroups where the step IS the chain wave's time: the chains split by LANE between two
// chain waves -- wave 4 and an idle wave, first of the same SIMD, then of another -- each owning its chains from the first
// tile to the last, no hand-over. Bit-identical and SLOWER: N = 2048 22.0 / 23.2 us against 18.4, N = 1024 13.9 / 13.6
// against 11.9. A ds_read_b128 costs the workgroup's one LDS pipe the same whether 24 or 12 lanes carry a chain, so two
// chain waves double the LDS time of the ordered sums instead of overlapping one wave's reads with the other's adds. What
// would keep the read count is a split by TILE with the 3 * WB partial sums handed over between the waves every tile.)
// Returns on chain-wave lane ch < 48: component ch%3 of body i0 + ch/3. All 320 threads must call it.
template <int LAYOUT, int WB = kWgBodies, typename PosPtr>
__device__ __forceinline__ double wg_force(PosPtr pos, int n, int i0, double init, double *C, int tid, int dbg = 0) {
    static_assert(WB == kWgBodies || LAYOUT == 5, "8- and 4-body workgroups exist in the default layout only");
    constexpr int kRows = 3 * WB, kBuf = kRows * kRow;   // chains of the chain wave, doubles per LDS tile buffer
    const int lane = tid & 63, wave = tid >> 6;
    const int tiles = (n + kTile - 1) / kTile;
    const int tdiag = i0 / kTile;
    // Workgroups of 8 / 4 bodies for target counts that would leave CUs without a 16-body workgroup (<= 2048 / <= 1024
    // targets): the chain wave's cost per tile does not depend on how many of its lanes carry a chain, so with one
    // workgroup per CU the step takes the chain wave's time; one body per pair wave, SIMD 0 left to the chain wave.
    if constexpr (wg_tile_split(WB)) return wg_force_split<WB>(pos, n, i0, init, C, tid, dbg);
    if constexpr (WB == 8) {
        switch (wave) {
            case 1: wg_pair_wave_big<1>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 2: wg_pair_wave_big<1>(pos, n, i0, 1, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 3: wg_pair_wave_big<1>(pos, n, i0, 2, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 5: wg_pair_wave_big<1>(pos, n, i0, 3, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 6: wg_pair_wave_big<1>(pos, n, i0, 4, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 7: wg_pair_wave_big<1>(pos, n, i0, 5, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 9: wg_pair_wave_big<1>(pos, n, i0, 6, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 10: wg_pair_wave_big<1>(pos, n, i0, 7, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 0: case 8: case 11: __syncthreads(); for (int T = 0; T < big_count(tiles); ++T) WG_LOOP_BARRIER(); return 0.0;
            default: break;
        }
    } else if constexpr (WB == 4) {
        switch (wave) {
            case 1: wg_pair_wave_big<1>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 2: wg_pair_wave_big<1>(pos, n, i0, 1, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 3: wg_pair_wave_big<1>(pos, n, i0, 2, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 5: wg_pair_wave_big<1>(pos, n, i0, 3, C, lane, tiles, tdiag, dbg, kBuf); return 0.0;
            case 4: break;
            default: __syncthreads(); for (int T = 0; T < big_count(tiles); ++T) WG_LOOP_BARRIER(); return 0.0;
        }
    } else if constexpr (LAYOUT == 5) {
        // (waves of a workgroup are placed on the four SIMDs round-robin: wave k on SIMD k % 4)
        switch (wave) {
            case 0: wg_pair_wave_big<1>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 8: __syncthreads(); for (int T = 0; T < big_count(tiles); ++T) WG_LOOP_BARRIER(); return 0.0;   // TB + 1 barriers
            case 1: wg_pair_wave_big<2>(pos, n, i0, 1, C, lane, tiles, tdiag, dbg); return 0.0;
            case 5: wg_pair_wave_big<2>(pos, n, i0, 3, C, lane, tiles, tdiag, dbg); return 0.0;
            case 9: wg_pair_wave_big<1>(pos, n, i0, 5, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave_big<2>(pos, n, i0, 6, C, lane, tiles, tdiag, dbg); return 0.0;
            case 6: wg_pair_wave_big<2>(pos, n, i0, 8, C, lane, tiles, tdiag, dbg); return 0.0;
            case 10: wg_pair_wave_big<1>(pos, n, i0, 10, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave_big<2>(pos, n, i0, 11, C, lane, tiles, tdiag, dbg); return 0.0;
            case 7: wg_pair_wave_big<2>(pos, n, i0, 13, C, lane, tiles, tdiag, dbg); return 0.0;
            case 11: wg_pair_wave_big<1>(pos, n, i0, 15, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    } else if constexpr (LAYOUT == 6) {
        switch (wave) {
            case 0: wg_pair_wave_big<1>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 8: wg_pair_wave_big<1>(pos, n, i0, 1, C, lane, tiles, tdiag, dbg); return 0.0;
            case 1: wg_pair_wave_big<2>(pos, n, i0, 2, C, lane, tiles, tdiag, dbg); return 0.0;
            case 5: wg_pair_wave_big<2>(pos, n, i0, 4, C, lane, tiles, tdiag, dbg); return 0.0;
            case 9: wg_pair_wave_big<1>(pos, n, i0, 6, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave_big<2>(pos, n, i0, 7, C, lane, tiles, tdiag, dbg); return 0.0;
            case 6: wg_pair_wave_big<2>(pos, n, i0, 9, C, lane, tiles, tdiag, dbg); return 0.0;
            case 10: wg_pair_wave_big<1>(pos, n, i0, 11, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave_big<2>(pos, n, i0, 12, C, lane, tiles, tdiag, dbg); return 0.0;
            case 7: wg_pair_wave_big<1>(pos, n, i0, 14, C, lane, tiles, tdiag, dbg); return 0.0;
            case 11: wg_pair_wave_big<1>(pos, n, i0, 15, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    } else if constexpr (LAYOUT == 4) {
        switch (wave) {
            case 0: __syncthreads(); for (int T = 0; T < big_count(tiles); ++T) WG_LOOP_BARRIER(); return 0.0;   // TB + 1 barriers
            case 1: wg_pair_wave_big<3>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave_big<3>(pos, n, i0, 3, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave_big<3>(pos, n, i0, 6, C, lane, tiles, tdiag, dbg); return 0.0;
            case 5: wg_pair_wave_big<3>(pos, n, i0, 9, C, lane, tiles, tdiag, dbg); return 0.0;
            case 6: wg_pair_wave_big<2>(pos, n, i0, 12, C, lane, tiles, tdiag, dbg); return 0.0;
            case 7: wg_pair_wave_big<2>(pos, n, i0, 14, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    } else if constexpr (LAYOUT == 3) {
        switch (wave) {
            case 0: wg_pair_wave_big<1>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 1: wg_pair_wave_big<3>(pos, n, i0, 1, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave_big<3>(pos, n, i0, 4, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave_big<3>(pos, n, i0, 7, C, lane, tiles, tdiag, dbg); return 0.0;
            case 5: wg_pair_wave_big<2>(pos, n, i0, 10, C, lane, tiles, tdiag, dbg); return 0.0;
            case 6: wg_pair_wave_big<2>(pos, n, i0, 12, C, lane, tiles, tdiag, dbg); return 0.0;
            case 7: wg_pair_wave_big<2>(pos, n, i0, 14, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    } else if constexpr (LAYOUT == 2) {
        switch (wave) {
            case 0: for (int t = 0; t <= tiles; ++t) __syncthreads(); return 0.0;   // tiles + 1 barriers, like every wave
            case 1: wg_pair_wave<3>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave<3>(pos, n, i0, 3, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave<3>(pos, n, i0, 6, C, lane, tiles, tdiag, dbg); return 0.0;
            case 5: wg_pair_wave<3>(pos, n, i0, 9, C, lane, tiles, tdiag, dbg); return 0.0;
            case 6: wg_pair_wave<2>(pos, n, i0, 12, C, lane, tiles, tdiag, dbg); return 0.0;
            case 7: wg_pair_wave<2>(pos, n, i0, 14, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    } else if constexpr (LAYOUT == 1) {
        switch (wave) {
            case 0: wg_pair_wave<1>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 1: wg_pair_wave<3>(pos, n, i0, 1, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave<3>(pos, n, i0, 4, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave<3>(pos, n, i0, 7, C, lane, tiles, tdiag, dbg); return 0.0;
            case 5: wg_pair_wave<2>(pos, n, i0, 10, C, lane, tiles, tdiag, dbg); return 0.0;
            case 6: wg_pair_wave<2>(pos, n, i0, 12, C, lane, tiles, tdiag, dbg); return 0.0;
            case 7: wg_pair_wave<2>(pos, n, i0, 14, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    } else {
        switch (wave) {
            case 0: wg_pair_wave<2>(pos, n, i0, 0, C, lane, tiles, tdiag, dbg); return 0.0;
            case 1: wg_pair_wave<5>(pos, n, i0, 2, C, lane, tiles, tdiag, dbg); return 0.0;
            case 2: wg_pair_wave<5>(pos, n, i0, 7, C, lane, tiles, tdiag, dbg); return 0.0;
            case 3: wg_pair_wave<4>(pos, n, i0, 12, C, lane, tiles, tdiag, dbg); return 0.0;
            default: break;
        }
    }
    // chain wave. Its dependent adds issue ahead of the one-body pair wave of its SIMD in the twelve-wave layout (s_setprio; the
    // same library with and without, alternating on one box: 36.3 against 36.8 us per step at N = 4096 on two boxes of the pool,
    // 36.2 either way on a third; nothing at the chain-bound sizes; layout 6 39.4 -> 37.9, still behind -- SIMD 0 has no room for
    // a second body either way. In layout 0 the same priority was 4-5 us SLOWER per evaluation.)
    if (LAYOUT == 5 || (dbg & 8)) __builtin_amdgcn_s_setprio(3);
    const int ch = lane < kRows ? lane : kRows - 1;
    const double *row = C + ch * kRow;
    const int gself = (i0 % kTile) / WB;
    const int li = (i0 % kTile) + ch / 3;
    double acc = init, accL = 0.0;
    double2 q[4][8];
    long long t_work = 0, t_bar = 0;
    const long long c_start = __builtin_readcyclecounter();
    __syncthreads();                                  // B_0: tiles 0 and 1 ready
    if constexpr (!(EPH_CHAIN_ASM && wg_big(LAYOUT))) {
        load_chunk(row, 0, q[0]);
        load_chunk(row, 1, q[1]);
    }
    if constexpr (wg_big(LAYOUT)) {
        const int TB = big_count(tiles);
        if ((dbg & 4) && blockIdx.x == 7 && lane == 0) g_wg_cycles[4] = __builtin_readcyclecounter() - c_start;   // wait for B_0
        for (int T = 0; T < TB; ++T) {
            const long long c0 = EPH_WG_ACCOUNT ? __builtin_readcyclecounter() : 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int t = big_start(T) + hf;
                if (t >= tiles || (T < 2 && hf)) break;
                const double *r = row + (t % 6) * kBuf;
                const double *rn = row + ((t + 1) % 6) * kBuf;   // complete since the previous barrier
                const int cnt = min(kTile, n - t * kTile);
                if constexpr (EPH_WG_SIDE == 1) {      // tuning: chain wave idle, the pair side alone
                } else if (t != tdiag && cnt == kTile) {
                    if constexpr (EPH_CHAIN_ASM) acc = chain_full_asm(r, acc);   // experimental build: pinned order
                    else acc = chain_full_pf(r, rn, q, acc);
                } else {
                    chain_masked<WB>(r, cnt, t == tdiag ? gself : -1, li, acc, accL);
                    if constexpr (!EPH_CHAIN_ASM) {
                        load_chunk(rn, 0, q[0]);
                        load_chunk(rn, 1, q[1]);
                    }
                }
            }
            const long long c1 = EPH_WG_ACCOUNT ? __builtin_readcyclecounter() : 0;
            WG_LOOP_BARRIER();                          // big tile T consumed, big tile T + 2 ready
            if constexpr (EPH_WG_ACCOUNT) { t_work += c1 - c0; t_bar += __builtin_readcyclecounter() - c1; }
        }
        if constexpr (EPH_WG_ACCOUNT) {
            if (blockIdx.x == 7 && lane == 0) { g_wg_cycles[4] = t_work; g_wg_cycles[5] = t_bar; g_wg_cycles[6] = tiles; g_wg_cycles[7] = __builtin_readcyclecounter() - c_start; }
            return accL + acc;
        }
        if ((dbg & 4) && blockIdx.x == 7 && lane == 0) { g_wg_cycles[6] = tiles; g_wg_cycles[7] = __builtin_readcyclecounter() - c_start; }
        return accL + acc;
    } else {
    for (int t = 0; t < tiles; ++t) {
        const long long c0 = __builtin_readcyclecounter();
        const double *r = row + (t % kWgBufs) * kWgBuf;
        const double *rn = row + ((t + 1) % kWgBufs) * kWgBuf;   // complete since the previous barrier
        const int cnt = min(kTile, n - t * kTile);
        if (dbg & 1) {
        } else if (t != tdiag && cnt == kTile) {
            acc = chain_full_pf(r, rn, q, acc);
        } else {
            chain_masked<kWgBodies>(r, cnt, t == tdiag ? gself : -1, li, acc, accL);
            load_chunk(rn, 0, q[0]);
            load_chunk(rn, 1, q[1]);
        }
        const long long c1 = __builtin_readcyclecounter();
        __syncthreads();                              // tile t consumed, tile t+2 ready
        const long long c2 = __builtin_readcyclecounter();
        t_work += c1 - c0;
        t_bar += c2 - c1;
    }
    if ((dbg & 4) && blockIdx.x == 7 && lane == 0) { g_wg_cycles[4] = t_work; g_wg_cycles[5] = t_bar; g_wg_cycles[6] = tiles; g_wg_cycles[7] = __builtin_readcyclecounter() - c_start; }
    return accL + acc;
    }
}

template <int LAYOUT, int WB = kWgBodies>
__global__ void __launch_bounds__(wg_threads(LAYOUT)) k_accel_wg(int n, int npad, const Body4 *__restrict__ pos,
                                                         const double *__restrict__ acc_init,
                                                         double *__restrict__ acc_out, int dbg, int lo, int hi,
                                                         KickDrift kd) {
    __shared__ __attribute__((aligned(16))) double C[wg_bufs(LAYOUT) * 3 * WB * kRow];
    const int tid = threadIdx.x, lane = tid & 63;
    const int i0 = lo + blockIdx.x * WB;
    const int my_i = i0 + lane / 3, cc = lane % 3;
    const bool owner = (tid >> 6) == kWgPairWaves && lane < 3 * WB && my_i < hi;
    const double init = (owner && acc_init) ? acc_init[cc * npad + my_i] : 0.0;
    const double a = wg_force<LAYOUT, WB>(pos, n, i0, init, C, tid, dbg);
    if (owner) {
        acc_out[cc * npad + my_i] = a;
        if (kd.v) kick_drift_one(kd, (size_t)cc * npad + my_i, my_i, cc, a);
    }
}

// One launch per integrator step, workgroup-specialised force (see k_lm_step for the step structure).
template <int L, int LAYOUT, int WB = kWgBodies>
__global__ void __launch_bounds__(wg_threads(LAYOUT)) k_lm_step_wg(const LmArgs a) {
    constexpr int kWgRows = 3 * WB;                    // (shadows the 16-body constant)
    __shared__ __attribute__((aligned(16))) double C[wg_bufs(LAYOUT) * kWgRows * kRow];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool chain_wave = (tid >> 6) == kWgPairWaves;
    // the wave that does the integrator's work around the force; wave-uniform by construction, and told so (a scalar
    // branch keeps the history registers out of the other roles' live ranges)
    const bool tail_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == wg_tail_wave(LAYOUT);
    const int i0 = a.lo + blockIdx.x * WB;
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = tail_wave && lane < kWgRows && my_i < a.hi;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + (owner ? my_i : 0);

    // history of this lane's (body, component): loaded before the force loop and used after it (layout 6: loaded after)
    auto load_history = [&](double (&yv)[L], double (&av)[L]) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int slot = (a.cur + j) % L;
            yv[j] = a.Y[slot * lvl + off];
            av[j] = j > 0 ? a.A[slot * lvl + off] : 0.0;
        }
    };
    auto finish = [&](double (&yv)[L], double (&av)[L], double anew) {   // the step around the force, owner lanes only
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
    };
    if constexpr (wg_tail_wave(LAYOUT) != kWgPairWaves) {
        // the tail wave's whole life is this branch, so its history registers are live across this code only (through
        // wg_force's role switch the allocator would keep them alive in every role and spill)
        if (tail_wave) {
            double yv[L], av[L];
            load_history(yv, av);
            // (Forming everything that does not need the new acceleration here, ahead of the barriers -- the predictor's
            // position sum, the products of both acceleration sums, Cowell's difference quotient -- was built and measured:
            // 37.55 vs 37.2 us per step at N = 4096. The early arithmetic takes issue slots from the pair wave and the chain
            // wave of this SIMD when they are the critical path, and the tail's work was not on it; gpurun_out r03 A/B.)
            const int tiles = (a.n + kTile - 1) / kTile;
            if constexpr (wg_tile_split(WB)) {
                for (int t = 0; t <= tiles; ++t) __syncthreads();                             // one barrier per tile there
            } else {
                __syncthreads(); for (int T = 0; T < big_count(tiles); ++T) WG_LOOP_BARRIER();   // the idle wave's role: TB + 1 barriers
            }
            __syncthreads();                              // the chain wave's result is in LDS
            if (owner) finish(yv, av, C[lane]);
        } else {
            const double anew = wg_force<LAYOUT, WB>(a.pos_cur, a.n, i0, 0.0, C, tid, a.wg_flags);
            if (chain_wave && lane < kWgRows) C[lane] = anew;    // every tile buffer is dead after the loop's last barrier
            __syncthreads();
        }
    } else {
        double yv[L], av[L];
        if (tail_wave && LAYOUT != 6) load_history(yv, av);
        const long long t_entry = (a.wg_flags & 4) ? __builtin_readcyclecounter() : 0;
        const double anew = wg_force<LAYOUT, WB>(a.pos_cur, a.n, i0, 0.0, C, tid, a.wg_flags);
        const long long t_force = (a.wg_flags & 4) ? __builtin_readcyclecounter() : 0;
        if ((a.wg_flags & 4) && chain_wave && lane == 0) {       // tuning: where a launch spends its time (EPH_DEBUG_WG=4)
            if (blockIdx.x == 7) { g_wg_cycles[0] = t_force - t_entry; }
            if (blockIdx.x < 1024) { g_wg_span[0][blockIdx.x] = t_entry; g_wg_span[1][blockIdx.x] = t_force; }
        }
        if (!owner) return;
        if constexpr (LAYOUT == 6) load_history(yv, av);
        finish(yv, av, anew);
    }
}

// ------------------------------------------------------------------------------------------------------
// OPT-IN FAST PATH (eph_nbody_set_path(.., EPH_PATH_FAST)): the same pair arithmetic (IEEE sqrt / divide, no
// contraction), but NOT the reference's summation order -- SURVEY §7 "hard parts", north_star's "tile-parallel
// partial sums". It exists to measure what bit-exactness costs; the default path stays the ordered one.
//
// Work split: lane = target body (a block of 64 consecutive bodies per wave), every lane of a wave works on the SAME
// source body, fetched with scalar loads (s_load_dwordx8 through the constant address space: no LDS, no vector
// loads, no transposition in the loop). The sources are cut into S slices; wave (block, slice) accumulates its
// slice in source order into three registers per lane. A workgroup = 4 slices of one block (one wave per SIMD).
// The S partial sums of a body are combined in slice order
// by a second small launch (k_fast_finish), which also does the Cowell velocity, the solout sample and the
// predictor -- deterministic: the value never depends on which wave finishes first.
//   a_i = ((p_0 + p_1) + ... + p_{S-1}),  p_s = ((0 + c(i, j0)) + c(i, j0 + 1)) + ...   (j over slice s, j != i)
// ------------------------------------------------------------------------------------------------------
constexpr int kFastWaves = 4;                          // waves (= slices) per workgroup
constexpr int kFastMaxSlices = 64;

// 1 / r^3 WITHOUT the IEEE square root and division (EPH_PATH_FAST_RSQ): y = v_rsq_f64(n2) refined by two Newton steps
// (relative error ~1e-16, not correctly rounded), then y * y * y. 15 VALU operations instead of 22 and one
// transcendental instead of two. n2 = 0 (the body itself) gives NaN here too; the caller masks that source.
__device__ __forceinline__ double inv_r3_approx(double n2) {
    double y = __builtin_amdgcn_rsq(n2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double a = n2 * y;
        const double r = __builtin_fma(-a, 0.5 * y, 0.5);      // 0.5 * (1 - n2 * y^2)
        y = __builtin_fma(y, r, y);
    }
    return y * y * y;
}
template <bool DIAG, int kFastUnroll, bool APPROX>
__device__ __forceinline__ void fast_slice(const __attribute__((address_space(4))) Body4 *src, int j0, int j1, int n, int i,
                                           double xi, double yi, double zi, double &ax, double &ay, double &az) {
    auto fetch = [&](int j, Body4 (&p)[kFastUnroll]) {
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) { p[u].x = src[j + u].x; p[u].y = src[j + u].y; p[u].z = src[j + u].z; p[u].mu = src[j + u].mu; }
    };
    Body4 nxt[kFastUnroll];
    fetch(j0, nxt);
    for (int j = j0; j < j1; j += kFastUnroll) {       // j1 - j0 is a multiple of kFastUnroll; sources >= n are padding
        Body4 pj[kFastUnroll];
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) pj[u] = nxt[u];
        fetch(min(j + kFastUnroll, j1 - kFastUnroll), nxt);   // next group's scalar loads in flight under this one's arithmetic
        PairPre pre[kFastUnroll];
        unsigned worst = 0u;
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) {
            pre[u] = pair_pre(xi, yi, zi, pj[u]);
            worst = max(worst, range_key(pre[u].n2));
        }
        double c[3 * kFastUnroll];
        if (APPROX) {
#pragma unroll
            for (int u = 0; u < kFastUnroll; ++u) {
                const double sc = pj[u].mu * inv_r3_approx(pre[u].n2);
                c[3 * u] = pre[u].dx * sc; c[3 * u + 1] = pre[u].dy * sc; c[3 * u + 2] = pre[u].dz * sc;
            }
        } else if (__builtin_amdgcn_ballot_w64(worst >= kRangeSpan) == 0) {
#pragma unroll
            for (int u = 0; u < kFastUnroll; ++u) pair_finish<true>(pre[u], pj[u].mu, c[3 * u], c[3 * u + 1], c[3 * u + 2]);
        } else {
#pragma unroll
            for (int u = 0; u < kFastUnroll; ++u) pair_finish<false>(pre[u], pj[u].mu, c[3 * u], c[3 * u + 1], c[3 * u + 2]);
        }
#pragma unroll
        for (int u = 0; u < kFastUnroll; ++u) {
            if (DIAG && j + u == i) continue;          // the body itself (n2 = 0 -> NaN): not a source
            // padding rows are zeros at the ORIGIN with mu = 0: a real body sitting exactly there (the central body of a
            // heliocentric system) would get n2 = 0 -> 0 * inf = NaN from them. Wave-uniform test, last slice only.
            if (j + u >= n) continue;
            ax = ax + c[3 * u];
            ay = ay + c[3 * u + 1];
            az = az + c[3 * u + 2];
        }
    }
}

// partial: [S][3][npad] scratch. Two launches per step: the kernel boundary is the release/acquire between the
// slice sums and their combination. (First version: one launch with a per-block arrival ticket, the last workgroup
// of a block combining -- measured 66 / 96 / 166 us per step at 16 / 32 / 64 slices, N = 4096: the agent-scope
// fence each workgroup needs before its ticket costs ~0.13 us and they serialise; gpurun_out r02a.)
template <int UNROLL, bool APPROX>
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_partial(int n, int npad, const Body4 *__restrict__ pos,
                                                                  int S, int slice_len, double *__restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs_per_block = S / kFastWaves;
    const int block = blockIdx.x / wgs_per_block;
    const int slice = (blockIdx.x % wgs_per_block) * kFastWaves + wave;
    const int i = block * 64 + lane;
    const int ic = min(i, n - 1);
    const auto *src = (const __attribute__((address_space(4))) Body4 *)(unsigned long long)pos;
    const double xi = pos[ic].x, yi = pos[ic].y, zi = pos[ic].z;
    const int j0 = slice * slice_len, j1 = min(j0 + slice_len, npad);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j0 < j1) {
        if (j0 < block * 64 + 64 && j1 > block * 64) fast_slice<true, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        else fast_slice<false, UNROLL, APPROX>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
    }
    double *pp = partial + (size_t)slice * 3 * npad + i;
    pp[0] = ax;
    pp[(size_t)npad] = ay;
    pp[(size_t)2 * npad] = az;
}

// ------------------------------------------------------------------------------------------------------
// OPT-IN MIXED PRECISION (eph_nbody_set_path(.., EPH_PATH_F32_PAIRS); BASELINE.json configs[4] "65 536-body f32 system"):
// the pair arithmetic in binary32 -- differences of positions rounded to f32, n2 by fma, v_rsq_f32 + one Newton step,
// y^3, mu y^3, the three products -- two sources at a time in the packed f32 instructions (v_pk_add / v_pk_mul /
// v_pk_fma_f32: the only VALU form that runs at twice the f64 rate), every contribution then converted to f64 and
// ACCUMULATED in f64 in the fast path's slice order; Cowell, predictor and the whole integrator state stay f64 (a
// twelfth-order multistep recurrence cannot hold its state in binary32). The reference has no f32
// path (ephemeris/src/propagators/nbody.rs:13,19): no parity claim, never the default, for large systems only.
// ------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
struct BodyF { float x, y, z, mu; };
__global__ void __launch_bounds__(256) k_pos_to_f32(int n, int npad, const Body4 *__restrict__ pos, BodyF *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    BodyF b{0.f, 0.f, 0.f, 0.f};
    if (i < n) { const Body4 p = pos[i]; b = BodyF{(float)p.x, (float)p.y, (float)p.z, (float)p.mu}; }
    out[i] = b;
}
template <bool DIAG>
__device__ __forceinline__ void f32_slice(const __attribute__((address_space(4))) BodyF *src, int j0, int j1, int n, int i,
                                          float xi, float yi, float zi, double &ax, double &ay, double &az) {
    constexpr int U = 4;                               // sources per iteration: two packed pairs
    const v2f x2{xi, xi}, y2{yi, yi}, z2{zi, zi};
    for (int j = j0; j < j1; j += U) {                 // j1 - j0 is a multiple of U; sources >= n are padding
        BodyF p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { p[u].x = src[j + u].x; p[u].y = src[j + u].y; p[u].z = src[j + u].z; p[u].mu = src[j + u].mu; }
        v2f cx[2], cy[2], cz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const BodyF &pa = p[2 * h], &pb = p[2 * h + 1];
            const v2f dx = v2f{pa.x, pb.x} - x2, dy = v2f{pa.y, pb.y} - y2, dz = v2f{pa.z, pb.z} - z2;
            v2f n2 = dx * dx;
            n2 = __builtin_elementwise_fma(dy, dy, n2);
            n2 = __builtin_elementwise_fma(dz, dz, n2);
            v2f y{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)};
            const v2f hn = n2 * v2f{0.5f, 0.5f};
            const v2f r = __builtin_elementwise_fma(-(hn * y), y, v2f{0.5f, 0.5f});   // 0.5 (1 - n2 y^2)
            y = __builtin_elementwise_fma(y, r, y);
            const v2f sc = v2f{pa.mu, pb.mu} * (y * y * y);
            cx[h] = dx * sc; cy[h] = dy * sc; cz[h] = dz * sc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (DIAG && j + u == i) continue;          // the body itself (n2 = 0 -> NaN): not a source
            if (j + u >= n) continue;                  // padding rows
            ax = ax + (double)cx[u >> 1][u & 1];
            ay = ay + (double)cy[u >> 1][u & 1];
            az = az + (double)cz[u >> 1][u & 1];
        }
    }
}
__global__ void __launch_bounds__(64 * kFastWaves) k_fast_partial_f32(int n, int npad, const BodyF *__restrict__ posf, int S,
                                                                      int slice_len, double *__restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgs_per_block = S / kFastWaves;
    const int block = blockIdx.x / wgs_per_block;
    const int slice = (blockIdx.x % wgs_per_block) * kFastWaves + wave;
    const int i = block * 64 + lane;
    const int ic = min(i, n - 1);
    const auto *src = (const __attribute__((address_space(4))) BodyF *)(unsigned long long)posf;
    const float xi = posf[ic].x, yi = posf[ic].y, zi = posf[ic].z;
    const int j0 = slice * slice_len, j1 = min(j0 + slice_len, npad);
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j0 < j1) {
        if (j0 < block * 64 + 64 && j1 > block * 64) f32_slice<true>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
        else f32_slice<false>(src, j0, j1, n, i, xi, yi, zi, ax, ay, az);
    }
    double *pp = partial + (size_t)slice * 3 * npad + i;
    pp[0] = ax;
    pp[(size_t)npad] = ay;
    pp[(size_t)2 * npad] = az;
}

// thread per (component, body): partial sums combined in slice order, then the rest of the fused step
template <int L>
__global__ void __launch_bounds__(256) k_fast_finish(const LmArgs a, int S, const double *__restrict__ partial) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * a.npad) return;
    const int cc = t / a.npad, my_i = t % a.npad;               // consecutive threads = consecutive bodies: coalesced
    if (my_i >= a.n) return;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + my_i;
    double yv[L], av[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = a.Y[slot * lvl + off];
        av[j] = j > 0 ? a.A[slot * lvl + off] : 0.0;
    }
    // all loads of a group of 16 slices in flight before the ordered adds (one load per add would pay the memory
    // latency S times: measured 11 us for this kernel at S = 32)
    double anew = 0.0;
    for (int base = 0; base < S; base += 16) {
        double pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = base + u < S ? partial[(size_t)(base + u) * lvl + off] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (base + u < S) anew = anew + pv[u];
    }
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

// k_lm_predict: the predictor alone (first step of a batch): thread per (component, body)
template <int L>
__global__ void __launch_bounds__(256) k_lm_predict(const LmArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * (a.hi - a.lo)) return;
    const int my_i = a.lo + t / 3, cc = t % 3;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + my_i;
    double yv[L], av[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = a.Y[slot * lvl + off];
        av[j] = a.A[slot * lvl + off];
    }
    const double ynext = lm_predict<L>(yv, av, a.wa, a.wb, a.hh);
    const int nslot = (a.cur + L - 1) % L;
    a.Y[(size_t)nslot * lvl + off] = ynext;
    reinterpret_cast<double *>(a.pos_next + my_i)[cc] = ynext;
}

// ------------------------------------------------------------------------------------------------------
// k_lm_persistent: n <= 64 (one tile). The whole system lives in one workgroup's LDS and registers and the
// kernel runs `nsteps` integrator steps per launch (a 32-body step is ~1e3 pair interactions: launch latency
// would dominate a per-step launch). 8 waves (two per SIMD, 256 VGPRs each); wave w owns bodies w*BPW..; lane ch of that wave owns the
// (body, component) chain ch for the force, the velocity, the history ring and the predictor, so the only data
// shared between threads are the packed positions sP (two barriers per step).
// On entry slot `cur` is a COMPLETE level (Y, A, V); on exit slot (cur - nsteps) mod L is.
// ------------------------------------------------------------------------------------------------------
template <int BPW, int L>
__global__ void __launch_bounds__(512) k_lm_persistent(const LmArgs a, long long nsteps) {
    constexpr int kWaves = 8;
    __shared__ __attribute__((aligned(16))) double C[kWaves][3 * BPW * kRow];
    __shared__ __attribute__((aligned(32))) Body4 sP[kTile];
    __shared__ double ringY[L][3 * kTile];   // [slot][body*3 + comp]
    __shared__ double ringA[L][3 * kTile];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = w * BPW;
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = lane < 3 * BPW && my_i < a.n;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + (owner ? my_i : 0);
    const int ro = owner ? my_i * 3 + cc : 0;

    if (tid < kTile) sP[tid] = a.pos_cur[tid < a.n ? tid : a.n - 1];
    if (owner) {
#pragma unroll
        for (int s = 0; s < L; ++s) {
            ringY[s][ro] = a.Y[s * lvl + off];
            ringA[s][ro] = a.A[s * lvl + off];
        }
    }
    double v = owner ? a.V[off] : 0.0;
    int cur = a.cur;
    __syncthreads();

    for (long long s = 1; s <= nsteps; ++s) {
        double yv[L], av[L];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int slot = (cur + j) % L;
            yv[j] = ringY[slot][ro];
            av[j] = ringA[slot][ro];
        }
        const double ynew = lm_predict<L>(yv, av, a.wa, a.wb, a.hh);
        const int nslot = (cur + L - 1) % L;
        if (owner) {
            ringY[nslot][ro] = ynew;
            reinterpret_cast<double *>(&sP[my_i])[cc] = ynew;
        }
        __syncthreads();   // new positions visible to every wave
        const double anew = wave_force<BPW, true>(sP, a.n, i0, 0.0, C[w], lane);
        if (owner) {
            ringA[nslot][ro] = anew;
            v = lm_cowell<L>(anew, av, ynew, yv[0], a.cw, a.h, a.hc);
            maybe_sample(a.samp, my_i, cc, (uint32_t)s, ynew);
        }
        cur = nslot;
        __syncthreads();   // every wave done reading sP before the next predictor overwrites it
    }

    if (owner) {
#pragma unroll
        for (int s = 0; s < L; ++s) {
            a.Y[s * lvl + off] = ringY[s][ro];
            a.A[s * lvl + off] = ringA[s][ro];
        }
        a.V[off] = v;
        // leave both packed buffers consistent with the newest level
        reinterpret_cast<double *>(a.pos_next + my_i)[cc] = ringY[cur][ro];
        reinterpret_cast<double *>(const_cast<Body4 *>(a.pos_cur) + my_i)[cc] = ringY[cur][ro];
    }
}

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
constexpr int kSmallMaxN = 32;           // bodies (the reference's shipped system has exactly 32); 33..64 -> k_lm_persistent
constexpr int kSmallRow = 32 + 2;        // doubles per row: 16-byte aligned rows an odd number of 16-byte units apart
constexpr int kSmallRows = 3 * kSmallMaxN;
// -DEPH_SMALL_ACCOUNT=1 (tuning build, scripts/build_exp.sh): thread 0 of k_lm_small accumulates shader-clock ticks per phase
// of a step into g_wg_cycles[3..7] (wait at barrier A | sum1 + pair | wait at barrier B | row sums | sum2 + hand-over)
#ifndef EPH_SMALL_ACCOUNT
#define EPH_SMALL_ACCOUNT 0
#endif
// 1: ONE copy of the step, the history shifted through the registers every step (24 v_mov_b64) instead of twelve copies of
// the step, one per ring rotation (45 KB of code; two CUs share an instruction cache)
#ifndef EPH_SMALL_ROLLED
#define EPH_SMALL_ROLLED 0
#endif
#define SMALL_TICK(k) do { if constexpr (EPH_SMALL_ACCOUNT) { const long long now_ = (long long)__builtin_readcyclecounter(); acct[k] += now_ - acct_t; acct_t = now_; } } while (0)
// the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2] on both halves of the double)
__device__ __forceinline__ double dpp_xor1(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
    const int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    return __hiloint2double(h2, l2);
}
// A workgroup-wide barrier for LDS hand-offs only: __syncthreads() also waits for every outstanding GLOBAL access
// (vmcnt(0)) -- the solout's sample stores would stall all eight waves for a memory round trip at every sampled step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
template <int L, bool MULTI>
__global__ void __launch_bounds__(512) k_lm_small(const LmArgs a0, const LmArgs *__restrict__ argv, long long nsteps) {
    __shared__ __attribute__((aligned(16))) double U[kSmallRows][kSmallRow];    // [body*3 + comp][source]: sources after the body
    __shared__ __attribute__((aligned(16))) double Lw[kSmallRows][kSmallRow];   //                        sources before the body
    __shared__ __attribute__((aligned(32))) Body4 sP[kTile];
    const LmArgs &a = MULTI ? argv[blockIdx.x] : a0;

    const int tid = threadIdx.x, n = a.n;
    // chain threads: tid = (body*3 + comp)*2 + half   (half 0 = sources before, 1 = sources after)
    const int chain = tid >> 1, half = tid & 1;
    const bool chain_thread = chain < 3 * n;
    const bool owner = chain_thread && half == 0;     // the (body, comp) thread: history, velocity, predictor
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
    // this thread's unordered pair (i < j), row-major over the strict upper triangle (n <= 32: at most 496 pairs for 512
    // threads), decoded once. A thread without one runs pair (0, 1) again and stores nothing: straight-line code.
    int pi0 = 0, pj0 = 1;
    const bool live0 = tid < npairs && !(wg_flags & 1);    // (wg_flags: tuning switches, EPH_DEBUG_SMALL; 0 in normal runs)
    if (tid < npairs) {
        int i = 0;
        while ((i + 1) * (2 * n - i - 2) / 2 <= tid) ++i;
        pi0 = i;
        pj0 = i + 1 + (tid - i * (2 * n - i - 1) / 2);
    }
    __syncthreads();

    // Every thread runs the pair arithmetic: straight-line code, so the scheduler interleaves it with the predictor's
    // position chain (sum1) instead of executing one exec-masked region after the other. The wrapper-free sqrt /
    // reciprocal sequences run first and unconditionally; the range test that validates them (device_math.h) is decided
    // behind them, where the branch no longer stalls the wave, and an out-of-range operand anywhere in the wave redoes
    // the term in the full IEEE form.
    auto pair = [&]() {
        const double4 vi = *reinterpret_cast<const double4 *>(&sP[pi0]), vj = *reinterpret_cast<const double4 *>(&sP[pj0]);
        const double dx = vj.x - vi.x, dy = vj.y - vi.y, dz = vj.z - vi.z;
        const double n2 = dx * dx + dy * dy + dz * dz;
        double ax, ay, az, bx, by, bz;
        {
            const PairDen den = pair_den<true>(n2);
            pair_apply<true>(den, dx, dy, dz, vj.w, ax, ay, az);
            pair_apply<true>(den, -dx, -dy, -dz, vi.w, bx, by, bz);
        }
        if (__builtin_amdgcn_ballot_w64(!in_range(n2)) != 0) {
            const PairDen den = pair_den<false>(n2);
            pair_apply<false>(den, dx, dy, dz, vj.w, ax, ay, az);
            pair_apply<false>(den, -dx, -dy, -dz, vi.w, bx, by, bz);
        }
        if (live0) {
            U[pi0 * 3 + 0][pj0] = ax;
            U[pi0 * 3 + 1][pj0] = ay;
            U[pi0 * 3 + 2][pj0] = az;
            Lw[pj0 * 3 + 0][pi0] = bx;
            Lw[pj0 * 3 + 1][pi0] = by;
            Lw[pj0 * 3 + 2][pi0] = bz;
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
        double p1[L], q2[L];
        p1[0] = ynew * wa[0];
#pragma unroll
        for (int j = 1; j < L; ++j) { p1[j] = yv[(R + j - 1) % L] * wa[j]; q2[j] = av[(R + j - 1) % L] * wb[j]; }
        __builtin_amdgcn_sched_barrier(0);
        double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < L; ++j) s1 = s1 + p1[j];
        // ---- pairs (i < j): one reciprocal cube per unordered pair, both directed contributions
        pair();
        asm volatile("" : "+v"(s1));                   // computed HERE (the compiler would sink it behind the force)
#pragma unroll
        for (int j = 1; j < L; ++j) asm volatile("" : "+v"(q2[j]));
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
            if constexpr (EPH_SMALL_ROLLED) {              // one copy of the step: the history moves through the registers
#pragma unroll
                for (int j = L - 1; j > 0; --j) { yv[j] = yv[j - 1]; av[j] = av[j - 1]; }
                yv[0] = ynew;
                av[0] = anew;
            } else {
                yv[Rn] = ynew;
                av[Rn] = anew;
            }
            ynew = ynext;
        }
        if constexpr (EPH_SMALL_ACCOUNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SMALL_TICK(4);
        // the predictor writes sP after every pair thread of this step passed the barrier above; U / Lw are
        // rewritten only after the next "positions visible" barrier
    };
    int rot = 0;                                       // rotation after the steps taken so far
    if ((wg_flags & 4) && tid == 0 && blockIdx.x < 1024) {   // tuning: where the dispatcher put this workgroup (HW_ID, XCC_ID)
        g_wg_span[0][blockIdx.x] = (long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        g_wg_span[1][blockIdx.x] = (long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11)) + 1;
        g_wg_span[2][blockIdx.x] = (long long)wall_clock64();
    }
    const long long dbg_c0 = (wg_flags & 4) ? (long long)__builtin_readcyclecounter() : 0;
    const long long dbg_w0 = (wg_flags & 4) ? (long long)wall_clock64() : 0;
    if constexpr (EPH_SMALL_ROLLED) {
        for (long long s = 1; s <= nsteps; ++s) step(std::integral_constant<int, 0>{}, s);
    } else {
        long long s = 1;
        bool more = nsteps >= 1;
        while (more) small_steps(step, s, nsteps, rot, more, std::make_integer_sequence<int, L>{});
    }
    if ((wg_flags & 4) && tid == 0 && blockIdx.x < 1024) g_wg_span[3][blockIdx.x] = (long long)wall_clock64();
    if ((wg_flags & 4) && tid == 0 && blockIdx.x == 0) {   // tuning (EPH_DEBUG_SMALL=4): shader-clock ticks, 100 MHz ticks, steps
        g_wg_cycles[0] = (long long)__builtin_readcyclecounter() - dbg_c0;
        g_wg_cycles[1] = (long long)wall_clock64() - dbg_w0;
        g_wg_cycles[2] = nsteps;
        if constexpr (EPH_SMALL_ACCOUNT)
            for (int q = 0; q < 5; ++q) g_wg_cycles[3 + q] = acct[q];
    }

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

// ------------------------------------------------------------------------------------------------------
// small element-wise kernels (start-up path, staging)
// ------------------------------------------------------------------------------------------------------
__global__ void k_pack(int n, int npad, const double *__restrict__ Y, const double *__restrict__ mu, Body4 *pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Body4 p;
    p.x = Y[i];
    p.y = Y[npad + i];
    p.z = Y[2 * (size_t)npad + i];
    p.mu = mu[i];
    pos[i] = p;
}
__global__ void k_copy3(int n, int npad, const double *__restrict__ src, double *__restrict__ dst) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * n) return;
    const size_t o = (size_t)(t / n) * npad + (t % n);
    dst[o] = src[o];
}
// SRKN stage: *dy = *dy + *ddy * (h * B[s]); *y = *y + *dy * (h * A[s])   symplectic.rs:90-97
__global__ void k_kick_drift(int n, int npad, const double *__restrict__ acc, double *v, double *y, double hb,
                             double ha, const double *__restrict__ mu, Body4 *pos_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t o = (size_t)c * npad + i;
        const double vn = v[o] + acc[o] * hb;
        v[o] = vn;
        r[c] = y[o] + vn * ha;
        y[o] = r[c];
    }
    Body4 p;
    p.x = r[0]; p.y = r[1]; p.z = r[2]; p.mu = mu[i];
    pos_out[i] = p;
}
// solout sample of the newest level for the regimes that do not run the fused kernel (start-up, SRKN methods)
__global__ void k_sample(int n, int npad, const double *__restrict__ Y, SampleArgs sa, uint32_t step) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * n) return;
    const int b = t / 3, c = t % 3;
    maybe_sample(sa, b, c, step, Y[(size_t)c * npad + b]);
}
// after the fits: move the samples of the unfinished window of every body to the front of its log region
__global__ void k_carry(int n, const uint64_t *__restrict__ region, const uint32_t *__restrict__ src,
                        const uint32_t *__restrict__ cnt, double *log) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n || src[b] == 0) return;
    double *base = log + region[b] * 3;
    for (uint32_t k = 0; k < cnt[b] * 3; ++k) base[k] = base[(size_t)src[b] * 3 + k];
}
// sharded propagator: window q of this rank's fit -> exchange record [24 coefficients, ncoef] of its slice
__global__ void k_pack_records(long long nwin, const double *__restrict__ co, const int32_t *__restrict__ nc,
                               double *__restrict__ rec) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int R = kDiv * 3 + 1;
    if (t >= nwin * R) return;
    const long long q = t / R;
    const int k = (int)(t % R);
    rec[t] = k < kDiv * 3 ? co[q * kDiv * 3 + k] : (double)nc[q];
}
__global__ void k_aos_to_soa(int n, int npad, const double *__restrict__ aos, double *__restrict__ soa) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * n) return;
    soa[(size_t)(t % 3) * npad + t / 3] = aos[t];
}
__global__ void k_soa_to_aos(int n, int npad, const double *__restrict__ soa, double *__restrict__ aos) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * n) return;
    aos[t] = soa[(size_t)(t % 3) * npad + t / 3];
}

// ------------------------------------------------------------------------------------------------------
// LeastSquaresFit::interpolate  ephemeris_explorer/src/dynamics/celestial.rs:24-135 (Forsythe recurrence,
// unit weights) on 9 samples at tau_k = k/8 (Forward) or 1 - k/8 (Backward), nbody.rs:422-442.
// The reference carries gamma, b, c and the basis polynomials as DVec3 with three identical components;
// scalars here, same operations. Thread per window, all three components.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_lsq_fit(long long nwin, const uint64_t *__restrict__ first,
                                                const uint8_t *__restrict__ degree_of, int backward,
                                                const double *__restrict__ log, double *__restrict__ coeffs,
                                                int32_t *__restrict__ ncoef) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    constexpr int M = kDiv + 1;
    double ts[M], xs[M][3];
    const double *src = log + first[w] * 3;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        ts[k] = backward ? 1.0 - (double)k / (double)kDiv : (double)k / (double)kDiv;
        xs[k][0] = src[k * 3 + 0];
        xs[k][1] = src[k * 3 + 1];
        xs[k][2] = src[k * 3 + 2];
    }
    int degree = degree_of[w];
    degree = degree < M - 1 ? degree : M - 1;
    if (degree > kDiv - 1) degree = kDiv - 1;   // Polynomial storage is 8 coefficients (host rejects degree > 7)

    double d0[3] = {0.0, 0.0, 0.0}, gamma0 = 0.0, b0 = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        d0[0] += xs[k][0]; d0[1] += xs[k][1]; d0[2] += xs[k][2];
        gamma0 += 1.0;
        b0 += ts[k];
    }
    b0 /= gamma0;
    d0[0] /= gamma0; d0[1] /= gamma0; d0[2] /= gamma0;

    double pd[kDiv][3];
    double pa[kDiv + 1], pb[kDiv + 1];
#pragma unroll
    for (int i = 0; i < kDiv; ++i) { pd[i][0] = pd[i][1] = pd[i][2] = 0.0; }
#pragma unroll
    for (int i = 0; i <= kDiv; ++i) { pa[i] = 0.0; pb[i] = 0.0; }
    pd[0][0] = d0[0]; pd[0][1] = d0[1]; pd[0][2] = d0[2];
    int nco = 1;
    if (degree > 0) {
        nco = degree + 1;
        double *p_km1 = pa, *p_k = pb;
        p_k[0] = 1.0;
        double gamma_k = gamma0, b_k = b0, minus_c_k = 0.0;
        int kp1 = 1;
        for (;;) {
            for (int i = 0; i < kp1; ++i) p_km1[i] = minus_c_k * p_km1[i] - b_k * p_k[i];
            for (int i = 0; i < kp1; ++i) p_km1[i + 1] += p_k[i];
            double d[3] = {0.0, 0.0, 0.0}, g = 0.0, bs = 0.0;
            for (int k = 0; k < M; ++k) {
                double px = 0.0;
                for (int c = kp1; c >= 0; --c) px = px * ts[k] + p_km1[c];
                d[0] += xs[k][0] * px; d[1] += xs[k][1] * px; d[2] += xs[k][2] * px;
                const double pp = px * px;
                g += pp;
                bs += ts[k] * pp;
            }
            if (g == 0.0) break;
            d[0] /= g; d[1] /= g; d[2] /= g;
            for (int i = 0; i < kp1 + 1; ++i) {
                pd[i][0] += d[0] * p_km1[i]; pd[i][1] += d[1] * p_km1[i]; pd[i][2] += d[2] * p_km1[i];
            }
            if (kp1 == degree) break;
            bs /= g;
            kp1 += 1;
            b_k = bs;
            minus_c_k = -(g / gamma_k);
            gamma_k = g;
            double *t = p_k; p_k = p_km1; p_km1 = t;
        }
    }
    // Polynomial::trim  ephemeris/src/trajectory.rs:387-395 (not applied on the degree == 0 early return)
    if (degree > 0)
        while (nco > 0 && pd[nco - 1][0] == 0.0 && pd[nco - 1][1] == 0.0 && pd[nco - 1][2] == 0.0) --nco;
    double *dst = coeffs + w * kDiv * 3;
    for (int i = 0; i < kDiv; ++i) {
        const bool keep = i < nco;
        dst[i * 3 + 0] = keep ? pd[i][0] : 0.0;
        dst[i * 3 + 1] = keep ? pd[i][1] : 0.0;
        dst[i * 3 + 2] = keep ? pd[i][2] : 0.0;
    }
    ncoef[w] = nco;
}

// ------------------------------------------------------------------------------------------------------
// UniformSpline::state_vector  ephemeris/src/trajectory.rs:459-470 (get_polynomial :551-561,
// get_index_local_exclusive :600-607, index_local_exclusive :614-617, eval_and_deriv :368-385)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_spline_eval(long long m, const double *__restrict__ at, double start,
                                                     double interval, long long npoly,
                                                     const double *__restrict__ coeffs,
                                                     const int32_t *__restrict__ ncoef, double *__restrict__ pos,
                                                     double *__restrict__ vel, uint8_t *__restrict__ inside) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const double local = at[q] - start;
    const double span = interval * (double)npoly;            // Duration::scaled
    bool ok = !(__builtin_signbit(local) || local > span);   // is_negative() is the sign bit
    unsigned long long idx = 0;
    if (ok) {
        const double c = ceil(local / interval);
        const unsigned long long ci = c <= 0.0 ? 0ull : (c >= 18446744073709551616.0 ? ~0ull : (unsigned long long)c);
        idx = ci == 0 ? 0 : ci - 1;                          // saturating_sub(1)
        ok = idx < (unsigned long long)npoly;
    }
    inside[q] = ok ? 1 : 0;
    if (!ok) {
        for (int c = 0; c < 3; ++c) { pos[q * 3 + c] = 0.0; if (vel) vel[q * 3 + c] = 0.0; }
        return;
    }
    const double tau = (local - interval * (double)idx) / interval;
    const double *co = coeffs + idx * kDiv * 3;
    const int nc = ncoef[idx];
    for (int c = 0; c < 3; ++c) {
        if (vel) {
            const double first = nc ? co[c] : 0.0;
            const double last = nc ? co[(nc - 1) * 3 + c] : 0.0;
            double e = last, d = last;
            for (int k = nc - 2; k >= 1; --k) {
                e = e * tau + co[k * 3 + c];
                d = d * tau + e;
            }
            e = e * tau + first;
            pos[q * 3 + c] = e;
            vel[q * 3 + c] = d / interval;
        } else {
            double r = 0.0;                                   // eval_slice_horner :398-410
            for (int k = nc - 1; k >= 0; --k) r = r * tau + co[k * 3 + c];
            pos[q * 3 + c] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
static int done(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(what, e);
        return EPH_ERR_HIP;
    }
    return EPH_OK;
}

// bodies per wave: enough waves to cover the 1024 SIMDs of the chip, as many bodies per wave as that allows
// (phase B's ordered adds cost the same for 1 or 21 chains, so more bodies per wave is cheaper per body)
int lm_bodies_per_wave(int n) {
    static const int forced = [] {
        const char *e = getenv("EPH_BPW");             // tuning override (1, 2, 4, 8)
        return e ? atoi(e) : 0;
    }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    if (n >= 8 * 1024) return 8;
    if (n >= 4 * 1024) return 4;
    if (n >= 2 * 1024) return 2;
    return 1;
}

// workgroup kernel tuning knobs (read once): EPH_WG_LAYOUT = 0 | 1 (role layout), EPH_DEBUG_WG bit mask
// (1 no chain, 2 no pair work, 4 cycle accounting into g_wg_cycles, 8 chain wave at raised priority)
static int wg_layout() {
    static const int v = [] { const char *e = getenv("EPH_WG_LAYOUT"); return e ? atoi(e) : kWgDefaultLayout; }();
    return v >= 1 && v <= 6 ? v : 0;
}
// bodies per workgroup of the default layout: 16, or 8 / 4 when 16-body workgroups would leave CUs idle (one workgroup
// per CU either way, and the chain wave's time per tile does not depend on how many chains it carries)
static int wg_bodies(int nt) {
    static const int forced = [] { const char *e = getenv("EPH_WG_BODIES"); return e ? atoi(e) : 0; }();
    if (wg_layout() != 5) return kWgBodies;
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    return nt <= 1024 ? 4 : nt <= 2048 ? 8 : kWgBodies;
}
static int wg_debug_flags() {
    static const int v = [] { const char *e = getenv("EPH_DEBUG_WG"); return e ? atoi(e) : 0; }();
    return v;
}
// which per-step force kernel: 1 = one wave per block (wave_force), 2 = workgroup-specialised (wg_force)
int force_kernel_kind(int n, int requested) {
    if (requested == 1 || requested == 2) return requested;
    static const int forced = [] {
        const char *e = getenv("EPH_FORCE");           // tuning override: "wave" | "wg"
        if (!e) return 0;
        return e[1] == 'a' ? 1 : 2;
    }();
    if (forced) return forced;
    // measured on MI355X (us per step, QT12; wg = layout 5 with 4 / 8 / 16 bodies per workgroup | wave): n=512 9.0 | 9.05,
    // 640 9.7 | 10.7, 768 10.5 | 11.7, 896 11.2 | 12.9, 1024 11.9 | 14.2, 1536 15.4 | 22.2, 2048 18.3 | 25.1, 4096 36.9 | 54, 8192 134 | 143, 12288 292 | 412,
    // 16384 511 | 552, 32768 1995 | 2174, 65536 7864 | 8823 (gpurun_out r02s, r02t, r02ab). (With round 2's first
    // workgroup layout the wave form still won outside 2048 <= n < 8192.)
    return n > 512 ? 2 : 1;
}

int launch_accel(hipStream_t s, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out, int kind,
                 int lo, int hi, const KickDrift *kdp) {
    if (hi < 0) hi = n;
    const KickDrift kd = kdp ? *kdp : KickDrift{nullptr, nullptr, 0.0, 0.0, nullptr};
    const int nt = hi - lo;                            // targets of this launch; the kernel choice follows them
    if (n <= 0 || nt <= 0) return EPH_OK;
    if (force_kernel_kind(nt, kind) == 2) {
        const int dbg = wg_debug_flags();
        const int wb = wg_bodies(nt);
        const dim3 grid((nt + wb - 1) / wb);
        const int lay = wg_layout();
        if (wb == 8)
            hipLaunchKernelGGL((k_accel_wg<5, 8>), grid, dim3(wg_threads(5)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (wb == 4)
            hipLaunchKernelGGL((k_accel_wg<5, 4>), grid, dim3(wg_threads(5)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (lay == 6)
            hipLaunchKernelGGL(k_accel_wg<6>, grid, dim3(wg_threads(6)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (lay == 5)
            hipLaunchKernelGGL(k_accel_wg<5>, grid, dim3(wg_threads(5)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (lay == 4)
            hipLaunchKernelGGL(k_accel_wg<4>, grid, dim3(wg_threads(4)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (lay == 3)
            hipLaunchKernelGGL(k_accel_wg<3>, grid, dim3(wg_threads(3)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (lay == 2)
            hipLaunchKernelGGL(k_accel_wg<2>, grid, dim3(wg_threads(2)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else if (lay == 1)
            hipLaunchKernelGGL(k_accel_wg<1>, grid, dim3(wg_threads(1)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        else
            hipLaunchKernelGGL(k_accel_wg<0>, grid, dim3(wg_threads(0)), 0, s, n, npad, pos, acc_init, acc_out, dbg, lo, hi, kd);
        return done("k_accel_wg");
    }
    const int bpw = lm_bodies_per_wave(nt);
    const dim3 grid((nt + bpw - 1) / bpw), block(64);
    switch (bpw) {
        case 1: hipLaunchKernelGGL(k_accel<1>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd); break;
        case 2: hipLaunchKernelGGL(k_accel<2>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd); break;
        case 4: hipLaunchKernelGGL(k_accel<4>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd); break;
        default: hipLaunchKernelGGL(k_accel<8>, grid, block, 0, s, n, npad, pos, acc_init, acc_out, lo, hi, kd); break;
    }
    return done("k_accel");
}

template <int L>
static int launch_lm_step_L(hipStream_t s, const LmArgs &a) {
    const int nt = a.hi - a.lo;
    const int bpw = lm_bodies_per_wave(nt);
    const dim3 grid((nt + bpw - 1) / bpw), block(64);
    switch (bpw) {
        case 1: hipLaunchKernelGGL((k_lm_step<1, L>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lm_step<2, L>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lm_step<4, L>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_lm_step<8, L>), grid, block, 0, s, a); break;
    }
    return done("k_lm_step");
}
int launch_lm_step(hipStream_t s, const LmArgs &a) {
    if (a.n <= 0 || a.hi <= a.lo) return EPH_OK;
    if (force_kernel_kind(a.hi - a.lo, a.kind) == 2) {
        const int wb = wg_bodies(a.hi - a.lo);
        const dim3 grid((a.hi - a.lo + wb - 1) / wb);
        LmArgs b = a;
        b.wg_flags = wg_debug_flags() & 12;            // priority knob + launch-level accounting (the per-tile one is k_accel_wg's)
        const int lay = wg_layout();
        if (a.L == 12 && wb == 8) hipLaunchKernelGGL((k_lm_step_wg<12, 5, 8>), grid, dim3(wg_threads(5)), 0, s, b);
        else if (a.L == 13 && wb == 8) hipLaunchKernelGGL((k_lm_step_wg<13, 5, 8>), grid, dim3(wg_threads(5)), 0, s, b);
        else if (a.L == 12 && wb == 4) hipLaunchKernelGGL((k_lm_step_wg<12, 5, 4>), grid, dim3(wg_threads(5)), 0, s, b);
        else if (a.L == 13 && wb == 4) hipLaunchKernelGGL((k_lm_step_wg<13, 5, 4>), grid, dim3(wg_threads(5)), 0, s, b);
        else if (a.L == 12 && lay == 6) hipLaunchKernelGGL((k_lm_step_wg<12, 6>), grid, dim3(wg_threads(6)), 0, s, b);
        else if (a.L == 13 && lay == 6) hipLaunchKernelGGL((k_lm_step_wg<13, 6>), grid, dim3(wg_threads(6)), 0, s, b);
        else if (a.L == 12 && lay == 5) hipLaunchKernelGGL((k_lm_step_wg<12, 5>), grid, dim3(wg_threads(5)), 0, s, b);
        else if (a.L == 13 && lay == 5) hipLaunchKernelGGL((k_lm_step_wg<13, 5>), grid, dim3(wg_threads(5)), 0, s, b);
        else if (a.L == 12 && lay == 4) hipLaunchKernelGGL((k_lm_step_wg<12, 4>), grid, dim3(wg_threads(4)), 0, s, b);
        else if (a.L == 13 && lay == 4) hipLaunchKernelGGL((k_lm_step_wg<13, 4>), grid, dim3(wg_threads(4)), 0, s, b);
        else if (a.L == 12 && lay == 3) hipLaunchKernelGGL((k_lm_step_wg<12, 3>), grid, dim3(wg_threads(3)), 0, s, b);
        else if (a.L == 13 && lay == 3) hipLaunchKernelGGL((k_lm_step_wg<13, 3>), grid, dim3(wg_threads(3)), 0, s, b);
        else if (a.L == 12 && lay == 2) hipLaunchKernelGGL((k_lm_step_wg<12, 2>), grid, dim3(wg_threads(2)), 0, s, b);
        else if (a.L == 13 && lay == 2) hipLaunchKernelGGL((k_lm_step_wg<13, 2>), grid, dim3(wg_threads(2)), 0, s, b);
        else if (a.L == 12 && lay == 1) hipLaunchKernelGGL((k_lm_step_wg<12, 1>), grid, dim3(wg_threads(1)), 0, s, b);
        else if (a.L == 12) hipLaunchKernelGGL((k_lm_step_wg<12, 0>), grid, dim3(wg_threads(0)), 0, s, b);
        else if (a.L == 13 && lay == 1) hipLaunchKernelGGL((k_lm_step_wg<13, 1>), grid, dim3(wg_threads(1)), 0, s, b);
        else if (a.L == 13) hipLaunchKernelGGL((k_lm_step_wg<13, 0>), grid, dim3(wg_threads(0)), 0, s, b);
        else return EPH_ERR_UNSUPPORTED;
        return done("k_lm_step_wg");
    }
    if (a.L == 12) return launch_lm_step_L<12>(s, a);
    if (a.L == 13) return launch_lm_step_L<13>(s, a);
    return EPH_ERR_UNSUPPORTED;
}
// slices of the fast path: enough waves for two per SIMD (2048), a multiple of the workgroup's 4, at most 64
int fast_slices(int npad) {
    static const int forced = [] { const char *e = getenv("EPH_FAST_SLICES"); return e ? atoi(e) : 0; }();
    int S = forced > 0 ? forced : 2048 / (npad / 64);
    S = std::max(kFastWaves, std::min(kFastMaxSlices, S));
    return (S + kFastWaves - 1) / kFastWaves * kFastWaves;
}
int launch_lm_step_fast(hipStream_t s, const LmArgs &a, double *partial, bool approx, float *posf) {
    if (a.n <= 0) return EPH_OK;
    if (a.lo != 0 || a.hi != a.n) return EPH_ERR_UNSUPPORTED;          // the fast path is not sharded
    static const int unroll = [] { const char *e = getenv("EPH_FAST_UNROLL"); return e && atoi(e) == 8 ? 8 : 4; }();
    const int S = fast_slices(a.npad);
    int slice_len = (a.npad + S - 1) / S;
    const int un = approx ? 4 : unroll;
    slice_len = (slice_len + un - 1) / un * un;
    const dim3 pgrid((unsigned)(a.npad / 64 * (S / kFastWaves))), pblock(64 * kFastWaves);
    if (posf) {                                                         // EPH_PATH_F32_PAIRS
        BodyF *pf = reinterpret_cast<BodyF *>(posf);
        hipLaunchKernelGGL(k_pos_to_f32, dim3((unsigned)((a.npad + 255) / 256)), dim3(256), 0, s, a.n, a.npad, a.pos_cur, pf);
        hipLaunchKernelGGL(k_fast_partial_f32, pgrid, pblock, 0, s, a.n, a.npad, (const BodyF *)pf, S, slice_len, partial);
    } else if (approx)
        hipLaunchKernelGGL((k_fast_partial<4, true>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    else if (unroll == 8 && a.npad % 8 == 0)
        hipLaunchKernelGGL((k_fast_partial<8, false>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    else
        hipLaunchKernelGGL((k_fast_partial<4, false>), pgrid, pblock, 0, s, a.n, a.npad, a.pos_cur, S, slice_len, partial);
    const dim3 grid((3 * a.npad + 255) / 256), block(256);
    if (a.L == 12) hipLaunchKernelGGL(k_fast_finish<12>, grid, block, 0, s, a, S, partial);
    else if (a.L == 13) hipLaunchKernelGGL(k_fast_finish<13>, grid, block, 0, s, a, S, partial);
    else return EPH_ERR_UNSUPPORTED;
    return done("k_fast_partial / k_fast_finish");
}
int launch_lm_predict(hipStream_t s, const LmArgs &a) {
    if (a.n <= 0 || a.hi <= a.lo) return EPH_OK;
    const dim3 grid((3 * (a.hi - a.lo) + 255) / 256), block(256);
    if (a.L == 12) hipLaunchKernelGGL(k_lm_predict<12>, grid, block, 0, s, a);
    else if (a.L == 13) hipLaunchKernelGGL(k_lm_predict<13>, grid, block, 0, s, a);
    else return EPH_ERR_UNSUPPORTED;
    return done("k_lm_predict");
}
template <int L>
static int launch_lm_persistent_L(hipStream_t s, const LmArgs &a, int64_t nsteps) {
    static const int old_design = [] { const char *e = getenv("EPH_SMALL"); return e && e[0] == '1'; }();
    if (!old_design && a.n <= kSmallMaxN) {
        static const int dbg = [] { const char *e = getenv("EPH_DEBUG_SMALL"); return e ? atoi(e) : 0; }();
        LmArgs b = a;
        b.wg_flags = dbg;                              // 1: no pair stage, 2: no chain stage (timing breakdown only)
        hipLaunchKernelGGL((k_lm_small<L, false>), dim3(1), dim3(512), 0, s, b, (const LmArgs *)nullptr, (long long)nsteps);
        return done("k_lm_small");
    }
    const int per_wave = (a.n + 7) / 8;
    const dim3 grid(1), block(512);
    if (per_wave <= 1) hipLaunchKernelGGL((k_lm_persistent<1, L>), grid, block, 0, s, a, (long long)nsteps);
    else if (per_wave <= 2) hipLaunchKernelGGL((k_lm_persistent<2, L>), grid, block, 0, s, a, (long long)nsteps);
    else if (per_wave <= 4) hipLaunchKernelGGL((k_lm_persistent<4, L>), grid, block, 0, s, a, (long long)nsteps);
    else hipLaunchKernelGGL((k_lm_persistent<8, L>), grid, block, 0, s, a, (long long)nsteps);
    return done("k_lm_persistent");
}
int launch_lm_persistent(hipStream_t s, const LmArgs &a, int64_t nsteps) {
    if (a.n <= 0 || nsteps <= 0) return EPH_OK;
    if (a.n > kSmallN) return EPH_ERR_UNSUPPORTED;
    if (a.L == 12) return launch_lm_persistent_L<12>(s, a, nsteps);
    if (a.L == 13) return launch_lm_persistent_L<13>(s, a, nsteps);
    return EPH_ERR_UNSUPPORTED;
}

int launch_lm_small_many(hipStream_t s, const LmArgs *argv_dev, int count, int L, int64_t nsteps) {
    static_assert(kGangMaxN == kSmallMaxN, "the gang launch is k_lm_small's");
    if (count <= 0 || nsteps <= 0) return EPH_OK;
    const LmArgs none{};
    // EPH_SMALL_LDS_PAD (tuning): extra dynamic LDS per workgroup, so that two workgroups do not fit one CU
    static const unsigned pad = [] { const char *e = getenv("EPH_SMALL_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    if (L == 12) hipLaunchKernelGGL((k_lm_small<12, true>), dim3((unsigned)count), dim3(512), pad, s, none, argv_dev, (long long)nsteps);
    else if (L == 13) hipLaunchKernelGGL((k_lm_small<13, true>), dim3((unsigned)count), dim3(512), pad, s, none, argv_dev, (long long)nsteps);
    else return EPH_ERR_UNSUPPORTED;
    return done("k_lm_small (gang)");
}

int launch_pack(hipStream_t s, int n, int npad, const double *Yslot, const double *mu, Body4 *pos) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, s, n, npad, Yslot, mu, pos);
    return done("k_pack");
}
int launch_copy3(hipStream_t s, int n, int npad, const double *src, double *dst) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_copy3, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, src, dst);
    return done("k_copy3");
}
int launch_kick_drift(hipStream_t s, int n, int npad, const double *a, double *v, double *y, double hb, double ha,
                      const double *mu, Body4 *pos_out) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_kick_drift, dim3((n + 255) / 256), dim3(256), 0, s, n, npad, a, v, y, hb, ha, mu, pos_out);
    return done("k_kick_drift");
}
int launch_aos_to_soa(hipStream_t s, int n, int npad, const double *aos, double *soa) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, aos, soa);
    return done("k_aos_to_soa");
}
int launch_soa_to_aos(hipStream_t s, int n, int npad, const double *soa, double *aos) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, soa, aos);
    return done("k_soa_to_aos");
}
int debug_wg_cycles(long long *out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_cycles), sizeof(long long) * 8);
    if (e != hipSuccess) { set_last_error("hipMemcpyFromSymbol", e); return EPH_ERR_HIP; }
    // step kernel: out[2] = earliest workgroup entry, out[3] = latest force completion over the grid (s_memtime ticks)
    static long long span[4][1024];
    e = hipMemcpyFromSymbol(span, HIP_SYMBOL(g_wg_span), sizeof(span));
    if (e != hipSuccess) { set_last_error("hipMemcpyFromSymbol", e); return EPH_ERR_HIP; }
    if (getenv("EPH_DEBUG_PLACEMENT")) {               // k_lm_small gang: HW_ID / XCC_ID of every workgroup -> workgroups per CU
        int per_cu[8][128] = {};
        int wgs = 0, cus = 0, worst = 0;
        for (int b = 0; b < 1024; ++b) {
            if (span[1][b] == 0) continue;
            const unsigned hw = (unsigned)span[0][b], xcc = (unsigned)(span[1][b] - 1) & 7u;
            const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 3u;   // gfx9 HW_ID fields
            int &d = per_cu[xcc][(se << 5) | (sh << 4) | cu];
            if (d++ == 0) ++cus;
            worst = std::max(worst, d);
            ++wgs;
        }
        fprintf(stderr, "placement: %d workgroups on %d distinct (xcc, se, cu), at most %d on one\n", wgs, cus, worst);
        // per workgroup: start (relative to the earliest) and duration in 100 MHz ticks -> us; per XCC min / mean / max duration
        long long first = 0;
        for (int b = 0; b < 1024; ++b) if (span[1][b] && (first == 0 || span[2][b] < first)) first = span[2][b];
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
        if (getenv("EPH_DEBUG_PLACEMENT")[0] == '2')
            for (int b = 0; b < 1024; ++b)
                if (span[1][b]) fprintf(stderr, "  wg %4d xcc %d hw %05x start +%.0f us dur %.0f us\n", b, (int)((span[1][b] - 1) & 7), (unsigned)span[0][b] & 0xfffff,
                                        (double)(span[2][b] - first) / 100.0, (double)(span[3][b] - span[2][b]) / 100.0);
        std::memset(span, 0, sizeof(span));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_span), span, sizeof(span));
        return EPH_OK;
    }
    long long lo = 0, hi = 0;
    for (int b = 0; b < 1024; ++b) {
        if (span[0][b] == 0 || span[2][b] != 0) continue;   // (span[2] != 0: k_lm_small's placement words, not the step kernel's ticks)
        if (lo == 0 || span[0][b] < lo) lo = span[0][b];
        if (span[1][b] > hi) hi = span[1][b];
    }
    if (lo) { out[2] = lo; out[3] = hi; }
    return EPH_OK;
}
int launch_debug_inv_r3(hipStream_t s, int64_t n, const double *n2, double *fast, double *ieee) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_debug_inv_r3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (long long)n, n2, fast, ieee);
    return done("k_debug_inv_r3");
}
int launch_debug_inv_r3_sweep(hipStream_t s, uint64_t seed, int64_t n, unsigned long long *out2) {
    const int per = 4096;
    const int64_t threads = (n + per - 1) / per;
    if (threads <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_debug_inv_r3_sweep, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                       (unsigned long long)seed, per, out2);
    return done("k_debug_inv_r3_sweep");
}
int launch_sample(hipStream_t s, int n, int npad, const double *Yslot, const SampleArgs &sa, uint32_t step) {
    if (n <= 0 || !sa.period) return EPH_OK;
    hipLaunchKernelGGL(k_sample, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, Yslot, sa, step);
    return done("k_sample");
}
int launch_carry(hipStream_t s, int n, const uint64_t *region, const uint32_t *src, const uint32_t *cnt, double *log) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_carry, dim3((n + 255) / 256), dim3(256), 0, s, n, region, src, cnt, log);
    return done("k_carry");
}
int launch_pack_records(hipStream_t s, int64_t nwin, const double *co, const int32_t *nc, double *rec) {
    if (nwin <= 0) return EPH_OK;
    const long long tot = nwin * (kDiv * 3 + 1);
    hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (long long)nwin, co, nc, rec);
    return done("k_pack_records");
}
int launch_lsq_fit(hipStream_t s, int64_t nwin, const uint64_t *first_sample, const uint8_t *degree, int backward,
                   const double *log, double *coeffs, int32_t *ncoef) {
    if (nwin <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_lsq_fit, dim3((unsigned)((nwin + 63) / 64)), dim3(64), 0, s, (long long)nwin, first_sample,
                       degree, backward, log, coeffs, ncoef);
    return done("k_lsq_fit");
}
int launch_spline_eval(hipStream_t s, int64_t m, const double *at, double start, double interval, int64_t npoly,
                       const double *coeffs, const int32_t *ncoef, double *pos, double *vel, uint8_t *inside) {
    if (m <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_spline_eval, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (long long)m, at, start,
                       interval, (long long)npoly, coeffs, ncoef, pos, vel, inside);
    return done("k_spline_eval");
}

}  // namespace eph
