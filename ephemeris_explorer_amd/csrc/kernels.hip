// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64). No MFMA: the path is f64
// VALU (sqrt + divide) bound, see DESIGN.md. Compiled with -ffp-contract=off: the reference (Rust) never
// fuses a*b+c, and parity is defined bit-for-bit, so every sum below is written in the reference's
// operation order and must stay un-fused.
//
// Reference citations are relative to the reference repository root.
#include "eph_internal.h"

namespace eph {

// ------------------------------------------------------------------------------------------------------
// Pair interaction: acceleration on a body at (xi,yi,zi) from source body pj = {x,y,z,mu}.
// `particular::gravity::newtonian` acceleration_paired / acceleration_at with softening 0
// (call sites ephemeris/src/propagators/nbody.rs:29, ephemeris_explorer/src/dynamics/spacecraft.rs:73).
// For the pair (k, i), k < i, the reference computes -(p_i - p_k) * (mu_k * inv); (p_k - p_i) * (mu_k * inv)
// is the same f64 (negation is exact), so one directed formula serves both triangles.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_accel(double xi, double yi, double zi, const Body4 &pj, double &cx, double &cy,
                                           double &cz) {
    const double dx = pj.x - xi, dy = pj.y - yi, dz = pj.z - zi;
    const double n2 = dx * dx + dy * dy + dz * dz;   // glam DVec3::length_squared, left to right
    const double inv = 1.0 / (n2 * sqrt(n2));        // IEEE correctly rounded f64 sqrt and divide
    const double s = pj.mu * inv;
    cx = dx * s;
    cy = dy * s;
    cz = dz * s;
}

__device__ __forceinline__ void wave_lds_fence() {
    // same-wave LDS hand-off (lane-per-source writes -> lane-per-chain reads): DS ops of one wave execute in
    // order; this only stops the compiler from moving them across.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------
// wave_force<BPW>: one wave64 computes the accelerations of BPW consecutive bodies i0..i0+BPW-1 in EXACTLY the
// reference's summation order (NewtonianGravity::eval, nbody.rs:22-38):
//     ddy[i] = ((init + c(0,i)) + ... + c(i-1,i))  +  ((0 + c(i,i+1)) + ... + c(i,n-1))
// Phase A: lane = source body j of the current 64-body tile; BPW independent interactions per lane go to a
//          wave-private LDS tile C[chain][j]  (chain = body*3 + component, row stride kRow doubles).
// Phase B: lane = chain (< 3*BPW); walks its row in j order with one dependent v_add_f64 per source.
// The sqrt/divide-heavy phase A is fully parallel; only the 3 adds per interaction are ordered.
// Returns, on lane `ch` < 3*BPW, component ch%3 of body i0 + ch/3.
// ------------------------------------------------------------------------------------------------------
template <int BPW, typename PosPtr>
__device__ __forceinline__ double wave_force(PosPtr pos, int n, int i0, double init, double *C, int lane) {
    double xi[BPW], yi[BPW], zi[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
        const int ii = min(i0 + b, n - 1);
        xi[b] = pos[ii].x;
        yi[b] = pos[ii].y;
        zi[b] = pos[ii].z;
    }
    const int ch = lane < 3 * BPW ? lane : 3 * BPW - 1;
    const int my_i = i0 + ch / 3;
    const double *row = C + ch * kRow;
    double acc = init;   // lower chain (sources before the body), continues from the caller's value
    double accL = 0.0;

    for (int j0 = 0; j0 < n; j0 += kTile) {
        const int j = j0 + lane;
        const Body4 pj = pos[j < n ? j : n - 1];
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            double cx, cy, cz;
            pair_accel(xi[b], yi[b], zi[b], pj, cx, cy, cz);   // NaN at j == i: never read back
            C[(b * 3 + 0) * kRow + lane] = cx;
            C[(b * 3 + 1) * kRow + lane] = cy;
            C[(b * 3 + 2) * kRow + lane] = cz;
        }
        wave_lds_fence();
        const int cnt = min(kTile, n - j0);
        const bool diag = (i0 >= j0) && (i0 < j0 + kTile);   // wave-uniform: BPW divides 64 and i0 % BPW == 0
        if (!diag && cnt == kTile) {
#pragma unroll
            for (int jl = 0; jl < kTile; jl += 2) {
                const double2 c2 = *reinterpret_cast<const double2 *>(row + jl);   // ds_read_b128
                acc = acc + c2.x;
                acc = acc + c2.y;
            }
        } else if (!diag) {
            for (int jl = 0; jl < cnt; ++jl) acc = acc + row[jl];
        } else {
            for (int jl = 0; jl < cnt; ++jl) {
                const double c = row[jl];
                const bool self = (j0 + jl == my_i);
                accL = self ? acc : accL;          // lower chain complete
                acc = self ? 0.0 : acc + c;        // upper chain starts from V::default()
            }
        }
        wave_lds_fence();
    }
    return accL + acc;   // ddy[i] += output_i
}

// ------------------------------------------------------------------------------------------------------
// k_accel: a = init + sum, SoA [3][npad] output. One wave per block, BPW bodies per wave.
// ------------------------------------------------------------------------------------------------------
template <int BPW>
__global__ void __launch_bounds__(64) k_accel(int n, int npad, const Body4 *__restrict__ pos,
                                              const double *__restrict__ acc_init, double *__restrict__ acc_out) {
    __shared__ __attribute__((aligned(16))) double C[3 * BPW * kRow];
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * BPW;
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = lane < 3 * BPW && my_i < n;
    const double init = (owner && acc_init) ? acc_init[cc * npad + my_i] : 0.0;
    const double a = wave_force<BPW>(pos, n, i0, init, C, lane);
    if (owner) acc_out[cc * npad + my_i] = a;
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
    __shared__ __attribute__((aligned(16))) double C[3 * BPW * kRow];
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * BPW;
    const int cb = lane / 3, cc = lane % 3;
    const int my_i = i0 + cb;
    const bool owner = lane < 3 * BPW && my_i < a.n;
    const size_t lvl = (size_t)3 * a.npad;
    const size_t off = (size_t)cc * a.npad + (owner ? my_i : 0);

    double yv[L], av[L];   // yv[j]/av[j]: level (new - j); av[0] is filled after the force
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int slot = (a.cur + j) % L;
        yv[j] = owner ? a.Y[slot * lvl + off] : 0.0;
        av[j] = (owner && j > 0) ? a.A[slot * lvl + off] : 0.0;
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

// k_lm_predict: the predictor alone (first step of a batch): thread per (component, body)
template <int L>
__global__ void __launch_bounds__(256) k_lm_predict(const LmArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * a.n) return;
    const int my_i = t / 3, cc = t % 3;
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
// would dominate a per-step launch). 16 waves; wave w owns bodies w*BPW..; lane ch of that wave owns the
// (body, component) chain ch for the force, the velocity, the history ring and the predictor, so the only data
// shared between threads are the packed positions sP (two barriers per step).
// On entry slot `cur` is a COMPLETE level (Y, A, V); on exit slot (cur - nsteps) mod L is.
// ------------------------------------------------------------------------------------------------------
template <int BPW, int L>
__global__ void __launch_bounds__(1024) k_lm_persistent(const LmArgs a, long long nsteps) {
    constexpr int kWaves = 16;
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
        const double anew = wave_force<BPW>(sP, a.n, i0, 0.0, C[w], lane);
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
    if (n >= 16 * 1024) return 16;
    if (n >= 8 * 1024) return 8;
    if (n >= 4 * 1024) return 4;
    if (n >= 2 * 1024) return 2;
    return 1;
}

int launch_accel(hipStream_t s, int n, int npad, const Body4 *pos, const double *acc_init, double *acc_out) {
    if (n <= 0) return EPH_OK;
    const int bpw = lm_bodies_per_wave(n);
    const dim3 grid((n + bpw - 1) / bpw), block(64);
    switch (bpw) {
        case 1: hipLaunchKernelGGL(k_accel<1>, grid, block, 0, s, n, npad, pos, acc_init, acc_out); break;
        case 2: hipLaunchKernelGGL(k_accel<2>, grid, block, 0, s, n, npad, pos, acc_init, acc_out); break;
        case 4: hipLaunchKernelGGL(k_accel<4>, grid, block, 0, s, n, npad, pos, acc_init, acc_out); break;
        case 8: hipLaunchKernelGGL(k_accel<8>, grid, block, 0, s, n, npad, pos, acc_init, acc_out); break;
        default: hipLaunchKernelGGL(k_accel<16>, grid, block, 0, s, n, npad, pos, acc_init, acc_out); break;
    }
    return done("k_accel");
}

template <int L>
static int launch_lm_step_L(hipStream_t s, const LmArgs &a) {
    const int bpw = lm_bodies_per_wave(a.n);
    const dim3 grid((a.n + bpw - 1) / bpw), block(64);
    switch (bpw) {
        case 1: hipLaunchKernelGGL((k_lm_step<1, L>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lm_step<2, L>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lm_step<4, L>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_lm_step<8, L>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_lm_step<16, L>), grid, block, 0, s, a); break;
    }
    return done("k_lm_step");
}
int launch_lm_step(hipStream_t s, const LmArgs &a) {
    if (a.n <= 0) return EPH_OK;
    if (a.L == 12) return launch_lm_step_L<12>(s, a);
    if (a.L == 13) return launch_lm_step_L<13>(s, a);
    return EPH_ERR_UNSUPPORTED;
}
int launch_lm_predict(hipStream_t s, const LmArgs &a) {
    if (a.n <= 0) return EPH_OK;
    const dim3 grid((3 * a.n + 255) / 256), block(256);
    if (a.L == 12) hipLaunchKernelGGL(k_lm_predict<12>, grid, block, 0, s, a);
    else if (a.L == 13) hipLaunchKernelGGL(k_lm_predict<13>, grid, block, 0, s, a);
    else return EPH_ERR_UNSUPPORTED;
    return done("k_lm_predict");
}
template <int L>
static int launch_lm_persistent_L(hipStream_t s, const LmArgs &a, int64_t nsteps) {
    const int per_wave = (a.n + 15) / 16;
    const dim3 grid(1), block(1024);
    if (per_wave <= 1) hipLaunchKernelGGL((k_lm_persistent<1, L>), grid, block, 0, s, a, (long long)nsteps);
    else if (per_wave <= 2) hipLaunchKernelGGL((k_lm_persistent<2, L>), grid, block, 0, s, a, (long long)nsteps);
    else hipLaunchKernelGGL((k_lm_persistent<4, L>), grid, block, 0, s, a, (long long)nsteps);
    return done("k_lm_persistent");
}
int launch_lm_persistent(hipStream_t s, const LmArgs &a, int64_t nsteps) {
    if (a.n <= 0 || nsteps <= 0) return EPH_OK;
    if (a.n > kSmallN) return EPH_ERR_UNSUPPORTED;
    if (a.L == 12) return launch_lm_persistent_L<12>(s, a, nsteps);
    if (a.L == 13) return launch_lm_persistent_L<13>(s, a, nsteps);
    return EPH_ERR_UNSUPPORTED;
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
