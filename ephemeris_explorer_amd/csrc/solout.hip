// solout.hip -- the kernels of the path that contain no point-mass term (compiled once): staging and start-up element-wise
// kernels, the predictor alone, the solout's sampling / least-squares fit / carry, the evaluator of a UniformSpline.
// Reference citations are relative to the reference repository root.
#include "eph_internal.h"
#include "force_common.h"

namespace eph {

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

// ---- launchers ---------------------------------------------------------------------------------------------------------------
int launch_lm_predict(hipStream_t s, const LmArgs &a) {
    if (a.n <= 0 || a.hi <= a.lo) return EPH_OK;
    const dim3 grid((3 * (a.hi - a.lo) + 255) / 256), block(256);
    if (a.L == 12) hipLaunchKernelGGL(k_lm_predict<12>, grid, block, 0, s, a);
    else if (a.L == 13) hipLaunchKernelGGL(k_lm_predict<13>, grid, block, 0, s, a);
    else return EPH_ERR_UNSUPPORTED;
    return launched("k_lm_predict");
}
int launch_pack(hipStream_t s, int n, int npad, const double *Yslot, const double *mu, Body4 *pos) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, s, n, npad, Yslot, mu, pos);
    return launched("k_pack");
}
int launch_copy3(hipStream_t s, int n, int npad, const double *src, double *dst) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_copy3, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, src, dst);
    return launched("k_copy3");
}
int launch_kick_drift(hipStream_t s, int n, int npad, const double *a, double *v, double *y, double hb, double ha,
                      const double *mu, Body4 *pos_out) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_kick_drift, dim3((n + 255) / 256), dim3(256), 0, s, n, npad, a, v, y, hb, ha, mu, pos_out);
    return launched("k_kick_drift");
}
int launch_aos_to_soa(hipStream_t s, int n, int npad, const double *aos, double *soa) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, aos, soa);
    return launched("k_aos_to_soa");
}
int launch_soa_to_aos(hipStream_t s, int n, int npad, const double *soa, double *aos) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, soa, aos);
    return launched("k_soa_to_aos");
}
int launch_sample(hipStream_t s, int n, int npad, const double *Yslot, const SampleArgs &sa, uint32_t step) {
    if (n <= 0 || !sa.period) return EPH_OK;
    hipLaunchKernelGGL(k_sample, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, npad, Yslot, sa, step);
    return launched("k_sample");
}
int launch_carry(hipStream_t s, int n, const uint64_t *region, const uint32_t *src, const uint32_t *cnt, double *log) {
    if (n <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_carry, dim3((n + 255) / 256), dim3(256), 0, s, n, region, src, cnt, log);
    return launched("k_carry");
}
int launch_pack_records(hipStream_t s, int64_t nwin, const double *co, const int32_t *nc, double *rec) {
    if (nwin <= 0) return EPH_OK;
    const long long tot = nwin * (kDiv * 3 + 1);
    hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (long long)nwin, co, nc, rec);
    return launched("k_pack_records");
}
int launch_lsq_fit(hipStream_t s, int64_t nwin, const uint64_t *first_sample, const uint8_t *degree, int backward,
                   const double *log, double *coeffs, int32_t *ncoef) {
    if (nwin <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_lsq_fit, dim3((unsigned)((nwin + 63) / 64)), dim3(64), 0, s, (long long)nwin, first_sample,
                       degree, backward, log, coeffs, ncoef);
    return launched("k_lsq_fit");
}
int launch_spline_eval(hipStream_t s, int64_t m, const double *at, double start, double interval, int64_t npoly,
                       const double *coeffs, const int32_t *ncoef, double *pos, double *vel, uint8_t *inside) {
    if (m <= 0) return EPH_OK;
    hipLaunchKernelGGL(k_spline_eval, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (long long)m, at, start,
                       interval, (long long)npoly, coeffs, ncoef, pos, vel, inside);
    return launched("k_spline_eval");
}

}  // namespace eph
