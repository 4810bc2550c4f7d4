// debug_api.cpp -- the extern "C" entry points of the test and tuning hooks (csrc/eph_debug.h). Linked into the test-hooks library
// and into tuning builds only: the product library libephemeris_amd.so exports the drop-in boundary and nothing else.
#include <new>

#include "eph_debug.h"
#include "host.h"

using namespace eph;

#define EPH_GUARD_BEGIN try {
#define EPH_GUARD_END } catch (const std::bad_alloc &) { return EPH_ERR_OUT_OF_MEMORY; } catch (...) { return EPH_ERR_HIP; }

#pragma GCC visibility push(default)
extern "C" {

int32_t eph_debug_inv_r3(int64_t n, const double *n2, double *fast, double *ieee) {
    EPH_GUARD_BEGIN
    if (n < 0 || (n > 0 && (!n2 || !fast || !ieee))) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    if (n == 0) return EPH_OK;
    DevBuf<double> a, b, c;
    if ((st = a.alloc(n)) || (st = b.alloc(n)) || (st = c.alloc(n))) return st;
    EPH_HIP(hipMemcpy(a.p, n2, sizeof(double) * n, hipMemcpyHostToDevice));
    if ((st = launch_debug_inv_r3(default_pair_variant(), nullptr, n, a.p, b.p, c.p))) return st;
    EPH_HIP(hipMemcpy(fast, b.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    EPH_HIP(hipMemcpy(ieee, c.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return EPH_OK;
    EPH_GUARD_END
}

// a / (x * sqrt(x)) through the division forms' seeded reciprocal + Markstein step, and through the compiler's IEEE expansions
int32_t eph_debug_quot(int64_t n, const double *x, const double *a, double *fast, double *ieee) {
    EPH_GUARD_BEGIN
    if (n < 0 || (n > 0 && (!x || !a || !fast || !ieee))) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    if (n == 0) return EPH_OK;
    DevBuf<double> dx, da, b, c;
    if ((st = dx.alloc(n)) || (st = da.alloc(n)) || (st = b.alloc(n)) || (st = c.alloc(n))) return st;
    EPH_HIP(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
    EPH_HIP(hipMemcpy(da.p, a, sizeof(double) * n, hipMemcpyHostToDevice));
    if ((st = launch_debug_quot(default_pair_variant(), nullptr, n, dx.p, da.p, b.p, c.p))) return st;
    EPH_HIP(hipMemcpy(fast, b.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    EPH_HIP(hipMemcpy(ieee, c.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return EPH_OK;
    EPH_GUARD_END
}

int32_t eph_debug_inv_r3_sweep(uint64_t seed, int64_t n, uint64_t *mismatches, uint64_t *example_bits) {
    EPH_GUARD_BEGIN
    if (n < 0 || !mismatches || !example_bits) return EPH_ERR_BAD_ARGUMENT;
    int st = check_device();
    if (st) return st;
    DevBuf<unsigned long long> out;
    if ((st = out.alloc(2))) return st;
    EPH_HIP(hipMemset(out.p, 0, 2 * sizeof(unsigned long long)));
    if ((st = launch_debug_inv_r3_sweep(default_pair_variant(), nullptr, seed, n, out.p))) return st;
    unsigned long long h[2];
    EPH_HIP(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
    *mismatches = h[0];
    *example_bits = h[1];
    return EPH_OK;
    EPH_GUARD_END
}

int32_t eph_debug_wg_cycles(int64_t *out8) {
    if (!out8) return EPH_ERR_BAD_ARGUMENT;
    return debug_wg_cycles(default_pair_variant(), (long long *)out8);
}

int32_t eph_debug_div(int64_t n, const double *a, const double *b, double *fast, double *ieee) { return debug_div_device(n, a, b, fast, ieee); }
int32_t eph_debug_rsq(int64_t n, const double *x, double *rsq, double *h) { return debug_rsq_device(n, x, rsq, h); }
int32_t eph_debug_pow(int64_t n, const double *x, double y, double *out) { return debug_pow_device(n, x, y, out); }

}  // extern "C"
#pragma GCC visibility pop
