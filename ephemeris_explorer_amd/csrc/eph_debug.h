/* eph_debug.h -- test and tuning hooks. NOT part of the drop-in boundary (include/ephemeris_amd.h): no trait of the reference
 * corresponds to them. Their extern "C" entry points are compiled in debug_api.cpp, which is linked into
 *   - libephemeris_amd_testhooks.so (the product's objects + debug_api.o; tests/hooks.py loads it), and
 *   - tuning builds (scripts/build_exp.sh NAME -DEPH_EXPERIMENTS=1 ...),
 * never into libephemeris_amd.so. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: the hardware's v_rsq_f64(x[i]) and the h ~ 0.5/sqrt(x[i]) left by the square root's coupled refinement
 * step -- the inputs of the error-bound note on inv_r3_seeded (csrc/pair_term.h) */
int32_t eph_debug_rsq(int64_t n, const double *x, double *rsq, double *h);
/* Test hook: the step-size controller's correctly rounded pow(x[i], y) on the device */
int32_t eph_debug_pow(int64_t n, const double *x, double y, double *out);
/* test hook: a[i] / b[i] through the shared-reciprocal division of k_craft_wave and through the compiler's IEEE
 * division */
int32_t eph_debug_div(int64_t n, const double *a, const double *b, double *fast, double *ieee);
/* Test hook: 1/(x*sqrt(x)) for n inputs computed by the kernel's in-range fast sequences (NaN where the range
 * guard would send the tile to the IEEE form) and by the compiler's IEEE sqrt/divide expansions. */
int32_t eph_debug_inv_r3(int64_t n, const double *n2, double *fast, double *ieee);
/* a / (x * sqrt(x)): the division forms' shared-reciprocal quotient (csrc/pair_term.h) beside the compiler's IEEE division */
int32_t eph_debug_quot(int64_t n, const double *x, const double *a, double *fast, double *ieee);
/* Test hook: the same comparison over n operands generated on the device (splitmix64(seed + index): random mantissa,
 * exponent uniform over the guarded range; n is rounded up to a multiple of 2^20). *mismatches = operands whose two
 * results differ in any bit; *example_bits = the IEEE bits of one of them (0 when none). */
int32_t eph_debug_inv_r3_sweep(uint64_t seed, int64_t n, uint64_t *mismatches, uint64_t *example_bits);
/* Tuning hook: zeros from the product library. A build with -DEPH_EXPERIMENTS=1 (scripts/build_exp.sh) returns the
 * single-workgroup kernel's per-phase tick accounting of its last launch (EPH_DEBUG_SMALL=4). */
int32_t eph_debug_wg_cycles(int64_t *out8);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
namespace eph {                 /* the device halves that live beside their kernels (craft.hip) */
int debug_div_device(int64_t n, const double *a, const double *b, double *fast, double *ieee);
int debug_rsq_device(int64_t n, const double *x, double *rsq, double *h);
int debug_pow_device(int64_t n, const double *x, double y, double *out);
}
#endif
