// ieee_seq.h -- IEEE-exact f64 sqrt / reciprocal / division sequences without the range-scaling wrappers (device only), the
// parts that do NOT depend on the evaluation order of the point-mass term (that one is pair_term.h).
//
// sqrt, the reciprocal and the quotient on the path must be the IEEE correctly rounded results (the CPU's sqrtsd / divsd). The
// compiler's f64 expansions are: v_rsq_f64 / v_rcp_f64 seed + fma refinement, wrapped in range scaling (v_ldexp, v_div_scale,
// v_div_fmas, v_div_fixup) that only acts for operands near the ends of the exponent range. `*_inrange` / `*_refined` are exactly
// those refinement sequences without the wrappers: bit-identical whenever the scaling would have been a no-op, which the
// callers' range tests guarantee. tests/test_gpu_parity.py::test_inrange_sqrt_and_reciprocal_sequences_are_ieee and
// tests/test_gpu_craft.py::test_shared_reciprocal_division_is_ieee check them against the host.
#pragma once
#include <hip/hip_runtime.h>

namespace eph {

__device__ __forceinline__ double sqrt_inrange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double rcp_inrange(double p) {
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0);
    return __builtin_fma(e, r, r);
}

// a / b, IEEE correctly rounded, with the reciprocal refinement shared between numerators: the compiler's f64
// division is  r = rcp(b) + two Newton steps;  q = a*r;  e = fma(-b, q, a);  q = fma(e, r, q)  inside the scaling
// wrappers. rcp_refined(b) is the first half, div_refined the second; bit-identical to a / b whenever div_scale /
// div_fixup would have been no-ops: b in [2^-200, 2^200) and a either 0 or in that range too (quotient within
// [2^-400, 2^400)). tests/test_gpu_craft.py::test_shared_reciprocal_division_is_ieee.
__device__ __forceinline__ bool in_range_div(double x) {   // biased exponent in [823, 1223)
    return (unsigned)(__double2hiint(x) - 0x33700000) < 0x19000000u;
}
__device__ __forceinline__ double rcp_refined(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double div_refined(double a, double b, double r) {
    const double q = a * r;
    const double e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, r, q);
}
// a / b with the shared reciprocal where it is exact (b_ok: b in range), the compiler's division otherwise
__device__ __forceinline__ double div_shared(double a, double b, double r, bool b_ok) {
    if (b_ok && (a == 0.0 || in_range_div(a))) return div_refined(a, b, r);
    return a / b;
}

}  // namespace eph
