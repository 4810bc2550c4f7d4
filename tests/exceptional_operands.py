"""Operands x for which the reciprocal's denominator p = RN(x * RN(sqrt(x))) has a significand at the very top of its
binade, p = 2^(E+1) - k ulp for small k. k = 1 (all ones) is THE exceptional significand of the reciprocal's closing
residual step (Markstein): 1/p = 2^-(E+1) (1 + 2^-53 + 2^-106 + ...) sits 2^-106 (relative) above a rounding boundary,
so an iterate that approaches 1/p from below ends on an exact tie. csrc/pair_term.h (inv_r3_seeded / rcp_biased)
explains how the sequences get this case right; these operands are how the tests check it."""
import numpy as np


def top_of_binade_operands(kmax=8, e_lo=-298, e_hi=298):
    """x in [2^e_lo, 2^e_hi) with mantissa(p) >= 2 - kmax * 2^-52, found by scanning the few x around
    (2^(E+1))^(2/3) for every binade E of p. Returns (x, k) arrays."""
    xs, ks = [], []
    for E in range(int(1.5 * e_lo) - 1, int(1.5 * e_hi) + 2):
        target = np.ldexp(1.0, E + 1)                      # p just below this power of two
        x0 = target ** (2.0 / 3.0)
        if not (np.ldexp(1.0, e_lo) <= x0 < np.ldexp(1.0, e_hi)):
            continue
        x = x0
        for _ in range(6):                                  # walk to the largest x with p(x) < target
            x = np.nextafter(x, np.inf)
        cand = []
        for _ in range(12 + 3 * kmax):
            cand.append(x)
            x = np.nextafter(x, 0.0)
        cand = np.array(cand)
        p = cand * np.sqrt(cand)
        k = np.round((target - p) / np.spacing(np.nextafter(target, 0.0))).astype(np.int64)
        keep = (p < target) & (k >= 1) & (k <= kmax)
        xs.extend(cand[keep])
        ks.extend(k[keep])
    return np.array(xs), np.array(ks)
