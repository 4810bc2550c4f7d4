"""Hard operands for the shared-reciprocal division of the division-form pair orders (csrc/pair_term.h: quot_seeded).

The kernels evaluate  a / p,  p = RN(x * RN(sqrt x)),  as
    r = RN(1 / p)            (inv_r3_seeded: correctly rounded, pair_term.h's argument)
    t = RN(a * r) ; e = a - p * t (one fma, exact) ; q = RN(t + e * r)
-- Markstein's division step. q = RN(a / p) needs t to be close enough to a / p and the residual to be exact; the quotients
that could break it are the ones closest to a rounding boundary. For a given p (significand B, odd part taken) and a small
integer k the numerator significand A with
    A * 2^54 - M * B = k      (quotient in [1/2, 1): M the odd 54-bit midpoint numerator)      or
    A * 2^53 - M * B = k      (quotient in [1, 2))
puts a / p within |k| * 2^-106 / B (relative: about |k| 2^-107) of the midpoint M: no pair (a, p) can come closer than |k| = 1.
`hard_numerators(p)` solves these congruences; tests run them through the device sequence against IEEE division, and
`emulate()` is the same sequence in exact rational arithmetic (checked here on the CPU for every case generated)."""
import math
import struct
from fractions import Fraction


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def p_of(x):
    return x * math.sqrt(x)


def rn(fr):
    return float(fr)                      # Fraction -> float rounds to nearest even


def emulate(a, p):
    """the device sequence in exact arithmetic: returns (q, exact_residual_representable)"""
    r = rn(Fraction(1) / Fraction(p))
    t = a * r
    e_exact = Fraction(a) - Fraction(p) * Fraction(t)
    e = rn(e_exact)
    q = rn(Fraction(t) + Fraction(e) * Fraction(r))
    return q, Fraction(e) == e_exact


def hard_numerators(p, ks=(1, -1, 2, -2, 3, -3, 5, -5)):
    """numerators a (as doubles in [1, 2) scaled to p's binade neighbourhood) whose quotient a / p is within |k| 2^-106 / B
    of a rounding boundary"""
    m, ex = math.frexp(p)                  # p = m * 2^ex, m in [0.5, 1)
    B = int(m * (1 << 53))                 # 53-bit significand
    out = []
    for shift, mbits in ((54, 54), (53, 54)):
        # A * 2^shift - M * B = k, M odd, 2^(mbits-1) <= M < 2^mbits (a midpoint between 53-bit neighbours)
        g = B & -B                         # power of two dividing B
        Bo = B // g
        mod = 1 << shift
        try:
            inv = pow(Bo, -1, mod)
        except ValueError:
            continue
        for k in ks:
            if k % g:
                continue
            M = (-(k // g) * inv) % mod
            if not (M & 1) or not ((1 << (mbits - 1)) <= M < (1 << mbits)):
                continue
            num = M * B + k
            if num % mod:
                continue
            A = num // mod
            if not ((1 << 52) <= A < (1 << 53)):
                continue
            out.append(math.ldexp(float(A), ex - 53))      # any binade will do: the sequence is scale invariant in range
    return out


if __name__ == "__main__":
    import random
    rng = random.Random(1)
    cases = bad = inexact = 0
    worst_gap = 1.0
    for _ in range(20000):
        x = math.ldexp(1.0 + rng.random(), rng.randrange(-40, 40))
        p = p_of(x)
        for a in hard_numerators(p) + [math.ldexp(1.0 + rng.random(), rng.randrange(-60, 60)) for _ in range(4)]:
            for s in (a, -a):
                q, exact = emulate(s, p)
                cases += 1
                bad += q != s / p
                inexact += not exact
    print(f"{cases} cases, {bad} wrong quotients, {inexact} inexact residuals")
