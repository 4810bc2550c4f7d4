"""Host side of the error-bound note on inv_r3_seeded (csrc/pair_term.h, step 5): the denominator p = RN(x RN(sqrt x))
never has an all-ones significand -- the one significand for which a reciprocal's closing residual step can end on a
tie. No device needed: sqrt and multiply here are the IEEE operations the note talks about."""
import numpy as np

from exceptional_operands import top_of_binade_operands


def test_p_is_never_all_ones():
    x, k = top_of_binade_operands(kmax=16)
    assert len(x) > 3000 and k.min() == 2                 # k = 1 (all ones) does not occur in any binade
    # the three classes: which k occur just below 2^(E+1) depends only on (E+1) mod 3
    p = x * np.sqrt(x)
    e1 = np.frexp(p)[1]                                    # p in [2^(e1-1), 2^e1)
    smallest = {r: int(k[e1 % 3 == r].min()) for r in range(3)}
    assert sorted(smallest.values()) == [2, 2, 3], smallest


def test_the_scale_invariance_the_enumeration_rests_on():
    rng = np.random.default_rng(3)
    x = np.ldexp(rng.uniform(1.0, 2.0, 100_000), rng.integers(-200, 200, 100_000))
    p = x * np.sqrt(x)
    p4 = (4.0 * x) * np.sqrt(4.0 * x)
    assert np.array_equal(p4, 8.0 * p)                    # x -> 4x maps p -> 8p exactly: three binade classes cover all
