"""GPU probe behind the error-bound note of inv_r3_seeded (csrc/pair_term.h):
  1. accuracy of v_rsq_f64 and of h after the coupled step, over random and structured operands;
  2. the device's own f64 division and the stripped sequences on denominators with an all-ones significand
     (the exceptional case of the reciprocal's closing residual step);
  3. 1/(x sqrt(x)) on the operands whose p = x sqrt(x) comes closest to the top of its binade."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ephemeris_explorer_amd as ea
import hooks as _hooks
hk = _hooks.load()          # the eph_debug_* hooks live in libephemeris_amd_testhooks.so (tests/hooks.py)
from exceptional_operands import top_of_binade_operands

rng = np.random.default_rng(5)
x = np.ldexp(rng.uniform(1.0, 2.0, 4_000_000), rng.integers(-299, 299, 4_000_000))
x = np.concatenate([x, np.ldexp(1.0 + np.arange(1, 4097) * 2.0 ** -12, 0), np.ldexp(1.0 + np.arange(1, 4097) * 2.0 ** -12, 1)])
y, h = hk.debug_rsq(x)
xl = x.astype(np.longdouble)
e0 = np.abs((y.astype(np.longdouble) * np.sqrt(xl) - 1).astype(np.float64))
eh = np.abs((h.astype(np.longdouble) * 2 * np.sqrt(xl) - 1).astype(np.float64))
print("v_rsq_f64 max rel err 2^%.2f ; h after coupled step max rel err 2^%.2f" % (np.log2(e0.max()), np.log2(eh.max())))
b = np.ldexp(np.nextafter(2.0, 0), np.arange(-200, 200))
for a in (1.0, 3.0, np.nextafter(2.0, 0)):
    fast, ieee = hk.debug_div(np.full_like(b, a), b)
    host = a / b
    print("a=%r / all-ones b: compiler division == host: %s ; shared-reciprocal == host: %s" % (
        a, np.array_equal(ieee, host), np.array_equal(fast, host)))
f, i = hk.debug_inv_r3(b)
host = 1.0 / (b * np.sqrt(b))
print("inv_r3 on all-ones x: fast==host", np.array_equal(f, host), "ieee==host", np.array_equal(i, host))
xs, ks = top_of_binade_operands(kmax=64)
f, i = hk.debug_inv_r3(xs)
host = 1.0 / (xs * np.sqrt(xs))
print("top-of-binade p: %d operands, k in %s..%s; fast==host %s ieee==host %s" % (len(xs), ks.min(), ks.max(), np.array_equal(f, host), np.array_equal(i, host)))
