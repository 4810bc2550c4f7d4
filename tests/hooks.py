"""The eph_debug_* test and tuning hooks (csrc/eph_debug.h). They are NOT in the product library: `load()` opens
libephemeris_amd_testhooks.so (the product's objects + debug_api.o, built by ephemeris_explorer_amd.build) or, for the scripts that
read a tuning build's accounting, the library named by `path`. TEST INFRASTRUCTURE ONLY."""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
HOOKS_LIB = ROOT / "ephemeris_explorer_amd" / "libephemeris_amd_testhooks.so"
_dp = C.POINTER(C.c_double)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class Hooks:
    def __init__(self, path=None):
        path = Path(path) if path else HOOKS_LIB
        if not path.exists():
            raise ImportError(f"{path} is missing: python -c 'import __graft_entry__ as g; g.build()'")
        L = C.CDLL(str(path))
        i64, f64 = C.c_int64, C.c_double
        L.eph_debug_inv_r3.argtypes = [i64, _dp, _dp, _dp]
        L.eph_debug_quot.argtypes = [i64, _dp, _dp, _dp, _dp]
        L.eph_debug_inv_r3_sweep.argtypes = [C.c_uint64, i64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.eph_debug_pow.argtypes = [i64, _dp, f64, _dp]
        L.eph_debug_div.argtypes = [i64, _dp, _dp, _dp, _dp]
        L.eph_debug_rsq.argtypes = [i64, _dp, _dp, _dp]
        L.eph_debug_wg_cycles.argtypes = [C.POINTER(C.c_int64)]
        L.eph_status_string.restype = C.c_char_p
        L.eph_status_string.argtypes = [C.c_int32]
        self.L = L

    def _check(self, st, where):
        if st < 0:
            raise RuntimeError(f"{where}: {self.L.eph_status_string(st).decode()}")

    def set_pair_variant(self, k):
        """the hooks library is a library of its own: ITS default evaluation order (eph_set_pair_variant of the hooks library),
        which is what eph_debug_inv_r3 / eph_debug_quot evaluate"""
        self._check(self.L.eph_set_pair_variant(int(k)), "eph_set_pair_variant")

    def debug_quot(self, x, a):
        """(fast, ieee) device evaluations of a / (x * sqrt(x)) -- the division forms' shared-reciprocal quotient"""
        x, a = _f64(x), _f64(a)
        fast, ieee = np.zeros_like(x), np.zeros_like(x)
        self._check(self.L.eph_debug_quot(x.size, _p(x), _p(a), _p(fast), _p(ieee)), "eph_debug_quot")
        return fast, ieee

    def debug_inv_r3(self, n2):
        """(fast, ieee) device evaluations of 1/(x*sqrt(x))"""
        n2 = _f64(n2)
        fast, ieee = np.zeros_like(n2), np.zeros_like(n2)
        self._check(self.L.eph_debug_inv_r3(n2.size, _p(n2), _p(fast), _p(ieee)), "eph_debug_inv_r3")
        return fast, ieee

    def debug_inv_r3_sweep(self, seed, n):
        """(mismatches, bits of one mismatching operand) of the in-range 1/(x*sqrt(x)) sequence against the IEEE expansion over n
        device-generated operands"""
        bad, ex = C.c_uint64(), C.c_uint64()
        self._check(self.L.eph_debug_inv_r3_sweep(int(seed), int(n), C.byref(bad), C.byref(ex)), "eph_debug_inv_r3_sweep")
        return bad.value, ex.value

    def debug_div(self, a, b):
        """(shared-reciprocal quotient, compiler IEEE quotient) of a / b on the device"""
        a, b = _f64(a).ravel(), _f64(b).ravel()
        fast, ieee = np.zeros_like(a), np.zeros_like(a)
        self._check(self.L.eph_debug_div(a.size, _p(a), _p(b), _p(fast), _p(ieee)), "eph_debug_div")
        return fast, ieee

    def debug_rsq(self, x):
        """(v_rsq_f64(x), h after the square root's coupled step) on the device"""
        x = _f64(x).ravel()
        y, h = np.zeros_like(x), np.zeros_like(x)
        self._check(self.L.eph_debug_rsq(x.size, _p(x), _p(y), _p(h)), "eph_debug_rsq")
        return y, h

    def debug_pow(self, x, y):
        """the controller's correctly rounded pow on the device"""
        x = _f64(x)
        out = np.zeros_like(x)
        self._check(self.L.eph_debug_pow(x.size, _p(x), float(y), _p(out)), "eph_debug_pow")
        return out

    def debug_wg_cycles(self):
        """k_lm_small's eight tick counters (zeros unless the library was built with -DEPH_EXPERIMENTS=1)"""
        out = (C.c_int64 * 8)()
        self._check(self.L.eph_debug_wg_cycles(out), "eph_debug_wg_cycles")
        return list(out)


_cache = {}


def load(path=None):
    key = str(path) if path else ""
    if key not in _cache:
        _cache[key] = Hooks(path)
    return _cache[key]
