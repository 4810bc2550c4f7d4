"""CPU: the C-ABI library builds for gfx950 without a GPU, loads, exports every symbol the header declares, and
fails LOUDLY (no fallback) when there is no device. Host-only entry points are checked against the oracle."""
import ctypes
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import orc


def header_functions():
    text = (ROOT / "include" / "ephemeris_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eph_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(product_lib):
    names = header_functions()
    assert len(names) >= 40
    lib = ctypes.CDLL(str(product_lib.LIB_PATH))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ephemeris_amd.h but not exported"
    assert sorted(product_lib.ABI_SYMBOLS) == names
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(product_lib.LIB_PATH)]).decode()
    exported = set(re.findall(r" T (eph_[a-z0-9_]+)", out))
    assert exported == set(names), exported ^ set(names)


def test_code_object_is_gfx950(product_lib):
    blob = product_lib.LIB_PATH.read_bytes()
    assert b"gfx950" in blob and b"k_lm_step" in blob and b"k_lm_persistent" in blob


def test_status_strings(product_lib):
    L = product_lib._lib()
    assert L.eph_status_string(3) == b"integration bound reached"     # StepError Display, lib.rs:323-331
    assert L.eph_status_string(1) == b"step size underflow"
    assert b"no CPU path" in L.eph_status_string(-2)


def test_coefficient_entry_points_match_oracle(product_lib):
    for name in ("BlanesMoan6B", "BlanesMoan14A", "McLachlanSS17", "Pefrl"):
        A, B, f = product_lib.srkn_coeffs(name)
        Ao, Bo, fo = orc.srkn_coeffs(name)
        assert f == fo and np.array_equal(A, Ao) and np.array_equal(B, Bo)
    for name in ("QuinlanTremaine12", "Stormer13"):
        a, b = product_lib.elm2_coeffs(name), orc.elm2_coeffs(name)
        assert a["order"] == b["order"] and a["inv_beta_d"] == b["inv_beta_d"] and a["inv_cowell_d"] == b["inv_cowell_d"]
        for k in ("w_alpha", "w_beta", "cowell"):
            assert np.array_equal(a[k], b[k])


def test_no_device_fails_loudly(product_lib):
    if product_lib.device_count() > 0:
        pytest.skip("a device is visible here")
    with pytest.raises(product_lib.EphemerisError) as e:
        product_lib.accel_eval(np.zeros((2, 3)), np.ones(2))
    assert e.value.status == product_lib.ERR_NO_DEVICE
    with pytest.raises(product_lib.EphemerisError):
        product_lib.NBodyIntegration(np.zeros((2, 3)), np.zeros((2, 3)), np.ones(2), 0.0, 1.0)
    with pytest.raises(product_lib.EphemerisError):
        product_lib.least_squares_fit(3, np.zeros((1, 9, 3)))


def test_product_never_imports_the_oracle():
    import ephemeris_explorer_amd
    pkg = ROOT / "ephemeris_explorer_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        text = p.read_text()
        assert "import orc" not in text and "eph_oracle" not in text and "pyoracle" not in text, p
