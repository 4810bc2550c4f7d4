"""-m gpu: the point-mass evaluation order as a RUN-TIME choice (csrc/pair_term.h, eph_set_pair_variant / EPH_PAIR_VARIANT).

The reference takes 1/r^3 from the crate `particular` (0.8.0-dev @ d490707a), whose source is not in its tree; the
library's default restates the published crate's form (order 0). Should the pinned revision evaluate it in another order,
the fix is one call (tools/identify_pair_variant.py says which), and this file shows every order bit-identical to the CPU
restatement switched to the same order (orc.set_pair_variant) on every kernel family that evaluates the term: k_accel (wave
and workgroup forms), the fused multistep kernels, the single-workgroup kernel and the spacecraft sweeps; on the committed
probe operands of the identification kit; with handles of different orders alive in one process; and the division forms'
shared-reciprocal quotient on the numerators closest to a rounding boundary (tests/division_hard_cases.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
k = int(sys.argv[2])
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system, load_ship
from ephemeris_explorer_amd.workloads import plummer
from oracle import orc
if sys.argv[3] == "call":
    ea.set_pair_variant(k)
assert ea.pair_variant() == k                # ("env": EPH_PAIR_VARIANT in the environment)
orc.set_pair_variant(k)
orc.set_pair_variant(k, native=True)
same = lambda a, b: np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))
root = sys.argv[1] + "/tests/golden/systems/"
# seam 1 at the sizes of the three force kernels
rng = np.random.default_rng(k)
for n in (40, 1000, 4096):
    pos, mu = rng.normal(size=(n, 3)) * 1e7, rng.uniform(1.0, 1e5, n)
    assert same(ea.accel_eval(pos, mu), orc.gravity(pos, mu)), ("accel", n)
# operands that leave the wrapper-free division of the division forms (pair_term.h pair_quot): a zero separation
# component (numerator +-0), a massless source (mu = 0), numerators below 2^-200 -- lane-level IEEE fallback
for n in (40, 300, 1000):
    epos, emu = rng.normal(size=(n, 3)) * 1e7, rng.uniform(1.0, 1e5, n)
    epos[:8, 0] = epos[0, 0]
    epos[8:16, 1] = epos[8, 1]
    epos[16:20, 2] = 0.0
    emu[3], emu[5], emu[17] = 0.0, 1e-70, 1e-68
    assert same(ea.accel_eval(epos, emu), orc.gravity(epos, emu)), ("accel edge", n)
# the variant differs from variant 0 somewhere (the flag does something)
orc.set_pair_variant(0)
base = orc.gravity(pos, mu)
orc.set_pair_variant(k)
assert not same(base, orc.gravity(pos, mu))
# single-workgroup multistep kernel + solout (32 bodies), per-step kernels (300 and 4096 bodies)
s = load_system(root + "full_solar_system_2433282.5")
g = ea.NBodyPropagator.from_system(s)
o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
g.step_n(400)
for _ in range(400):
    assert o.step() == 0
assert same(g.state()[0], o.state()[0]) and same(g.state()[1], o.state()[1]), "k_lm_small"
sg, so = g.take_solution(), o.take_solution()
for b in range(s.n):
    assert sg.info(b) == so.info(b) and same(sg.coeffs(b)[0], so.coeffs(b)[0]), ("spline", b)
orc.set_gravity_threads(8, native=True)
for n, steps in ((300, 12 + 20), (4096, 12 + 3)):
    pos, vel, mu = plummer(n)
    g2 = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    o2 = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0, native=True)
    g2.advance(steps)
    assert o2.advance(steps) == 0
    assert same(g2.state()[0], o2.state()[0]) and same(g2.state()[1], o2.state()[1]), ("k_lm_step", n)
# the spacecraft sweep (thread-per-craft and wave-per-craft kernels)
ship = load_ship(root + "full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
g.step_to(ship.start + 2 * 86400.0)
assert o.step_to(ship.start + 2 * 86400.0) == 0
sg2, so2 = g.take_solution(), o.take_solution()
sg.append(sg2)
assert so.append(so2)
eph = ea.Ephemeris(sg, s.mu)
for ncraft in (2, 20000):
    pos = np.repeat(ship.pos[None], ncraft, 0) + np.arange(ncraft)[:, None] * 1e-2
    vel = np.repeat(ship.vel[None], ncraft, 0)
    batch = ea.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=256)
    batch.propagate(ship.start + 3 * 3600.0)
    assert (batch.status()["status"] == 0).all()
    for i in (0, ncraft - 1):
        c = orc.Craft(so, s.mu, ship.start, pos[i], vel[i], "Verner87")
        assert c.step_to(ship.start + 3 * 3600.0) == 0
        kt, kp, kv = batch.knots(i)
        ot, op, ov = c.knots()
        assert same(kt, ot) and same(kp, op) and same(kv, ov), ("craft", ncraft, i)
# what the order costs at the metric's size (recorded in profiles/r03_pair_variants.md)
import time
pos, vel, mu = plummer(4096)
g3 = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
g3.advance(12 + 50)
g3.state()
t0 = time.perf_counter()
g3.advance(500)
g3.state()
print("variant %d us_per_step_4096 %.2f" % (k, (time.perf_counter() - t0) / 500 * 1e6))
print("variant", k, "ok")
'''


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6])
def test_pair_variant_matches_oracle_variant(gpu, variant):
    how = "env" if variant % 2 else "call"
    env = dict(os.environ)
    env.pop("EPH_PAIR_VARIANT", None)
    if how == "env":
        env["EPH_PAIR_VARIANT"] = str(variant)
    r = subprocess.run([sys.executable, "-c", SCRIPT, str(ROOT), str(variant), how], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"variant {variant} ok" in r.stdout
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    with open(out / "pair_variant_times.txt", "a") as f:
        f.write([ln for ln in r.stdout.splitlines() if "us_per_step_4096" in ln][0] + "\n")


def test_default_is_order_zero_and_bad_orders_are_refused(gpu):
    assert gpu.pair_variant() == 0
    for bad in (-1, 7, 100):
        with pytest.raises(Exception):
            gpu.set_pair_variant(bad)
    assert gpu.pair_variant() == 0


def _same(a, b):
    import numpy as np
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def test_probe_operands_give_the_committed_bits_in_every_order(gpu):
    """tests/golden/pair_probe.json through seam 1 (two bodies): what the identification kit promises the maintainer"""
    import json

    import numpy as np
    doc = json.loads((ROOT / "tests/golden/pair_probe.json").read_text())
    f = lambda h: np.array([int(x, 16) for x in h], dtype=np.uint64).view(np.float64)
    try:
        for k in range(7):
            gpu.set_pair_variant(k)
            for i, p in enumerate(doc["pairs"]):
                pos = np.array([f(p["pi"]), f(p["pj"])])
                mu = np.array([f([p["mui"]])[0], f([p["muj"]])[0]])
                got = gpu.accel_eval(pos, mu).reshape(-1).view(np.uint64)
                want = np.array([int(x, 16) for x in doc["expected"][str(k)][i]], dtype=np.uint64)
                want[want == np.uint64(1 << 63)] = 0          # accumulated into +0: a -0 term arrives as +0
                assert np.array_equal(got, want), (k, i)
    finally:
        gpu.set_pair_variant(0)


def test_handles_of_different_orders_coexist(gpu):
    """the order is fixed per handle at creation: a handle made under order 4 keeps it after the default moved on"""
    import numpy as np

    from ephemeris_explorer_amd.workloads import plummer
    from oracle import orc
    pos, vel, mu = plummer(700)
    try:
        gpu.set_pair_variant(4)
        g4 = gpu.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
        gpu.set_pair_variant(0)
        g0 = gpu.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
        c4 = g4.clone()                                   # a clone inherits its parent's order, not the default
        for g in (g4, g0, c4):
            g.advance(12 + 6)
        for k, gs in ((4, (g4, c4)), (0, (g0,))):
            orc.set_pair_variant(k, native=True)
            o = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0, native=True)
            assert o.advance(12 + 6) == 0
            for g in gs:
                assert _same(g.state()[0], o.state()[0]) and _same(g.state()[1], o.state()[1]), k
        assert not _same(g4.acc(), g0.acc())               # (the orders differ in last bits of the accelerations)
    finally:
        gpu.set_pair_variant(0)
        orc.set_pair_variant(0, native=True)


def test_seeded_quotient_on_hard_cases(gpu, hooks):
    """a / p, p = x sqrt(x), through r = RN(1/p) and Markstein's step against the device's IEEE division: numerators whose
    quotient lies k 2^-106 / p' from a rounding boundary (k = 1, 2, 3, 5), and random ones"""
    import math
    import random

    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import division_hard_cases as dh
    rng = random.Random(20260927)
    xs, as_ = [], []
    while len(xs) < 400000:
        x = math.ldexp(1.0 + rng.random(), rng.randrange(-130, 130))
        p = dh.p_of(x)
        for a in dh.hard_numerators(p) + [math.ldexp(1.0 + rng.random(), rng.randrange(-190, 190)) for _ in range(2)]:
            for sgn in (1.0, -1.0):
                xs.append(x)
                as_.append(sgn * a)
    try:
        hooks.set_pair_variant(4)
        fast, ieee = hooks.debug_quot(np.array(xs), np.array(as_))
    finally:
        hooks.set_pair_variant(0)
    assert _same(fast, ieee)
    host = np.array(as_) / np.array([dh.p_of(x) for x in xs])
    assert _same(ieee, host)
