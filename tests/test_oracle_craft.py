"""CPU: pins the massless-path oracle: the reference's doc-test known answers, its spacecraft test re-expressed on the
committed fixtures (same assertions), an independent Python restatement (bit for bit), and the pinned definition of
the controller's powf."""
import math

import numpy as np
import pytest

from conftest import SYSTEMS, load_system
from ephemeris_explorer_amd.systems import load_ship, parse_epoch
from oracle import orc, pyoracle as po


def test_doc_test_known_answers():
    """integration/src/lib.rs:32-56 (RK4, h = 1e-3) and :60-93 (DormandPrince54, atol 1e-8 / rtol 1e-10, h_init 1e-3,
    h_max 0.2): y' = -y, y(0) = 1 integrated to the bound t = 5; both assert |y - e^-5| < 1e-6."""
    y, steps = orc.doc_test_decay("RK4", False, 0.001)
    assert steps == 5000 and abs(y - math.exp(-5.0)) < 1e-6
    y, steps = orc.doc_test_decay("DormandPrince54", True, 1e-3, 0.2, 1e-8, 1e-10)
    assert abs(y - math.exp(-5.0)) < 1e-6 and 20 < steps < 100
    for name in ("CashKarp45", "Fehlberg45", "Tsitouras75", "Verner87", "Verner98", "DormandPrince87"):
        y, _ = orc.doc_test_decay(name, True, 1e-3, 0.2, 1e-8, 1e-10)
        assert abs(y - math.exp(-5.0)) < 1e-6, name


@pytest.fixture(scope="module")
def scenario():
    s = load_system("simple_solar_system_2433282.5")
    pr = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert pr.step_to(parse_epoch("1952-01-01 00:00:00")) == 0
    eph = pr.take_solution()
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    burns = [(b.start, b.start + b.duration, b.acceleration, s.names.index(b.reference) if b.reference else -1)
             for b in ship.burns]
    return s, eph, ship, burns


def test_reference_spacecraft_scenario(scenario):
    """ephemeris/tests/spacecraft_propagation.rs:401-483 on the committed 10-body 1950 system: Verner87, tol 1e-3 km,
    four burns in Earth / Sun / Mars TNB frames; the spacecraft must be within 10 000 km of Earth at t0 and +15 min and
    of Mars at 1950-07-27 15:45 and 1951-01-01 (:476-480)."""
    s, eph, ship, burns = scenario
    for mode in (0, 1):                      # pinned correctly rounded powf, then this host's libm pow
        orc.set_pow_mode(mode)
        try:
            c = orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, ship.integrator, tol_pos=ship.tolerance,
                          tol_vel=ship.tolerance, burns=burns)
            assert c.step_to(parse_epoch("1951-01-01 00:00:00")) == 0
        finally:
            orc.set_pow_mode(0)
        kt, kp, kv = c.knots()
        assert 10000 < len(kt) < 20000

        def distance(body, when):
            t = parse_epoch(when)
            return np.linalg.norm(orc.hermite_eval(kt, kp, kv, t)[0] - eph.eval(s.names.index(body), t)[0])

        assert distance("Earth", "1950-01-01 00:00:00") < 10_000.0
        assert distance("Earth", "1950-01-01 00:15:00") < 10_000.0
        assert distance("Mars", "1950-07-27 15:45:00") < 10_000.0
        assert distance("Mars", "1951-01-01 00:00:00") < 10_000.0


@pytest.mark.parametrize("method", ["Verner87", "DormandPrince54", "CashKarp45", "Fine45", "Tsitouras75Nystrom"])
def test_c_oracle_equals_python_restatement_massless(scenario, method):
    s, eph, ship, burns = scenario
    if method == "Tsitouras75Nystrom":       # ERKN: y'' = f(t, y) -- the ship's burns re-expressed in the inertial frame
        with pytest.raises(ValueError):      # a frame-relative burn makes the right-hand side velocity dependent
            orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, method, burns=burns)
        burns = [(b0, b1, acc, -1) for b0, b1, acc, _ in burns]
    pe = []
    for b in range(s.n):
        st, iv, n = eph.info(b)
        co, nc = eph.coeffs(b)
        pe.append({"start": st, "interval": iv, "polys": [[po.Vec(*co[p, k]) for k in range(nc[p])] for p in range(n)]})
    orc.set_pow_mode(1)                      # the Python restatement calls math.pow (libm)
    try:
        c = orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, method, tol_pos=1e-3, tol_vel=1e-3, burns=burns)
        p = po.Craft(pe, s.mu, ship.start, ship.pos, ship.vel, method, 1e-3, burns)
        for _ in range(400):                 # crosses the first two burns (integrator resets) and rejections
            assert c.step() == 0 and p.step() == 0
    finally:
        orc.set_pow_mode(0)
    kt, kp, kv = c.knots()
    assert len(p.knots) == len(kt)
    for i, (t, y) in enumerate(p.knots):
        assert kt[i] == t and tuple(kp[i]) == y[:3] and tuple(kv[i]) == y[3:], (method, i)


def test_controller_pow_is_correctly_rounded():
    """The pinned powf: equal to the exact power rounded to nearest (mpmath, 200 bits); this host's libm differs from
    that in a small fraction of calls (glibc documents < 0.52 ULP), which is why the definition is pinned."""
    import mpmath as mp
    mp.mp.prec = 200
    rng = np.random.default_rng(2)
    xs = np.exp(rng.uniform(np.log(1e-14), np.log(1e8), 3000))
    libm_diff = 0
    for k in (4, 5, 7, 8):
        y = -(1.0 / k)
        for x in xs:
            v = orc.cr_pow(x, y)
            assert v == float(mp.power(mp.mpf(float(x)), mp.mpf(y)))
            libm_diff += v != math.pow(x, y)
    assert libm_diff < 0.01 * 4 * len(xs)
    assert orc.cr_pow(0.0, -1 / 7) == math.inf and orc.cr_pow(math.inf, -0.2) == 0.0 and orc.cr_pow(1.0, -0.2) == 1.0


def test_hermite_spline_restatement():
    t = np.array([0.0, 1.0, 3.0])
    p = np.array([[0.0, 0, 0], [1.0, 2, 3], [5.0, 1, 0]])
    v = np.array([[1.0, 0, 0], [1.0, 1, 1], [0.0, 0, 0]])
    for k in range(3):                       # exact at the knots (binary_search Ok branch)
        r = orc.hermite_eval(t, p, v, t[k])
        assert np.array_equal(r[0], p[k]) and np.array_equal(r[1], v[k])
    assert orc.hermite_eval(t, p, v, -0.1) is None and orc.hermite_eval(t, p, v, 3.1) is None
    r = orc.hermite_eval(t, p, v, 0.5)       # cubic through (0,1) with end slopes: position 0.5 + ... closed form
    h00, h10, h01, h11 = 0.5, 0.125, 0.5, -0.125
    assert np.allclose(r[0], h00 * p[0] + h10 * v[0] + h01 * p[1] + h11 * v[1], atol=1e-15)


def test_spacecraft_solout_events_c_vs_python(scenario):
    """The app's SpacecraftSolout (dynamics/spacecraft.rs:91-162,514-587): SOI transitions and apsides per accepted
    step. C oracle == independent Python restatement, and the events are the physical ones of the reference's Mars
    transfer: starts inside Earth's sphere, leaves it into the Sun's on the third day, perigee of the parking orbit
    below the initial 7000 km."""
    from ephemeris_explorer_amd.systems import soi_radii
    s, eph, ship, burns = scenario
    soi = soi_radii(s)
    assert math.isinf(soi[s.names.index("Sun")]) and 9.0e5 < soi[s.names.index("Earth")] < 9.3e5
    pe = []
    for b in range(s.n):
        st, iv, n = eph.info(b)
        co, nc = eph.coeffs(b)
        pe.append({"start": st, "interval": iv, "polys": [[po.Vec(*co[p, k]) for k in range(nc[p])] for p in range(n)]})
    orc.set_pow_mode(1)
    try:
        c = orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, "Verner87", tol_pos=1e-3, tol_vel=1e-3, burns=burns,
                      soi_radius=soi)
        p = po.Craft(pe, s.mu, ship.start, ship.pos, ship.vel, "Verner87", 1e-3, burns, soi=list(soi))
        end = ship.start + 3.5 * 86400.0
        while c.knots()[0][-1] < end:
            assert c.step() == 0 and p.step() == 0
    finally:
        orc.set_pow_mode(0)
    tt, tb = c.transitions()
    at, ad, ab, ak = c.apsides()
    assert [(float(t), int(b)) for t, b in zip(tt, tb)] == p.transitions
    assert [(float(t), float(d), int(b), int(k)) for t, d, b, k in zip(at, ad, ab, ak)] == p.apsides
    earth, sun = s.names.index("Earth"), s.names.index("Sun")
    assert [int(b) for b in tb] == [earth, sun] and tt[0] == ship.start
    assert 2.0 * 86400.0 < tt[1] - ship.start < 3.0 * 86400.0
    assert len(at) >= 2 and all(int(b) == earth for b in ab[:2]) and ad[1] < 7000.0 and int(ak[1]) == 0


def _golden_methods():
    import json
    from conftest import GOLDEN
    return json.loads((GOLDEN / "craft_golden.json").read_text())


@pytest.mark.parametrize("method", ["Verner87", "Fine45", "DormandPrince54"])
def test_oracle_reproduces_the_committed_spacecraft_knots(scenario, method):
    """tests/golden/craft_golden.json (make_golden.py): the Mars transfer to 1951-01-01, knot count, counters, sampled
    knots, SOI transitions and first apsides -- pins the massless path (controller pow included) across changes."""
    from ephemeris_explorer_amd.systems import soi_radii
    s, eph, ship, burns = scenario
    gold = _golden_methods()
    soi = soi_radii(s)
    assert [float(x).hex() for x in soi] == gold["soi_radius"]
    gm = gold["methods"][method]
    c = orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, method, tol_pos=ship.tolerance, tol_vel=ship.tolerance,
                  burns=burns, soi_radius=soi)
    assert c.step_to(parse_epoch("1951-01-01 00:00:00")) == 0
    kt, kp, kv = c.knots()
    st = c.state()
    assert (len(kt), st["steps"], st["attempts"], float(st["next_h"]).hex()) == \
        (gm["knots"], gm["steps"], gm["attempts"], gm["next_h"])
    for smp in gm["sample"]:
        i = smp["i"]
        assert float(kt[i]).hex() == smp["t"]
        assert [float(x).hex() for x in kp[i]] == smp["pos"] and [float(x).hex() for x in kv[i]] == smp["vel"]
    tt, tb = c.transitions()
    assert [[float(t).hex(), int(b)] for t, b in zip(tt, tb)] == gm["transitions"]
    at, ad, ab, ak = c.apsides()
    assert len(at) == gm["apsides"]
    assert [[float(t).hex(), float(d).hex(), int(b), int(k)] for t, d, b, k in list(zip(at, ad, ab, ak))[:8]] == \
        gm["first_apsides"]


@pytest.mark.parametrize("method", ["Verner87", "DormandPrince54", "Fine45"])
@pytest.mark.parametrize("which", ["n_max_5", "n_max_40", "h_max_100", "factors", "tolerances", "tiny_h_init", "impossible_tolerance"])
def test_controller_edge_cases_c_vs_python(scenario, method, which):
    """The adaptive pair's error paths and non-default parameters in BOTH restatements (runge_kutta/mod.rs:225-243,414-439;
    spacecraft.rs:479-485,598-609): MaxIterationsReached with the counter reset at manoeuvre boundaries, the h_max clamp, other
    controller factors, unequal tolerances, StepSizeUnderflow from a tiny h_init and from a tolerance nothing meets. The GPU is
    compared with the C restatement on the same cases in tests/test_gpu_edge_cases.py."""
    s, eph, ship, burns = scenario
    pe = []
    for b in range(s.n):
        st, iv, n = eph.info(b)
        co, nc = eph.coeffs(b)
        pe.append({"start": st, "interval": iv, "polys": [[po.Vec(*co[p, k]) for k in range(nc[p])] for p in range(n)]})
    kw = dict(h_init=60.0, h_max=1.7976931348623157e308, tol_pos=1e-3, tol_vel=1e-3, fac_min=0.2, fac_max=5.0, fac=0.9, n_max=1_000_000)
    kw.update({"n_max_5": dict(n_max=5), "n_max_40": dict(n_max=40), "h_max_100": dict(h_max=100.0),
               "factors": dict(fac_min=0.5, fac_max=2.0, fac=0.8), "tolerances": dict(tol_pos=1e-6, tol_vel=1.0),
               "tiny_h_init": dict(h_init=np.spacing(abs(ship.start)) / 4.0),
               "impossible_tolerance": dict(tol_pos=1e-300, tol_vel=1e-300)}[which])
    t0 = ship.start
    earth = s.names.index("Earth")
    short = [(t0 + 150.0, t0 + 400.0, [2e-4, 1e-4, 0.0], earth), (t0 + 900.0, t0 + 1000.0, [0.0, -1e-4, 2e-5], -1)]
    orc.set_pow_mode(1)                      # the Python restatement calls math.pow (libm)
    try:
        c = orc.Craft(eph, s.mu, t0, ship.pos, ship.vel, method, burns=short, **kw)
        p = po.Craft(pe, s.mu, t0, ship.pos, ship.vel, method, 1e-3, short, h_init=kw["h_init"], n_max=kw["n_max"])
        p.h_max, p.fac_min, p.fac_max, p.fac = kw["h_max"], kw["fac_min"], kw["fac_max"], kw["fac"]
        p.tol_pos, p.tol_vel = kw["tol_pos"], kw["tol_vel"]
        last = (0, 0)
        for _ in range(120):
            last = (c.step(), p.step())
            assert last[0] == last[1], (which, last)
            if last[0]:
                break
    finally:
        orc.set_pow_mode(0)
    expect = {"n_max_5": 2, "n_max_40": 2, "tiny_h_init": 1, "impossible_tolerance": 1}.get(which, 0)
    assert last == (expect, expect)
    cs = c.state()
    assert cs["attempts"] == p.n and cs["next_h"] == p.next_h and cs["t"] == p.t
    kt, kp, kv = c.knots()
    assert len(p.knots) == len(kt)
    for i, (t, y) in enumerate(p.knots):
        assert kt[i] == t and tuple(kp[i]) == y[:3] and tuple(kv[i]) == y[3:], (which, method, i)
