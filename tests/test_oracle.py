"""CPU: pins the oracle -- against the reference's coefficient constants, an independent Python restatement,
committed golden vectors, closed-form / known-answer problems and the reference tests' own thresholds re-expressed
on the committed systems fixtures."""
import json
import math

import numpy as np
import pytest

from conftest import GOLDEN, load_system
from oracle import orc, pyoracle as po

TABLES = json.loads((GOLDEN / "coeff_tables.json").read_text())


def fromhex(xs):
    return np.array([float.fromhex(x) for x in xs])


# ---- coefficients --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["BlanesMoan6B", "BlanesMoan11B", "BlanesMoan14A", "ForestRuth", "McLachlanO4",
                                  "McLachlanSS17", "Pefrl", "Ruth"])
def test_srkn_coefficients_match_reference_ratios(name):
    A, B, fsal = orc.srkn_coeffs(name)
    t = TABLES["methods"][name]
    assert fsal == t["FSAL"]
    assert np.array_equal(A, fromhex(t["A"]["f64hex"])) and np.array_equal(B, fromhex(t["B"]["f64hex"]))
    # consistency: sum(A) = sum(B) = 1 for a consistent splitting method
    assert abs(A.sum() - 1) < 1e-14 and abs(B.sum() - 1) < 1e-14


def test_ratio_from_f64_restatement():
    """Ratio::from_f64 (integration/src/ratio.rs:75-103): the C restatement agrees with the generator's emulation,
    including the 7 literals that do NOT round-trip."""
    lits = {"BlanesMoan6B": [0.245298957184271, 0.60487266571108, -0.350171622895351, 0.0829844064174052,
                             0.396309801498368, -0.0390563049223486, 0.1195241940131508, 0.0],
            "BlanesMoan14A": [0.09171915262446165, -0.19641146648645422]}
    for v in lits["BlanesMoan6B"]:
        n, d, f = orc.ratio_from_f64(v)
        assert f == v and (d == 1 or n % 2 or d % 2) and float(n) / float(d) == v
    assert orc.ratio_from_f64(0.09171915262446165)[2] == 0.09171915262446163
    assert orc.ratio_from_f64(-0.19641146648645422)[2] == -0.1964114664864542
    # every SRKN ratio of the tables is reproduced from its own f64 image or is one of the known exceptions
    for name in ("BlanesMoan6B", "BlanesMoan11B", "ForestRuth"):
        t = TABLES["methods"][name]
        for (n, d), hx in zip(t["A"]["ratio"] + t["B"]["ratio"], t["A"]["f64hex"] + t["B"]["f64hex"]):
            n2, d2, f2 = orc.ratio_from_f64(float.fromhex(hx))
            assert (n2, d2) == (int(n), int(d)) and f2 == float.fromhex(hx)


def test_elm2_and_cowell_coefficients():
    c = orc.elm2_coeffs("QuinlanTremaine12")
    t = TABLES["methods"]["QuinlanTremaine12"]
    assert c["order"] == 12
    assert list(c["w_alpha"]) == [float(-int(a)) for a in t["ALPHA"][1:]] == [2, -2, 1, 0, 0, 0, 0, 0, 1, -2, 2, -1]
    assert list(c["w_beta"]) == [float(int(b)) for b in t["BETA_N"][1:]]
    assert c["inv_beta_d"] == 1.0 / 53222400.0
    cw = TABLES["cowell"]["Cowell<12>"]
    assert list(c["cowell"]) == [float(int(b)) for b in cw["BETA_N"]] and c["inv_cowell_d"] == 1.0 / 435891456000.0
    # consistency of a second-order multistep method: rho(1) = rho'(1) = 0 and sigma(1) = rho''(1) / 2, in integers
    al = [int(a) for a in t["ALPHA"]]                       # coefficient of zeta^(12-k)
    k = len(al) - 1
    assert sum(al) == 0 and sum((k - i) * a for i, a in enumerate(al)) == 0
    assert 2 * sum(int(b) for b in t["BETA_N"]) == sum((k - i) * (k - i - 1) * a for i, a in enumerate(al)) * int(t["BETA_D"])
    s13 = orc.elm2_coeffs("Stormer13")
    assert s13["order"] == 13 and list(s13["w_alpha"][:2]) == [2.0, -1.0]


@pytest.mark.parametrize("name", ["RK4", "CashKarp45", "DormandPrince54", "DormandPrince87", "Fehlberg45", "Verner87",
                                  "Verner98", "Tsitouras75"])
def test_erk_tables(name):
    c = orc.erk_coeffs(name)
    t = TABLES["methods"][name]
    assert np.array_equal(c["B"], fromhex(t["B"]["f64hex"])) and np.array_equal(c["C"], fromhex(t["C"]["f64hex"]))
    for row, ref in zip(c["A"], t["A"]["f64hex"]):
        assert np.array_equal(row, fromhex(ref))
    assert abs(c["B"].sum() - 1) < 1e-14
    for i, row in enumerate(c["A"]):            # row-sum condition c_i = sum_j a_ij
        assert abs(row.sum() - c["C"][i]) < 1e-13
    if "E" in t:
        assert np.array_equal(c["E"], fromhex(t["E"]["f64hex"])) and abs(c["E"].sum()) < 1e-14


# ---- two independent restatements agree bit for bit ---------------------------------------------------------
@pytest.mark.parametrize("name,steps", [("sun_earth_moon_2433282.5", 120), ("simple_solar_system_2433282.5", 40)])
@pytest.mark.parametrize("sign", [1, -1])
def test_c_oracle_equals_python_restatement(name, steps, sign):
    s = load_system(name)
    nb = orc.NBody(s.pos, s.vel, s.mu, s.epoch, sign * s.dt)
    pr = po.Problem(s.pos, s.vel, s.mu, s.epoch)
    lm = po.LinearMultistep2("QuinlanTremaine12", sign * s.dt, pr)
    for k in range(steps):
        assert nb.advance(1) == 0
        lm.advance()
        p, v, t, sc = nb.state()
        assert np.array_equal(p, np.array(pr.y)) and np.array_equal(v, np.array(pr.dy)), (name, k)
        assert t == pr.time and sc == lm.step_count()
    assert nb.eval_count() == pr.evals == steps - 12 + 302      # 302 evals of start-up


@pytest.mark.parametrize("direction", [1, -1])
def test_c_oracle_solout_equals_python_restatement(direction):
    s = load_system("sun_earth_moon_2433282.5")
    a = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, s.count, s.degree)
    b = po.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, s.count, s.degree)
    for _ in range(200):
        assert a.step() == 0
        b.step()
    sa, sb = a.take_solution(), b.take_solution()
    for body in range(s.n):
        st, iv, n = sa.info(body)
        co, nc = sa.coeffs(body)
        assert (st, iv, n) == (sb[body]["start"], sb[body]["interval"], len(sb[body]["polys"]))
        for p in range(n):
            assert nc[p] == len(sb[body]["polys"][p])
            assert np.array_equal(co[p, : nc[p]], np.array(sb[body]["polys"][p]))
        for at in np.linspace(st - iv / 3, st + iv * n + iv / 3, 23):
            ea = sa.eval(body, at)
            eb = po.spline_eval(sb[body]["start"], sb[body]["interval"], sb[body]["polys"], at)
            assert (ea is None) == (eb is None)
            if ea is not None:
                assert np.array_equal(ea[0], np.array(eb[0])) and np.array_equal(ea[1], np.array(eb[1]))
    # partially filled windows back-date the next solution's start (nbody.rs:455-468)
    sa2, sb2 = a.take_solution(), b.take_solution()
    for body in range(s.n):
        assert sa2.info(body)[0] == sb2[body]["start"] and sa2.info(body)[2] == 0


# ---- committed golden vectors ---------------------------------------------------------------------------------
GOLD = json.loads((GOLDEN / "nbody_golden.json").read_text())


@pytest.mark.parametrize("name", list(GOLD["systems"]))
def test_oracle_reproduces_golden_states(name):
    s = load_system(name)
    g = GOLD["systems"][name]
    for sign, key in ((1, "forward"), (-1, "backward")):
        nb = orc.NBody(s.pos, s.vel, s.mu, s.epoch, sign * s.dt)
        done = 0
        for mark in sorted(int(k) for k in g[key]):
            assert nb.advance(mark - done) == 0
            done = mark
            p, v, t, sc = nb.state()
            e = g[key][str(mark)]
            assert t == float.fromhex(e["t"]) and sc == e["step_count"]
            assert np.array_equal(p.ravel(), fromhex(e["pos"])) and np.array_equal(v.ravel(), fromhex(e["vel"]))


@pytest.mark.parametrize("name", ["sun_earth_moon_2433282.5", "simple_solar_system_2433282.5"])
def test_oracle_reproduces_golden_splines(name):
    s = load_system(name)
    for d, key in ((1, "forward"), (-1, "backward")):
        g = GOLD["systems"][name]["splines"][key]
        pr = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, d, s.count, s.degree)
        for _ in range(g["steps"]):
            assert pr.step() == 0
        assert pr.time() == float.fromhex(g["time"])
        sol = pr.take_solution()
        for b, e in enumerate(g["bodies"]):
            st, iv, n = sol.info(b)
            assert (st, iv, n) == (float.fromhex(e["start"]), float.fromhex(e["interval"]), e["npoly"])
            co, nc = sol.coeffs(b)
            keep = len(e["ncoef"])
            assert list(nc[:keep]) == e["ncoef"] and np.array_equal(co[:keep].ravel(), fromhex(e["coeffs"]))


# ---- known answers ------------------------------------------------------------------------------------------
def kepler_state(mu, a, e, t):
    """closed-form two-body relative orbit in the plane (integration/examples/plot_work_precision.rs:196-242)."""
    n = math.sqrt(mu / a ** 3)
    M = n * t
    E = M
    for _ in range(50):
        E -= (E - e * math.sin(E) - M) / (1 - e * math.cos(E))
    r = np.array([a * (math.cos(E) - e), a * math.sqrt(1 - e * e) * math.sin(E), 0.0])
    rd = n * a / (1 - e * math.cos(E))
    v = np.array([-rd * math.sin(E), rd * math.sqrt(1 - e * e) * math.cos(E), 0.0])
    return r, v


@pytest.mark.parametrize("method,tol", [("QuinlanTremaine12", 2e-9), ("Stormer13", 2e-9), ("BlanesMoan14A", 1e-7),
                                        ("BlanesMoan6B", 1e-4)])
def test_kepler_closed_form(method, tol):
    mu_tot, a, e = 1.0, 1.0, 0.3
    m1, m2 = 0.75, 0.25
    r0, v0 = kepler_state(mu_tot, a, e, 0.0)
    pos = np.array([-m2 / mu_tot * r0, m1 / mu_tot * r0])
    vel = np.array([-m2 / mu_tot * v0, m1 / mu_tot * v0])
    period = 2 * math.pi
    steps = 2000
    nb = orc.NBody(pos, vel, [m1, m2], 0.0, period / steps, method)
    assert nb.advance(3 * steps) == 0
    p, v, t, _ = nb.state()
    r, rv = kepler_state(mu_tot, a, e, t)
    assert np.abs((p[1] - p[0]) - r).max() < tol and np.abs((v[1] - v[0]) - rv).max() < tol * 5


def test_qt12_convergence_order():
    """Halving h shrinks the Kepler error by between 2^6 (the BlanesMoan6B start-up, run at h/4, is 6th order and
    its error is carried along) and 2^12 (the multistep proper)."""
    mu_tot, a, e = 1.0, 1.0, 0.5
    r0, v0 = kepler_state(mu_tot, a, e, 0.0)
    errs = []
    for steps in (300, 600):
        nb = orc.NBody([[0, 0, 0], r0], [[0, 0, 0], v0], [1.0, 0.0], 0.0, 2 * math.pi / steps)
        nb.advance(steps)
        p, _, t, _ = nb.state()
        errs.append(np.abs(p[1] - p[0] - kepler_state(mu_tot, a, e, t)[0]).max())
    order = math.log2(errs[0] / errs[1])
    assert 6.0 < order < 14.5, (errs, order)


def test_invariants_full_solar_system():
    s = load_system("full_solar_system_2433282.5")
    nb = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)

    def invariants(p, v):
        ke = 0.5 * np.sum(s.mu * np.sum(v * v, axis=1))
        d = p[:, None, :] - p[None, :, :]
        r = np.sqrt((d * d).sum(-1)) + np.eye(s.n)
        pe = -0.5 * np.sum(np.outer(s.mu, s.mu) / r * (1 - np.eye(s.n)))
        return ke + pe, (s.mu[:, None] * v).sum(0), (s.mu[:, None] * np.cross(p, v)).sum(0)

    e0, p0, l0 = invariants(s.pos, s.vel)
    assert nb.advance(20000) == 0          # 139 days
    p, v, _, _ = nb.state()
    e1, p1, l1 = invariants(p, v)
    assert abs((e1 - e0) / e0) < 1e-8
    assert np.abs(p1 - p0).max() < 1e-9 * np.abs(s.mu[:, None] * s.vel).sum()
    assert np.abs(l1 - l0).max() < 1e-8 * np.abs(l0).max()   # Cowell velocities carry truncation error


def test_self_convergence_is_roundoff_limited():
    """ephemeris/tests/solar_system_convergence.rs:268,336-360 asserts QuinlanTremaine12 at h = 10 min is within
    10 m / 1 m/s of the h/2 run after a year -- but with its compensated `Double<DVec3>` state type (:12-110). The
    app (and this path) integrates plain DVec3 barycentric coordinates, where the uncompensated multistep sums
    accumulate round-off like ulp(|r|) * steps^1.5. Re-expressed offline on the committed 32-body system over
    120 days: inner bodies agree to tens of metres, the outer system to ~1 km -- which is why GPU parity is
    defined bit-for-bit against the oracle rather than by a physical tolerance (DESIGN.md "Parity")."""
    s = load_system("full_solar_system_2433282.5")
    span = 120 * 86400.0
    a = orc.NBody(s.pos, s.vel, s.mu, s.epoch, 600.0)
    b = orc.NBody(s.pos, s.vel, s.mu, s.epoch, 300.0)
    assert a.advance(int(span / 600)) == 0 and b.advance(int(span / 300)) == 0
    pa, va, ta, _ = a.state()
    pb, vb, tb, _ = b.state()
    assert ta == tb
    d = np.linalg.norm(pa - pb, axis=1)
    assert d[:6].max() < 0.05 and d.max() < 3.0                 # km: Sun..Mars, then everything
    assert np.linalg.norm(va - vb, axis=1).max() < 1e-3         # km/s
    # the round-off estimate that explains the outer-system figure
    steps = span / 300
    assert d.max() < 10 * np.spacing(np.abs(pa).max()) * steps ** 1.5


def test_bound_and_underflow_errors():
    s = load_system("sun_earth_moon_2433282.5")
    nb = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
    nb.set_bound(s.epoch + 3.5 * s.dt)
    # the start-up sub-steps (h/4) check the bound too: the 4th macro step stops half way (runge_kutta/mod.rs:113-115)
    assert nb.advance(10) == orc.BOUND_REACHED and nb.state()[3] == 3 and nb.state()[2] == s.epoch + 3.5 * s.dt
    tiny = orc.NBody(s.pos, s.vel, s.mu, 1e30, 1.0)
    assert tiny.advance(1) == orc.STEP_SIZE_UNDERFLOW


def test_least_squares_fit_reproduces_polynomials():
    ts = [k / 8.0 for k in range(9)]
    rng = np.random.default_rng(3)
    for deg in range(0, 8):
        c = rng.normal(size=(deg + 1, 3))
        xs = np.array([sum(c[k] * t ** k for k in range(deg + 1)) for t in ts])
        co, n = orc.least_squares_fit(deg, ts, xs)
        assert n == deg + 1 and np.abs(co[: deg + 1] - c).max() < 1e-5
    co, n = orc.least_squares_fit(5, ts, np.zeros((9, 3)))
    assert n == 0                                               # trim() removes exact zeros


def test_f64_recurrence_vs_exact_arithmetic():
    """how far a correct f64 evaluation of the QuinlanTremaine12 recurrence sits from the same recurrence
    in exact (40-digit) arithmetic -- the Python restatement run over mpmath numbers. sun_earth_moon, dt = 6 h, 3000
    steps (2 years): ~3e-4 km = 2e-12 AU; with round-off growing like steps^1.5 that is ~4e-10 AU at 1e5 steps, so the
    north star's 1e-9 AU bound separates "a correct f64 implementation" from "a different algorithm". (The GPU does
    not need the margin: it reproduces the oracle's bits.)"""
    import mpmath as mp
    s = load_system("sun_earth_moon_2433282.5")
    steps = 3000
    with mp.workdps(40):
        p = po.Problem(s.pos, s.vel, s.mu, s.epoch, num=mp.mpf, sqrt=mp.sqrt)
        lm = po.LinearMultistep2("QuinlanTremaine12", s.dt, p, num=mp.mpf)
        for _ in range(steps):
            lm.advance()
        exact = np.array([[float(c) for c in r] for r in p.y])
    o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
    assert o.advance(steps) == 0
    d = np.abs(o.state()[0] - exact).max()
    au = 1.495978707e8
    assert 0.0 < d / au < 2e-11
    assert d / au * (1e5 / steps) ** 1.5 < 1e-9


def test_target_partitioned_openmp_gravity_has_the_same_bits():
    """bench.py's "all_cores" CPU line evaluates the sums partitioned by target body over OpenMP threads (all N^2
    directed interactions); it must be the serial pair loop's result bit for bit, through a whole integration."""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(300)
    a = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0)
    assert a.advance(12 + 20) == 0
    orc.set_gravity_threads(4)
    try:
        b = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0)
        assert b.advance(12 + 20) == 0
        acc_par = orc.gravity(pos, mu)
    finally:
        orc.set_gravity_threads(0)
    assert np.array_equal(a.state()[0], b.state()[0]) and np.array_equal(a.state()[1], b.state()[1])
    assert np.array_equal(acc_par, orc.gravity(pos, mu))


def test_pair_formula_variants_stay_bounded(capsys):
    """What the one unpinned choice can cost: the `particular` crate's source is absent, so the order in
    which the point-mass term is evaluated is a restatement. Six other plausible orders -- three "one reciprocal"
    forms (1/(r*r*r); s*s*s with s = 1/sqrt(n2); (1/n2)*(1/sqrt(n2))) and the three DIVISION forms ((d*mu)/p,
    d*(mu/p), (d/p)*mu with p = n2*sqrt(n2)) -- differ from the pinned one by round-off only; after 1e5
    QuinlanTremaine12 steps that is a few 1e-9 AU on sun_earth_moon (68 years, worst body the Moon) and ~2e-8 AU on
    the fast moons of the full system (1.9 years) -- pure along-track round-off growth, the same size as the
    f64-vs-exact-arithmetic gap. So agreement with the Rust binary to 1e-9 AU needs the same operation order; what this
    repository guarantees is bit-identity with the committed restatement IN THE SELECTED ORDER. The measured
    displacement of each order against order 0 is printed and recorded in profiles/r03_pair_variants.md."""
    au = 1.495978707e8
    bounds = {"sun_earth_moon_2433282.5": 1e-8, "full_solar_system_2433282.5": 2e-7}
    try:
        for name, bound in bounds.items():
            s = load_system(name)
            states = []
            for variant in range(7):
                orc.set_pair_variant(variant)
                o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
                assert o.advance(100_000) == 0
                states.append(o.state()[0])
            for variant in range(1, 7):
                d = np.abs(states[variant] - states[0]).max() / au
                with capsys.disabled():
                    print(f"\n  {name}: variant {variant} vs 0 after 1e5 steps: {d:.3e} AU", end="")
                assert 0.0 < d < bound, (name, variant, d)
    finally:
        orc.set_pair_variant(0)


@pytest.mark.parametrize("variant", range(1, 7))
def test_c_and_python_restatements_agree_in_every_pair_variant(variant):
    """The switch exists in both restatements (eph_oracle.c point_mass_term, pyoracle.point_mass_term): same bits in
    the same order, different bits from order 0 somewhere (otherwise the switch does nothing)."""
    s = load_system("simple_solar_system_2433282.5")
    ref = orc.gravity(s.pos, s.mu)
    try:
        orc.set_pair_variant(variant)
        po.set_pair_variant(variant)
        acc = orc.gravity(s.pos, s.mu)
        y = [po.Vec(*p) for p in s.pos]
        pa = np.array(po.gravity(y, list(s.mu), 0.0))
        assert np.array_equal(acc, pa)
        assert not np.array_equal(acc, ref)
        assert np.abs(acc / ref - 1.0).max() < 1e-12      # round-off only (cancellation in the sums included)
        nb = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
        pr = po.Problem(s.pos, s.vel, s.mu, s.epoch)
        lm = po.LinearMultistep2("QuinlanTremaine12", s.dt, pr)
        assert nb.advance(20) == 0
        for _ in range(20):
            lm.advance()
        assert np.array_equal(nb.state()[0], np.array(pr.y))
    finally:
        orc.set_pair_variant(0)
        po.set_pair_variant(0)


def test_erkn_table_satisfies_the_nystrom_order_conditions():
    """Tsitouras75Nystrom (integration/src/methods.rs:1417-1520) as the exact Ratio pairs of the reference: the
    Runge-Kutta-Nystrom order conditions a transcription slip would break (published rationals: exact to ~1e-18)."""
    from fractions import Fraction as Fr
    t = TABLES["methods"]["Tsitouras75Nystrom"]
    r = lambda nd: Fr(int(nd[0]), int(nd[1]))   # noqa: E731
    A = [[r(x) for x in row] for row in t["A"]["ratio"]]
    BP, BV, C = ([r(x) for x in t[k]["ratio"]] for k in ("BP", "BV", "C"))
    EP, EV = ([r(x) for x in t[k]["ratio"]] for k in ("EP", "EV"))
    tol = Fr(1, 10 ** 16)
    assert abs(sum(BV) - 1) < tol and abs(sum(BP) - Fr(1, 2)) < tol
    assert abs(sum(b * c for b, c in zip(BV, C)) - Fr(1, 2)) < tol
    assert abs(sum(b * c for b, c in zip(BP, C)) - Fr(1, 6)) < tol
    assert abs(sum(b * c * c for b, c in zip(BV, C)) - Fr(1, 3)) < tol
    for row, c in zip(A, C):
        assert abs(sum(row) - c * c / 2) < tol                 # stage positions are second-order Taylor consistent
    for bp, bv, c in zip(BP, BV, C):
        assert abs(bp - bv * (1 - c)) < tol                    # the simplifying assumption b'_i = b_i (1 - c_i)
    assert abs(sum(EP)) < tol and abs(sum(EV)) < tol           # both solutions are consistent: their difference is O(h^p)
    assert t["FSAL"] and int(t["ORDER"]) == 7 and int(t["ORDER_EMBEDDED"]) == 5 and C[-1] == 1
    assert BP[:6] == A[6] and BP[6] == 0                       # FSAL: the last stage is the new point


# ---- error paths: the two restatements against each other (the GPU is compared with the C one in tests/test_gpu_edge_cases.py) ----
def _py_state(pr):
    return (np.array([[float(c) for c in r] for r in pr.y]), np.array([[float(c) for c in r] for r in pr.dy]))


@pytest.mark.parametrize("t0,h,method,expect", [
    (1e20, 1.0, "QuinlanTremaine12", (1, 0.0, 0)),                   # refused before anything moves
    (2.0 ** 52, 1.0, "QuinlanTremaine12", (1, 0.0, 0)),              # the first sub-step of h/4
    (2.0 ** 52 - 1.0, 2.0, "QuinlanTremaine12", (1, 1.0, 0)),        # the third sub-step of the first macro step
    (2.0 ** 52 - 9.0, 2.0, "Stormer13", (1, 9.0, 4)),                # the third sub-step of the fifth macro step
])
def test_underflow_inside_the_starter_c_vs_python(t0, h, method, expect):
    """multistep/mod.rs:201-218 + runge_kutta/mod.rs:112-125: the main test passes, a sub-step of the Substepper fails, and the `?`
    leaves the problem partly advanced. Both restatements must agree on where: status, time, step count, state -- twice."""
    rng = np.random.default_rng(5)
    n = 6
    pos, vel, mu = rng.normal(size=(n, 3)) * 1e7, rng.normal(size=(n, 3)), rng.uniform(1.0, 1e5, n)
    c = orc.NBody(pos, vel, mu, t0, h, method)
    pr = po.Problem(pos, vel, mu, t0)
    lm = po.LinearMultistep2(method, h, pr)
    for calls in (30, 1):
        st_c = c.advance(calls)
        st_p = 0
        for _ in range(calls):
            st_p = lm.advance()
            if st_p:
                break
        assert st_c == st_p == expect[0]
        pc, vc, tc, sc = c.state()
        pp, vp = _py_state(pr)
        assert tc == pr.time == t0 + expect[1] and sc == lm.step_count() == expect[2]
        assert np.array_equal(pc, pp) and np.array_equal(vc, vp)
        assert c.eval_count() == pr.evals


def test_bound_inside_the_starter_c_vs_python():
    """the bound is tested by the sub-steps too (runge_kutta/mod.rs:113-115): a bound half way through the fourth macro step"""
    s = load_system("sun_earth_moon_2433282.5")
    c = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
    c.set_bound(s.epoch + 3.5 * s.dt)
    pr = po.Problem(s.pos, s.vel, s.mu, s.epoch)
    pr.bound = s.epoch + 3.5 * s.dt
    lm = po.LinearMultistep2("QuinlanTremaine12", s.dt, pr)
    st = 0
    for _ in range(10):
        st = lm.advance()
        if st:
            break
    assert c.advance(10) == st == orc.BOUND_REACHED
    pc, vc, tc, sc = c.state()
    pp, vp = _py_state(pr)
    assert tc == pr.time and sc == lm.step_count() == 3 and np.array_equal(pc, pp) and np.array_equal(vc, vp)


@pytest.mark.parametrize("dt,counts", [(0.1, [3]), (0.7, [10]), (0.7, [3, 10, 1, 7])])
def test_sampling_trigger_that_stops_firing_c_vs_python(dt, counts):
    """nbody.rs:389-391: `last_sample_time += delta; if last_sample_time == sample_period` on accumulated f64 sums: 0.1 x 3 is
    reached exactly, 0.7 x 10 is stepped over and that body is never sampled again -- in both restatements."""
    s = load_system("sun_earth_moon_2433282.5")
    count = np.array([counts[b % len(counts)] for b in range(s.n)], dtype=np.uint32)
    degree = np.array([5] * s.n, dtype=np.uint32)
    a = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, dt, 1, count, degree)
    b = po.Propagator(s.pos, s.vel, s.mu, s.epoch, dt, 1, count, degree)
    for _ in range(400):
        assert a.step() == 0 and b.step() == 0
    sa, sb = a.take_solution(), b.take_solution()
    sampled = 0
    for body in range(s.n):
        start, interval, npoly = sa.info(body)
        assert start == sb[body]["start"] and interval == sb[body]["interval"] and npoly == len(sb[body]["polys"])
        co, nc = sa.coeffs(body)
        for k, poly in enumerate(sb[body]["polys"]):
            assert nc[k] == len(poly)
            assert np.array_equal(co[k, :nc[k]], np.array([[float(x) for x in v] for v in poly]))
        sampled += npoly > 0
        acc, fires = 0.0, False
        for _ in range(4 * int(count[body]) + 8):
            acc += dt
            if acc == dt * float(count[body]):
                fires = True
                break
            if acc > dt * float(count[body]):
                break
        assert fires == (npoly > 0)
    if (dt, counts) == (0.7, [10]):
        assert sampled == 0
