"""-m gpu: the HIP path against the CPU oracle, through the C ABI. Bit-exact (f64, same operation order)."""
import numpy as np
import pytest

from conftest import load_system
from oracle import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_same_bits(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    if not np.array_equal(bits(a), bits(b)):
        bad = np.argwhere(bits(a) != bits(b))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ, first at {bad[0]}: "
                             f"{a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}")


def random_system(n, seed):
    rng = np.random.default_rng(seed)
    pos = rng.normal(size=(n, 3)) * 1e7
    vel = rng.normal(size=(n, 3))
    mu = rng.uniform(1.0, 1e5, size=n)
    return pos, vel, mu


@pytest.mark.parametrize("n", [1, 2, 3, 32, 63, 64, 65, 100, 257, 1000, 2048, 4096, 4100])
def test_accel_matches_oracle_bitwise(gpu, n):
    pos, _, mu = random_system(n, 100 + n)
    assert_same_bits(gpu.accel_eval(pos, mu), orc.gravity(pos, mu), f"accel n={n}")


def test_accel_accumulates_into_nonzero_ddy(gpu):
    pos, _, mu = random_system(40, 7)
    init = np.random.default_rng(1).normal(size=(40, 3)) * 1e-9
    ref = init.copy()
    orc.lib().orc_newtonian_gravity_eval(40, orc._ptr(pos), orc._ptr(mu), orc._ptr(ref))
    assert_same_bits(gpu.accel_eval(pos, mu, init), ref, "accumulate")


def test_accel_empty(gpu):
    assert gpu.accel_eval(np.zeros((0, 3)), np.zeros(0)).shape == (0, 3)


@pytest.mark.parametrize("name,steps", [("sun_earth_moon_2433282.5", 400), ("simple_solar_system_2433282.5", 300),
                                        ("full_solar_system_2433282.5", 300)])
@pytest.mark.parametrize("sign", [1, -1])
@pytest.mark.parametrize("path", [1, 2])
def test_qt12_state_bitwise(gpu, name, steps, sign, path):
    """QuinlanTremaine12 (start-up through BlanesMoan6B sub-steps, then the multistep) on the reference's own
    systems; path 1 = one launch per step, path 2 = persistent single-workgroup kernel."""
    s = load_system(name)
    g = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, sign * s.dt)
    g.set_path(path)
    o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, sign * s.dt)
    done = 0
    for chunk in (1, 5, 6, 1, 7, steps - 20):       # crosses the start-up boundary inside and between calls
        g.advance(chunk)
        assert o.advance(chunk) == 0
        done += chunk
        pg, vg, tg, cg = g.state()
        po, vo, to, co = o.state()
        assert (tg, cg) == (to, co)
        assert_same_bits(pg, po, f"{name} pos after {done}")
        assert_same_bits(vg, vo, f"{name} vel after {done}")
    assert_same_bits(g.acc(), o.acc(), "current_ddy")
    assert g.eval_count() == o.eval_count()


@pytest.mark.parametrize("method", ["Stormer13", "BlanesMoan6B", "BlanesMoan11B", "BlanesMoan14A", "ForestRuth",
                                    "McLachlanO4", "McLachlanSS17", "Pefrl", "Ruth"])
def test_other_methods_bitwise(gpu, method):
    """Both ELM2 tables and every SRKN table the reference defines (methods.rs) as the integrator."""
    s = load_system("simple_solar_system_2433282.5")
    g = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, 3600.0, method)
    o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, 3600.0, method)
    g.advance(40)
    assert o.advance(40) == 0
    pg, vg, tg, cg = g.state()
    po, vo, to, co = o.state()
    assert (tg, cg) == (to, co)
    assert_same_bits(pg, po, method)
    assert_same_bits(vg, vo, method)


@pytest.mark.parametrize("n,steps", [(100, 30), (1024, 16), (4096, 4)])
def test_plummer_qt12_bitwise(gpu, n, steps):
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(n)
    h = 1.0 / 1024.0
    g = gpu.NBodyIntegration(pos, vel, mu, 0.0, h)
    o = orc.NBody(pos, vel, mu, 0.0, h, native=True)
    g.advance(12 + steps)
    assert o.advance(12 + steps) == 0
    pg, vg, tg, cg = g.state()
    po, vo, to, co = o.state()
    assert (tg, cg) == (to, co)
    assert_same_bits(pg, po, f"plummer {n} pos")
    assert_same_bits(vg, vo, f"plummer {n} vel")


def test_bound_and_clone(gpu):
    s = load_system("sun_earth_moon_2433282.5")
    g = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
    g.set_bound(s.epoch + 30.5 * s.dt)
    o.set_bound(s.epoch + 30.5 * s.dt)
    with pytest.raises(gpu.StepError) as e:
        g.advance(100)
    assert e.value.status == gpu.BOUND_REACHED
    assert o.advance(100) == orc.BOUND_REACHED
    assert g.state()[2:] == o.state()[2:]
    assert_same_bits(g.state()[0], o.state()[0], "state at the bound")
    g.set_bound(np.inf)
    o.set_bound(np.inf)
    c = g.clone()
    g.advance(25)
    c.advance(25)
    o.advance(25)
    assert_same_bits(c.state()[0], g.state()[0], "clone resumes identically")
    assert_same_bits(g.state()[0], o.state()[0], "after clone")


@pytest.mark.parametrize("name,steps", [("sun_earth_moon_2433282.5", 500), ("full_solar_system_2433282.5", 2000)])
@pytest.mark.parametrize("direction", [1, -1])
def test_propagator_splines_bitwise(gpu, name, steps, direction):
    """NBodyPropagator + SplineInterpolators solout + LeastSquaresFit: spline starts, intervals, polynomial
    coefficients and trimmed lengths, time(), has_reached(), across take_solution() with partial windows."""
    s = load_system(name)
    g = gpu.NBodyPropagator.from_system(s, direction)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, s.count, s.degree)
    for chunk in (3, 20, steps // 2, steps - steps // 2 - 23):
        g.step_n(chunk)
        for _ in range(chunk):
            assert o.step() == 0
        assert g.time() == o.time()
        assert g.integrator_time() == o.integrator_time()
        sg, so = g.take_solution(), o.take_solution()
        for b in range(s.n):
            assert sg.info(b) == so.info(b), (b, sg.info(b), so.info(b))
            cg, ng = sg.coeffs(b)
            co, no = so.coeffs(b)
            assert np.array_equal(ng, no)
            assert_same_bits(cg, co, f"{name} body {b} coefficients")
    assert_same_bits(g.state()[0], o.state()[0], "state")


def test_step_to_and_eval(gpu):
    s = load_system("sun_earth_moon_2433282.5")
    for direction in (1, -1):
        g = gpu.NBodyPropagator.from_system(s, direction)
        o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, s.count, s.degree)
        target = s.epoch + direction * 40 * 86400.0
        g.step_to(target)
        assert o.step_to(target) == 0
        assert g.has_reached(target) and o.has_reached(target)
        assert g.time() == o.time() and g.state()[3] == o.state()[3]
        sg, so = g.take_solution(), o.take_solution()
        for b in range(s.n):
            st, iv, n = sg.info(b)
            at = np.concatenate([np.linspace(st - iv, st + iv * (n + 1), 257), st + iv * np.arange(n + 1),
                                 [np.nextafter(st, -np.inf), np.nextafter(st + iv * n, np.inf)]])
            pos, vel, inside = sg.eval(b, at)
            ponly, _, inside2 = sg.eval(b, at, with_velocity=False)
            assert np.array_equal(inside, inside2)
            for k, t in enumerate(at):
                r = so.eval(b, t)
                assert (r is not None) == bool(inside[k]), (b, t)
                if r is not None:
                    assert_same_bits(pos[k], r[0], "state_vector position")
                    assert_same_bits(vel[k], r[1], "state_vector velocity")
                    assert_same_bits(ponly[k], so.eval(b, t, with_velocity=False), "position")


def test_least_squares_fit_kernel(gpu):
    rng = np.random.default_rng(5)
    samples = rng.normal(size=(64, 9, 3)) * 1e8
    samples[3] = 0.0                      # all-zero window: every coefficient trims away
    samples[4] = samples[4][:1]           # constant window
    for degree in range(0, 8):
        for backward in (False, True):
            co, nc = gpu.least_squares_fit(degree, samples, backward)
            ts = [1.0 - k / 8.0 if backward else k / 8.0 for k in range(9)]
            for w in range(len(samples)):
                ref, n = orc.least_squares_fit(degree, ts, samples[w])
                assert n == nc[w], (degree, w)
                assert_same_bits(co[w], ref, f"fit degree {degree} window {w}")


def test_inrange_sqrt_and_reciprocal_sequences_are_ieee(gpu, hooks):
    """The pair kernel's stripped sqrt / reciprocal sequences (no range-scaling wrappers) must give the IEEE
    correctly rounded results inside the guarded range: compared with the compiler's full expansions on the
    device AND with the host's sqrt/divide (x86 sqrtsd/divsd), over random mantissas across the whole guarded
    exponent range plus adversarial mantissas (all ones, 1+ulp, perfect squares, powers of two)."""
    rng = np.random.default_rng(11)
    mant = np.concatenate([rng.uniform(1.0, 2.0, 2_000_000), [1.0, np.nextafter(2.0, 0), np.nextafter(1.0, 2),
                           1.5, 1.25, 1.9999999999999991, 1.0000000000000004]])
    expo = rng.integers(-299, 299, size=mant.size)
    x = np.ldexp(mant, expo)
    x = np.concatenate([x, np.arange(1, 4097, dtype=np.float64) ** 2, np.ldexp(1.0, np.arange(-300, 300)),
                        np.ldexp(np.nextafter(2.0, 0), np.arange(-300, 299))])
    # the operands whose denominator p = x sqrt(x) comes closest to the top of its binade, p = 2^(E+1) - k ulp: k = 1 (all
    # ones) is the one significand for which the reciprocal's closing residual step can fail, and it cannot occur
    # (csrc/pair_term.h, note on inv_r3_seeded); k = 2 .. 64 in every binade of the guarded range are here
    from exceptional_operands import top_of_binade_operands
    xt, kt = top_of_binade_operands(kmax=64)
    assert len(xt) > 15000 and kt.min() == 2
    x = np.concatenate([x, xt])
    fast, ieee = hooks.debug_inv_r3(x)
    host = 1.0 / (x * np.sqrt(x))
    ok = ~np.isnan(fast)
    assert ok.mean() > 0.999                       # only the range ends fall to the IEEE form
    assert_same_bits(ieee, host, "device IEEE expansions vs host")
    assert_same_bits(fast[ok], host[ok], "in-range sequences vs host")
    # the seed the error bound starts from: v_rsq_f64 within its documented 2^-23, h after the coupled step within 2^-45
    y, h = hooks.debug_rsq(x[:500_000])
    xl = x[:500_000].astype(np.longdouble)
    assert np.abs((y.astype(np.longdouble) * np.sqrt(xl) - 1).astype(np.float64)).max() < 2.0 ** -23
    assert np.abs((h.astype(np.longdouble) * 2 * np.sqrt(xl) - 1).astype(np.float64)).max() < 2.0 ** -45
    # the device's own division (and the shared-reciprocal form of it) on all-ones denominators, the exceptional
    # significand of ITS residual step: correctly rounded on this hardware in every binade
    b = np.ldexp(np.nextafter(2.0, 0), np.arange(-199, 199))
    for a in (1.0, 3.0, np.nextafter(2.0, 0), 1.0 + 2.0 ** -52):
        qf, qi = hooks.debug_div(np.full_like(b, a), b)
        assert_same_bits(qi, a / b, "compiler division, all-ones denominator")
        assert_same_bits(qf, a / b, "shared-reciprocal division, all-ones denominator")
    # outside the guard the fast form is not used
    out = np.array([0.0, 1e-300, 1e300, np.inf, 5e-324])
    f2, _ = hooks.debug_inv_r3(out)
    assert np.isnan(f2).all()


def test_inrange_sequence_sweep_on_the_device(gpu, hooks):
    """The same comparison over 2^33 device-generated operands (random mantissa, exponent uniform over the guarded range):
    the in-range 1/(x*sqrt(x)) -- whose reciprocal is seeded from the square root's refinement, pair_term.h -- must
    equal the compiler's IEEE expansion in every bit. (`git show d7efbfa:scripts/r02_twelfth.sh` ran 2.7e11 operands: no mismatch.)"""
    for seed in (1, 0xDEADBEEF):
        bad, example = hooks.debug_inv_r3_sweep(seed, 1 << 32)
        assert bad == 0, f"{bad} mismatches, e.g. operand bits {example:#x}"


@pytest.mark.parametrize("name,steps", [("sun_earth_moon_2433282.5", 100_000), ("full_solar_system_2433282.5", 100_000)])
def test_1e5_steps_bitwise(gpu, name, steps):
    """The north-star's horizon: 1e5 steps of QuinlanTremaine12 on the reference's systems, positions must be within
    1e-9 AU of the CPU path -- they are identical."""
    s = load_system(name)
    g = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt, native=True)
    g.advance(steps)
    assert o.advance(steps) == 0
    assert g.state()[2:] == o.state()[2:]
    assert_same_bits(g.state()[0], o.state()[0], f"{name} positions after {steps} steps")
    assert_same_bits(g.state()[1], o.state()[1], f"{name} velocities after {steps} steps")


def test_config2_million_steps_with_solout(gpu):
    """BASELINE configs[1]: full_solar_system (32 bodies, dt 10 min), 1e6 steps (19 years) with the file's
    count/degree sampling: final state and the last polynomials of every body, bit for bit."""
    s = load_system("full_solar_system_2433282.5")
    g = gpu.NBodyPropagator.from_system(s)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree, native=True)
    steps = 1_000_000
    g.step_n(steps)
    for _ in range(steps):
        assert o.step() == 0
    assert g.time() == o.time()
    assert_same_bits(g.state()[0], o.state()[0], "positions after 1e6 steps")
    sg, so = g.take_solution(), o.take_solution()
    for b in range(s.n):
        assert sg.info(b) == so.info(b)
        cg, ng = sg.coeffs(b)
        co, no = so.coeffs(b)
        assert np.array_equal(ng, no)
        assert_same_bits(cg[-3:], co[-3:], f"body {b} last polynomials")
        assert_same_bits(cg[::997], co[::997], f"body {b} sampled polynomials")


def test_full_size_two_kernels_agree(gpu):
    """BASELINE configs[2] at full size: 4096-body Plummer sphere. The oracle needs 42 ms per step, so beyond the 16
    steps checked against it (test_plummer_qt12_bitwise) the two independent force kernels (one wave per block vs.
    workgroup-specialised) are run for 2000 steps each and must agree bit for bit; total momentum stays at round-off."""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(4096)
    a = gpu.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    b = gpu.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    a.set_path(1)
    b.set_path(3)
    a.advance(2012)
    b.advance(2012)
    pa, va, ta, ca = a.state()
    pb, vb, tb, cb = b.state()
    assert (ta, ca) == (tb, cb)
    assert_same_bits(pa, pb, "positions: wave kernel vs workgroup kernel")
    assert_same_bits(va, vb, "velocities: wave kernel vs workgroup kernel")
    assert np.abs((mu[:, None] * va).sum(0)).max() < 1e-12


def test_degenerate_sizes(gpu):
    """Empty and tiny systems through every seam: no bodies at all (nothing to do, no error), one body (no pairs:
    uniform motion through the same integrator arithmetic), two bodies."""
    z3 = np.zeros((0, 3))
    assert gpu.accel_eval(z3, np.zeros(0)).shape == (0, 3)
    g = gpu.NBodyIntegration(z3, z3, np.zeros(0), 0.0, 60.0)
    g.advance(30)
    p, v, t, sc = g.state()
    assert p.shape == (0, 3) and sc == 30 and t == 30 * 60.0
    rng = np.random.default_rng(5)
    for n in (1, 2):
        pos, vel, mu = rng.normal(0, 1e5, (n, 3)), rng.normal(0, 1.0, (n, 3)), rng.uniform(1e3, 1e5, n)
        g = gpu.NBodyIntegration(pos, vel, mu, 10.0, 60.0)
        o = orc.NBody(pos, vel, mu, 10.0, 60.0)
        g.advance(12 + 50)
        assert o.advance(12 + 50) == 0
        gp, gv, gt, gs = g.state()
        op, ov, ot, os_ = o.state()
        assert gt == ot and gs == os_ and np.array_equal(gp, op) and np.array_equal(gv, ov)
        pr = gpu.NBodyPropagator(pos, vel, mu, 10.0, 60.0, gpu.FORWARD, np.full(n, 2, np.uint32), np.full(n, 5, np.uint32))
        opr = orc.Propagator(pos, vel, mu, 10.0, 60.0, 1, np.full(n, 2, np.uint32), np.full(n, 5, np.uint32))
        sol = pr.propagate(10.0 + 100 * 60.0)
        assert opr.step_to(10.0 + 100 * 60.0) == 0
        osol = opr.take_solution()
        for b in range(n):
            assert sol.info(b) == osol.info(b)
            assert np.array_equal(sol.coeffs(b)[0], osol.coeffs(b)[0])


def test_workgroup_kernel_at_every_tile_count(gpu):
    """The role-specialised workgroup kernel is chosen for n > 512 (9+ source tiles; workgroups of 4 bodies up to 1024
    targets, 8 in six waves up to 2048, 16 above -- all three sizes occur below); its barrier schedule (single
    tiles first, then pairs of tiles, six LDS buffers) is exercised here at the tile counts it never sees by default -- 2, 3,
    4, 5, 7, 17 tiles, ragged last tiles -- by forcing it (EPH_FORCE=wg, read once per process: hence the subprocess),
    at the size-dependent workgroup choice and with each workgroup size forced at EVERY n, accelerations and a few fused
    steps against the oracle."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
from oracle import orc
same = lambda a, b: np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))
rng = np.random.default_rng(5)
for n in (65, 128, 129, 200, 256, 300, 448, 1030, 2100):
    pos, mu = rng.normal(size=(n, 3)) * 1e7, rng.uniform(1.0, 1e5, n)
    assert same(ea.accel_eval(pos, mu), orc.gravity(pos, mu)), ("accel", n)
for n in (130, 300, 1030, 2100):
    pos, vel, mu = plummer(n)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    o = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0, native=True)
    g.advance(12 + 7)
    assert o.advance(12 + 7) == 0
    assert same(g.state()[0], o.state()[0]) and same(g.state()[1], o.state()[1]), ("steps", n)
print("ok")
'''
    for bodies in (None, "4", "8", "9", "16"):             # 9 = the six-wave 8-body form (the default for 1024 < targets <= 2048)
        env = dict(os.environ, EPH_FORCE="wg")
        if bodies:
            env["EPH_WG_BODIES"] = bodies
        r = subprocess.run([sys.executable, "-c", script, str(ROOT)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (bodies, r.stdout[-1000:], r.stderr[-3000:])


def test_kernel_choice_boundaries(gpu):
    """Accelerations at the sizes where the kernel choice changes: wave form up to 512 targets, then workgroups of 4 bodies
    (up to 1024), 8 (up to 2048) and 16 -- one body either side of each boundary, plus ragged sizes in between."""
    rng = np.random.default_rng(3)
    for n in (511, 512, 513, 520, 576, 700, 1023, 1024, 1025, 1500, 2047, 2048, 2049):
        pos, mu = rng.normal(size=(n, 3)) * 1e7, rng.uniform(1.0, 1e5, n)
        assert_same_bits(gpu.accel_eval(pos, mu), orc.gravity(pos, mu), f"accelerations, n = {n}")


@pytest.mark.parametrize("n", [1025, 1100, 1536, 2047, 2048, 2049])
@pytest.mark.parametrize("method", ["QuinlanTremaine12", "Stormer13"])
def test_six_wave_workgroup_sizes(gpu, n, method):
    """1024 < targets <= 2048 take the six-wave 8-body workgroup (four pair waves of two bodies, chain, tail: round 5); either side
    of its boundaries and ragged counts inside, both ring lengths, steps in two calls"""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(n)
    g = gpu.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0, method)
    o = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0, method, native=True)
    for k in (13 + 1, 9):
        g.advance(k)
        assert o.advance(k) == 0
        assert_same_bits(g.state()[0], o.state()[0], f"{method} n={n} positions")
        assert_same_bits(g.state()[1], o.state()[1], f"{method} n={n} velocities")
    assert_same_bits(g.acc(), o.acc(), f"{method} n={n} accelerations")


@pytest.mark.parametrize("name", ["sun_earth_moon_2433282.5", "full_solar_system_2433282.5"])
@pytest.mark.parametrize("direction", [1, -1])
def test_single_steps_are_deferred_but_indistinguishable(gpu, name, direction):
    """The app's calling pattern (ephemeris_explorer/src/prediction.rs:422-443): `step()` in a loop, `has_reached()` and
    `time()` after EVERY step, now and then a snapshot (take_solution + clone). eph_prop_step queues steady-state steps and
    answers time() / has_reached() from host bookkeeping; everything observable must equal the oracle stepping one at a time."""
    s = load_system(name)
    g = gpu.NBodyPropagator.from_system(s, direction)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, s.count, s.degree)
    target = s.epoch + direction * 400 * s.dt
    reached_at = None
    twin = None
    for k in range(1, 12 + 900 + 1):
        g.step()
        assert o.step() == 0
        assert g.time() == o.time(), k
        assert g.has_reached(target) == o.has_reached(target), k
        if reached_at is None and g.has_reached(target):
            reached_at = k
        if k in (5, 40, 137, 500):                    # a snapshot: inside the start-up, then with steps queued
            sg, so = g.take_solution(), o.take_solution()
            for b in range(s.n):
                assert sg.info(b) == so.info(b), (k, b)
                assert np.array_equal(sg.coeffs(b)[1], so.coeffs(b)[1])
                assert_same_bits(sg.coeffs(b)[0], so.coeffs(b)[0], f"step {k} body {b}")
            assert g.time() == o.time()
        if k == 300:
            twin = g.clone()                          # Clone with steps queued: the clone resumes identically
    if s.n == 3:                                      # (the 32-body system's slowest spline needs 3600 steps per polynomial)
        assert reached_at is not None and reached_at > 300
    for _ in range(100):
        twin.step()
    g_state, o_state = g.state(), o.state()
    assert g_state[2:] == o_state[2:]
    assert_same_bits(g_state[0], o_state[0], "positions")
    assert_same_bits(g_state[1], o_state[1], "velocities")
    assert twin.state()[3] == 300 + 100
    sg, so = g.take_solution(), o.take_solution()
    for b in range(s.n):
        assert sg.info(b) == so.info(b)
        assert_same_bits(sg.coeffs(b)[0], so.coeffs(b)[0], f"final body {b}")


def test_borrowed_integration_view_sees_the_queued_steps(gpu):
    """eph_prop_integrator hands out a view of the propagator's Integration; eph_prop_step only QUEUES steady-state steps.
    Every eph_nbody_* call through the view runs the queue first: its state, its clone, and a bound set through it are
    those of a propagator that executed every step when it was asked to."""
    s = load_system("sun_earth_moon_2433282.5")
    g = gpu.NBodyPropagator.from_system(s)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    view = g.integration()
    g.step_n(20)
    for _ in range(7):
        g.step()                                       # queued
    for _ in range(27):
        assert o.step() == 0
    p, v, t, c = view.state()
    po, vo, to, co = o.state()
    assert (t, c) == (to, co) and c == 27
    assert_same_bits(p, po, "view positions") and assert_same_bits(v, vo, "view velocities")
    for _ in range(3):
        g.step()
    twin = view.clone()                                # an Integration clone taken through the view: 30 steps in
    assert twin.state()[3] == 30
    for _ in range(4):
        g.step()                                       # 34 queued / run
    view.set_bound(s.epoch + 35.5 * s.dt)              # the queued steps ran against the old bound first
    g.step()                                           # 35
    g.step()                                           # 36: t = 35 dt < bound
    with pytest.raises(gpu.StepError) as e:
        for _ in range(3):
            g.step()                                   # t = 36 dt >= bound: BoundReached, reported by the step itself
    assert e.value.status == gpu.BOUND_REACHED
    assert view.state()[3] == 36
