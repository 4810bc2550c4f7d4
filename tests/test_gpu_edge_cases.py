"""-m gpu: the reference's edge cases, driven ON THE DEVICE and compared with the CPU oracle bit for bit.

The device code has carried these branches since round 1; until round 5 only the oracle was tested on them. Each test
names the reference lines whose behaviour it pins:

  (a) StepError::StepSizeUnderflow   integration/src/multistep/mod.rs:201-210 (LinearMultistepIntegrator::advance),
                                     integration/src/runge_kutta/mod.rs:112-120 (FixedRungeKuttaIntegrator::advance, which
                                     is also the Substepper's inner integrator and the adaptive pair's inner one)
  (b) StepError::MaxIterationsReached  runge_kutta/mod.rs:414-419; `n` counts ATTEMPTS over the integrator's lifetime and
                                     is reset by SpacecraftPropagator::reset_integrator at a manoeuvre boundary
                                     (ephemeris/src/propagators/spacecraft.rs:479-485,598-609)
  (c) non-default AdaptiveMethodParams: the h_max clamp and fac_min / fac_max / fac of IController::step
                                     (runge_kutta/mod.rs:225-243), unequal position / velocity tolerances
                                     (ephemeris_explorer/src/dynamics/spacecraft.rs:609-641)
  (d) the exact-equality sampling trigger's FAILURE branch: `last_sample_time += delta; if last_sample_time ==
      sample_period` (ephemeris/src/propagators/nbody.rs:389-391) with a dt whose accumulated sum steps over the period:
      that body is never sampled again.

The craft tests run in this process on the wave-per-craft kernel (k_craft_wave: the batches are small) and again, in child
processes, on the thread-per-craft static kernel (k_craft_propagate) and the queue kernel (k_craft_queue)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SYSTEMS, load_system
from ephemeris_explorer_amd.systems import load_ship, parse_epoch
from oracle import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


def random_system(n, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(n, 3)) * 1e7, rng.normal(size=(n, 3)), rng.uniform(1.0, 1e5, size=n)


# body counts that reach every step-kernel family: k_lm_small (<= 32), k_lm_persistent (<= 64), k_lm_step (wave_force, <= 512),
# k_lm_step_wg<L,4> (<= 1024 targets), <L,8> (<= 2048), <L,16>
SIZES = [3, 32, 48, 200, 700, 1500, 2100]
TWO52, TWO53 = 2.0 ** 52, 2.0 ** 53


def _advance_both(gpu, g, o, k):
    """(status the product reports, status of the oracle) for advance(k)"""
    try:
        g.advance(k)
        sg = 0
    except gpu.StepError as e:
        sg = e.status
    return sg, o.advance(k)


def _assert_same_integration(g, o, what):
    pg, vg, tg, cg = g.state()
    po, vo, to, co = o.state()
    assert (tg, cg) == (to, co), f"{what}: time / step_count {(tg, cg)} vs {(to, co)}"
    assert same(pg, po) and same(vg, vo), f"{what}: state differs"
    assert g.eval_count() == o.eval_count(), f"{what}: eval_count {g.eval_count()} vs {o.eval_count()}"


# ------------------------------------------------------------------------------------------------------------------------
# (a) StepSizeUnderflow, massive bodies
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["QuinlanTremaine12", "Stormer13", "BlanesMoan6B"])
@pytest.mark.parametrize("n", SIZES)
def test_nbody_underflow_on_the_first_step(gpu, n, method):
    """time + h == time before anything moves (multistep/mod.rs:207-208, runge_kutta/mod.rs:118-119): Err, nothing evaluated,
    and the same again on the next call."""
    pos, vel, mu = random_system(n, 500 + n)
    g = gpu.NBodyIntegration(pos, vel, mu, 1e20, 1.0, method)
    o = orc.NBody(pos, vel, mu, 1e20, 1.0, method)
    for k in (1, 25):
        assert _advance_both(gpu, g, o, k) == (gpu.STEP_SIZE_UNDERFLOW, orc.STEP_SIZE_UNDERFLOW)
        _assert_same_integration(g, o, f"{method} n={n}")
    assert g.state()[2] == 1e20 and g.state()[3] == 0 and same(g.state()[0], pos)


@pytest.mark.parametrize("t0,h,why", [(TWO52, 1.0, "the first sub-step: t + h != t but t + h/4 == t"),
                                      (TWO52 - 1.0, 2.0, "the third sub-step of the first macro step (2^52 + 0.5 ties to even)"),
                                      (TWO52 - 9.0, 2.0, "the third sub-step of the FIFTH macro step")])
@pytest.mark.parametrize("n", SIZES)
def test_nbody_underflow_inside_the_starter(gpu, n, t0, h, why):
    """The multistep methods start through Substepper<4, BlanesMoan6B> (multistep/mod.rs:98-108,211-218): the main test
    passes (t + h != t) and one of the four sub-steps of h/4 fails in FixedRungeKuttaIntegrator::advance. The reference
    returns the error with the problem PARTLY advanced (the sub-steps before the failing one stay applied, time included),
    and a further call re-enters the start-up from there."""
    pos, vel, mu = random_system(n, 900 + n)
    g = gpu.NBodyIntegration(pos, vel, mu, t0, h)
    o = orc.NBody(pos, vel, mu, t0, h)
    sg, so = _advance_both(gpu, g, o, 30)
    assert so == orc.STEP_SIZE_UNDERFLOW, why
    assert sg == gpu.STEP_SIZE_UNDERFLOW
    _assert_same_integration(g, o, f"n={n} {why}")
    assert same(g.acc(), o.acc())
    sg, so = _advance_both(gpu, g, o, 1)          # and once more: the same error from the same place
    assert (sg, so) == (gpu.STEP_SIZE_UNDERFLOW, orc.STEP_SIZE_UNDERFLOW)
    _assert_same_integration(g, o, f"n={n} {why} (second call)")


@pytest.mark.parametrize("method", ["BlanesMoan6B", "Ruth", "McLachlanSS17"])
@pytest.mark.parametrize("n", SIZES)
def test_nbody_underflow_in_the_middle_of_a_call(gpu, n, method):
    """A fixed-step SRKN from t0 = 2^53 - 8 with h = 1: eight steps are representable, the ninth is not (2^53 + 1 ties to
    even). advance(20) stops after 8 with StepSizeUnderflow; time, step_count and state are those of the 8 steps."""
    pos, vel, mu = random_system(n, 1300 + n)
    g = gpu.NBodyIntegration(pos, vel, mu, TWO53 - 8.0, 1.0, method)
    o = orc.NBody(pos, vel, mu, TWO53 - 8.0, 1.0, method)
    assert _advance_both(gpu, g, o, 3) == (0, 0)
    assert _advance_both(gpu, g, o, 20) == (gpu.STEP_SIZE_UNDERFLOW, orc.STEP_SIZE_UNDERFLOW)
    _assert_same_integration(g, o, f"{method} n={n}")
    assert g.state()[2] == TWO53 and g.state()[3] == 8


def test_gang_with_underflowing_members(gpu):
    """eph_nbody_advance_many (one workgroup per system in k_lm_small): a healthy system, one that underflows at once, one
    that underflows inside the starter and one SRKN system that underflows after 8 steps. The call reports the error; every
    member is where its own separate advance would have left it = where the oracle is."""
    full = load_system("full_solar_system_2433282.5")
    sem = load_system("sun_earth_moon_2433282.5")
    specs = [(full, full.epoch, full.dt, "QuinlanTremaine12"), (full, 1e20, 1.0, "QuinlanTremaine12"),
             (sem, TWO52 - 1.0, 2.0, "QuinlanTremaine12"), (full, TWO53 - 8.0, 1.0, "BlanesMoan6B"),
             (sem, TWO52, 1.0, "Stormer13")]
    gang = [gpu.NBodyIntegration(s.pos, s.vel, s.mu, t0, h, m) for s, t0, h, m in specs]
    solo = [gpu.NBodyIntegration(s.pos, s.vel, s.mu, t0, h, m) for s, t0, h, m in specs]
    orcs = [orc.NBody(s.pos, s.vel, s.mu, t0, h, m) for s, t0, h, m in specs]
    with pytest.raises(gpu.StepError) as e:
        gpu.advance_many(gang, 40)
    assert e.value.status == gpu.STEP_SIZE_UNDERFLOW
    expect = [0, 1, 1, 1, 1]
    for g, s, o, want in zip(gang, solo, orcs, expect):
        sg, so = _advance_both(gpu, s, o, 40)
        assert (sg, so) == (want, want)
        _assert_same_integration(s, o, "solo")
        _assert_same_integration(g, o, "gang member")
    gpu.advance_many(gang[:1], 100)                    # the healthy member goes on
    assert orcs[0].advance(100) == 0
    _assert_same_integration(gang[0], orcs[0], "healthy member afterwards")


@pytest.mark.parametrize("n", [3, 32, 300])
def test_propagator_reports_the_integrator_underflow(gpu, n):
    """NBodyPropagator::step -> Integration::advance -> Err(StepSizeUnderflow) (nbody.rs:196-205, integration/src/lib.rs:
    496-503): no sample is taken, the solution stays empty, time() is the start."""
    pos, vel, mu = random_system(n, 77 + n)
    count, degree = np.full(n, 2, np.uint32), np.full(n, 5, np.uint32)
    for t0, h in ((1e20, 1.0), (TWO52, 1.0), (TWO52 - 9.0, 2.0)):
        g = gpu.NBodyPropagator(pos, vel, mu, t0, h, 1, count, degree)
        o = orc.Propagator(pos, vel, mu, t0, h, 1, count, degree)
        for k in (50, 1):
            with pytest.raises(gpu.StepError) as e:
                g.step_n(k)
            assert e.value.status == gpu.STEP_SIZE_UNDERFLOW
            so = 0
            for _ in range(k):
                so = o.step()
                if so:
                    break
            assert so == orc.STEP_SIZE_UNDERFLOW
            pg, vg, tg, cg = g.state()
            po, vo, to, co = o.state()
            assert (tg, cg) == (to, co) and same(pg, po) and same(vg, vo)
            assert g.time() == o.time() and g.integrator_time() == o.integrator_time()
            assert g.has_reached(t0) == o.has_reached(t0)
        sg, so_ = g.take_solution(), o.take_solution()
        for b in range(n):
            assert sg.info(b) == so_.info(b)
            cg_, ng = sg.coeffs(b)
            co_, no = so_.coeffs(b)
            assert np.array_equal(ng, no) and same(cg_, co_)


# ------------------------------------------------------------------------------------------------------------------------
# (d) the sampling trigger that stops firing
# ------------------------------------------------------------------------------------------------------------------------
def _trigger(dt, count):
    """the step on which `last_sample_time == sample_period` first holds, or 0 when the sum steps over it (never)"""
    s, period = 0.0, dt * float(count)
    for k in range(1, 4 * count + 8):
        s += dt
        if s == period:
            return k
        if s > period:
            return 0
    return 0


@pytest.mark.parametrize("dt,counts", [(0.1, [3]), (0.7, [10]), (0.7, [3, 10, 1, 7]), (0.1, [3, 6, 10, 1])])
@pytest.mark.parametrize("system", ["sun_earth_moon_2433282.5", "full_solar_system_2433282.5", "plummer300"])
@pytest.mark.parametrize("direction", [1, -1])
def test_sampling_trigger_with_a_step_that_is_not_representable(gpu, system, dt, counts, direction):
    """nbody.rs:389-391 compares an ACCUMULATED f64 sum with dt * count for equality. 0.1 * 3 is reached exactly
    (0.1 + 0.1 + 0.1 == 0.30000000000000004 == 0.1 * 3); 0.7 * 10 is stepped over (ten additions of 0.7 give
    6.999999999999999, the eleventh 7.699999999999999 > 7.0): such a body is never sampled, its spline stays empty,
    time() stays at the start and has_reached() stays false. Whatever the reference's arithmetic says per body, the
    product says the same: polynomials, time(), has_reached(), state."""
    if system == "plummer300":
        from ephemeris_explorer_amd.workloads import plummer
        pos, vel, mu = plummer(300)
        t0 = 0.0
    else:
        s = load_system(system)
        pos, vel, mu, t0 = s.pos, s.vel, s.mu, s.epoch
    n = len(mu)
    count = np.array([counts[b % len(counts)] for b in range(n)], dtype=np.uint32)
    degree = np.array([3 + b % 5 for b in range(n)], dtype=np.uint32)
    fires = [_trigger(dt, int(c)) for c in counts]
    if (dt, counts) == (0.1, [3]):
        assert fires == [3]
    if dt == 0.7:
        assert 0 in fires                                   # the scenario is what it claims: some body never samples
    g = gpu.NBodyPropagator(pos, vel, mu, t0, dt, direction, count, degree)
    o = orc.Propagator(pos, vel, mu, t0, dt, direction, count, degree)
    done = 0
    for k in (1, 30, 1, 200, 97):                           # single steps (deferred on the device) and batches
        g.step_n(k)
        for _ in range(k):
            assert o.step() == 0
        done += k
        assert g.time() == o.time() and g.integrator_time() == o.integrator_time()
        for probe in (t0, t0 + direction * dt * 8 * min(counts), t0 + direction * dt * done):
            assert g.has_reached(probe) == o.has_reached(probe)
    if 0 in fires:
        assert g.time() == t0 and not g.has_reached(t0 + direction * dt)
    pg, vg, tg, cg = g.state()
    po, vo, to, co = o.state()
    assert (tg, cg) == (to, co) and same(pg, po) and same(vg, vo)
    sg, so = g.take_solution(), o.take_solution()
    sampled = 0
    for b in range(n):
        assert sg.info(b) == so.info(b), f"body {b} (count {count[b]})"
        cg_, ng = sg.coeffs(b)
        co_, no = so.coeffs(b)
        assert np.array_equal(ng, no) and same(cg_, co_), f"body {b}"
        sampled += len(ng) > 0
        if _trigger(dt, int(count[b])) == 0:
            assert len(ng) == 0
    assert sampled == sum(1 for b in range(n) if _trigger(dt, int(count[b])))
    # and the run goes on after the hand-over (the interpolators keep their partial windows)
    g.step_n(150)
    for _ in range(150):
        assert o.step() == 0
    sg, so = g.take_solution(), o.take_solution()
    for b in range(n):
        assert sg.info(b) == so.info(b)
        assert same(sg.coeffs(b)[0], so.coeffs(b)[0])


# ------------------------------------------------------------------------------------------------------------------------
# the massless side: (a) underflow, (b) MaxIterationsReached, (c) non-default parameters
# ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def simple_system(gpu):
    """10-body 1950 system, QuinlanTremaine12 6 h, 60 days of ephemeris: on the GPU and in the oracle (bit-identical,
    test_gpu_parity.py)."""
    s = load_system("simple_solar_system_2433282.5")
    end = s.epoch + 60 * 86400.0
    g = gpu.NBodyPropagator.from_system(s)
    sol = g.propagate(end)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(end) == 0
    osol = o.take_solution()
    for b in range(s.n):
        assert sol.info(b) == osol.info(b)
    return s, sol, gpu.Ephemeris(sol, s.mu), osol


def _ship():
    return load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")


def _params(gpu, h_init=60.0, h_max=1.7976931348623157e308, tol_pos=1e-3, tol_vel=1e-3, fac_min=0.2, fac_max=5.0, fac=0.9,
            n_max=1_000_000):
    return gpu.AdaptiveParams(h_init, h_max, tol_pos, tol_vel, fac_min, fac_max, fac, n_max)


def _oracle_craft(osol, s, t0, pos, vel, method, p, burns=()):
    return orc.Craft(osol, s.mu, t0, pos, vel, method, h_init=p.h_init, h_max=p.h_max, tol_pos=p.tol_position,
                     tol_vel=p.tol_velocity, fac_min=p.fac_min, fac_max=p.fac_max, fac=p.fac, n_max=p.n_max, burns=burns)


def _fleet(ship, n, seed):
    rng = np.random.default_rng(seed)
    pos = ship.pos + rng.normal(0.0, 50.0, size=(n, 3))
    vel = ship.vel + rng.normal(0.0, 0.005, size=(n, 3))
    pos[0], vel[0] = ship.pos, ship.vel
    return pos, vel


def _compare_craft(batch, i, c, st_expected, what):
    """status, attempt counter n, accepted steps, time, state, next_h and every knot of craft i against the oracle craft"""
    st = batch.status()
    cs = c.state()
    assert st["status"][i] == st_expected, f"{what}: status {st['status'][i]} vs {st_expected}"
    assert st["attempts"][i] == cs["attempts"], f"{what}: n {st['attempts'][i]} vs {cs['attempts']}"
    assert st["steps"][i] == cs["steps"], f"{what}: steps {st['steps'][i]} vs {cs['steps']}"
    gs = batch.state()
    assert bits(gs["t"][i]) == bits(cs["t"]), f"{what}: time {gs['t'][i]!r} vs {cs['t']!r}"
    assert same(gs["pos"][i], cs["pos"]) and same(gs["vel"][i], cs["vel"]), f"{what}: state"
    assert bits(gs["next_h"][i]) == bits(cs["next_h"]), f"{what}: next_h {gs['next_h'][i]!r} vs {cs['next_h']!r}"
    kt, kp, kv = batch.knots(i)
    ot, op, ov = c.knots()
    assert len(kt) == len(ot) == st["nknots"][i], f"{what}: {len(kt)} vs {len(ot)} knots"
    assert same(kt, ot) and same(kp, op) and same(kv, ov), f"{what}: knots differ"


@pytest.mark.parametrize("method", ["Verner87", "DormandPrince54", "Fine45", "Tsitouras75"])
def test_craft_edge_underflow(gpu, simple_system, method):
    """runge_kutta/mod.rs:118-119 inside the adaptive pair (mod.rs:414-439). (1) h_init below half the spacing of doubles at
    t0: the first attempt is refused, n = 0, one knot (the initial one). (2) a tolerance no step can meet: every attempt is
    rejected, the controller shrinks h by fac_min each time, until time + h == time: StepSizeUnderflow after the same
    number of attempts, with the same next_h, and the state restored to the last accepted one. (3) healthy craft in the
    same batch are untouched by their neighbours' errors."""
    s, sol, eph, osol = simple_system
    ship = _ship()
    t0 = ship.start
    spacing = np.spacing(abs(t0))
    n = 5
    pos, vel = _fleet(ship, n, 11)
    end = t0 + 3600.0
    # (1)
    p = _params(gpu, h_init=spacing / 4.0)
    batch = gpu.SpacecraftBatch(eph, t0, pos, vel, method, p, max_knots=512)
    batch.propagate(end)
    for i in range(n):
        c = _oracle_craft(osol, s, t0, pos[i], vel[i], method, p)
        assert c.step_to(end) == orc.STEP_SIZE_UNDERFLOW
        _compare_craft(batch, i, c, gpu.STEP_SIZE_UNDERFLOW, f"{method} tiny h_init craft {i}")
        assert batch.status()["attempts"][i] == 0 and batch.status()["nknots"][i] == 1
    batch.propagate(end)                                    # again: still the same error, nothing moved
    c = _oracle_craft(osol, s, t0, pos[0], vel[0], method, p)
    assert c.step_to(end) == orc.STEP_SIZE_UNDERFLOW and c.step_to(end) == orc.STEP_SIZE_UNDERFLOW
    _compare_craft(batch, 0, c, gpu.STEP_SIZE_UNDERFLOW, f"{method} tiny h_init, second call")
    # (2)
    p = _params(gpu, tol_pos=1e-300, tol_vel=1e-300)
    batch = gpu.SpacecraftBatch(eph, t0, pos, vel, method, p, max_knots=512)
    batch.propagate(end)
    for i in range(n):
        c = _oracle_craft(osol, s, t0, pos[i], vel[i], method, p)
        assert c.step_to(end) == orc.STEP_SIZE_UNDERFLOW
        assert c.state()["attempts"] > 8
        _compare_craft(batch, i, c, gpu.STEP_SIZE_UNDERFLOW, f"{method} impossible tolerance craft {i}")
    # (2b) the same after some accepted steps: a burn of absurd size makes the error estimate explode later on
    p = _params(gpu)
    burns = [(t0 + 600.0, t0 + 700.0, [1e200, 0.0, 0.0], -1)]
    batch = gpu.SpacecraftBatch(eph, t0, pos[:2], vel[:2], method, p, [burns, []], max_knots=512)
    batch.propagate(end)
    c0 = _oracle_craft(osol, s, t0, pos[0], vel[0], method, p, burns)
    c1 = _oracle_craft(osol, s, t0, pos[1], vel[1], method, p)
    st0 = c0.step_to(end)
    assert st0 != 0 and c1.step_to(end) == 0
    _compare_craft(batch, 0, c0, st0, f"{method} craft with the absurd burn")
    _compare_craft(batch, 1, c1, 0, f"{method} its healthy neighbour")          # (3)


@pytest.mark.parametrize("n_max", [5, 40])
@pytest.mark.parametrize("method", ["Verner87", "DormandPrince54", "Fine45"])
def test_craft_edge_max_iterations(gpu, simple_system, method, n_max):
    """`if self.n > self.n_max { return Err(MaxIterationsReached) }` (runge_kutta/mod.rs:417-419): n counts attempts (accepted
    and rejected) since the integrator was created, so n_max + 1 attempts succeed and the next one fails -- but
    reset_integrator at every manoeuvre boundary (spacecraft.rs:479-485,606-609) makes a new integrator with n = 0. A
    timeline with several short segments therefore gets further than n_max attempts in total; the last, unbounded segment
    is where it stops. Also in two legs (stop before the failure, resume into it) and with a craft without burns beside it."""
    s, sol, eph, osol = simple_system
    ship = _ship()
    t0 = ship.start
    earth = s.names.index("Earth")
    burns = [(t0 + 150.0, t0 + 400.0, [2e-4, 1e-4, 0.0], earth), (t0 + 900.0, t0 + 1000.0, [0.0, -1e-4, 2e-5], -1),
             (t0 + 1000.0, t0 + 1700.0, [1e-5, 0.0, 0.0], earth)]
    n = 4
    pos, vel = _fleet(ship, n, 12)
    blist = [burns, [], burns[:1], burns]
    p = _params(gpu, n_max=n_max)
    end = t0 + 5 * 86400.0
    one = gpu.SpacecraftBatch(eph, t0, pos, vel, method, p, blist, max_knots=512)
    two = gpu.SpacecraftBatch(eph, t0, pos, vel, method, p, blist, max_knots=512)
    one.propagate(end)
    two.propagate(t0 + 500.0)
    two.propagate(end)
    totals = []
    for i in range(n):
        c = _oracle_craft(osol, s, t0, pos[i], vel[i], method, p, blist[i])
        assert c.step_to(end) == orc.MAX_ITERATIONS
        _compare_craft(one, i, c, gpu.MAX_ITERATIONS_REACHED, f"{method} n_max={n_max} craft {i}")
        _compare_craft(two, i, c, gpu.MAX_ITERATIONS_REACHED, f"{method} n_max={n_max} craft {i}, two legs")
        assert c.state()["attempts"] == n_max + 1                     # the counter of the LAST integrator
        totals.append(len(c.knots()[0]) - 1)
    if n_max == 5:
        assert totals[0] > n_max + 1 and totals[0] > totals[1]          # the resets bought the craft with burns extra steps
    # step by step (IncrementalPropagator::step): the error arrives on the same step
    steps = gpu.SpacecraftBatch(eph, t0, pos[:1], vel[:1], method, p, blist[:1], max_knots=512)
    c = _oracle_craft(osol, s, t0, pos[0], vel[0], method, p, blist[0])
    for k in range(totals[0] + 3):
        steps.step_n(1)
        so = c.step()
        assert steps.status()["status"][0] == so, f"step {k}"
        if so:
            break
    assert so == orc.MAX_ITERATIONS
    _compare_craft(steps, 0, c, gpu.MAX_ITERATIONS_REACHED, f"{method} n_max={n_max} single steps")


PARAM_SETS = {
    "h_max_300": dict(h_max=300.0),
    "h_max_100": dict(h_max=100.0),
    "h_max_300_tight": dict(h_max=300.0, tol_pos=1e-6, tol_vel=1e-6),
    "factors": dict(fac_min=0.5, fac_max=2.0, fac=0.8),
    "tol_pos_lt_vel": dict(tol_pos=1e-6, tol_vel=1.0),
    "tol_vel_lt_pos": dict(tol_pos=1.0, tol_vel=1e-6),
    "loose": dict(tol_pos=1.0, tol_vel=1.0),
    "everything": dict(h_init=7.5, h_max=300.0, tol_pos=1e-6, tol_vel=1e-4, fac_min=0.5, fac_max=2.0, fac=0.8),
    "h_init_above_h_max": dict(h_init=900.0, h_max=120.0),
}


@pytest.mark.parametrize("which", sorted(PARAM_SETS))
@pytest.mark.parametrize("method", ["Verner87", "DormandPrince54", "Fine45"])
def test_craft_edge_non_default_params(gpu, simple_system, method, which):
    """IController::step with every parameter away from the app's INITIAL_ADAPTIVE_PARAMS (runge_kutta/mod.rs:225-243;
    load/mod.rs:472-486): the h_max clamp active on every step (a low orbit wants ~60-900 s), fac_min / fac_max / fac,
    AbsTol with different position and velocity tolerances (dynamics/spacecraft.rs:609-641), tolerances 1e-6 and 1.0, an
    h_init above h_max (the clamp acts on the controller's output only: the first attempt is h_init). Knots bit for bit,
    with burns (one in a TNB frame), in two legs."""
    s, sol, eph, osol = simple_system
    ship = _ship()
    t0 = ship.start
    earth = s.names.index("Earth")
    burns = [(t0 + 1800.0, t0 + 1890.0, [2e-4, 1e-4, 0.0], earth), (t0 + 40000.0, t0 + 40100.0, [0.0, 1e-4, 0.0], -1)]
    n = 6
    pos, vel = _fleet(ship, n, 13)
    blist = [burns if i % 2 == 0 else [] for i in range(n)]
    p = _params(gpu, **PARAM_SETS[which])
    mid, end = t0 + 20000.0, t0 + 86400.0
    batch = gpu.SpacecraftBatch(eph, t0, pos, vel, method, p, blist, max_knots=8192)
    batch.propagate(mid)
    batch.propagate(end)
    assert (batch.status()["status"] == 0).all()
    for i in range(n):
        c = _oracle_craft(osol, s, t0, pos[i], vel[i], method, p, blist[i])
        assert c.step_to(mid) == 0 and c.step_to(end) == 0
        _compare_craft(batch, i, c, 0, f"{method} {which} craft {i}")
    d = np.diff(batch.knots(1)[0])
    if "h_max" in PARAM_SETS[which]:
        assert d[1:].max() <= PARAM_SETS[which]["h_max"]            # (the first attempt is h_init whatever h_max says)
    # the clamp was live on most steps, not just present (at 1e-3 this orbit wants ~400 s from Verner87, ~155 s from the 5(4) pairs)
    if which in ("h_max_100", "h_init_above_h_max") or (which == "h_max_300" and method == "Verner87"):
        assert (d == PARAM_SETS[which]["h_max"]).sum() > len(d) // 2


@pytest.mark.parametrize("form", ["thread-static", "thread-queue", "thread-static-undealt"])
def test_craft_edge_cases_on_the_other_sweep_kernels(gpu, form):
    """The craft tests above run on k_craft_wave (small batches). The same tests again on the thread-per-craft kernels:
    k_craft_propagate (static; craft dealt to the lanes, and craft i on lane i) and k_craft_queue (persistent grid + work
    queue). The kernel form is read once per process, hence child processes."""
    env = dict(os.environ, EPH_CRAFT_FORM="thread", EPH_CRAFT_QUEUE="1" if form == "thread-queue" else "0",
               EPH_CRAFT_SORT="0" if form.endswith("undealt") else "1")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_edge_cases.py"), "-q", "-x", "-m", "gpu",
                        "-k", "craft_edge and not other_sweep_kernels"], env=env, cwd=str(ROOT), capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
