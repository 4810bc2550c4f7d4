"""-m gpu: seeded differential runs -- the HIP path against the CPU oracle on scenarios nobody wrote by hand. Every scenario is drawn
from a fixed seed (the failures, if any, are reproducible by their seed), every comparison is on bit patterns.

  * massive bodies: random body counts across the kernel families, every fixed-step method, positive and negative steps, advance()
    in random chunks, a bound somewhere inside (StepError::BoundReached, multistep/mod.rs:201-203), clones;
  * spacecraft: random embedded pair, tolerances over twelve decades (unequal for position and velocity), h_init, h_max, controller
    factors, n_max small enough to trip now and then, up to four burns in inertial and body-relative TNB frames, propagation in
    random legs -- status, attempt counter, next_h, state and every knot (runge_kutta/mod.rs:225-243,414-439; spacecraft.rs:598-615)."""
import os

import numpy as np
import pytest

from conftest import SYSTEMS, load_system
from ephemeris_explorer_amd.systems import load_ship
from oracle import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


FIXED_METHODS = ["QuinlanTremaine12", "Stormer13", "BlanesMoan6B", "BlanesMoan11B", "BlanesMoan14A", "ForestRuth", "McLachlanO4",
                 "McLachlanSS17", "Pefrl", "Ruth"]


def _seeds(var, default):
    first, count = (int(x) for x in os.environ.get(var, default).split(":"))
    return range(first, first + count)


@pytest.mark.parametrize("seed", _seeds("EPH_FUZZ_NBODY_SEEDS", "0:40"))      # ("first:count": wider draws for soak runs)
def test_nbody_random_scenarios(gpu, seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([2, 3, 5, 17, 32, 33, 47, 64, 65, 130, 300, 513, 700, 1100, 2050]))
    method = FIXED_METHODS[int(rng.integers(len(FIXED_METHODS)))]
    pos = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(3, 8)
    vel = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 1)
    mu = 10.0 ** rng.uniform(-3, 6, size=n)
    if rng.random() < 0.3:
        mu[rng.integers(n)] = 0.0                          # a massless body among the massive ones
    h = float(10.0 ** rng.uniform(-2, 3)) * (1.0 if rng.random() < 0.7 else -1.0)
    t0 = float(rng.normal() * 1e8)
    g = gpu.NBodyIntegration(pos, vel, mu, t0, h, method)
    o = orc.NBody(pos, vel, mu, t0, h, method, native=n > 300)
    total = int(rng.integers(14, 60 if n <= 700 else 24))
    bound_at = int(rng.integers(3, total + 10)) if (h > 0 and rng.random() < 0.4) else None
    if bound_at is not None:
        b = t0 + (bound_at + 0.5) * h
        g.set_bound(b)
        o.set_bound(b)
    done, twin = 0, None
    while done < total:
        k = int(min(total - done, rng.choice([1, 1, 2, 3, 5, 8, 13])))
        try:
            g.advance(k)
            sg = 0
        except gpu.StepError as e:
            sg = e.status
        so = o.advance(k)
        assert sg == so, f"seed {seed}: status {sg} vs {so} after {done} steps ({method}, n={n})"
        pg, vg, tg, cg = g.state()
        po, vo, to, co = o.state()
        assert (tg, cg) == (to, co) and same(pg, po) and same(vg, vo), f"seed {seed}: state after {done}+{k} steps ({method}, n={n})"
        if sg:
            assert sg == gpu.BOUND_REACHED and bound_at is not None
            break
        done += k
        if twin is None and done >= total // 2 and (n <= 64 or n > 512 or True):
            twin = (g.clone(), o.clone())
    assert same(g.acc(), o.acc()) and g.eval_count() == o.eval_count()
    if twin is not None:                                   # the clone resumes exactly like the original did
        tg_, to_ = twin
        try:
            tg_.advance(3)
            s1 = 0
        except gpu.StepError as e:
            s1 = e.status
        assert s1 == to_.advance(3)
        assert same(tg_.state()[0], to_.state()[0]) and same(tg_.state()[1], to_.state()[1])


@pytest.mark.parametrize("seed", _seeds("EPH_FUZZ_PROP_SEEDS", "0:20"))
def test_propagator_random_scenarios(gpu, seed):
    """NBodyPropagator with the SplineInterpolators solout (nbody.rs:65-235,371-400; celestial.rs:19-135): random systems, sample
    counts and polynomial degrees per body, both directions, the two multistep methods, step / step_n / step_to / take_solution in
    random order -- polynomials, spline starts, time(), has_reached(), state."""
    rng = np.random.default_rng(8000 + seed)
    n = int(rng.choice([2, 3, 9, 32, 40, 64, 70, 200]))
    pos = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(4, 8)
    vel = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-1, 1)
    mu = 10.0 ** rng.uniform(0, 6, size=n)
    dt = float(rng.choice([0.25, 1.0, 60.0, 600.0, 900.0, 3600.0]))
    direction = 1 if rng.random() < 0.6 else -1
    count = rng.choice([1, 2, 3, 4, 6, 7, 12], size=n).astype(np.uint32)
    degree = rng.integers(2, 8, size=n).astype(np.uint32)
    method = "QuinlanTremaine12" if rng.random() < 0.7 else "Stormer13"
    t0 = float(rng.integers(-10**9, 10**9))
    g = gpu.NBodyPropagator(pos, vel, mu, t0, dt, direction, count, degree, method)
    o = orc.Propagator(pos, vel, mu, t0, dt, direction, count, degree, method)

    def compare_solutions(sg, so, what):
        for b in range(n):
            assert sg.info(b) == so.info(b), f"seed {seed} {what}: body {b} info {sg.info(b)} vs {so.info(b)}"
            cg, ng = sg.coeffs(b)
            co, no = so.coeffs(b)
            assert np.array_equal(ng, no) and same(cg, co), f"seed {seed} {what}: body {b} polynomials"

    steps = 0
    for _ in range(int(rng.integers(4, 9))):
        op = rng.choice(["step", "step_n", "step_to", "take"])
        if op == "step":
            for _ in range(int(rng.integers(1, 4))):
                g.step()
                assert o.step() == 0
                steps += 1
        elif op == "step_n":
            k = int(rng.integers(1, 400 if n <= 70 else 60))
            g.step_n(k)
            assert o.step_n(k) == 0
            steps += k
        elif op == "step_to":
            target = o.time() + direction * dt * float(rng.uniform(0, 8 * 12 * 2))
            if steps > 8 * int(count.max()) and np.isfinite(target):
                g.step_to(target)
                assert o.step_to(target) == 0
        else:
            compare_solutions(g.take_solution(), o.take_solution(), f"take after {steps} steps")
        assert g.time() == o.time() and g.integrator_time() == o.integrator_time(), f"seed {seed}: time after {op}"
        probe = o.time() + direction * dt * float(rng.uniform(-3, 3))
        assert g.has_reached(probe) == o.has_reached(probe)
    pg, vg, tg, cg = g.state()
    po, vo, to, co = o.state()
    assert (tg, cg) == (to, co) and same(pg, po) and same(vg, vo), f"seed {seed}: final state"
    compare_solutions(g.take_solution(), o.take_solution(), "final take")


@pytest.fixture(scope="module")
def simple_system(gpu):
    s = load_system("simple_solar_system_2433282.5")
    end = s.epoch + 45 * 86400.0
    sol = gpu.NBodyPropagator.from_system(s).propagate(end)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(end) == 0
    return s, gpu.Ephemeris(sol, s.mu), o.take_solution()


PAIRS = ["CashKarp45", "DormandPrince54", "DormandPrince87", "Fehlberg45", "Tsitouras75", "Verner87", "Verner98", "Fine45"]


# EPH_FUZZ_CRAFT_SEEDS="first:count" widens the draw for soak runs (default 0:64); EPH_CRAFT_FORM / EPH_CRAFT_QUEUE / EPH_CRAFT_SORT pick the
# sweep kernel as everywhere (the default for these one-to-five-craft batches is the wave-per-craft kernel)
_FIRST, _COUNT = (int(x) for x in os.environ.get("EPH_FUZZ_CRAFT_SEEDS", "0:64").split(":"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + _COUNT))
def test_craft_random_scenarios(gpu, simple_system, seed):
    s, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    rng = np.random.default_rng(9000 + seed)
    method = PAIRS[int(rng.integers(len(PAIRS)))]
    t0 = ship.start + float(rng.uniform(0, 5 * 86400.0))
    n = int(rng.integers(1, 6))
    pos = ship.pos + rng.normal(0.0, 10.0 ** rng.uniform(0, 3.5), size=(n, 3))
    vel = ship.vel + rng.normal(0.0, 10.0 ** rng.uniform(-4, -1), size=(n, 3))
    p = gpu.AdaptiveParams(float(10.0 ** rng.uniform(-1, 3.3)),                                    # h_init
                           float(10.0 ** rng.uniform(1.5, 5)) if rng.random() < 0.5 else 1.7976931348623157e308,   # h_max
                           float(10.0 ** rng.uniform(-9, 3)), float(10.0 ** rng.uniform(-9, 3)),      # tolerances
                           float(rng.uniform(0.05, 0.9)), float(rng.uniform(1.1, 10.0)), float(rng.uniform(0.5, 0.99)),
                           int(rng.choice([3, 17, 60, 400, 1_000_000])))
    earth, sun = s.names.index("Earth"), s.names.index("Sun")
    blist = []
    for i in range(n):
        burns, t = [], t0
        for _ in range(int(rng.integers(0, 5))):
            t += float(10.0 ** rng.uniform(1.5, 4.3))
            dur = float(10.0 ** rng.uniform(0.5, 3))
            acc = (rng.normal(size=3) * 10.0 ** rng.uniform(-6, -3)).tolist()
            burns.append((t, t + dur, acc, int(rng.choice([-1, earth, sun]))))
            t += dur * float(rng.choice([1.0, 1.0, 3.0]))                # back-to-back burns now and then
        blist.append(burns)
    span = float(10.0 ** rng.uniform(3, 5.2))
    legs = sorted(set([t0 + span] + [t0 + span * float(x) for x in rng.random(int(rng.integers(0, 3)))]))
    batch = gpu.SpacecraftBatch(eph, t0, pos, vel, method, p, blist, max_knots=30000)
    crafts = [orc.Craft(osol, s.mu, t0, pos[i], vel[i], method, h_init=p.h_init, h_max=p.h_max, tol_pos=p.tol_position,
                        tol_vel=p.tol_velocity, fac_min=p.fac_min, fac_max=p.fac_max, fac=p.fac, n_max=p.n_max, burns=blist[i])
              for i in range(n)]
    ost = [0] * n
    for t_end in legs:
        batch.propagate(t_end)
        for i, c in enumerate(crafts):
            if ost[i] == 0:
                ost[i] = c.step_to(t_end)
    st, gs = batch.status(), batch.state()
    for i, c in enumerate(crafts):
        cs = c.state()
        what = f"seed {seed} craft {i} ({method}, tol {p.tol_position:.1e}/{p.tol_velocity:.1e}, n_max {p.n_max}, {len(blist[i])} burns)"
        if st["status"][i] == gpu.KNOTS_FULL:          # (a soak seed whose craft takes > 30 000 steps, e.g. 4617: the slab's knots are the
            kt, kp, kv = batch.knots(i)                #  oracle's first 30 000 -- the test does not drain slabs)
            ot, op, ov = c.knots()
            assert len(kt) == 30000 < len(ot) and same(kt, ot[:30000]) and same(kp, op[:30000]) and same(kv, ov[:30000]), what
            continue
        assert st["status"][i] == ost[i], f"{what}: status {st['status'][i]} vs {ost[i]}"
        assert st["attempts"][i] == cs["attempts"] and st["steps"][i] == cs["steps"], what
        assert bits(gs["t"][i]) == bits(cs["t"]) and bits(gs["next_h"][i]) == bits(cs["next_h"]), what
        assert same(gs["pos"][i], cs["pos"]) and same(gs["vel"][i], cs["vel"]), what
        kt, kp, kv = batch.knots(i)
        ot, op, ov = c.knots()
        assert len(kt) == len(ot) and same(kt, ot) and same(kp, op) and same(kv, ov), what
