"""-m gpu: the massless sweep (one device thread per spacecraft) against the CPU oracle, through the C ABI.

Every f64 operation on this path is performed in the oracle's order; the single libm call of the reference, powf in
the step-size controller (integration/src/runge_kutta/mod.rs:239), is evaluated correctly rounded by the same
double-double sequence on both sides. Knots (times, positions, velocities) must be bit-identical."""
import numpy as np
import pytest

from conftest import SYSTEMS, load_system
from ephemeris_explorer_amd.systems import load_ship, parse_epoch
from oracle import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def simple_system(gpu):
    """10-body 1950 system, QuinlanTremaine12 6 h, two years of ephemeris: on the GPU and in the oracle (bit-identical,
    test_gpu_parity.py), like ephemeris/tests/spacecraft_propagation.rs:401-409."""
    s = load_system("simple_solar_system_2433282.5")
    end = parse_epoch("1952-01-01 00:00:00")
    g = gpu.NBodyPropagator.from_system(s)
    sol = g.propagate(end)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(end) == 0
    osol = o.take_solution()
    for b in range(s.n):
        assert sol.info(b) == osol.info(b)
    return s, sol, gpu.Ephemeris(sol, s.mu), osol


def ship_burns(ship, names):
    return [(b.start, b.start + b.duration, b.acceleration, names.index(b.reference) if b.reference else -1)
            for b in ship.burns]


def compare_knots(g, o, what):
    gt, gp, gv = g
    ot, op, ov = o
    assert len(gt) == len(ot), f"{what}: {len(gt)} vs {len(ot)} knots"
    exact = (np.array_equal(bits(gt), bits(ot)) and np.array_equal(bits(gp), bits(op)) and
             np.array_equal(bits(gv), bits(ov)))
    if not exact:
        first = int(np.argmax((bits(gt) != bits(ot)) | (bits(gp) != bits(op)).any(axis=1)))
        raise AssertionError(f"{what}: knots differ from #{first}: t {gt[first]!r} vs {ot[first]!r}")
    return exact


def test_mars_transfer_scenario(gpu, simple_system):
    """The reference's own spacecraft test re-expressed on committed fixtures: Verner87, tol 1e-3, four burns in the
    Earth / Sun / Mars TNB frames; asserts of ephemeris/tests/spacecraft_propagation.rs:476-480."""
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    burns = ship_burns(ship, s.names)
    end = parse_epoch("1951-01-01 00:00:00")
    params = gpu.AdaptiveParams.default(ship.tolerance)
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], ship.integrator, params, [burns], max_knots=20000)
    batch.propagate(end)
    st = batch.status()
    assert st["status"][0] == 0
    c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, ship.integrator, tol_pos=ship.tolerance,
                  tol_vel=ship.tolerance, burns=burns)
    assert c.step_to(end) == 0
    assert st["steps"][0] == c.state()["steps"] and st["attempts"][0] == c.state()["attempts"]
    kt, kp, kv = batch.knots(0)
    assert compare_knots((kt, kp, kv), c.knots(), "Mars transfer"), "knots are expected to be bit-identical"
    gs = batch.state()
    assert gs["t"][0] == c.state()["t"] and gs["next_h"][0] == c.state()["next_h"]

    def distance(body, when):
        t = parse_epoch(when)
        p, _, inside = gpu.hermite_eval(kt, kp, kv, [t])
        bp, _, ins2 = sol.eval(s.names.index(body), [t], with_velocity=False)
        assert inside[0] and ins2[0]
        return np.linalg.norm(p[0] - bp[0])

    assert distance("Earth", "1950-01-01 00:00:00") < 10_000.0
    assert distance("Earth", "1950-01-01 00:15:00") < 10_000.0
    assert distance("Mars", "1950-07-27 15:45:00") < 10_000.0       # enters and stays in Mars' orbit
    assert distance("Mars", "1951-01-01 00:00:00") < 10_000.0


@pytest.mark.parametrize("method", ["CashKarp45", "DormandPrince54", "DormandPrince87", "Fehlberg45", "Tsitouras75",
                                    "Verner87", "Verner98", "Fine45"])
def test_every_embedded_pair(gpu, simple_system, method):
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    burns = ship_burns(ship, s.names)[:2]
    end = ship.start + 3 * 86400.0
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], method, gpu.AdaptiveParams.default(1e-3),
                                [burns], max_knots=20000)
    batch.propagate(end)
    assert batch.status()["status"][0] == 0
    c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, method, burns=burns)
    assert c.step_to(end) == 0
    assert compare_knots(batch.knots(0), c.knots(), method)


def test_erkn_tsitouras75nystrom(gpu, simple_system):
    """ERKN (integration/src/runge_kutta/nystrom/explicit.rs; Tsitouras75Nystrom): y'' = f(t, y), i.e. the spacecraft
    model with inertial-frame burns only. Runs the ERKNG kernels with the velocity-stage matrix zeroed; bit-identical
    to the oracle's separate restatement of ERKN::advance. Both kernel forms (one wave per craft / thread per craft)."""
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    relative = ship_burns(ship, s.names)[:2]
    inertial = [(b0, b1, acc, -1) for b0, b1, acc, _ in relative]
    end = ship.start + 3 * 86400.0
    with pytest.raises(gpu.EphemerisError):                    # a TNB burn needs the velocity: not a SecondOrderODE
        gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "Tsitouras75Nystrom",
                            gpu.AdaptiveParams.default(1e-3), [relative], max_knots=64)
    for n in (1, 13000):                                       # wave form, thread form
        pos = np.repeat(ship.pos[None], n, 0) + np.arange(n)[:, None] * 1e-3
        vel = np.repeat(ship.vel[None], n, 0)
        batch = gpu.SpacecraftBatch(eph, ship.start, pos, vel, "Tsitouras75Nystrom", gpu.AdaptiveParams.default(1e-3),
                                    [inertial] * n, max_knots=4096)
        batch.propagate(end)
        assert (batch.status()["status"] == 0).all()
        for i in sorted({0, n - 1}):
            c = orc.Craft(osol, s.mu, ship.start, pos[i], vel[i], "Tsitouras75Nystrom", burns=inertial)
            assert c.step_to(end) == 0
            assert compare_knots(batch.knots(i), c.knots(), f"ERKN craft {i} of {n}")


def test_batch_of_perturbed_craft_and_resume(gpu, simple_system):
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    rng = np.random.default_rng(20260926)
    n = 96
    pos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
    vel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
    burns = [ship_burns(ship, s.names)[:1] if i % 3 == 0 else [] for i in range(n)]
    batch = gpu.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", burns=burns, max_knots=8192)
    mid, end = ship.start + 2 * 86400.0, ship.start + 5 * 86400.0   # low Earth orbit: ~50 accepted steps per revolution
    batch.propagate(mid)            # step_to(mid), then resume to `end`: same knots as going straight through
    batch.propagate(end)
    st = batch.status()
    assert (st["status"] == 0).all()
    exact = 0
    for i in range(n):
        c = orc.Craft(osol, s.mu, ship.start, pos[i], vel[i], "Verner87", burns=burns[i])
        assert c.step_to(mid) == 0 and c.step_to(end) == 0
        exact += compare_knots(batch.knots(i, st["nknots"][i]), c.knots(), f"craft {i}")
    assert exact == n


def test_errors_are_per_craft_values(gpu, simple_system):
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    # craft 1 runs past the end of the ephemeris -> EvalFailed (spacecraft.rs:264-281); craft 0 is fine;
    # craft 2 fills its knot slab
    far = parse_epoch("1953-01-01 00:00:00")
    burns = ship_burns(ship, s.names)          # the transfer leaves Earth orbit, so steps grow to hours
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos] * 2, [ship.vel] * 2, "Verner87", burns=[burns, burns],
                                max_knots=100000)
    batch.propagate(ship.start + 86400.0)
    assert (batch.status()["status"] == 0).all()
    batch.propagate(far)
    assert (batch.status()["status"] == gpu.EVAL_FAILED).all()
    c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, "Verner87", burns=burns)
    assert c.step_to(far) == orc.EVAL_FAILED
    assert batch.status()["nknots"][0] == len(c.knots()[0])
    assert compare_knots(batch.knots(0), c.knots(), "up to the failure")
    small = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "Verner87", max_knots=16)
    small.propagate(ship.start + 86400.0)
    assert small.status()["status"][0] == gpu.KNOTS_FULL and small.status()["nknots"][0] == 16
    with pytest.raises(gpu.EphemerisError):
        gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "RK4")          # no embedded pair


def test_hermite_eval_kernel(gpu):
    rng = np.random.default_rng(4)
    t = np.cumsum(rng.uniform(1.0, 100.0, 50))
    p, v = rng.normal(size=(50, 3)) * 1e6, rng.normal(size=(50, 3))
    at = np.concatenate([t, rng.uniform(t[0] - 10, t[-1] + 10, 500), [t[0], t[-1]]])
    op, ov, inside = gpu.hermite_eval(t, p, v, at)
    for k, x in enumerate(at):
        r = orc.hermite_eval(t, p, v, x)
        assert (r is not None) == bool(inside[k])
        if r is not None:
            assert np.array_equal(bits(op[k]), bits(r[0])) and np.array_equal(bits(ov[k]), bits(r[1]))


def test_controller_pow_is_the_same_correctly_rounded_value(gpu, hooks):
    rng = np.random.default_rng(8)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-16), np.log(1e10), 200000)), [0.0, 1.0, np.inf, 1e-300, 5e-324]])
    for k in (4, 5, 7, 8):
        y = -(1.0 / k)
        dev = hooks.debug_pow(x, y)
        ref = np.array([orc.cr_pow(v, y) for v in x])
        assert np.array_equal(bits(dev), bits(ref))


def test_headless_cli_on_a_reference_system(gpu, tmp_path, capsys):
    """system directory -> +-ephemeris -> ships, plus a state.json export that loads back."""
    import json
    from ephemeris_explorer_amd import cli
    from ephemeris_explorer_amd.systems import load_system as ls
    out_state = tmp_path / "state.json"
    rc = cli.main([str(SYSTEMS / "sun_earth_moon_2433282.5"), "--years", "0.2", "--export-state",
                   "1950-02-01 00:00:00", str(out_state)])
    assert rc == 0
    rep = json.loads(capsys.readouterr().out)
    assert rep["bodies"] == 3 and rep["forward"]["polynomials"] > 0 and rep["backward"]["polynomials"] > 0
    ship = rep["ships"][0]
    assert ship["name"] == "Earth Station" and ship["status"] == 0 and ship["knots"] > 10
    (tmp_path / "ephemeris.json").write_text((SYSTEMS / "sun_earth_moon_2433282.5" / "ephemeris.json").read_text())
    back = ls(tmp_path)
    assert back.n == 3 and back.epoch == parse_epoch("1950-02-01 00:00:00")
    assert np.linalg.norm(back.pos[1] - back.pos[0]) > 1.4e8        # Earth is 1 au from the Sun


def test_spacecraft_solout_events(gpu, simple_system):
    """The app's SpacecraftSolout (dynamics/spacecraft.rs:514-587): SOI transitions and apsides of the whole Mars
    transfer, bit-identical to the oracle (times, distances, bodies, kinds); the event search runs on the steps of
    each propagate call, so a propagation in two legs gives the same lists."""
    from ephemeris_explorer_amd.systems import soi_radii
    s, sol, eph, osol = simple_system
    soi = soi_radii(s)
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    burns = ship_burns(ship, s.names)
    end = parse_epoch("1951-01-01 00:00:00")
    params = gpu.AdaptiveParams.default(ship.tolerance)
    c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, ship.integrator, tol_pos=ship.tolerance,
                  tol_vel=ship.tolerance, burns=burns, soi_radius=soi)
    assert c.step_to(end) == 0
    ott, otb = c.transitions()
    oat, oad, oab, oak = c.apsides()
    names = [s.names[b] for b in otb]
    assert names == ["Earth", "Sun", "Mars"] and len(oat) > 1000     # parking orbit, cruise, Mars orbit
    snap = None
    for legs in ([end], [ship.start + 40 * 86400.0, parse_epoch("1950-07-27 00:00:00"), end], "resume a clone"):
        if legs == "resume a clone":                      # Clone + resume (prediction.rs:224-229,378)
            batch = snap
            legs = [parse_epoch("1950-07-27 00:00:00"), end]
            for leg in legs:
                batch.propagate(leg)
            assert batch.status()["status"][0] == 0
            ntr, nap, est = batch.event_counts()
            (tt, tb), (at, ad, ab, ak) = batch.events(0)
            assert np.array_equal(bits(tt), bits(ott)) and np.array_equal(bits(at), bits(oat)) and np.array_equal(ak, oak)
            assert compare_knots(batch.knots(0), c.knots(), "resumed clone")
            continue
        batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], ship.integrator, params, [burns],
                                    max_knots=20000).enable_events(soi, max_transitions=16, max_apsides=4096)
        for k, leg in enumerate(legs):
            batch.propagate(leg)
            if len(legs) == 3 and k == 0:
                snap = batch.clone()                      # taken after the first leg, resumed later
        assert batch.status()["status"][0] == 0
        ntr, nap, est = batch.event_counts()
        assert est[0] == 0 and ntr[0] == len(ott) and nap[0] == len(oat)
        (tt, tb), (at, ad, ab, ak) = batch.events(0)
        assert np.array_equal(bits(tt), bits(ott)) and np.array_equal(tb, otb)
        assert np.array_equal(bits(at), bits(oat)) and np.array_equal(bits(ad), bits(oad))
        assert np.array_equal(ab, oab) and np.array_equal(ak, oak)


def test_event_slab_overflow_is_reported(gpu, simple_system):
    from ephemeris_explorer_amd.systems import soi_radii
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    params = gpu.AdaptiveParams.default(ship.tolerance)
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos] * 3, [ship.vel] * 3, ship.integrator, params, None,
                                max_knots=4096).enable_events(soi_radii(s), max_transitions=4, max_apsides=5)
    batch.propagate(ship.start + 86400.0)                            # ~15 revolutions of the parking orbit
    ntr, nap, est = batch.event_counts()
    assert list(est) == [7, 7, 7] and list(nap) == [4, 4, 4] and list(ntr) == [1, 1, 1]   # full = < 2 free entries
    assert list(batch.status()["status"]) == [0, 0, 0]               # the trajectories themselves are complete
    # drain and resume: read, reset_events, propagate again (no new steps: the search continues over the stored
    # knots); the concatenated apsides equal the oracle's
    c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, ship.integrator, tol_pos=ship.tolerance,
                  tol_vel=ship.tolerance, soi_radius=soi_radii(s))
    assert c.step_to(ship.start + 86400.0) == 0
    oat, oad, oab, oak = c.apsides()
    at_all, ad_all = [], []
    for _ in range(100):
        (tt, tb), (at, ad, ab, ak) = batch.events(0)
        at_all.append(at); ad_all.append(ad)
        if batch.event_counts()[2][0] == 0:
            break
        batch.reset_events()
        batch.propagate(ship.start + 86400.0)
    assert np.array_equal(bits(np.concatenate(at_all)), bits(oat))
    assert np.array_equal(bits(np.concatenate(ad_all)), bits(oad))
    assert batch.event_counts()[0][0] == 1 and np.array_equal(batch.events(0)[0][1], c.transitions()[1][-1:])


def test_draining_the_knot_slab(gpu, simple_system):
    """A long propagation through a small slab: propagate -> read knots -> reset_knots, repeated. The stitched
    pieces (each starts with the previous piece's last knot) and the events equal the oracle's single run."""
    from ephemeris_explorer_amd.systems import soi_radii
    s, sol, eph, osol = simple_system
    soi = soi_radii(s)
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    end = ship.start + 2 * 86400.0                      # no burns: the parking orbit, ~210 knots per day
    c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, "Verner87", soi_radius=soi)
    assert c.step_to(end) == 0
    ot, op, ov = c.knots()
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "Verner87", max_knots=100).enable_events(soi, 16, 4096)
    ts, ps, vs, rounds = [], [], [], 0
    while True:
        batch.propagate(end)
        st = batch.status()["status"][0]
        kt, kp, kv = batch.knots(0)
        first = 0 if not ts else 1                          # knot 0 repeats the previous piece's last knot
        if ts:
            assert kt[0] == ts[-1][-1]
        ts.append(kt[first:]); ps.append(kp[first:]); vs.append(kv[first:])
        rounds += 1
        if st == 0:
            break
        assert st == gpu.KNOTS_FULL and len(kt) == 100
        batch.reset_knots()
        assert batch.status()["nknots"][0] == 1 and batch.status()["status"][0] == 0
    assert rounds > 3
    assert compare_knots((np.concatenate(ts), np.concatenate(ps), np.concatenate(vs)), (ot, op, ov), "stitched")
    (tt, tb), (at, ad, ab, ak) = batch.events(0)
    ott, otb = c.transitions()
    oat, oad, oab, oak = c.apsides()
    assert np.array_equal(bits(tt), bits(ott)) and np.array_equal(tb, otb)
    assert np.array_equal(bits(at), bits(oat)) and np.array_equal(bits(ad), bits(oad)) and np.array_equal(ak, oak)


def test_interpolation_error_scan(gpu, simple_system):
    """The debug window's scan (ephemeris_explorer/src/ui/windows/debug.rs:182-238): QuinlanTremaine12 re-integration
    with the ephemeris dt, max over steps of |position - spline position| * 1e3 per body. Device result == the same
    loop over the oracle's integrator and splines, bit for bit; and the fit errors are the small numbers the
    ephemeris settings are tuned for."""
    s, sol, eph, osol = simple_system
    steps = 400
    g = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt, "QuinlanTremaine12")
    g.set_bound(s.epoch + 300 * s.dt)                      # the scan ends at the bound (BoundReached), not at n_steps
    err, done = eph.interpolation_errors(g, steps)
    assert done == 300
    o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, s.dt)
    want = np.full(s.n, -1.0)
    for _ in range(done):
        assert o.advance(1) == 0
        pos, _, t, _ = o.state()
        for b in range(s.n):
            tp = osol.eval(b, t)[0]
            d = pos[b] - tp
            e = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) * 1e3
            want[b] = e if want[b] < 0.0 else max(want[b], e)
    assert np.array_equal(bits(err), bits(want))
    assert 0.0 < err.max() < 50.0                          # metres


def test_config4_full_size_sweep(gpu):
    """BASELINE.json configs[3] at its full width on one GPU: the full_solar_system ephemeris (32 bodies) and 1e6
    spacecraft (the Mars Transfer Ship state perturbed per component by normal(0, 100 km / 0.01 km/s), seed 20260926,
    as SURVEY 8(d) prescribes), Verner87, tol 1e-3, bounded to 0.1 day. Size-independent properties: every craft
    finishes; a craft's trajectory does not depend on the batch it is in (batch of 1e6 == batch of 64 == the CPU
    oracle, bit for bit, for a spread sample); the sweep is deterministic."""
    s = load_system("full_solar_system_2433282.5")
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    days = 0.1
    end_eph = s.epoch + 5 * 86400.0
    sol = gpu.NBodyPropagator.from_system(s).propagate(end_eph)
    eph = gpu.Ephemeris(sol, s.mu)
    n = 1_000_000
    rng = np.random.default_rng(20260926)
    pos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
    vel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
    t_end = ship.start + days * 86400.0
    big = gpu.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=96)
    big.propagate(t_end)
    st = big.status()
    assert (st["status"] == 0).all() and st["steps"].min() > 10 and st["nknots"].max() < 96
    fin = big.state()
    assert (fin["t"] >= t_end).all()
    sample = np.concatenate([np.arange(8), rng.integers(0, n, 48), np.arange(n - 8, n)])
    small = gpu.SpacecraftBatch(eph, ship.start, pos[sample], vel[sample], "Verner87", max_knots=96)
    small.propagate(t_end)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(end_eph) == 0
    osol = o.take_solution()
    for k, i in enumerate(sample):
        a, b = big.knots(int(i)), small.knots(k)
        assert compare_knots(a, b, f"craft {i}: batch of 1e6 vs batch of 64")
        if k % 8 == 0:
            c = orc.Craft(osol, s.mu, ship.start, pos[i], vel[i], "Verner87")
            assert c.step_to(t_end) == 0
            assert compare_knots(a, c.knots(), f"craft {i} vs oracle")
    slab_t, slab_y = big.knot_slabs()                       # the bulk read agrees with the per-craft gather
    for i in sample[:6]:
        kt, kp, kv = big.knots(int(i))
        nk = len(kt)
        assert np.array_equal(slab_t[:nk, i], kt) and np.array_equal(slab_y[:nk, :3, i], kp)
        assert np.array_equal(slab_y[:nk, 3:, i], kv)
    again = gpu.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=96)
    again.propagate(t_end)
    f2 = again.state()
    assert np.array_equal(bits(f2["pos"]), bits(fin["pos"])) and np.array_equal(bits(f2["t"]), bits(fin["t"]))


def test_thread_form_on_small_batches(gpu):
    """Batches of up to 12288 spacecraft run one wave per craft (k_craft_wave, k_craft_events<true>), larger ones one
    thread per craft. The scenario tests above are small, so they exercise the wave form; this runs the same
    bit-parity scenarios again with the thread form forced (EPH_CRAFT_FORM is read once per process)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, EPH_CRAFT_FORM="thread")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_craft.py"), "-q", "-x", "-m", "gpu",
                        "-k", "mars_transfer or every_embedded or solout_events or slab_overflow or draining"],
                       env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_shared_reciprocal_division_is_ieee(gpu, hooks):
    """k_craft_wave divides by a body's spline interval through one refined reciprocal per lane (pair_term.h:
    rcp_refined / div_refined / div_shared). Quotients must equal IEEE division -- the compiler's on the device and the
    host's -- for the intervals of the committed systems and random ones, numerators spanning the epochs and residues
    that occur plus adversarial values (exact multiples, one ulp around them, zero, tiny and huge values that must
    take the fallback)."""
    rng = np.random.default_rng(7)
    s = load_system("full_solar_system_2433282.5")
    intervals = np.unique(np.concatenate([8.0 * s.dt * s.count.astype(np.float64), rng.uniform(1.0, 4.0e6, 64),
                                          2.0 ** rng.integers(-20, 40, 16).astype(np.float64), [1e-70, 1e70]]))
    a_parts, b_parts = [], []
    for b in intervals:
        m = rng.integers(0, 1 << 20, 20000).astype(np.float64)
        exact = m * b
        a = np.concatenate([rng.uniform(0.0, 1.0e10, 40000), rng.uniform(0.0, min(b, 1e300), 20000), exact,
                            np.nextafter(exact, np.inf), np.nextafter(exact, 0.0),
                            [0.0, b, 1.0e-7, 1.0e11, 5e-324, 1e-310, 1e-70, 1e70, 1e300]])
        a_parts.append(a)
        b_parts.append(np.full_like(a, b))
    a, b = np.concatenate(a_parts), np.concatenate(b_parts)
    fast, ieee = hooks.debug_div(a, b)
    with np.errstate(over="ignore", under="ignore"):
        want = a / b
    assert np.array_equal(bits(ieee), bits(want))             # device IEEE division == host division
    assert np.array_equal(bits(fast), bits(want)), int((bits(fast) != bits(want)).sum())


def test_more_than_64_bodies(gpu):
    """An ephemeris of 96 bodies (the 32 of the full system + 64 light test bodies on displaced copies of the planets'
    states): the wave form's lane-per-body evaluation then runs in two chunks with the ordered sum carried across
    them. Knots bit-identical to the oracle; a 20 000-craft batch (thread form) agrees on a sample."""
    s = load_system("full_solar_system_2433282.5")
    rng = np.random.default_rng(96)
    extra = 64
    src = rng.integers(1, s.n, extra)
    pos = np.vstack([s.pos, s.pos[src] + rng.normal(0.0, 5.0e6, (extra, 3))])
    vel = np.vstack([s.vel, s.vel[src] + rng.normal(0.0, 0.5, (extra, 3))])
    mu = np.concatenate([s.mu, rng.uniform(1e-3, 1.0, extra)])
    count = np.concatenate([s.count, rng.integers(4, 40, extra).astype(np.uint32)])
    degree = np.concatenate([s.degree, rng.integers(5, 8, extra).astype(np.uint32)])
    end = s.epoch + 2.0 * 86400.0
    g = gpu.NBodyPropagator(pos, vel, mu, s.epoch, s.dt, gpu.FORWARD, count, degree)
    sol = g.propagate(end)
    o = orc.Propagator(pos, vel, mu, s.epoch, s.dt, 1, count, degree)
    assert o.step_to(end) == 0
    osol = o.take_solution()
    eph = gpu.Ephemeris(sol, mu)
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    t_end = ship.start + 0.5 * 86400.0
    cpos = ship.pos + rng.normal(0.0, 100.0, (20000, 3))
    cvel = ship.vel + rng.normal(0.0, 0.01, (20000, 3))
    small = gpu.SpacecraftBatch(eph, ship.start, cpos[:3], cvel[:3], "Verner87", max_knots=512)     # wave form
    small.propagate(t_end)
    big = gpu.SpacecraftBatch(eph, ship.start, cpos, cvel, "Verner87", max_knots=512)               # thread form
    big.propagate(t_end)
    assert (small.status()["status"] == 0).all() and (big.status()["status"] == 0).all()
    for i in range(3):
        c = orc.Craft(osol, mu, ship.start, cpos[i], cvel[i], "Verner87")
        assert c.step_to(t_end) == 0
        assert compare_knots(small.knots(i), c.knots(), f"wave form, craft {i}, 96 bodies")
        assert compare_knots(big.knots(i), c.knots(), f"thread form, craft {i}, 96 bodies")


@pytest.mark.parametrize("method", ["Verner87", "Fine45", "DormandPrince54"])
def test_committed_spacecraft_knots(gpu, simple_system, method):
    """The device against tests/golden/craft_golden.json directly (no oracle in the loop): the Mars transfer to
    1951-01-01 with the app's event search."""
    import json
    from conftest import GOLDEN
    from ephemeris_explorer_amd.systems import soi_radii
    s, sol, eph, osol = simple_system
    gm = json.loads((GOLDEN / "craft_golden.json").read_text())["methods"][method]
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], method, gpu.AdaptiveParams.default(ship.tolerance),
                                [ship_burns(ship, s.names)], max_knots=40000).enable_events(soi_radii(s), 16, 8192)
    batch.propagate(parse_epoch("1951-01-01 00:00:00"))
    st = batch.status()
    assert st["status"][0] == 0
    kt, kp, kv = batch.knots(0)
    assert (len(kt), int(st["steps"][0]), int(st["attempts"][0]), float(batch.state()["next_h"][0]).hex()) == \
        (gm["knots"], gm["steps"], gm["attempts"], gm["next_h"])
    for smp in gm["sample"]:
        i = smp["i"]
        assert float(kt[i]).hex() == smp["t"]
        assert [float(x).hex() for x in kp[i]] == smp["pos"] and [float(x).hex() for x in kv[i]] == smp["vel"]
    (tt, tb), (at, ad, ab, ak) = batch.events(0)
    assert [[float(t).hex(), int(b)] for t, b in zip(tt, tb)] == gm["transitions"]
    assert len(at) == gm["apsides"]
    assert [[float(t).hex(), float(d).hex(), int(b), int(k)] for t, d, b, k in list(zip(at, ad, ab, ak))[:8]] == \
        gm["first_apsides"]


def test_single_steps(gpu, simple_system):
    """IncrementalPropagator::step through eph_craft_batch_step_n: one accepted step (one knot) per call and craft,
    across burn boundaries, equal to the oracle's step(); then step_n(25) == 25 x step()."""
    s, sol, eph, osol = simple_system
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    burns = ship_burns(ship, s.names)
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos, ship.pos + 10.0], [ship.vel, ship.vel], "Verner87",
                                burns=[burns, []], max_knots=256)
    cs = [orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, "Verner87", burns=burns),
          orc.Craft(osol, s.mu, ship.start, ship.pos + 10.0, ship.vel, "Verner87")]
    for k in range(40):
        batch.step_n(1)
        for c in cs:
            assert c.step() == 0
        assert list(batch.status()["nknots"]) == [k + 2, k + 2]
    batch.step_n(25)
    for c in cs:
        for _ in range(25):
            assert c.step() == 0
    for i, c in enumerate(cs):
        assert compare_knots(batch.knots(i), c.knots(), f"single steps, craft {i}")
    assert list(batch.status()["steps"]) == [65, 65]


_QUEUE_SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system, load_ship
from ephemeris_explorer_amd.workloads import craft_population
root = sys.argv[1] + "/tests/golden/systems/full_solar_system_2433282.5"
s = load_system(root)
ship = load_ship(root + "/ships/Mars Transfer Ship.json")
sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + 45 * 86400.0)
eph = ea.Ephemeris(sol, s.mu)
n = 6000
pos, vel, fam = craft_population("mixed", n, s, ship)
earth = s.names.index("Earth")
burns = [[(ship.start + 3000.0, ship.start + 3090.0, [2e-4, 1e-4, 0.0], earth)] if i % 7 == 0 else [] for i in range(n)]
from ephemeris_explorer_amd.systems import soi_radii
b = ea.SpacecraftBatch(eph, ship.start, pos, vel, sys.argv[3], max_knots=4000, burns=burns)
b.enable_events(soi_radii(s), max_transitions=16, max_apsides=256)   # the app's solout reads the knot slabs on the device
b.propagate(ship.start + 0.6 * 86400.0)           # two legs: the second resumes craft from stored state
b.propagate(ship.start + 1.5 * 86400.0)
rec = b.summary()
kt, ky = b.knot_slabs(0, int(rec["nknots"].max()))
for i in range(n):                                 # entries beyond a craft's knots are unspecified
    kt[rec["nknots"][i]:, i] = 0.0
    ky[rec["nknots"][i]:, :, i] = 0.0
counts = b.event_counts()
ev = [np.concatenate([np.asarray(x, dtype=np.float64) for pair in b.events(i, counts) for x in pair]) for i in range(0, n, 37)]
# a clone taken here, the slab drained on the original, both resumed: single-craft read-out and the newest knots must agree
c = b.clone()
b.reset_knots()
b.propagate(ship.start + 1.6 * 86400.0)
c.propagate(ship.start + 1.6 * 86400.0)
rec2, rec2c = b.summary(), c.summary()
assert rec2["status"].tobytes() == rec2c["status"].tobytes() and rec2["steps"].tobytes() == rec2c["steps"].tobytes()
tails = []
for i in range(0, n, 53):
    tb, pb, vb = b.knots(i)
    tc, pc, vc = c.knots(i)
    m = len(tb)
    assert m >= 1 and np.array_equal(tb, tc[-m:]) and np.array_equal(pb, pc[-m:]) and np.array_equal(vb, vc[-m:]), i
    tails.append(np.concatenate([tb[-1:], pb[-1], vb[-1]]))
np.savez(sys.argv[2], rec=rec, kt=kt, ky=ky, pos=pos, vel=vel, ntr=counts[0], nap=counts[1], ev=np.concatenate(ev),
         tails=np.array(tails))
'''


@pytest.mark.parametrize("method", ["Verner87", "DormandPrince54", "Fine45"])
def test_queue_and_static_sweeps_give_the_same_bits(gpu, tmp_path, method):
    """k_craft_queue (persistent grid, work queue, one attempt per iteration) against k_craft_propagate (craft i on
    thread i) on a mixed population -- low orbits, transfer orbits, a heliocentric cruise, every seventh craft with a
    burn in the Earth's TNB frame -- in two legs: every record and every knot identical, and a craft of each family
    identical to the restatement."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    outs = []
    # (kernel form, lane assignment): static and queue kernels with the craft dealt to the lanes by dynamical time (the default
    # for a heterogeneous batch), and the static kernel with craft i on lane i (EPH_CRAFT_SORT=0)
    for q, srt in (("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")):
        out = tmp_path / f"q{q}s{srt}.npz"
        env = dict(os.environ, EPH_CRAFT_QUEUE=q, EPH_CRAFT_SORT=srt, EPH_CRAFT_FORM="thread")
        r = subprocess.run([sys.executable, "-c", _QUEUE_SCRIPT, str(ROOT), str(out), method], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(out))
    a, b = outs[0], outs[1]
    assert (a["rec"]["status"] == 0).all()
    for other in outs[1:]:
        assert a["rec"].tobytes() == other["rec"].tobytes()
        assert np.array_equal(a["kt"], other["kt"]) and np.array_equal(a["ky"], other["ky"])
        assert np.array_equal(a["ntr"], other["ntr"]) and np.array_equal(a["nap"], other["nap"])
        assert a["ev"].tobytes() == other["ev"].tobytes() and a["tails"].tobytes() == other["tails"].tobytes()
    assert a["nap"].sum() > 1000                               # (the low orbits pass many apsides: the search ran)
    steps = a["rec"]["steps"]
    assert steps.max() > 8 * steps.min()                       # the population is what it claims to be
    s = load_system("full_solar_system_2433282.5")
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(s.epoch + 45 * 86400.0) == 0
    osol = o.take_solution()
    earth = s.names.index("Earth")
    for i in (0, 1, 2, 3, 7):
        burns = [(ship.start + 3000.0, ship.start + 3090.0, [2e-4, 1e-4, 0.0], earth)] if i % 7 == 0 else []
        c = orc.Craft(osol, s.mu, ship.start, a["pos"][i], a["vel"][i], method, burns=burns)
        assert c.step_to(ship.start + 0.6 * 86400.0) == 0 and c.step_to(ship.start + 1.5 * 86400.0) == 0
        ot, op, ov = c.knots()
        nk = a["rec"]["nknots"][i]
        assert nk == len(ot)
        assert np.array_equal(b["kt"][:nk, i], ot) and np.array_equal(b["ky"][:nk, :3, i], op) and np.array_equal(b["ky"][:nk, 3:, i], ov)


def test_body_order_of_the_acceleration_sum(gpu):
    """eph_craft_batch_set_body_order: the app iterates an EntityHashMap (dynamics/spacecraft.rs:164-165,222-228), so WHICH order
    the massive bodies' terms are added in is the caller's to say. Thread-per-craft and wave-per-craft kernels against the
    restatement in the same order; a permutation changes bits; None restores the table's order; non-permutations are refused."""
    from conftest import ROOT
    from ephemeris_explorer_amd.systems import load_system as load_sys
    sysdir = ROOT / "tests/golden/systems/full_solar_system_2433282.5"
    s = load_sys(sysdir)
    ship = load_ship(sysdir / "ships" / "Mars Transfer Ship.json")
    g = gpu.NBodyPropagator.from_system(s)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    g.step_to(ship.start + 2 * 86400.0)
    assert o.step_to(ship.start + 2 * 86400.0) == 0
    sg, so = g.take_solution(), o.take_solution()
    eph = gpu.Ephemeris(sg, s.mu)
    order = np.random.default_rng(4).permutation(s.n).astype(np.int32)
    t_end = ship.start + 6 * 3600.0
    knots = {}
    for ncraft in (3, 20000):                          # wave-per-craft | thread-per-craft
        pos = np.repeat(ship.pos[None], ncraft, 0) + np.arange(ncraft)[:, None] * 1e-2
        vel = np.repeat(ship.vel[None], ncraft, 0)
        for which in ("table", "permuted"):
            b = gpu.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=256)
            if which == "permuted":
                b.set_body_order(order)
            tw = b.clone()                             # a clone inherits the order
            for x in (b, tw):
                x.propagate(t_end)
                assert (x.status()["status"] == 0).all()
            for i in (0, ncraft - 1):
                c = orc.Craft(so, s.mu, ship.start, pos[i], vel[i], "Verner87", body_order=order if which == "permuted" else None)
                assert c.step_to(t_end) == 0
                ot, op, ov = c.knots()
                for x in (b, tw):
                    kt, kp, kv = x.knots(i)
                    assert np.array_equal(kt, ot) and np.array_equal(kp, op) and np.array_equal(kv, ov), (ncraft, which, i)
            knots[(ncraft, which)] = b.knots(0)
        assert not np.array_equal(knots[(ncraft, "table")][1], knots[(ncraft, "permuted")][1])
    b = gpu.SpacecraftBatch(eph, ship.start, pos[:3], vel[:3], "Verner87", max_knots=256)
    b.set_body_order(order).set_body_order(None)
    b.propagate(t_end)
    assert np.array_equal(b.knots(0)[1], knots[(3, "table")][1])
    for bad in (np.zeros(s.n, dtype=np.int32), np.arange(s.n, dtype=np.int32) + 1):
        with pytest.raises(gpu.EphemerisError):
            b.set_body_order(bad)


def test_block_cache_follows_the_handles(gpu):
    """csrc/mem.cpp: device blocks of >= 64 MiB (the knot slabs of a spacecraft batch) are kept for the next batch of the same shape,
    per device and bounded; a reused block is cleared; eph_release_cached_memory returns them -- and when the library's LAST
    allocation on the device goes, so does the cache (a process that shares the GPU with another allocator is not left holding it).
    Runs in a child process: the accounting is process-wide and other tests' handles are alive in this one."""
    import subprocess
    import sys
    from conftest import ROOT
    script = r'''
import gc, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system, load_ship
root = sys.argv[1]
s = load_system(root + "/tests/golden/systems/simple_solar_system_2433282.5")
ship = load_ship(root + "/tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
prop = ea.NBodyPropagator.from_system(s)
sol = prop.propagate(s.epoch + 30 * 86400.0)
eph = ea.Ephemeris(sol, s.mu)
n = 20000                                                  # 20000 craft x 600 knots x 8 B = 96 MB per knot row set: cached size class
pos = ship.pos + np.random.default_rng(1).normal(0.0, 10.0, size=(n, 3))
vel = np.repeat(ship.vel[None], n, 0)
def sweep():
    b = ea.SpacecraftBatch(eph, ship.start, pos, vel, "DormandPrince54", max_knots=600)
    b.propagate(ship.start + 3600.0)
    st = b.status()
    assert (st["status"] == 0).all()
    kt, ky = b.knot_slabs(0, int(st["nknots"].max()) + 2)   # two rows beyond the last knot of every craft
    return st["nknots"].copy(), kt, ky
nk1, kt1, ky1 = sweep()
gc.collect()
nk2, kt2, ky2 = sweep()                                      # takes the first batch's blocks out of the cache
assert np.array_equal(nk1, nk2)
last = int(nk2.max())
assert np.array_equal(kt1[:last], kt2[:last]) and np.array_equal(ky1[:last], ky2[:last])
assert not kt2[last:].any() and not ky2[last:].any(), "rows beyond nknots of a REUSED block must be clear, not the previous batch's"
gc.collect()
held = ea.release_cached_memory()
assert held >= 64 << 20, held                               # the ephemeris is still alive: the batch's blocks were being kept
assert ea.release_cached_memory() == 0
sweep()
del prop, sol, eph
gc.collect()
assert ea.release_cached_memory() == 0, "the cache must have gone with the library's last allocation on the device"
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", script, str(ROOT)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
