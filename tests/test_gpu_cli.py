"""-m gpu: SURVEY 8(f)1 -- the headless flow over the reference's on-disk formats (state.json, ephemeris.json,
ships/*.json): what cli.run produces on the reference's own `full_solar_system_2433282.5` (32 bodies, 3 ships) compared
with the CPU restatement, not just counted: forward and backward ephemerides of every body (spline bounds, coefficients,
trimmed lengths) and, per ship, outcome, every knot and the app's SpacecraftSolout events -- bit for bit."""
import numpy as np
import pytest

from conftest import SYSTEMS
from oracle import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_cli_run_matches_the_restatement(gpu):
    from ephemeris_explorer_amd import cli
    from ephemeris_explorer_amd.systems import soi_radii
    years = 2.0
    r = cli.run(SYSTEMS / "full_solar_system_2433282.5", years)
    s = r.system
    assert s.n == 32 and len(r.ships) == 3
    span = years * cli.SEC_PER_YEAR
    oracle_solutions = {}
    for direction, sol in ((1, r.forward), (-1, r.backward)):
        o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, s.count, s.degree, native=True)
        assert o.step_to(s.epoch + direction * span) == 0
        osol = o.take_solution()
        oracle_solutions[direction] = osol
        polys = 0
        for b in range(s.n):
            assert sol.info(b) == osol.info(b), (direction, s.names[b])
            cg, ng = sol.coeffs(b)
            co, no = osol.coeffs(b)
            assert np.array_equal(ng, no) and np.array_equal(bits(cg), bits(co)), (direction, s.names[b])
            polys += len(ng)
        assert polys > 10000
    assert r.fwd_prop.time() >= s.epoch + span and r.bwd_prop.time() <= s.epoch - span
    # ships: the three of the reference's directory, with their burns in Earth / Sun / Mars / Moon / Jupiter ... frames
    soi = soi_radii(s)
    osol = oracle_solutions[1]
    seen = {}
    for ship, burns, batch, skipped in r.ships:
        assert skipped is None, skipped
        c = orc.Craft(osol, s.mu, ship.start, ship.pos, ship.vel, ship.integrator, tol_pos=ship.tolerance,
                      tol_vel=ship.tolerance, burns=burns, soi_radius=soi)
        want = c.step_to(ship.end)
        st = batch.status()
        assert int(st["status"][0]) == want, ship.name
        seen[ship.name] = want
        kt, kp, kv = batch.knots(0)
        ot, op, ov = c.knots()
        assert len(kt) == len(ot) and len(kt) > (50 if want == 0 else 0), ship.name
        assert np.array_equal(bits(kt), bits(ot)) and np.array_equal(bits(kp), bits(op)) and np.array_equal(bits(kv), bits(ov)), ship.name
        (tt, tb), (at, ad, ab, ak) = batch.events(0)
        ott, otb = c.transitions()
        oat, oad, oab, oak = c.apsides()
        assert np.array_equal(bits(tt), bits(ott)) and np.array_equal(tb, otb), ship.name
        assert np.array_equal(bits(at), bits(oat)) and np.array_equal(bits(ad), bits(oad)), ship.name
        assert np.array_equal(ab, oab) and np.array_equal(ak, oak), ship.name
    # the Voyager plan starts in 1977, outside the 1950 +-2-year ephemeris: EvalFailed at once (one knot), like the
    # reference's propagator
    assert seen == {"Mars Transfer Ship": 0, "Moon Transfer Ship": 0, "Voyager Style Ship": orc.EVAL_FAILED}


def test_cli_live_flow_equals_the_static_one(gpu):
    """`--live`: the app's own flow -- the bodies' task sends a snapshot every 30 days, each is merged into the LIVE device table,
    one task per ship chases it (three ship threads + the N-body thread, one shared eph_ephemeris): knots, events and statuses equal
    the static run's (ships against the finished table), which test_cli_run_matches_the_restatement pins to the oracle."""
    from ephemeris_explorer_amd import cli
    sysdir = SYSTEMS / "full_solar_system_2433282.5"
    static = cli.run(sysdir, 2.0, backward=False)
    for rep in range(2):
        live = cli.run_live(sysdir, 2.0, chunk_days=30.0)
        assert live.live_revision >= 23                                  # two years in 30-day snapshots, the first seeds the table
        for b in range(static.system.n):
            assert live.forward.info(b) == static.forward.info(b)
            assert np.array_equal(bits(live.forward.coeffs(b)[0]), bits(static.forward.coeffs(b)[0]))
        assert [s[0].name for s in live.ships] == [s[0].name for s in static.ships]
        for (ship, _, bl, skip_l), (_, _, bs, skip_s) in zip(live.ships, static.ships):
            assert skip_l is None and skip_s is None
            assert bl.status()["status"][0] == bs.status()["status"][0], ship.name
            for x, y in zip(bl.knots(0), bs.knots(0)):
                assert np.array_equal(bits(x), bits(y)), (rep, ship.name)
            (tl, bl_), (al, dl, abl, akl) = bl.events(0)
            (ts, bs_), (as_, ds, abs_, aks) = bs.events(0)
            assert np.array_equal(bits(tl), bits(ts)) and np.array_equal(bl_, bs_) and np.array_equal(bits(al), bits(as_))
            assert np.array_equal(bits(dl), bits(ds)) and np.array_equal(abl, abs_) and np.array_equal(akl, aks)


def test_cli_fine45_ship(gpu, tmp_path):
    """A ships/*.json selecting the app's ERKNG integrator (Fine45, dynamics/spacecraft.rs:797) goes through the headless
    flow like any other (it used to be skipped)."""
    import json
    import shutil
    from ephemeris_explorer_amd import cli
    src = SYSTEMS / "sun_earth_moon_2433282.5"
    dst = tmp_path / "sys"
    shutil.copytree(src, dst)
    ship = json.loads((dst / "ships" / "Earth Station.json").read_text())
    ship["integrator"] = "Fine45"
    (dst / "ships" / "Earth Station.json").write_text(json.dumps(ship))
    r = cli.run(dst, 0.1)
    (sh, burns, batch, skipped), = r.ships
    assert skipped is None and sh.integrator == "Fine45" and int(batch.status()["status"][0]) == 0
    o = orc.Propagator(r.system.pos, r.system.vel, r.system.mu, r.system.epoch, r.system.dt, 1, r.system.count, r.system.degree)
    assert o.step_to(r.system.epoch + 0.1 * cli.SEC_PER_YEAR) == 0
    c = orc.Craft(o.take_solution(), r.system.mu, sh.start, sh.pos, sh.vel, "Fine45", tol_pos=sh.tolerance, tol_vel=sh.tolerance)
    assert c.step_to(sh.end) == 0
    kt, kp, kv = batch.knots(0)
    ot, op, ov = c.knots()
    assert np.array_equal(bits(kt), bits(ot)) and np.array_equal(bits(kp), bits(op))


def test_c_example_runs_on_the_device(gpu, tmp_path):
    """examples/propagate.c: the propagator seam from plain C (what a cgo / Rust-FFI binding does), on the device; its
    printed Earth position at day 10 equals the oracle's (to the printed millimetre)."""
    import re
    import subprocess
    from conftest import ROOT, load_system
    exe = tmp_path / "propagate"
    libdir = ROOT / "ephemeris_explorer_amd"
    subprocess.check_call(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "propagate.c"), f"-L{libdir}",
                           "-lephemeris_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    m = re.search(r"r\(day 10\) = \(([-0-9.]+), ([-0-9.]+), ([-0-9.]+)\) km inside=1", r.stdout)
    assert m, r.stdout
    s = load_system("sun_earth_moon_2433282.5")
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(s.epoch + 30 * 86400.0) == 0
    want = o.take_solution().eval(1, s.epoch + 10 * 86400.0)[0]
    got = np.array([float(x) for x in m.groups()])
    assert np.abs(got - want).max() < 1e-3 + 1e-12 * np.abs(want).max()


def test_cpp_example_runs_on_the_device(gpu, tmp_path):
    """examples/propagate.cpp: the reference's trait surface (include/ephemeris_amd.hpp) from C++ -- NewtonianGravity::eval,
    a bounded NBodyPropagator::propagate, EvaluateTrajectory::state_vector, a SpacecraftPropagator::step_to with a burn.
    It prints hex floats: every value equals the CPU restatement's bit for bit."""
    import re
    import subprocess
    from conftest import ROOT, load_system
    exe = tmp_path / "propagate_cpp"
    libdir = ROOT / "ephemeris_explorer_amd"
    subprocess.check_call(["g++", "-std=c++17", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "propagate.cpp"), f"-L{libdir}",
                           "-lephemeris_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    hx = r"(-?0x[0-9a-fp.+-]+)"

    def floats(m):
        return np.array([float.fromhex(x) for x in m.groups()])
    s = load_system("sun_earth_moon_2433282.5")
    t0 = s.epoch
    m = re.search(rf"ddy\[2\] = {hx} {hx} {hx}", r.stdout)
    assert m, r.stdout
    assert np.array_equal(bits(floats(m)), bits(orc.gravity(s.pos, s.mu)[2]))
    o = orc.Propagator(s.pos, s.vel, s.mu, t0, s.dt, 1, s.count, s.degree)
    assert o.step_to(t0 + 40 * 86400.0) == 0
    m = re.search(rf"reached {hx}; Earth spline: (\d+) polynomials of (\d+) s; inside=1", r.stdout)
    assert m, r.stdout
    assert float.fromhex(m.group(1)) == o.time()
    eph = o.take_solution()
    start, interval, npoly = eph.info(1)
    assert int(m.group(2)) == npoly and float(m.group(3)) == interval
    m = re.search(rf"earth\(day 10\) = {hx} {hx} {hx} \| {hx} {hx} {hx}", r.stdout)
    assert m, r.stdout
    wp, wv = eph.eval(1, t0 + 10 * 86400.0)[:2]
    assert np.array_equal(bits(floats(m)), bits(np.concatenate([np.ravel(wp), np.ravel(wv)])))
    c = orc.Craft(eph, s.mu, t0, [-27204249.668775786, 132947582.43848978, 57641619.74241204],
                  [-22.207539106181895, -5.189518219791726, -2.2515617105336263], "Verner87",
                  burns=[(t0 + 7200.0, t0 + 7260.0, [5e-4, 0.0, 0.0], 1)])
    assert c.step_to(t0 + 3 * 86400.0) == 0
    kt, kp, kv = c.knots()
    m = re.search(rf"craft: status 0 \(\w+\), knots (\d+), last knot t = {hx} r = {hx} {hx} {hx}", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == len(kt)
    got = np.array([float.fromhex(x) for x in m.groups()[1:]])
    assert np.array_equal(bits(got), bits(np.concatenate([[kt[-1]], kp[-1]])))
    # the app's flow: a clone resumed, a propagation drained in two pieces and joined, the app's solout
    from ephemeris_explorer_amd.systems import soi_radii
    from oracle import pyoracle as po
    burn = (t0 + 7200.0, t0 + 7260.0, [5e-4, 0.0, 0.0], 1)
    c = orc.Craft(eph, s.mu, t0, [-27204249.668775786, 132947582.43848978, 57641619.74241204],
                  [-22.207539106181895, -5.189518219791726, -2.2515617105336263], "Verner87", burns=[burn], soi_radius=soi_radii(s))
    assert c.step_to(t0 + 1.5 * 86400.0) == 0
    first_leg = len(c.knots()[0])
    assert c.step_to(t0 + 3 * 86400.0) == 0
    kt, kp, kv = c.knots()
    m = re.search(rf"joined: knots (\d+) \(first leg (\d+)\), inside=1, r\(day 2\) = {hx} {hx} {hx}", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == len(kt) and int(m.group(2)) == first_leg
    want = orc.hermite_eval(kt, kp, kv, t0 + 2 * 86400.0)[0]
    assert np.array_equal(bits([float.fromhex(x) for x in m.groups()[2:]]), bits(want))
    trt, trb = c.transitions()
    apt, apd, apb, apk = c.apsides()
    m = re.search(rf"events: status 0, transitions (\d+), apsides (\d+), first apsis at {hx}", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == len(trt) and int(m.group(2)) == len(apt) and len(apt) > 0 and float.fromhex(m.group(3)) == apt[0]
    edit = po.timeline_divergence_time_before([burn, (t0 + 2 * 86400.0, t0 + 2 * 86400.0 + 30.0, [0.0, 1e-4, 0.0], -1)], [burn], t0 + 3 * 86400.0)
    assert f"flight plan edit restarts at t0 + {edit - t0:.1f} s" in r.stdout, r.stdout
    # auto-extend: a ship runs off the table's end, the bodies' next snapshot is merged into the live table, the SAME propagator resumes
    c = orc.Craft(eph, s.mu, t0, [-27204249.668775786, 132947582.43848978, 57641619.74241204],
                  [-22.207539106181895, -5.189518219791726, -2.2515617105336263], "DormandPrince54")
    target = t0 + 50 * 86400.0
    assert c.step_to(target) == orc.EVAL_FAILED
    failed_at = len(c.knots()[0])
    assert o.step_to(t0 + 60 * 86400.0) == 0
    assert eph.append(o.take_solution())
    assert c.step_to(target) == 0
    kt, kp, kv = c.knots()
    m = re.search(rf"auto-extend: (failed to evaluate ODE) after (\d+) knots \(table valid at target: 0 -> 1, revision 1\); resumed: (\w+), knots (\d+), "
                  rf"last knot t = {hx} r = {hx} {hx} {hx}", r.stdout)
    assert m, r.stdout
    assert int(m.group(2)) == failed_at and int(m.group(4)) == len(kt) > failed_at
    assert np.array_equal(bits([float.fromhex(x) for x in m.groups()[4:]]), bits(np.concatenate([[kt[-1]], kp[-1]])))


def test_c_spacecraft_example_runs_on_the_device(gpu, tmp_path):
    """examples/craft.c: INTEGRATION.md 4b's batch-of-one spacecraft propagator driven from plain C -- create, the app's
    solout, a step_n loop until has_reached, knots + events. Everything it prints equals the restatement stepped the same
    number of times (the burn in the Earth's TNB frame included)."""
    import re
    import subprocess
    from conftest import ROOT, load_system
    from ephemeris_explorer_amd.systems import soi_radii
    exe = tmp_path / "craft"
    libdir = ROOT / "ephemeris_explorer_amd"
    subprocess.check_call(["gcc", "-std=c99", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "craft.c"), f"-L{libdir}",
                           "-lephemeris_amd", f"-Wl,-rpath,{libdir}", "-lm", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    m = re.search(r"steps (\d+) in (\d+) calls, knots (\d+), t - t0 = ([-0-9.]+) s, r = \(([-0-9.]+), ([-0-9.]+), ([-0-9.]+)\) km", r.stdout)
    e = re.search(r"transitions (\d+) \(first: body (-?\d+) at ([-0-9.]+) s\), apsides (\d+) \(event status (\d+)\)", r.stdout)
    a = re.search(r"first apsis: (\w+) of body (\d+) at ([-0-9.]+) s, ([-0-9.]+) km", r.stdout)
    assert m and e and a, r.stdout
    steps, calls, knots = int(m.group(1)), int(m.group(2)), int(m.group(3))
    assert knots == steps + 1 and 0 < steps <= 64 * calls, r.stdout   # a call that fills the slab stops early
    s = load_system("sun_earth_moon_2433282.5")
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert o.step_to(s.epoch + 40 * 86400.0) == 0
    eph = o.take_solution()
    t0 = s.epoch
    c = orc.Craft(eph, s.mu, t0, [-27204249.668775786, 132947582.43848978, 57641619.74241204],
                  [-22.207539106181895, -5.189518219791726, -2.2515617105336263], "Verner87",
                  burns=[(t0 + 7200.0, t0 + 7260.0, [5e-4, 0.0, 0.0], 1)], soi_radius=soi_radii(s))
    for _ in range(steps):
        assert c.step() == 0
    st = c.state()
    assert st["t"] >= t0 + 3 * 86400.0 and abs((st["t"] - t0) - float(m.group(4))) < 1e-6
    assert np.abs(st["pos"] - np.array([float(m.group(k)) for k in (5, 6, 7)])).max() < 1e-6 + 1e-15 * np.abs(st["pos"]).max()
    trt, trb = c.transitions()
    apt, apd, apb, apk = c.apsides()
    assert len(trt) == int(e.group(1)) and len(apt) == int(e.group(4)) and int(e.group(5)) == 0
    assert trb[0] == int(e.group(2)) and abs((trt[0] - t0) - float(e.group(3))) < 1e-3
    assert ("apoapsis" if apk[0] else "periapsis") == a.group(1) and apb[0] == int(a.group(2))
    assert abs((apt[0] - t0) - float(a.group(3))) < 1e-3 and abs(apd[0] - float(a.group(4))) < 1e-6
    assert len(apt) > 40                                  # ~15 orbits a day around the Earth
