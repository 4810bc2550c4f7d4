"""-m gpu: the device ephemeris GROWS under living spacecraft batches, like the reference's context.

GravitationalBody.trajectory is Trajectory(Arc<RwLock<PredictionTrajectory>>) (ephemeris_explorer/src/dynamics/spacecraft.rs:52-74,
dynamics/mod.rs:84-85); merged N-body snapshots append to it (dynamics/celestial.rs:198-204 -> UniformSpline::append,
ephemeris/src/trajectory.rs:515-549), auto_extend requests more every frame (auto_extend.rs:182-202), and ships resume from their
STORED propagator (prediction.rs:378), whose context is that same live table. A reference propagator that returned EvalFailed at
the table's end (spacecraft.rs:264-281) therefore continues once the bodies' splines have grown. Here: eph_ephemeris_append /
_merge / _clear + eph_craft_batch_retry_failed, every result compared with the CPU oracle (orc.Craft over an orc.Solution that is
appended between calls) bit for bit: status, the attempt counter n (runge_kutta/mod.rs:427 returns BEFORE n += 1), next_h, state and
every knot -- including the FSAL quirk (explicit.rs:76-79 has swapped k[0] / k[S-1] before the failing stage: the retried step
starts from a stale first stage; restated, not repaired).

The tests run in this process on the wave-per-craft kernel (k_craft_wave) and again, in child processes, on k_craft_propagate
(dealt and undealt) and k_craft_queue."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SYSTEMS, load_system
from ephemeris_explorer_amd.systems import load_ship, soi_radii
from oracle import orc

pytestmark = pytest.mark.gpu
DAY = 86400.0
METHODS = ["Verner87", "DormandPrince54", "Fine45"]          # no FSAL | FSAL | FSAL on SecondOrderState (ERKNG)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    return np.array_equal(bits(a), bits(b))


@pytest.fixture(scope="module")
def pieces(gpu):
    """The 10-body 1950 system integrated ON THE DEVICE and in the oracle, handed out as four consecutive take_solution() pieces
    (bodies sampled every step or two, so that every spline's polynomials are at most 4 days long: the file's own periods reach 50
    days, and a test would need a year of ephemeris per piece)."""
    s = load_system("simple_solar_system_2433282.5")
    count = np.minimum(s.count, 2)
    g = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    out = []
    for k in (1, 2, 3, 4):
        t = s.epoch + (1 + 3 * k) * DAY
        g.step_to(t)
        assert o.step_to(t) == 0
        sg, so = g.take_solution(), o.take_solution()
        for b in range(s.n):
            assert sg.info(b) == so.info(b)
            assert same(sg.coeffs(b)[0], so.coeffs(b)[0])
        out.append((sg, so))
    return s, out


def _ship():
    return load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")


def _fleet(ship, n, seed):
    rng = np.random.default_rng(seed)
    pos = ship.pos + rng.normal(0.0, 50.0, size=(n, 3))
    vel = ship.vel + rng.normal(0.0, 0.005, size=(n, 3))
    pos[0], vel[0] = ship.pos, ship.vel
    return pos, vel


def _compare(batch, i, c, st_expected, what):
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


def _live(gpu, s, piece):
    """(device table, oracle table) holding the first piece; both are grown by the tests"""
    sg, so = piece
    return gpu.Ephemeris(sg, s.mu), so.clone()


@pytest.mark.parametrize("method", METHODS)
def test_live_resume_after_append(gpu, pieces, method):
    """(a) craft run to EvalFailed at the table's end; the propagator's next take_solution is appended; the SAME batch resumes.
    (b) a clone taken BEFORE the extension resumes after it (prediction.rs:378). Twice over, the second time after TWO failed
    attempts (a retry without growth fails again and moves the FSAL registers once more, as the reference's second step() does)."""
    s, pcs = pieces
    ship = _ship()
    eph, olive = _live(gpu, s, pcs[0])
    n = 4
    pos, vel = _fleet(ship, n, 3)
    end = s.epoch + 9.5 * DAY
    batch = gpu.SpacecraftBatch(eph, ship.start, pos, vel, method, max_knots=8192)
    crafts = [orc.Craft(olive, s.mu, ship.start, pos[i], vel[i], method) for i in range(n)]
    batch.propagate(end)
    for i, c in enumerate(crafts):
        assert c.step_to(end) == orc.EVAL_FAILED
        _compare(batch, i, c, gpu.EVAL_FAILED, f"{method} craft {i}: first table's end")
    first_failure = [len(c.knots()[0]) for c in crafts]
    snapshot = batch.clone()                              # stored propagator, taken before the extension
    # sticky: without re-arming, nothing moves -- a drain loop must not re-attempt failed craft
    batch.propagate(end)
    for i, c in enumerate(crafts):
        _compare(batch, i, c, gpu.EVAL_FAILED, f"{method} craft {i}: sticky")
    # the bodies' next snapshot is merged into the live table (both sides)
    rev = eph.revision
    eph.append(pcs[1][0])
    assert olive.append(pcs[1][1])
    assert eph.revision == rev + 1
    for b in range(s.n):
        assert eph.info(b) == olive.info(b)
    batch.retry_failed().propagate(end)
    for i, c in enumerate(crafts):
        assert c.step_to(end) == orc.EVAL_FAILED          # ... and off the second table's end
        _compare(batch, i, c, gpu.EVAL_FAILED, f"{method} craft {i}: second table's end")
    # a retry WITHOUT growth: one more failed attempt each (the reference remembers nothing)
    batch.retry_failed().propagate(end)
    for i, c in enumerate(crafts):
        assert c.step_to(end) == orc.EVAL_FAILED
        _compare(batch, i, c, gpu.EVAL_FAILED, f"{method} craft {i}: retried without growth")
    eph.append(pcs[2][0])
    assert olive.append(pcs[2][1])
    batch.retry_failed().propagate(end)
    for i, c in enumerate(crafts):
        assert c.step_to(end) == 0
        _compare(batch, i, c, 0, f"{method} craft {i}: the end")
    # (b) the clone: same failed state, resumed against the table as it is NOW (two pieces longer); the oracle's counterpart is
    # a craft that failed once at the first table's end and then saw all three pieces
    ref = pcs[0][1].clone()
    c0 = orc.Craft(ref, s.mu, ship.start, pos[0], vel[0], method)
    assert c0.step_to(end) == orc.EVAL_FAILED
    assert ref.append(pcs[1][1]) and ref.append(pcs[2][1])
    assert c0.step_to(end) == 0
    snapshot.retry_failed().propagate(end)
    _compare(snapshot, 0, c0, 0, f"{method}: clone resumed after the extension")
    # only where the reference's would: a craft that saw the long table from the start equals the resumed one for a pair
    # without FSAL, and differs right after the (odd number of) failed attempts for an FSAL pair
    whole = orc.Craft(ref, s.mu, ship.start, pos[0], vel[0], method)
    assert whole.step_to(end) == 0
    wt, wp, _ = whole.knots()
    kt, kp, _ = snapshot.knots(0)
    k = first_failure[0]
    assert same(wt[:k], kt[:k]) and same(wp[:k], kp[:k])
    if method == "Verner87":
        assert len(wt) == len(kt) and same(wp, kp)
    else:
        assert not same(wp[k:k + 3], kp[k:k + 3])


@pytest.mark.parametrize("method", METHODS)
def test_live_events_and_body_order(gpu, pieces, method):
    """(d) a batch with the app's SpacecraftSolout enabled (SOI transitions + apsides, dynamics/spacecraft.rs:514-587) and a
    permuted Bodies iteration order (the permuted table copy is re-gathered from the live table before every sweep)"""
    s, pcs = pieces
    ship = _ship()
    soi = soi_radii(s)
    order = np.array([3, 0, 4, 1, 2, 9, 8, 7, 6, 5], dtype=np.int32)
    eph, olive = _live(gpu, s, pcs[0])
    end = s.epoch + 6.5 * DAY
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], method, max_knots=8192)
    batch.enable_events(soi, max_transitions=16, max_apsides=2048).set_body_order(order)
    c = orc.Craft(olive, s.mu, ship.start, ship.pos, ship.vel, method, soi_radius=soi, body_order=order)
    batch.propagate(end)
    assert c.step_to(end) == orc.EVAL_FAILED
    _compare(batch, 0, c, gpu.EVAL_FAILED, f"{method}: first table's end")
    eph.append(pcs[1][0])
    assert olive.append(pcs[1][1])
    batch.retry_failed().propagate(end)
    assert c.step_to(end) == 0
    _compare(batch, 0, c, 0, f"{method}: resumed")
    ott, otb = c.transitions()
    oat, oad, oab, oak = c.apsides()
    ntr, nap, est = batch.event_counts()
    assert est[0] == 0 and ntr[0] == len(ott) and nap[0] == len(oat) and len(oat) > 50
    (tt, tb), (at, ad, ab, ak) = batch.events(0)
    assert same(tt, ott) and np.array_equal(tb, otb)
    assert same(at, oat) and same(ad, oad) and np.array_equal(ab, oab) and np.array_equal(ak, oak)


def test_live_batch_dealt_to_the_lanes(gpu, pieces):
    """192 craft: in the thread-per-craft forms the batch is dealt to the lanes by orbital period (craft_sort: lane != craft), so the
    FSAL registers and the re-arming travel through the permutation; every craft against its oracle."""
    s, pcs = pieces
    ship = _ship()
    eph, olive = _live(gpu, s, pcs[0])
    n = 192
    rng = np.random.default_rng(77)
    pos = ship.pos + rng.normal(0.0, 200.0, size=(n, 3))
    vel = ship.vel * (1.0 + rng.uniform(-0.02, 0.05, size=(n, 1)))       # a spread of orbital periods
    end = s.epoch + 5.5 * DAY
    batch = gpu.SpacecraftBatch(eph, ship.start, pos, vel, "DormandPrince54", max_knots=6144)
    crafts = [orc.Craft(olive, s.mu, ship.start, pos[i], vel[i], "DormandPrince54") for i in range(n)]
    batch.propagate(end)
    st = [c.step_to(end) for c in crafts]
    assert set(st) == {orc.EVAL_FAILED}
    assert np.array_equal(batch.status()["status"], np.full(n, gpu.EVAL_FAILED))
    eph.append(pcs[1][0])
    assert olive.append(pcs[1][1])
    batch.retry_failed().propagate(end)
    summ = batch.summary()
    kt, ky = batch.knot_slabs()
    for i, c in enumerate(crafts):
        assert c.step_to(end) == 0
        cs = c.state()
        assert summ["status"][i] == 0 and summ["attempts"][i] == cs["attempts"] and summ["steps"][i] == cs["steps"], i
        assert bits(summ["t"][i]) == bits(cs["t"]) and bits(summ["next_h"][i]) == bits(cs["next_h"]), i
        ot, op, ov = c.knots()
        nk = summ["nknots"][i]
        assert nk == len(ot), i
        assert same(kt[:nk, i], ot) and same(ky[:nk, 0:3, i], op) and same(ky[:nk, 3:6, i], ov), i


def test_live_many_small_appends(gpu, pieces):
    """The app's pattern: a snapshot every few steps (load/mod.rs:675). The bodies advance two steps at a time and every
    take_solution is merged into the live table (CelestialTrajectory::merge, dynamics/celestial.rs:198-204: most snapshots carry no
    new polynomial for most bodies); a craft chases the table's end, failing and resuming many times. In-place row appends and
    re-layouts of the device table both occur (the first region has room for as many polynomials again)."""
    s, pcs = pieces
    ship = _ship()
    count = np.minimum(s.count, 2)
    g = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    t = s.epoch + 4.5 * DAY
    g.step_to(t)
    assert o.step_to(t) == 0
    eph, olive = gpu.Ephemeris(g.take_solution(), s.mu), o.take_solution()
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos, ship.pos + 30.0], [ship.vel, ship.vel], "DormandPrince54", max_knots=24576)
    crafts = [orc.Craft(olive, s.mu, ship.start, ship.pos + 30.0 * i, ship.vel, "DormandPrince54") for i in range(2)]
    end = s.epoch + 40 * DAY
    failures = 0
    for k in range(60):
        g.step_n(2)
        for _ in range(2):
            assert o.step() == 0
        eph.merge(g.take_solution())
        tail = o.take_solution()
        for b in range(s.n):                             # clear_after(propagated.start()): a no-op here, the snapshot starts
            olive.clear_after(tail.info(b)[0], b)        # at the table's end (get_index: None for time >= span)
        assert olive.append(tail)
        batch.retry_failed().propagate(end)
        for i, c in enumerate(crafts):
            st = c.step_to(end)
            assert st == orc.EVAL_FAILED
            failures += 1
            if k % 10 == 9:
                _compare(batch, i, c, gpu.EVAL_FAILED, f"snapshot {k} craft {i}")
    for b in range(s.n):
        assert eph.info(b) == olive.info(b)
    assert failures == 120 and len(crafts[0].knots()[0]) > 3000
    _compare(batch, 0, crafts[0], gpu.EVAL_FAILED, "after 60 snapshots")
    # a table built in one piece from the same polynomials gives a new batch the same knots as the grown one
    whole = gpu.Ephemeris.from_image(eph.export_image())
    t_in = s.epoch + 30 * DAY
    b1 = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "Verner87", max_knots=16384)
    b2 = gpu.SpacecraftBatch(whole, ship.start, [ship.pos], [ship.vel], "Verner87", max_knots=16384)
    b1.propagate(t_in)
    b2.propagate(t_in)
    assert b1.status()["status"][0] == 0 and b1.status()["nknots"][0] > 1000
    for x, y in zip(b1.knots(0), b2.knots(0)):
        assert same(x, y)


def test_live_clear_and_prepend(gpu, pieces):
    """clear_before (what trims a long-running table's past, trajectory.rs:536-542 -- it drops the polynomial that CONTAINS `at` too:
    get_index_exclusive(at + interval)): a craft whose epoch falls before the new start gets EvalFailed exactly where the oracle
    does, one inside continues; clear_after (:544-549) cuts the future; refusals leave the table untouched."""
    s, pcs = pieces
    ship = _ship()
    eph, olive = _live(gpu, s, pcs[0])
    for k in (1, 2, 3):                                  # the table: [epoch, epoch + 16 d] (polynomials of 2 and 4 days)
        eph.append(pcs[k][0])
        assert olive.append(pcs[k][1])
    t_mid = s.epoch + 4.5 * DAY
    batch = gpu.SpacecraftBatch(eph, ship.start, [ship.pos, ship.pos + 10.0], [ship.vel, ship.vel], "Verner87", max_knots=8192)
    crafts = [orc.Craft(olive, s.mu, ship.start, ship.pos + 10.0 * i, ship.vel, "Verner87") for i in range(2)]
    batch.propagate(t_mid)
    for i, c in enumerate(crafts):
        assert c.step_to(t_mid) == 0
    at = s.epoch + 3.0 * DAY
    eph.clear_before(at)
    olive.clear_before(at)
    for b in range(s.n):
        assert eph.info(b) == olive.info(b)
        assert eph.info(b)[0] == s.epoch + 4.0 * DAY     # 3 d lies inside [0, 4 d] / [2 d, 4 d]: that polynomial goes as well
    assert not eph.is_valid_at(s.epoch) and not eph.is_valid_at(at) and eph.is_valid_at(t_mid)
    start = s.epoch + 4.0 * DAY
    assert eph.is_valid_at(start) and not eph.is_valid_at(np.nextafter(start, -np.inf))   # contains(start): the sign BIT
    end = s.epoch + 6.0 * DAY
    batch.propagate(end)
    for i, c in enumerate(crafts):
        assert c.step_to(end) == 0
        _compare(batch, i, c, 0, f"craft {i} after clear_before")
    # a new batch that starts before the trimmed table's start fails at its first evaluation, like the oracle's
    late = gpu.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "Verner87", max_knots=64)
    late.propagate(end)
    c = orc.Craft(olive, s.mu, ship.start, ship.pos, ship.vel, "Verner87")
    assert c.step_to(end) == orc.EVAL_FAILED
    _compare(late, 0, c, gpu.EVAL_FAILED, "start before the table")
    # clear_after: the table's future is cut at the start of the polynomial that contains `cut` (what merge does first)
    cut = s.epoch + 13.0 * DAY
    eph.clear_after(cut)
    olive.clear_after(cut)
    for b in range(s.n):
        assert eph.info(b) == olive.info(b)
        assert eph.info(b)[0] + eph.info(b)[1] * eph.info(b)[2] == pcs[3][0].info(b)[0]     # 12 d
    kt, kp, kv = crafts[0].knots()
    j = int(np.searchsorted(kt, t_mid))                  # a knot inside what is left of the table
    far = s.epoch + 13.0 * DAY
    b3 = gpu.SpacecraftBatch(eph, kt[j], [kp[j]], [kv[j]], "Verner87", max_knots=4096)
    c3 = orc.Craft(olive, s.mu, kt[j], kp[j], kv[j], "Verner87")
    b3.propagate(far)
    st3 = c3.step_to(far)
    assert st3 == orc.EVAL_FAILED and len(c3.knots()[0]) > 100     # ... runs off the cut
    _compare(b3, 0, c3, st3, "after clear_after")
    # refusals leave the table untouched: pieces that are not contiguous (trajectory.rs:517-518,530-531)
    rev = eph.revision
    with pytest.raises(ValueError):
        eph.append(pcs[1][0])
    with pytest.raises(ValueError):
        eph.merge(pcs[3][0], gpu.BACKWARD)
    assert eph.revision == rev
    for b in range(s.n):
        assert eph.info(b) == olive.info(b)
    # the continuation the cut made room for: merge = clear_after(propagated.start()) + append (dynamics/celestial.rs:198-204)
    eph.merge(pcs[3][0])
    for b in range(s.n):
        olive.clear_after(pcs[3][1].info(b)[0], b)
    assert olive.append(pcs[3][1])
    for b in range(s.n):
        assert eph.info(b) == olive.info(b)
    b3.retry_failed().propagate(far)
    assert c3.step_to(far) == 0
    _compare(b3, 0, c3, 0, "resumed after the merge")


def test_live_prepend_backward(gpu):
    """The Backward propagator's pieces are prepended (dynamics/celestial.rs:220-226): a table that starts at the epoch grows into
    the past, and a batch created afterwards starts there."""
    s = load_system("simple_solar_system_2433282.5")
    count = np.minimum(s.count, 2)
    fwd_g = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    fwd_o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    bwd_g = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, -1, count, s.degree)
    bwd_o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, -1, count, s.degree)
    fwd_g.step_to(s.epoch + 4.5 * DAY)
    assert fwd_o.step_to(s.epoch + 4.5 * DAY) == 0
    eph, olive = gpu.Ephemeris(fwd_g.take_solution(), s.mu), fwd_o.take_solution()
    for k in (1, 2):
        t = s.epoch - 4.5 * k * DAY
        bwd_g.step_to(t)
        assert bwd_o.step_to(t) == 0
        eph.merge(bwd_g.take_solution(), gpu.BACKWARD)
        assert olive.append(bwd_o.take_solution(), -1)
        for b in range(s.n):
            assert eph.info(b) == olive.info(b)
    ship = _ship()
    t0 = s.epoch - 7.0 * DAY
    assert eph.is_valid_at(t0)
    # the ship's state, placed a week earlier (any state will do: parity, not astronautics)
    earth = s.names.index("Earth")
    ep0, ev0 = olive.eval(earth, t0)
    ep1, ev1 = olive.eval(earth, ship.start)
    pos, vel = ship.pos - ep1 + ep0, ship.vel - ev1 + ev0
    batch = gpu.SpacecraftBatch(eph, t0, [pos], [vel], "Verner87", max_knots=8192)
    c = orc.Craft(olive, s.mu, t0, pos, vel, "Verner87")
    end = s.epoch + 1.0 * DAY
    batch.propagate(end)
    assert c.step_to(end) == 0
    _compare(batch, 0, c, 0, "started in the prepended part")


def test_live_export_import_round_trip(gpu, pieces):
    """eph_ephemeris_export / _import: the image of a grown table imports to a table with identical bounds, and exporting that
    one again gives the same bytes (every coefficient, every count)."""
    s, pcs = pieces
    eph, olive = _live(gpu, s, pcs[0])
    eph.append(pcs[1][0])
    img = eph.export_image()
    twin = gpu.Ephemeris.from_image(img)
    assert twin.n_bodies == s.n
    for b in range(s.n):
        assert twin.info(b) == eph.info(b)
    assert np.array_equal(twin.export_image(), img)
    bad = img.copy()
    bad[0] ^= 1
    with pytest.raises(gpu.EphemerisError):
        gpu.Ephemeris.from_image(bad)
    with pytest.raises(gpu.EphemerisError):
        gpu.Ephemeris.from_image(img[:len(img) - 8])


@pytest.mark.parametrize("form", ["thread-static", "thread-queue", "thread-static-undealt"])
def test_live_ephemeris_on_the_other_sweep_kernels(gpu, form):
    """The tests above run on k_craft_wave (small batches). The same tests again on the thread-per-craft kernels: k_craft_propagate
    (static; craft dealt to the lanes, and craft i on lane i) and k_craft_queue (persistent grid + work queue). The kernel form is
    read once per process, hence child processes."""
    env = dict(os.environ, EPH_CRAFT_FORM="thread", EPH_CRAFT_QUEUE="1" if form == "thread-queue" else "0",
               EPH_CRAFT_SORT="0" if form.endswith("undealt") else "1")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_live_ephemeris.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_live_ and not other_sweep_kernels"], env=env, cwd=str(ROOT), capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
