"""-m gpu: the header's threading contract, executed.

include/ephemeris_amd.h promises "a handle is not thread-safe; distinct handles may be used from distinct threads", and the app relies
on it: forward and backward N-body propagators and one task per ship run concurrently on Bevy's compute pool
(ephemeris_explorer/src/prediction.rs:385-391, load/mod.rs:673-687), each `loop { step(); if ready { take_solution(); clone(); send } }`
(prediction.rs:422-443), propagators are moved between the pool's threads, and the ships read the bodies' LIVE trajectories
(Arc<RwLock<..>>, dynamics/mod.rs:84-85) while merged snapshots grow them.

Two drivers: a C++ program (examples/threads.cpp, std::thread over include/ephemeris_amd.hpp -> the C ABI) and Python threads over
ctypes (ctypes.CDLL releases the GIL for the duration of every foreign call). Every thread's results must be bit-identical to the same
work done serially -- and the serial work is checked against the CPU oracle."""
import subprocess
import threading

import numpy as np
import pytest

from conftest import ROOT, SYSTEMS, load_system
from ephemeris_explorer_amd.systems import load_ship
from oracle import orc

pytestmark = pytest.mark.gpu
DAY = 86400.0
REPS = 20


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_cpp_threads_program(gpu, tmp_path):
    """examples/threads.cpp: nine concurrent tasks (forward / backward N-body tasks with take_solution + clone, two ship tasks on
    their own tables -- wave-per-craft and thread-per-craft kernels --, two seam-1/seam-2 loops through the staging pool, two ship
    tasks chasing a table that a writer thread grows, and handles created / used / destroyed on three different threads), 20
    repetitions, every digest equal to the serial run's."""
    exe = tmp_path / "threads"
    libdir = ROOT / "ephemeris_explorer_amd"
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-O1", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "threads.cpp"),
                           f"-L{libdir}", "-lephemeris_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    r = subprocess.run([str(exe), str(REPS)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert f"{REPS} repetitions x 9 concurrent tasks (13 threads): every result bit-identical to the serial run" in r.stdout, r.stdout
    assert "differs" not in r.stdout and "serial: forward" in r.stdout


# ---- the same contract from Python threads over ctypes ------------------------------------------------------------------------
def _nbody_task(gpu, s, direction, count):
    """the N-body task: step a few times, take_solution, clone; the clone replaces the stored propagator"""
    p = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, count, s.degree)
    out = []
    for k in range(16):
        p.step_n(6)
        sol = p.take_solution()
        out.append([(sol.info(b), sol.coeffs(b)) for b in range(s.n)])
        if k % 3 == 2:
            p = p.clone()
    out.append(p.state())
    return out


def _same_nbody(a, b):
    for x, y in zip(a[:-1], b[:-1]):
        for (ix, (cx, nx)), (iy, (cy, ny)) in zip(x, y):
            if ix != iy or not np.array_equal(nx, ny) or not np.array_equal(bits(cx), bits(cy)):
                return False
    (p0, v0, t0, c0), (p1, v1, t1, c1) = a[-1], b[-1]
    return t0 == t1 and c0 == c1 and np.array_equal(bits(p0), bits(p1)) and np.array_equal(bits(v0), bits(v1))


def _ship_task(gpu, eph, ship, pos, vel, method, end):
    """the ship task: step_to in four legs, a snapshot per leg that replaces the stored propagator"""
    b = gpu.SpacecraftBatch(eph, ship.start, pos, vel, method, max_knots=8192)
    for k in (1, 2, 3, 4):
        b.propagate(ship.start + (end - ship.start) * k / 4)
        b = b.clone()
    st = b.status()
    return st["status"].copy(), st["nknots"].copy(), [b.knots(i) for i in range(len(pos))]


def _same_ship(a, b):
    if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
        return False
    return all(np.array_equal(bits(x), bits(y)) for ka, kb in zip(a[2], b[2]) for x, y in zip(ka, kb))


def _seam_task(gpu, n, loops, seed):
    """seam 1 + seam 2 with the state read back after every step (the staging buffers they go through are pooled since round 6)"""
    rng = np.random.default_rng(seed)
    pos, vel, mu = rng.normal(0.0, 1e6, (n, 3)), rng.normal(0.0, 0.05, (n, 3)), rng.uniform(1.0, 10.0, n)
    g = gpu.NBodyIntegration(pos, vel, mu, 0.0, 10.0)
    out = []
    for _ in range(loops):
        g.advance(1)
        p, v, t, c = g.state()
        out.append((p, v, t, c, gpu.accel_eval(p, mu)))
    return out


def _same_seam(a, b):
    return all(x[2] == y[2] and x[3] == y[3] and all(np.array_equal(bits(x[i]), bits(y[i])) for i in (0, 1, 4)) for x, y in zip(a, b))


def test_python_threads_over_ctypes(gpu):
    s = load_system("simple_solar_system_2433282.5")
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    count = np.minimum(s.count, 2)
    # a third table for the ship task (thread C), and its oracle twin
    g = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    g.step_to(s.epoch + 8 * DAY)
    assert o.step_to(s.epoch + 8 * DAY) == 0
    so = o.take_solution()
    eph = gpu.Ephemeris(g.take_solution(), s.mu)
    rng = np.random.default_rng(5)
    pos = ship.pos + rng.normal(0.0, 40.0, (3, 3))
    vel = ship.vel + rng.normal(0.0, 0.004, (3, 3))
    end = ship.start + 2.0 * DAY

    tasks = {
        "forward": (lambda: _nbody_task(gpu, s, 1, count), _same_nbody),
        "backward": (lambda: _nbody_task(gpu, s, -1, count), _same_nbody),
        "ship": (lambda: _ship_task(gpu, eph, ship, pos, vel, "DormandPrince54", end), _same_ship),
        "seams": (lambda: _seam_task(gpu, 300, 25, 11), _same_seam),
        "seams-small": (lambda: _seam_task(gpu, 32, 40, 12), _same_seam),
    }
    serial = {k: f() for k, (f, _) in tasks.items()}

    # the serial run against the oracle: the N-body task's pieces and the ship's knots
    for direction, key in ((1, "forward"), (-1, "backward")):
        oo = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, count, s.degree)
        for k in range(16):
            for _ in range(6):
                assert oo.step() == 0
            sol = oo.take_solution()
            for b in range(s.n):
                info, (c, nc) = serial[key][k][b]
                assert info == sol.info(b)
                oc, onc = sol.coeffs(b)
                assert np.array_equal(nc, onc) and np.array_equal(bits(c), bits(oc))
    for i in range(3):
        c = orc.Craft(so, s.mu, ship.start, pos[i], vel[i], "DormandPrince54")
        assert c.step_to(end) == 0
        kt, kp, kv = serial["ship"][2][i]
        ot, op, ov = c.knots()
        # the legs end at has_reached(leg end): a craft stepped to `end` in one go takes the same steps as one stepped in four legs
        assert len(kt) == len(ot) and np.array_equal(bits(kt), bits(ot)) and np.array_equal(bits(kp), bits(op)) and np.array_equal(bits(kv), bits(ov))

    bad = []
    for rep in range(REPS):
        got, errors = {}, []

        def run(name, f):
            try:
                got[name] = f()
            except Exception as e:               # noqa: BLE001 -- reported below, with the task's name
                errors.append((name, repr(e)))
        th = [threading.Thread(target=run, args=(k, f)) for k, (f, _) in tasks.items()]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for k, (_, same) in tasks.items():
            if not same(got[k], serial[k]):
                bad.append((rep, k))
    assert not bad, bad


def test_handles_move_between_threads(gpu):
    """A propagator and a batch created on one thread, used on a second, destroyed on a third (the task pool moves propagators:
    they are Send); results equal the single-thread run's."""
    s = load_system("sun_earth_moon_2433282.5")
    box = {}

    def make():
        box["p"] = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)

    def use():
        box["p"].step_n(30)
        sol = box["p"].take_solution()
        box["eph"] = gpu.Ephemeris(sol, s.mu)
        box["state"] = box["p"].state()
        box["sol"] = [(sol.info(b), sol.coeffs(b)) for b in range(s.n)]

    def drop():
        del box["p"], box["eph"]

    for f in (make, use, drop):
        t = threading.Thread(target=f)
        t.start()
        t.join()
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    for _ in range(30):
        assert o.step() == 0
    so = o.take_solution()
    po, vo, to, co = o.state()
    pg, vg, tg, cg = box["state"]
    assert tg == to and cg == co and np.array_equal(bits(pg), bits(po)) and np.array_equal(bits(vg), bits(vo))
    for b in range(s.n):
        info, (c, nc) = box["sol"][b]
        assert info == so.info(b)
        assert np.array_equal(bits(c), bits(so.coeffs(b)[0]))


def test_live_table_shared_between_a_writer_and_readers(gpu):
    """The one handle that IS shared: eph_ephemeris carries the reference's RwLock. A writer thread merges the bodies' snapshots
    (dynamics/celestial.rs:198-204) while two ship tasks restart their stored propagators whenever the context has become valid at
    their next target (flight_plan.rs:363-395: `propagator.context().is_valid_at(..)`), re-layouts of the device table included.
    Every leg stays a margin inside the table it started against, so the knots are independent of the interleaving: equal to the
    oracle's against the full table."""
    s = load_system("simple_solar_system_2433282.5")
    ship = load_ship(SYSTEMS / "full_solar_system_2433282.5" / "ships" / "Mars Transfer Ship.json")
    count = np.minimum(s.count, 2)
    g = gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, count, s.degree)
    pieces, whole = [], None
    for k in range(1, 11):
        t = s.epoch + 4.0 * k * DAY
        g.step_to(t)
        assert o.step_to(t) == 0
        pieces.append(g.take_solution())
        piece = o.take_solution()
        if whole is None:
            whole = piece
        else:
            assert whole.append(piece)
    legs = [s.epoch + 4.0 * k * DAY - 1.5 * DAY for k in range(1, 10)]
    want = {}
    for name, method, dx in (("a", "Verner87", 0.0), ("b", "DormandPrince54", 25.0)):
        c = orc.Craft(whole, s.mu, ship.start, ship.pos + dx, ship.vel, method)
        assert c.step_to(legs[-1]) == 0
        want[name] = c.knots()

    for rep in range(4):
        eph = gpu.Ephemeris(pieces[0], s.mu)
        got, errors = {}, []

        def writer():
            try:
                for p in pieces[1:]:
                    eph.merge(p)
            except Exception as e:               # noqa: BLE001
                errors.append(("writer", repr(e)))

        def reader(name, method, dx):
            try:
                b = gpu.SpacecraftBatch(eph, ship.start, [ship.pos + dx], [ship.vel], method, max_knots=32768)
                for leg in legs:
                    while not eph.is_valid_at(leg + 1.0 * DAY):
                        if errors:
                            return
                    b.propagate(leg)
                assert b.status()["status"][0] == 0
                got[name] = b.knots(0)
            except Exception as e:               # noqa: BLE001
                errors.append((name, repr(e)))
        th = [threading.Thread(target=writer), threading.Thread(target=reader, args=("a", "Verner87", 0.0)),
              threading.Thread(target=reader, args=("b", "DormandPrince54", 25.0))]
        for t in reversed(th):                   # the readers first: they wait for the writer
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        assert eph.revision == len(pieces) - 1
        for name in ("a", "b"):
            for x, y in zip(got[name], want[name]):
                assert np.array_equal(bits(x), bits(y)), (rep, name)
