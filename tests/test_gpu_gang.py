"""-m gpu: several independent small systems advanced in ONE launch (eph_nbody_advance_many / eph_prop_step_n_many:
a workgroup per system in k_lm_small). The reference runs its forward and backward N-body propagators concurrently
(ephemeris_explorer/src/load/mod.rs:673-687); here they share launches, and the results are those of the separate calls --
bit for bit, and against the CPU restatement."""
import numpy as np
import pytest

from conftest import load_system
from oracle import orc

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def test_advance_many_equals_separate_advances(gpu):
    full = load_system("full_solar_system_2433282.5")
    sem = load_system("sun_earth_moon_2433282.5")
    specs = [(full, full.dt, "QuinlanTremaine12"), (full, -full.dt, "QuinlanTremaine12"), (sem, sem.dt, "QuinlanTremaine12"),
             (full, 300.0, "QuinlanTremaine12")]
    gang = [gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, h, m) for s, h, m in specs]
    solo = [gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, h, m) for s, h, m in specs]
    gpu.advance_many(gang, 7)                          # inside the start-up: every system on its own
    gpu.advance_many(gang, 5 + 400)                    # crosses into the steady state
    gpu.advance_many(gang, 1)
    gpu.advance_many(gang, 2999)
    for g in solo:
        g.advance(7 + 5 + 400 + 1 + 2999)
    for g, s in zip(gang, solo):
        (p, v, t, c), (p0, v0, t0, c0) = g.state(), s.state()
        assert t == t0 and c == c0 and _same(p, p0) and _same(v, v0) and _same(g.acc(), s.acc())
    o = orc.NBody(full.pos, full.vel, full.mu, full.epoch, -full.dt)
    assert o.advance(3412) == 0
    assert _same(gang[1].state()[0], o.state()[0]) and _same(gang[1].state()[1], o.state()[1])


def test_advance_many_with_systems_that_do_not_qualify(gpu):
    """a 300-body system, a Stormer13 system (another ring length) and a bound that ends inside the call: the call does
    what the separate advances do, including the StepError."""
    from ephemeris_explorer_amd.workloads import plummer
    full = load_system("full_solar_system_2433282.5")
    pos, vel, mu = plummer(300)

    def make():
        return [gpu.NBodyIntegration(full.pos, full.vel, full.mu, full.epoch, full.dt),
                gpu.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0),
                gpu.NBodyIntegration(full.pos, full.vel, full.mu, full.epoch, full.dt, "Stormer13")]
    gang, solo = make(), make()
    gpu.advance_many(gang, 40)
    for g in solo:
        g.advance(40)
    for g, s in zip(gang, solo):
        assert g.state()[2:] == s.state()[2:] and _same(g.state()[0], s.state()[0]) and _same(g.state()[1], s.state()[1])
    two = [gpu.NBodyIntegration(full.pos, full.vel, full.mu, full.epoch, full.dt) for _ in range(2)]
    gpu.advance_many(two, 20)
    two[1].set_bound(full.epoch + 25.5 * full.dt)
    with pytest.raises(gpu.StepError) as e:
        gpu.advance_many(two, 10)
    assert e.value.status == gpu.BOUND_REACHED
    assert two[0].state()[3] == 30 and two[1].state()[3] == 26
    with pytest.raises(gpu.EphemerisError):            # a system cannot be in the gang twice
        gpu.advance_many([two[0], two[0]], 1)


@pytest.mark.parametrize("count", [2, 9])
def test_step_n_many_builds_the_same_ephemerides(gpu, count):
    """forward and backward propagators (and a few more with other sampling periods) stepped together from creation:
    start-up alone, steady state in shared launches, every spline identical to the propagator stepped by itself."""
    s = load_system("full_solar_system_2433282.5")

    def make(i):
        direction = gpu.FORWARD if i % 2 == 0 else gpu.BACKWARD
        count_b = (s.count + i // 2).astype(np.uint32)
        return gpu.NBodyPropagator(s.pos, s.vel, s.mu, s.epoch, s.dt, direction, count_b, s.degree)
    gang, solo = [make(i) for i in range(count)], [make(i) for i in range(count)]
    for k in (5, 300, 1, 4000):
        gpu.step_n_many(gang, k)
    for p in solo:
        p.step_n(5 + 300 + 1 + 4000)
    for g, p in zip(gang, solo):
        assert g.time() == p.time()
        sg, sp = g.take_solution(), p.take_solution()
        for b in range(s.n):
            assert sg.info(b) == sp.info(b)
            (cg, ng), (cp, np_) = sg.coeffs(b), sp.coeffs(b)
            assert np.array_equal(ng, np_) and _same(cg, cp)
    o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, -1, s.count, s.degree)
    for _ in range(4306):
        assert o.step() == 0
    g1 = make(1)
    gpu.step_n_many([g1, make(0)], 4306)
    so, sg = o.take_solution(), g1.take_solution()
    for b in range(s.n):
        assert sg.info(b) == so.info(b) and _same(sg.coeffs(b)[0], so.coeffs(b)[0])


def test_gang_larger_than_the_chip(gpu):
    """More systems than the chip has CUs take the four-wave form of k_lm_small (two unordered pairs per thread, two workgroups
    per CU: step_small.hip, round 5): 300 systems -- the 32-body system forward and backward with different steps, the 10- and the
    3-body one -- in one launch, every member bit-identical to its own separate advance, a few of them to the restatement."""
    full, simple, sem = (load_system(n) for n in ("full_solar_system_2433282.5", "simple_solar_system_2433282.5",
                                                   "sun_earth_moon_2433282.5"))

    def spec(i):
        s = (full, simple, sem)[i % 3]
        return s, s.dt * (1.0 + 0.001 * (i // 3)) * (1 if i % 2 == 0 else -1)
    K = 300
    gang = [gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, h) for s, h in map(spec, range(K))]
    gpu.advance_many(gang, 12)
    gpu.advance_many(gang, 700)
    gpu.advance_many(gang, 1)
    for i in (0, 1, 2, 3, 150, 257, 298, 299):
        s, h = spec(i)
        solo = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, h)
        solo.advance(713)
        (p, v, t, c), (p0, v0, t0, c0) = gang[i].state(), solo.state()
        assert t == t0 and c == c0 and _same(p, p0) and _same(v, v0) and _same(gang[i].acc(), solo.acc()), i
    for i in (0, 1, 299):
        s, h = spec(i)
        o = orc.NBody(s.pos, s.vel, s.mu, s.epoch, h)
        assert o.advance(713) == 0
        assert _same(gang[i].state()[0], o.state()[0]) and _same(gang[i].state()[1], o.state()[1]), i
