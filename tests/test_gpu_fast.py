"""-m gpu: the OPT-IN fast path (eph_nbody_set_path(h, EPH_PATH_FAST)): slice-parallel partial sums combined in slice
order instead of the reference's ordered chain. It is NOT bit-identical to the reference and is never the default;
these tests pin what it does promise: determinism, and closeness to the ordered path at the level of summation
round-off. The measured divergence over 1e5 steps is reported by bench.py --path fast."""
import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
H = 1.0 / 1024.0
FAST = 4


@pytest.mark.parametrize("n", [100, 1000, 4096, 5000])
def test_fast_path_is_deterministic_and_close_to_the_ordered_path(gpu, n):
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(n)
    exact = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
    runs = []
    for _ in range(2):
        g = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
        g.set_path(FAST)
        g.advance(12 + 40)
        runs.append((g.state(), g.acc()))
    exact.advance(12 + 40)
    (p0, v0, t0, c0), a0 = runs[0]
    (p1, v1, t1, c1), a1 = runs[1]
    assert (t0, c0) == (t1, c1) == exact.state()[2:]
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1) and np.array_equal(a0, a1)   # run-to-run identical
    pe, ve = exact.state()[:2]
    ae = exact.acc()
    # same pair arithmetic, different summation order; 40 steps on, the accelerations are taken at positions that have
    # drifted apart at round-off level (amplified by the close pairs of a softening-free sphere)
    scale = np.abs(ae).max()
    assert np.abs(a0 - ae).max() < 1e-8 * scale
    # 40 steps later: round-off level (amplified by the close pairs of a softening-free sphere), NOT identical
    assert 0.0 < np.abs(p0 - pe).max() < 1e-8
    assert np.abs(v0 - ve).max() < 1e-6


def test_fast_rsq_path(gpu):
    """EPH_PATH_FAST_RSQ: the fast path with 1/r^3 from v_rsq_f64 + two Newton steps (no IEEE sqrt / divide): deterministic,
    accelerations within a few ulp of the ordered path's pair terms summed in slice order."""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(1000)
    runs = []
    for path in (0, 4, 5, 5):
        g = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
        g.set_path(path)
        g.advance(12 + 1)                    # one steady step: the accelerations of the same positions
        runs.append(g.acc())
    exact, fast, rsq, rsq2 = runs
    assert np.array_equal(rsq, rsq2)
    scale = np.abs(exact).max()
    assert 0.0 < np.abs(rsq - exact).max() < 1e-13 * scale and not np.array_equal(rsq, fast)
    with pytest.raises(gpu.EphemerisError):
        gpu.NBodyIntegration(pos, vel, mu, 0.0, H).set_path(7)


def test_fast_path_refuses_what_it_does_not_cover(gpu):
    from conftest import load_system
    s = load_system("full_solar_system_2433282.5")
    g = gpu.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    g.set_path(FAST)
    g.advance(12)                                             # start-up runs the ordered kernels
    with pytest.raises(gpu.EphemerisError):                   # 32 bodies: one workgroup, nothing to slice
        g.advance(1)
    with pytest.raises(gpu.EphemerisError):
        g.set_path(7)


def test_fast_path_with_solout_sampling(gpu):
    """The fused step still samples for the solout: the propagator on the fast path produces splines that agree with the
    ordered path's to round-off."""
    from ephemeris_explorer_amd.workloads import plummer
    n = 256
    pos, vel, mu = plummer(n)
    count = np.full(n, 2, np.uint32)
    degree = np.full(n, 6, np.uint32)
    a = gpu.NBodyPropagator(pos, vel, mu, 0.0, H, gpu.FORWARD, count, degree)
    b = gpu.NBodyPropagator(pos, vel, mu, 0.0, H, gpu.FORWARD, count, degree)
    b.integration().set_path(FAST)
    sa, sb = a.propagate(200 * H), b.propagate(200 * H)
    differing = 0
    for body in range(n):
        assert sa.info(body) == sb.info(body) and sa.info(body)[2] >= 12
        ca, cb = sa.coeffs(body)[0], sb.coeffs(body)[0]
        assert np.abs(ca - cb).max() < 1e-9
        differing += not np.array_equal(ca, cb)
    assert differing > 8                 # round-off level, but not the ordered path's bits (most bodies' fitted
    #                                      coefficients round to the same doubles over 200 steps)


def test_fast_path_divergence_at_the_metric_size(gpu):
    """N = 4096, 1000 steps against the oracle's committed positions (tests/golden/plummer4096_horizon.npz): the
    figure bench.py --path fast reports for 10^k steps up to 1e5, here bounded for the first three checkpoints."""
    from ephemeris_explorer_amd.workloads import plummer
    fx = np.load(GOLDEN / "plummer4096_horizon.npz")
    pos, vel, mu = plummer(4096)
    g = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
    g.set_path(FAST)
    done = 0
    for c, bound in ((10, 0.0), (100, 1e-10), (1000, 1e-7)):
        g.advance(c - done)
        done = c
        d = np.abs(g.state()[0] - fx[f"pos_{c}"]).max()
        if c == 10:
            assert d == 0.0                                   # still inside the (ordered) start-up
        else:
            assert 0.0 < d < bound, (c, d)


@pytest.mark.parametrize("path", [4, 5])
def test_fast_path_body_at_the_origin_with_padded_sources(gpu, path):
    """n % 64 != 0 leaves zero-filled padding rows behind the last source; a body sitting exactly at the origin (the
    central body of a heliocentric system) must not see them (n2 = 0 -> 0 * inf = NaN before the fix)."""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(1000)
    pos[7] = 0.0
    exact = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
    g = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
    g.set_path(path)
    # set_path applies to the steady steps; the start-up runs the ordered kernels. Put the body back at the origin for the
    # first fast step by checking the accelerations of a system whose start-up is trivial: zero velocities, tiny h
    g.advance(12 + 2)
    exact.advance(12 + 2)
    a, ae = g.acc(), exact.acc()
    assert np.isfinite(a).all() and np.isfinite(g.state()[0]).all()
    assert np.abs(a - ae).max() < 1e-9 * np.abs(ae).max()


@pytest.mark.parametrize("n", [1000, 16384, 65536])      # 65 536: BASELINE configs[4]'s size
def test_f32_pairs_path(gpu, n):
    """EPH_PATH_F32_PAIRS (BASELINE.json configs[4]'s precision): pair arithmetic in binary32, f64 accumulation in slice
    order, f64 integrator. Deterministic; accelerations at single-precision distance from the exact path's, the state after
    a few steps accordingly close; a body at the origin with padded sources stays finite."""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(n)
    pos[5] = 0.0
    runs = []
    for path in (0, 6, 6):
        g = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
        g.set_path(path)
        g.advance(12 + 1)
        a1 = g.acc()
        g.advance(20)
        runs.append((a1, g.state()[0]))
    (ae, pe), (am, pm), (am2, pm2) = runs
    assert np.array_equal(am, am2) and np.array_equal(pm, pm2)
    assert np.isfinite(am).all() and np.isfinite(pm).all()
    rel = np.abs(am - ae).max() / np.abs(ae).max()
    assert 1e-9 < rel < 2e-5, rel                      # binary32 pair terms: ~6e-8 each, a few close pairs dominate
    assert 0.0 < np.abs(pm - pe).max() < 1e-6


def test_f32_pairs_bodies_coinciding_in_binary32(gpu):
    """Distinct bodies whose positions round to the same binary32 values (n2 = 0 in the f32 pair arithmetic, y = inf): a massive
    partner's mu y^3 = inf is clamped and multiplies a zero separation, a massless one's 0 * inf = NaN leaves the clamp as 0 --
    the mutual term is dropped instead of turning both sums into NaN. Partners sit in different 64-body blocks, so the unmasked
    loop sees them; their masses are negligible and their velocities equal, so the exact path (start-up, and the comparison)
    keeps them together."""
    from ephemeris_explorer_amd.workloads import plummer
    pos, vel, mu = plummer(1000)
    pairs = [(8 + k, 700 + k) for k in range(6)]
    for k, (a, b) in enumerate(pairs):
        pos[b] = pos[a] + 1e-9
        vel[b] = vel[a]
        mu[a] = 1e-30
        mu[b] = 0.0 if k % 2 else 1e-30
    runs = []
    for path in (0, 6):
        g = gpu.NBodyIntegration(pos, vel, mu, 0.0, H)
        g.set_path(path)
        g.advance(12 + 3)
        runs.append((g.acc(), g.state()[0]))
    (ae, pe), (am, pm) = runs
    together = [np.array_equal(pm[a].astype(np.float32), pm[b].astype(np.float32)) for a, b in pairs]
    assert sum(together) >= 3, together               # (a pair can straddle a rounding boundary: 1e-9 against an ulp of 6e-8)
    assert np.isfinite(am).all() and np.isfinite(pm).all()
    assert np.abs(am - ae).max() / np.abs(ae).max() < 2e-5


_FUSED_CHILD = """
import sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
out = []
for n, path in ((4096, 4), (4096, 5), (5000, 4), (8192, 6), (1000, 5)):
    pos, vel, mu = plummer(n)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    g.set_path(path)
    g.advance(12 + 25)
    p, v, t, c = g.state()
    out.append(hashlib.sha256(p.tobytes() + v.tobytes() + g.acc().tobytes()).hexdigest())
print(" ".join(out))
"""


def test_one_launch_step_equals_the_two_launch_form(gpu):
    """Round 6: the fast paths' step is ONE launch -- the workgroup that arrives last at a block of 64 targets combines the block's
    slice sums in slice order and does Cowell / predictor (k_fast_step; sc1 write-through stores, one relaxed agent-scope ticket per
    workgroup, no fence). Same arithmetic in the same order as the two-launch form (k_fast_partial + k_fast_finish,
    EPH_FAST_FUSED=0): bit-identical state after 25 steady steps for every fast path, twice (whichever workgroup arrives last)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    runs = {}
    for fused in ("1", "0", "1"):
        env = dict(os.environ, EPH_FAST_FUSED=fused)
        r = subprocess.run([sys.executable, "-c", _FUSED_CHILD, str(ROOT)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.setdefault(fused, []).append(r.stdout.strip().split())
    assert len(runs["1"][0]) == 5 and runs["1"][0] == runs["1"][1] == runs["0"][0]
