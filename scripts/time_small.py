"""Steady QuinlanTremaine12 step of the small committed systems (k_lm_small: one workgroup per system), alone and as a
gang of K systems in one launch (eph_nbody_advance_many), and the 1e6-step propagator run of configs[1] with its solout.
usage (GPU box): python scripts/time_small.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system
STEPS = 500000
for name in ("full_solar_system_2433282.5", "simple_solar_system_2433282.5", "sun_earth_moon_2433282.5"):
    s = load_system(ROOT / "tests/golden/systems" / name)
    g = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    g.advance(12); g.advance(1000); g.sync()
    t = time.time(); g.advance(STEPS); g.sync(); w = time.time() - t
    print(f"{name}: {s.n} bodies, {w / STEPS * 1e6:.3f} us per step (no solout)", flush=True)
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
for K in (1, 2, 16, 256, 1024):
    gs = [ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt * (1 if i % 2 == 0 else -1)) for i in range(K)]
    ea.advance_many(gs, 12); ea.advance_many(gs, 1000)
    for g in gs: g.sync()
    n = 200000
    t = time.time(); ea.advance_many(gs, n)
    for g in gs: g.sync()
    w = time.time() - t
    print(f"gang of {K:4d} x 32 bodies: {w / n * 1e6:.3f} us per step of the gang, {K * s.n * n / w:.3e} body-steps/s", flush=True)
p = ea.NBodyPropagator.from_system(s)
t = time.time(); p.step_n(1_000_000); w = time.time() - t
print(f"configs[1]: 1e6 steps with solout and fits: {w:.3f} s", flush=True)
fw, bw = ea.NBodyPropagator.from_system(s), ea.NBodyPropagator.from_system(s, direction=ea.BACKWARD)
t = time.time(); ea.step_n_many([fw, bw], 1_000_000); w = time.time() - t
print(f"forward + backward, 1e6 steps each, stepped together: {w:.3f} s", flush=True)
