"""k_lm_small: one system, and gangs of K systems per launch (us per step of the gang)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
for K in (1, 2, 16, 128, 256):
    gs = [ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt * (1 if i % 2 == 0 else -1)) for i in range(K)]
    ea.advance_many(gs, 12); ea.advance_many(gs, 1000)
    for g in gs: g.sync()
    n = 200000
    best = 1e9
    for rep in range(3):
        t = time.time(); ea.advance_many(gs, n)
        for g in gs: g.sync()
        best = min(best, time.time() - t)
    print(f"gang of {K:4d} x 32 bodies: {best / n * 1e6:.3f} us per step of the gang, {K * s.n * n / best:.3e} body-steps/s", flush=True)
