"""Ad hoc: steady QuinlanTremaine12 step of the small committed systems (persistent kernel, no solout)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system
for name in ("full_solar_system_2433282.5", "simple_solar_system_2433282.5", "sun_earth_moon_2433282.5"):
    s = load_system(ROOT / "tests/golden/systems" / name)
    g = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    g.advance(12); g.advance(1000); g.sync()
    t = time.time(); g.advance(500000); g.sync(); w = time.time() - t
    print(f"{name}: {s.n} bodies, {w / 500000 * 1e6:.3f} us per step (no solout)", flush=True)
