"""Where a sweep's wall time goes: propagate (kernel + launch + wait) and summary (pack kernel + one copy), per batch.
usage (GPU box): [EPH_CRAFT_SORT=0] python scripts/time_sweep_parts.py [n_craft]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_ship, load_system
from ephemeris_explorer_amd.workloads import craft_population
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
sysdir = ROOT / "tests/golden/systems/full_solar_system_2433282.5"
s = load_system(sysdir); ship = load_ship(sysdir / "ships" / "Mars Transfer Ship.json")
sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + 41 * 86400.0)
eph = ea.Ephemeris(sol, s.mu)
pos, vel, fam = craft_population("transfer", n, s, ship)
t_end = ship.start + 0.25 * 86400.0
for rep in range(4):
    t0 = time.perf_counter()
    b = ea.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=364)
    t1 = time.perf_counter()
    b.propagate(t_end)
    t2 = time.perf_counter()
    st = b.summary()
    t3 = time.perf_counter()
    print(f"create {1e3 * (t1 - t0):.1f} ms, propagate {1e3 * (t2 - t1):.1f} ms (kernel {b.kernel_ms():.1f}), summary {1e3 * (t3 - t2):.1f} ms", flush=True)
