"""A/B of the sweep kernel between library builds on ONE box: python scripts/ab_craft.py [n_craft] [days] [reps]
(EPH_AMD_LIBRARY selects the library). Prints the kernel time of every repetition (a new batch each, one ephemeris)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system, load_ship

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
days = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
method = sys.argv[4] if len(sys.argv) > 4 else "Verner87"
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
ship = load_ship(ROOT / "tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + (days + 40.0) * 86400.0)
eph = ea.Ephemeris(sol, s.mu)
rng = np.random.default_rng(20260926)
pos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
vel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
ms, chk = [], None
for _ in range(reps):
    batch = ea.SpacecraftBatch(eph, ship.start, pos, vel, method, max_knots=int(1200 * days) + 64)
    batch.propagate(ship.start + days * 86400.0)
    ms.append(round(batch.kernel_ms(), 3))
    st = batch.status()
    chk = (int(st["steps"].sum()), int(st["attempts"].sum()), float(batch.state()["pos"].sum()))
    del batch
print(json.dumps({"lib": str(ea.LIB_PATH.name), "method": method, "n": n, "days": days, "kernel_ms": ms, "min": min(ms), "check": chk}))
