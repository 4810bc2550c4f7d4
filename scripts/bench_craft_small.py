"""Ad hoc: massless sweep at small batch sizes, both kernel forms (EPH_CRAFT_FORM=wave|thread), kernel ms only.
usage: python scripts/bench_craft_small.py  (runs itself once per form)"""
import os, subprocess, sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 1:
    import numpy as np
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.systems import load_system, load_ship
    s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
    ship = load_ship(ROOT / "tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
    sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + 45 * 86400.0)
    eph = ea.Ephemeris(sol, s.mu)
    rng = np.random.default_rng(20260926)
    sizes = [int(x) for x in os.environ.get("SIZES", "1,64,1024,4096,16384,32768,65536").split(",")]
    for n in sizes:
        pos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
        vel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
        days = 1.0 if n <= 1024 else 0.25
        b = ea.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=int(1200 * days) + 64)
        b.propagate(ship.start + days * 86400.0)
        st = b.status()
        steps = int(st["steps"].sum())
        print(f"form={os.environ.get('EPH_CRAFT_FORM', 'auto'):6s} n={n:6d}: {b.kernel_ms():9.2f} ms, "
              f"{steps / (b.kernel_ms() * 1e-3):.3e} craft-steps/s, {b.kernel_ms() * 1e3 / (steps / n):.2f} us per step of one craft",
              flush=True)
else:
    for form in ("wave", "thread"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, EPH_CRAFT_FORM=form))
