"""Massless sweep throughput (BASELINE.json configs[3], bounded): full_solar_system ephemeris + N spacecraft
(Mars Transfer Ship state perturbed by normal(0, 100 km / 0.01 km/s) per component, seed 20260926), Verner87,
tol 1e-3 km, no burns, `days` days. Prints craft-steps/s and RHS evaluations/s, and the CPU oracle beside it.
usage: python scripts/bench_craft.py [n_craft] [days]"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system, load_ship

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
days = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
ship = load_ship(ROOT / "tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
t0 = time.time()
prop = ea.NBodyPropagator.from_system(s)
sol = prop.propagate(s.epoch + (days + 40.0) * 86400.0)
t_eph = time.time() - t0
eph = ea.Ephemeris(sol, s.mu)
rng = np.random.default_rng(20260926)
pos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
vel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
max_knots = int(1200 * days) + 64
batch = ea.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=max_knots)
t0 = time.time()
batch.propagate(ship.start + days * 86400.0)
wall = time.time() - t0
st = batch.status()
ok = int((st["status"] == 0).sum())
steps = int(st["steps"].sum()); attempts = int(st["attempts"].sum())
ms = batch.kernel_ms()
out = {"n_craft": n, "days": days, "ok": ok, "accepted_steps": steps, "attempts": attempts,
       "kernel_ms": ms, "wall_s": wall, "ephemeris_build_s": t_eph,
       "craft_steps_per_s": steps / (ms * 1e-3), "rhs_evals_per_s": attempts * 13 / (ms * 1e-3),
       "body_evals_per_s": attempts * 13 * s.n / (ms * 1e-3)}
# CPU oracle on a few craft of the same batch
from oracle import orc
o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
o.step_to(s.epoch + (days + 40.0) * 86400.0)
osol = o.take_solution()
m = min(n, 16)
t0 = time.time(); csteps = 0
for i in range(m):
    c = orc.Craft(osol, s.mu, ship.start, pos[i], vel[i], "Verner87")
    assert c.step_to(ship.start + days * 86400.0) == 0
    csteps += c.state()["steps"]
    assert len(c.knots()[0]) == st["nknots"][i]
cpu = time.time() - t0
out["cpu_oracle_craft_steps_per_s"] = csteps / cpu
out["speedup_vs_1_core"] = out["craft_steps_per_s"] / out["cpu_oracle_craft_steps_per_s"]
print(json.dumps(out))
