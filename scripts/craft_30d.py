"""BASELINE.json configs[3] at its full specification on ONE GPU (builder-run; the 8-GPU form shards the craft):
full_solar_system ephemeris + 1e6 massless spacecraft (Mars Transfer Ship state perturbed by normal(0, 100 km / 0.01 km/s),
seed 20260926), Verner87, tol 1e-3 km, 30 days. The knots of 1e6 craft x 30 days (~1.4 TB) do not fit 288 GB, so the sweep
runs in one-day legs with the drain point between them (eph_craft_batch_reset_knots: the newest knot becomes knot 0 of an
empty slab) -- a consumer would copy each leg's slab out before the reset. Checks: every craft finishes; a sample of craft
equals the CPU oracle's uninterrupted 30-day propagation bit for bit (final state and step count).
usage: python scripts/craft_30d.py [n_craft] [days] -> one JSON line"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

import ephemeris_explorer_amd as ea  # noqa: E402
from ephemeris_explorer_amd.systems import load_ship, load_system  # noqa: E402
from oracle import orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
days = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
sysdir = ROOT / "tests/golden/systems/full_solar_system_2433282.5"
s = load_system(sysdir)
ship = load_ship(sysdir / "ships" / "Mars Transfer Ship.json")
end_eph = s.epoch + (days + 45.0) * 86400.0
sol = ea.NBodyPropagator.from_system(s).propagate(end_eph)
eph = ea.Ephemeris(sol, s.mu)
rng = np.random.default_rng(20260926)
pos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
vel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
leg_days = 1.0
batch = ea.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=1100)
t0 = time.perf_counter()
kernel_ms = 0.0
legs = int(round(days / leg_days))
for leg in range(1, legs + 1):
    batch.propagate(ship.start + leg * leg_days * 86400.0)
    st = batch.status()
    assert (st["status"] == 0).all(), (leg, np.unique(st["status"]))
    if leg < legs:
        batch.reset_knots()
elapsed = time.perf_counter() - t0
kernel_ms = batch.kernel_ms()
st, fin = batch.status(), batch.state()
steps = int(st["steps"].astype(np.int64).sum())
attempts = int(st["attempts"].astype(np.int64).sum())
# parity on a sample: the oracle's uninterrupted propagation
o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree, native=True)
assert o.step_to(end_eph) == 0
osol = o.take_solution()
sample = [0, 1, n // 3, n // 2, n - 2, n - 1]
same = 0
for i in sample:
    c = orc.Craft(osol, s.mu, ship.start, pos[i], vel[i], "Verner87")
    assert c.step_to(ship.start + days * 86400.0) == 0
    ct, cp, cv = c.knots()
    ok = ct[-1] == fin["t"][i] and np.array_equal(cp[-1], fin["pos"][i]) and np.array_equal(cv[-1], fin["vel"][i]) and \
        len(ct) - 1 == int(st["steps"][i])
    same += bool(ok)
print(json.dumps({
    "workload": f"full_solar_system ephemeris + {n} craft x {days} d, Verner87 tol 1e-3 (BASELINE.json configs[3], 1 GPU)",
    "legs": legs, "leg_days": leg_days, "craft_steps": steps, "attempts": attempts, "seconds": elapsed,
    "craft_steps_per_s": steps / elapsed, "kernel_seconds": kernel_ms * 1e-3, "craft_steps_per_s_kernel": steps / (kernel_ms * 1e-3),
    "fp64_tflops_kernel": attempts * 13.0 * 32 * 74.0 / (kernel_ms * 1e-3) / 1e12,
    "mean_steps_per_craft": steps / n, "sample_checked": len(sample), "sample_bit_identical_to_oracle": same,
    "knot_slab_gb_per_leg": n * 1100 * 56 / 1e9}))
assert same == len(sample)
