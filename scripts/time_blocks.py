"""What a call costs around its kernels: blocks of K QT12 steps, each followed by sync(), for K = 1, 20, 200 at N = 4096 and
K = 1, 100 at 32 bodies. Prints the median block in us and per step (the K = 200 figure is the kernel; the K = 20 block of the driver's
bench form carries the launch ramp and the wait once per 20 steps).   usage (GPU box): python scripts/time_blocks.py"""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea  # noqa: E402
from ephemeris_explorer_amd.systems import load_system  # noqa: E402
from ephemeris_explorer_amd.workloads import plummer  # noqa: E402

pos, vel, mu = plummer(4096)
big = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
small = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
tag = os.environ.get("TAG", "blocks")
for name, g, ks in (("plummer_4096", big, (1, 20, 200)), ("solar_32", small, (1, 100))):
    g.advance(12)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 1.0:
        g.advance(200)
        g.sync()
    for k in ks:
        reps = max(30, min(2000, int(0.25 / (k * (36e-6 if g is big else 1e-6) + 30e-6))))
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            g.advance(k)
            g.sync()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{tag} {name} K={k:4d}: block {med * 1e6:9.2f} us (min {ts[0] * 1e6:.2f}), {med / k * 1e6:8.3f} us per step, {reps} blocks", flush=True)
