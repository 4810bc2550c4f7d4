"""Propagator (solout on) vs bare integration at N = 4096: what sampling + least-squares fits + spline hand-over cost per step."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import ephemeris_explorer_amd as ea  # noqa: E402
from ephemeris_explorer_amd.workloads import plummer  # noqa: E402
n = 4096
pos, vel, mu = plummer(n)
H = 1.0 / 1024.0
for count in (1, 4, 32):
    p = ea.NBodyPropagator(pos, vel, mu, 0.0, H, ea.FORWARD, np.full(n, count, np.uint32), np.full(n, 6, np.uint32))
    p.step_n(12 + 100)
    p.integration().sync()
    t = time.perf_counter()
    p.step_n(4000)
    p.integration().sync()
    dt = time.perf_counter() - t
    sol = p.take_solution()
    print(f"propagator count={count}: {dt / 4000 * 1e6:.2f} us/step wall, {sol.info(0)[2]} polynomials for body 0", flush=True)
g = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
g.advance(112); g.sync()
t = time.perf_counter(); g.advance(4000); g.sync(); dt = time.perf_counter() - t
print(f"integration only: {dt / 4000 * 1e6:.2f} us/step wall")
