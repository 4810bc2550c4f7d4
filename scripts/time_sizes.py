"""Steady QuinlanTremaine12 step over system sizes (HIP events on the handle's stream + wall clock), after a prewarm.
usage (GPU box): python scripts/time_sizes.py [n ...]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192]
for n in sizes:
    pos, vel, mu = plummer(n)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    g.advance(12)
    steps = max(50, int(2000 * (4096 / n) ** 2 / 8))
    t = time.time()
    while time.time() - t < 1.0:
        g.advance(steps); g.sync()
    g.enable_timing(True)
    best = None
    for rep in range(5):
        t = time.time(); g.advance(steps); g.sync(); w = (time.time() - t) / steps * 1e6
        best = w if best is None or w < best else best
    ms, l = g.kernel_time()
    print(f"N={n}: {ms / l * 1e3:.2f} us per step (events, mean of 5 x {steps}), best wall {best:.2f}", flush=True)
