"""usage: run_case.py plummer N steps | system NAME steps [path]   -- a single timed case (used under rocprofv3)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
from ephemeris_explorer_amd.systems import load_system

kind = sys.argv[1]
if kind == "plummer":
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    pos, vel, mu = plummer(n)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
else:
    s = load_system(ROOT / "tests/golden/systems" / sys.argv[2])
    steps = int(sys.argv[3])
    g = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    if len(sys.argv) > 4:
        g.set_path(int(sys.argv[4]))
g.advance(12)
g.sync()
g.enable_timing(True)
t = time.time(); g.advance(steps); g.sync(); wall = time.time() - t
ms, l = g.kernel_time()
print(f"{sys.argv[1:]}: events {ms/steps*1e3:.3f} us/step, wall {wall/steps*1e6:.3f} us/step")
