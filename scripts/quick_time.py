"""Ad-hoc timing on the GPU box (not the bench contract): steady-state QT12 steps."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
from ephemeris_explorer_amd.systems import load_system

print(ea.device_name())
for n, steps in ((4096, 200), (1024, 200), (16384, 20)):
    pos, vel, mu = plummer(n)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    g.advance(12)
    g.enable_timing(True)
    g.advance(10)
    ms0, l0 = g.kernel_time()
    t = time.time(); g.advance(steps); g.state(); wall = time.time() - t
    ms, l = g.kernel_time()
    per = (ms - ms0) / (l - l0) * 1e3
    print(f"N={n}: {per:.2f} us/step (events), wall {wall/steps*1e6:.2f} us/step -> {n/per*1e6:.3e} body-steps/s")
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
for path in (2, 1):
    g = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    g.set_path(path)
    g.advance(12)
    steps = 200000 if path == 2 else 5000
    t = time.time(); g.advance(steps); g.state(); wall = time.time() - t
    print(f"N=32 path {path}: wall {wall/steps*1e6:.3f} us/step -> {32/(wall/steps):.3e} body-steps/s")
p = ea.NBodyPropagator.from_system(s)
p.step_n(12)
t = time.time(); p.step_n(200000); wall = time.time() - t
print(f"N=32 propagator (solout+fits): {wall/200000*1e6:.3f} us/step")
