"""Ad-hoc timing on the GPU box (not the bench contract): steady-state QT12 steps."""
import os, sys, time, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np

def run_n(ns):
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    for n, steps in ns:
        pos, vel, mu = plummer(n)
        g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
        g.advance(12)
        g.advance(10)
        g.enable_timing(True)
        t = time.time(); g.advance(steps); g.sync(); wall = time.time() - t
        ms, l = g.kernel_time()
        per = ms / l * 1e3
        print(f"BPW={os.environ.get('EPH_BPW','auto')} N={n}: {per:.2f} us/step (events), wall {wall/steps*1e6:.2f} us/step -> {n/per*1e6:.3e} body-steps/s", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "sub":
    run_n([(int(sys.argv[2]), int(sys.argv[3]))])
    sys.exit(0)

import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system
print(ea.device_name())
for bpw in ("1", "2", "4", "8"):
    for n, steps in ((4096, 200), (1024, 200), (16384, 20)):
        env = dict(os.environ, EPH_BPW=bpw)
        subprocess.run([sys.executable, __file__, "sub", str(n), str(steps)], env=env)
for name in ("full_solar_system_2433282.5", "sun_earth_moon_2433282.5"):
    s = load_system(ROOT / "tests/golden/systems" / name)
    for path in (2, 1):
        g = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
        g.set_path(path)
        g.advance(12)
        steps = 200000 if path == 2 else 5000
        t = time.time(); g.advance(steps); g.sync(); wall = time.time() - t
        print(f"{name} N={s.n} path {path}: wall {wall/steps*1e6:.3f} us/step -> {s.n/(wall/steps):.3e} body-steps/s")
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
p = ea.NBodyPropagator.from_system(s)
p.step_n(12)
t = time.time(); p.step_n(200000); wall = time.time() - t
print(f"N=32 propagator (solout+fits): {wall/200000*1e6:.3f} us/step")
