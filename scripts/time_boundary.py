"""What the boundary costs when the caller insists on host buffers ("PCIe-inclusive"): seam 1
(eph_accel_eval: positions and mu in, accelerations out, every call) and seam 2 driven one step at a time with the state read
back after every step, against the resident rate bench.py reports."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
for n in (32, 4096, 65536):
    pos, vel, mu = plummer(n) if n <= 4096 else (np.random.default_rng(1).normal(size=(n, 3)), np.zeros((n, 3)), np.full(n, 1.0 / n))
    ea.accel_eval(pos, mu)
    reps = 2000 if n == 32 else 200 if n == 4096 else 5
    t = time.time()
    for _ in range(reps):
        ea.accel_eval(pos, mu)
    w = (time.time() - t) / reps
    print(f"seam 1, N={n}: eph_accel_eval with host buffers {w * 1e6:.1f} us per evaluation ({n / w:.3e} body-evaluations/s)")
pos, vel, mu = plummer(4096)
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
g.advance(12 + 200); g.sync()
t = time.time(); g.advance(2000); g.sync(); w_res = (time.time() - t) / 2000
t = time.time()
for _ in range(500):
    g.advance(1); g.state()
w_host = (time.time() - t) / 500
t = time.time()
for _ in range(50):
    g.advance(10); g.state()
w_host10 = (time.time() - t) / 500
print(f"seam 2, N=4096: resident {w_res * 1e6:.1f} us per step ({4096 / w_res:.3e} body-steps/s); state read back to the host after every "
      f"step {w_host * 1e6:.1f} us per step ({4096 / w_host:.3e}); after every 10th step {w_host10 * 1e6:.1f} us per step ({4096 / w_host10:.3e})")
