import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
H = 1.0 / 1024.0
n = 256
pos, vel, mu = plummer(n)
count = np.full(n, 2, np.uint32); degree = np.full(n, 6, np.uint32)
a = ea.NBodyPropagator(pos, vel, mu, 0.0, H, ea.FORWARD, count, degree)
b = ea.NBodyPropagator(pos, vel, mu, 0.0, H, ea.FORWARD, count, degree)
b.integration().set_path(4)
for k in (12, 4, 20, 164):
    a.step_n(k); b.step_n(k)
    print(k, "state diff", np.abs(a.state()[0] - b.state()[0]).max(), np.abs(a.state()[1] - b.state()[1]).max())
sa, sb = a.take_solution(), b.take_solution()
for body in (0, 100, 255):
    print(body, sa.info(body), np.abs(sa.coeffs(body)[0] - sb.coeffs(body)[0]).max())
g = ea.NBodyIntegration(pos, vel, mu, 0.0, H); g.set_path(4); e = ea.NBodyIntegration(pos, vel, mu, 0.0, H)
g.advance(200); e.advance(200)
print("integration only:", np.abs(g.state()[0] - e.state()[0]).max())
