import sys, ctypes
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
n = int(sys.argv[1])
pos, vel, mu = plummer(n)
import time
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0, "BlanesMoan6B")
g.advance(2); g.sync()
t0 = time.time(); g.advance(20); g.sync(); dt = time.time() - t0
evals = 20 * 6
print(f"BlanesMoan6B: {dt/evals*1e6:.1f} us per force evaluation (+ kick/drift launch)")
out = (ctypes.c_int64 * 8)()
ea._lib().eph_debug_wg_cycles(out)
v = list(out)
t = v[6]
print(f"n={n} tiles={t}: pair wave(5 bodies) work {v[0]/t:.0f} wait {v[1]/t:.0f} | pair wave 0 (2 bodies) work {v[2]/t:.0f} wait {v[3]/t:.0f} | chain work {v[4]/t:.0f} wait {v[5]/t:.0f}  cycles/tile; chain wave total {v[7]} cycles -> if the kernel took T us, clock = {v[7]}/T MHz")
