"""Ad-hoc timing on the GPU box: steady QuinlanTremaine12 steps of the Plummer sphere on a chosen path.
usage: python scripts/time_path.py N STEPS PATH [METHOD]   (PATH 0 = default ordered, 3 = workgroup kernel, 4 = fast)"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

import ephemeris_explorer_amd as ea  # noqa: E402
from ephemeris_explorer_amd.workloads import plummer  # noqa: E402

n, steps, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
method = sys.argv[4] if len(sys.argv) > 4 else "QuinlanTremaine12"
pos, vel, mu = plummer(n)
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0, method)
g.set_path(path)
g.advance(12)
g.advance(20)
g.enable_timing(True)
t = time.time()
g.advance(steps)
g.sync()
wall = time.time() - t
ms, launches = g.kernel_time()
per = ms / launches * 1e3 if launches else wall / steps * 1e6   # SRKN steps are not event-timed: wall
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("EPH_"))
print(f"{method} path={path} N={n} {tag}: {per:.2f} us/step (events), wall {wall / steps * 1e6:.2f} us/step -> "
      f"{n / per * 1e6:.3e} body-steps/s", flush=True)
if path == 4:
    e = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    e.advance(12 + 20 + steps)
    print("   max |dpos| vs ordered path after", 32 + steps, "steps:", np.abs(e.state()[0] - g.state()[0]).max(), flush=True)
