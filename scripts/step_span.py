"""Where one launch of the default step kernel spends its time (EPH_DEBUG_WG=4): s_memtime ticks of workgroup 7's chain wave
(whole force, wait for the first tiles) and the span from the earliest wave entry to the latest force completion over the grid."""
import ctypes
import os
import sys
from pathlib import Path
os.environ.setdefault("EPH_DEBUG_WG", "4")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea  # noqa: E402
from ephemeris_explorer_amd.workloads import plummer  # noqa: E402
pos, vel, mu = plummer(4096)
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
g.advance(12 + 50)
g.sync()
out = (ctypes.c_int64 * 8)()
# reset min/max, then one more step
import numpy as np  # noqa: E402
g.enable_timing(True)
for _ in range(3):
    g.advance(1)
    g.sync()
    ea._lib().eph_debug_wg_cycles(out)
    v = list(out)
    print(f"block 7 chain wave: force {v[0]} ticks (wait for first tiles {v[4]}, loop+wait {v[7]}); grid: earliest entry -> latest force end "
          f"{v[3] - v[2]} ticks (min/max are cumulative over launches: meaningful on the first line only)")
ms, n = g.kernel_time()
print("events per step (us):", ms / n * 1e3)
