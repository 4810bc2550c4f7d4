"""Soak: the 4096-body Plummer sphere for many steps with two role layouts of the workgroup step kernel (separate processes:
EPH_WG_LAYOUT is read once), SHA-256 of positions and velocities at the end must agree. Layout 0 is round 1's kernel.
usage: python scripts/soak_layouts.py [steps [layouts, comma separated; default 0,5]]   -> one JSON line"""
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import ephemeris_explorer_amd as ea
    from ephemeris_explorer_amd.workloads import plummer
    steps = int(sys.argv[2])
    pos, vel, mu = plummer(4096)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    t = time.perf_counter()
    g.advance(steps)
    p, v, tt, sc = g.state()
    print(json.dumps({"sha_pos": hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest(),
                      "sha_vel": hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest(), "time": tt, "steps": sc,
                      "seconds": time.perf_counter() - t}))
    sys.exit(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
layouts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "5"]
out = {}
for layout in layouts:
    r = subprocess.run([sys.executable, __file__, "child", str(steps)], env=dict(os.environ, EPH_WG_LAYOUT=layout),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out[layout] = json.loads(r.stdout.strip().splitlines()[-1])
same = len({(o["sha_pos"], o["sha_vel"]) for o in out.values()}) == 1
print(json.dumps({"workload": "plummer_4096_f64_qt12", "steps": steps, "layouts": out, "bit_identical": same}))
assert same
