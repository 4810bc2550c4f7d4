"""One-off evidence run: Plummer N = 4096, QuinlanTremaine12, GPU vs the CPU oracle at 10^k-th steps (bit patterns).
usage: python scripts/soak_parity.py [steps]   (the oracle needs ~25 ms per step on one core)"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
from oracle import orc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pos, vel, mu = plummer(4096)
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
o = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0, native=True)
marks = sorted({12, 13, 22, 112, steps + 12} | {12 + 10 ** k for k in range(1, 7) if 10 ** k <= steps})
done, out = 0, []
t0 = time.time()
for m in marks:
    g.advance(m - done)
    assert o.advance(m - done) == 0
    done = m
    gp, gv, gt, _ = g.state()
    op, ov, ot, _ = o.state()
    out.append({"integrator_step": m, "max_abs_dpos": float(np.abs(gp - op).max()), "max_abs_dvel": float(np.abs(gv - ov).max()),
                "bits_equal": bool(np.array_equal(gp, op) and np.array_equal(gv, ov) and gt == ot)})
    print(out[-1], flush=True)
print(json.dumps({"n": 4096, "steps_after_startup": steps, "seconds": time.time() - t0, "checks": out}))
