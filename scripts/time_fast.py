"""us per step of the opt-in fast paths (EPH_FAST_FUSED=0|1 selects the two-launch / one-launch form): python scripts/time_fast.py"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
out = {"fused": os.environ.get("EPH_FAST_FUSED", "1")}
for n, path, name in ((4096, 4, "fast"), (4096, 5, "fast-rsq"), (4096, 6, "f32"), (65536, 6, "f32-65536"), (16384, 5, "fast-rsq-16384")):
    pos, vel, mu = plummer(n)
    g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
    g.set_path(path)
    g.advance(12 + 50); g.sync()
    k = 2000 if n <= 4096 else 200
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); g.advance(k); g.sync(); best = min(best, (time.perf_counter() - t0) / k * 1e6)
    out[name] = round(best, 3)
print(json.dumps(out))
