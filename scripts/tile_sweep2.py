"""BASELINE.json configs[2]: "synthetic 4096-body Plummer sphere f64, 1xMI355X, LDS-tile-size sweep with rocprof HBM GB/s".
What "tile" means in the two paths of this build:
  * ordered path (default, bit-identical): sources per workgroup barrier -- 64 (layouts 0-2) or 128 (layouts 3-6); the LDS
    tile the chain wave consumes is always 64 sources wide (one wave64 of pair results per row);
  * opt-in fast path: the source slice one wave accumulates before the partial sums are combined: N / S for S slices,
    64 ... 1024 sources (no LDS: sources arrive by scalar loads).
One JSON line per point: us per step (HIP events over 2000 steps), body-steps/s, algorithmic HBM GB/s (680 B per body-step).
usage (GPU box): python scripts/tile_sweep2.py > gpurun_out/r02_tile_sweep.jsonl"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
N = 4096


def run(env, path):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, str(ROOT / "scripts" / "time_path.py"), str(N), "2000", str(path)], env=e,
                         capture_output=True, text=True).stdout
    us = float(out.split(": ")[1].split("us/step")[0])
    return us


for layout, tile in ((0, 64), (1, 64), (2, 64), (3, 128), (4, 128), (5, 128), (6, 128)):
    us = run({"EPH_WG_LAYOUT": str(layout)}, 0)
    print(json.dumps({"path": "ordered", "kernel": f"k_lm_step_wg<12,{layout}>", "sources_per_barrier": tile, "us_per_step": us,
                      "body_steps_per_s": N / us * 1e6, "algorithmic_hbm_gbs": 680.0 * N / us / 1e3}), flush=True)
for bpw in (1, 2, 4, 8):
    us = run({"EPH_FORCE": "wave", "EPH_BPW": str(bpw)}, 0)
    print(json.dumps({"path": "ordered", "kernel": f"k_lm_step<{bpw},12>", "sources_per_tile": 64, "bodies_per_wave": bpw,
                      "us_per_step": us, "body_steps_per_s": N / us * 1e6, "algorithmic_hbm_gbs": 680.0 * N / us / 1e3}), flush=True)
for S in (4, 8, 16, 32, 64):
    us = run({"EPH_FAST_SLICES": str(S)}, 4)
    print(json.dumps({"path": "fast (opt-in)", "kernel": "k_fast_partial<4> + k_fast_finish<12>", "slices": S,
                      "sources_per_slice": N // S, "us_per_step": us, "body_steps_per_s": N / us * 1e6,
                      "algorithmic_hbm_gbs": 680.0 * N / us / 1e3}), flush=True)
