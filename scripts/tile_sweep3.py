"""BASELINE.json configs[2] ("4096-body Plummer sphere f64, LDS-tile-size sweep with rocprof HBM GB/s") at HEAD.
One JSON line per point: us per step (HIP events over STEPS steps), body-steps/s, algorithmic GB/s (680 B per body-step), and --
for the points marked pmc -- the fabric-side read traffic of the step kernel from `rocprofv3 --pmc FETCH_SIZE` (doubled: gfx950
tallies a 128-byte request as 64, MI355X_MICROARCH.md; Infinity-Cache hits are counted, so this bounds HBM reads from above).
What "tile" means here: sources per workgroup barrier (64 or 128), bodies per workgroup (chains per chain wave) in the ordered
path; sources per slice in the opt-in fast path.
usage (GPU box, from the repository root): python scripts/tile_sweep3.py [LIB] > gpurun_out/<tag>/tile_sweep.jsonl
LIB: a library built with -DEPH_EXPERIMENTS (the retired layouts live there); default: the product library (layout 5 only)."""
import csv, json, os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
N, STEPS = 4096, 2000
lib = sys.argv[1] if len(sys.argv) > 1 else None


def run(env, path, pmc=False):
    e = dict(os.environ, **env)
    if lib:
        e["EPH_AMD_LIBRARY"] = lib
    cmd = [sys.executable, str(ROOT / "scripts" / "time_path.py"), str(N), str(STEPS), str(path)]
    out = subprocess.run(cmd, env=e, capture_output=True, text=True).stdout
    us = float(out.split(": ")[1].split("us/step")[0])
    fetch = None
    if pmc:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            e2 = dict(e, TMPDIR="/tmp")
            cmd2 = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", "FETCH_SIZE", "-d", d, "-o", "p", "--",
                    sys.executable, str(ROOT / "scripts" / "time_path.py"), str(N), "200", str(path)]
            subprocess.run(cmd2, env=e2, capture_output=True, text=True, cwd="/tmp")
            tot, cnt = 0.0, 0
            for f in Path(d).rglob("*counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == "FETCH_SIZE" and ("k_lm_step" in r["Kernel_Name"] or "k_fast_partial" in r["Kernel_Name"]):
                        tot += float(r["Counter_Value"]); cnt += 1
            if cnt:
                fetch = 2.0 * tot / cnt * 1024.0          # bytes per launch (KiB counter, doubled)
    return us, fetch


def emit(d, us, fetch):
    d.update({"us_per_step": us, "body_steps_per_s": N / us * 1e6, "algorithmic_hbm_gbs": 680.0 * N / us / 1e3})
    if fetch is not None:
        d.update({"fetch_bytes_per_launch": fetch, "fetch_gbs": fetch / us / 1e3, "fetch_over_algorithmic": fetch / (680.0 * N)})
    print(json.dumps(d), flush=True)


us, f = run({}, 0, True)        # (the other role layouts of rounds 2-3 left the source in round 4: profiles/r03_tile_sweep.jsonl)
emit({"path": "ordered", "kernel": "k_lm_step_wg<12,16>", "sources_per_barrier": 128, "bodies_per_workgroup": 16}, us, f)
for wb in (8, 4):
    us, f = run({"EPH_WG_BODIES": str(wb)}, 0, True)
    emit({"path": "ordered", "kernel": f"k_lm_step_wg<12,{wb}>", "sources_per_barrier": 128 if wb == 8 else 64, "bodies_per_workgroup": wb}, us, f)
for bpw in (1, 2, 4, 8):
    us, f = run({"EPH_FORCE": "wave", "EPH_BPW": str(bpw)}, 0, bpw == 4)
    emit({"path": "ordered", "kernel": f"k_lm_step<{bpw},12>", "sources_per_tile": 64, "bodies_per_wave": bpw}, us, f)
for S in (4, 8, 16, 32, 64):
    us, f = run({"EPH_FAST_SLICES": str(S)}, 4, S == 32)
    emit({"path": "fast (opt-in)", "kernel": "k_fast_partial<4> + k_fast_finish<12>", "slices": S, "sources_per_slice": N // S}, us, f)
