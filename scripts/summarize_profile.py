#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>/ (rocprofv3 CSVs from scripts/round_profile.sh) into small tracked files under profiles/.

usage: python scripts/summarize_profile.py r01
writes profiles/<tag>_bench.json, profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, profiles/traffic.json"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
src = ROOT / "gpurun_out" / tag
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)

shutil.copy(src / "stats_kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")
bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
(dst / f"{tag}_bench.json").write_text(json.dumps(bench, indent=1) + "\n")


for extra in ("bench_fast", "bench_fast_rsq", "bench_craft", "bench_sharded", "bench_f32pairs", "bench_f32pairs_4096",
              "bench_craft_mixed", "bench_craft_mixed_queue", "bench_craft_mixed_static", "bench_craft_1m", "bench_gpus2_shared_device", "bench_gpus4_shared_device",
              "bench_20a", "bench_20b"):                                                              # the round's other bench lines
    f = src / f"{extra}.json"
    if f.exists() and f.read_text().strip():
        (dst / f"{tag}_{extra}.json").write_text(json.dumps(json.loads(f.read_text().strip().splitlines()[-1]), indent=1) + "\n")
for extra in ("time_small.txt", "time_sizes.txt"):
    f = src / extra
    if f.exists():
        shutil.copy(f, dst / f"{tag}_{extra}")
for extra in ("fast_stats", "craft_stats", "f32_stats"):
    f = src / f"{extra}_kernel_stats.csv"
    if f.exists():
        shutil.copy(f, dst / f"{tag}_{extra.replace('_stats', '')}_kernel_stats.csv")


def per_dispatch(name):
    rows = list(csv.DictReader(open(src / f"{name}_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: {"dispatches": len(disp[k]), **{c: v / len(disp[k]) for c, v in cs.items()}} for k, cs in agg.items()}


pmc = {}
for name in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for k, v in per_dispatch(name).items():
        if "lm_step" in k or "lm_persistent" in k:
            pmc.setdefault(k, {}).update(v)
if (src / "fast_pmc_sq_counter_collection.csv").exists():
    fast = {k: v for k, v in per_dispatch("fast_pmc_sq").items() if "k_fast" in k}
    for extra in ("fast_pmc_fetch", "fast_pmc_write"):
        if (src / f"{extra}_counter_collection.csv").exists():
            for k, v in per_dispatch(extra).items():
                if "k_fast" in k:
                    fast.setdefault(k, {}).update(v)
    (dst / f"{tag}_fast_pmc.json").write_text(json.dumps(fast, indent=1) + "\n")
    if all("FETCH_SIZE" in v for v in fast.values()) and fast:
        # one step of the fast path = one k_fast_partial + one k_fast_finish launch
        total = sum((2.0 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024.0 for v in fast.values())
        (dst / "traffic_fast.json").write_text(json.dumps({
            "source": f"profiles/{tag}_fast_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; sum of the step's two launches)",
            "traffic_bytes_per_launch": total}, indent=1) + "\n")
stats = {r["Name"]: r for r in csv.DictReader(open(src / "stats_kernel_stats.csv"))}
out = {"note": "per-dispatch averages; SQ_* counters are summed over the chip; FETCH_SIZE/WRITE_SIZE in KiB. "
               "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read, so "
               "traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB; Infinity-Cache hits are counted, so this is an upper "
               "bound on HBM bytes (the 3.7 MB working set is cache resident).",
       "kernels": {}}
for k, v in pmc.items():
    e = dict(v)
    if k in stats:
        e["avg_ns_kernel_trace"] = float(stats[k]["AverageNs"])
        e["calls"] = int(stats[k]["Calls"])
    if "FETCH_SIZE" in e:
        e["traffic_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e.get("WRITE_SIZE", 0.0)) * 1024.0
    out["kernels"][k] = e
(dst / f"{tag}_pmc.json").write_text(json.dumps(out, indent=1) + "\n")
main = max(out["kernels"].items(), key=lambda kv: kv[1].get("calls", 0) * kv[1].get("avg_ns_kernel_trace", 0))
(dst / "traffic.json").write_text(json.dumps({
    "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes)",
    "kernel": main[0], "traffic_bytes_per_launch": main[1].get("traffic_bytes_per_launch"),
    "valu_wave_insts_per_launch": main[1].get("SQ_INSTS_VALU"),
    "avg_ns_kernel_trace": main[1].get("avg_ns_kernel_trace")}, indent=1) + "\n")
print(json.dumps(out, indent=1)[:3000])
print("bench:", bench["value"], bench["roofline"]["launch_us"])
