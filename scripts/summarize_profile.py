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
sys.path.insert(0, str(ROOT))


def keep_measurement_commit(path, new):
    """A counter file is re-written every time this script runs; `profile_commit` is the commit the COUNTERS were taken at, so it is
    carried over from the existing file when the counters and the source hashes are the ones already recorded there."""
    try:
        old = json.loads(Path(path).read_text())
    except (OSError, ValueError):
        return new
    same = all(old.get(k) == new.get(k) for k in ("source_sha256_16", "avg_ns_kernel_trace", "traffic_bytes_per_launch",
                                                   "traffic_bytes_per_attempt", "valu_wave_insts_per_launch"))
    if same and old.get("profile_commit"):
        new = dict(new, profile_commit=old["profile_commit"])
    return new

from ephemeris_explorer_amd.workloads import profile_stamp      # sha256 of the kernel sources a counter file was taken with


def craft_code_object():
    """registers / spills / scratch / LDS / occupancy of the 13-stage sweep kernels as THIS tree compiles them (order 0):
    hipcc -Rpass-analysis=kernel-resource-usage on csrc/craft_sweep.hip"""
    import re
    import subprocess
    from ephemeris_explorer_amd import build as b
    cmd = [b.hipcc(), *b.FLAGS, "-DEPH_PAIR_VARIANT=0", "-x", "hip", "--offload-device-only", "-c", str(b.CSRC / "craft_sweep.hip"),
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    out = {"source": "hipcc -Rpass-analysis=kernel-resource-usage of csrc/craft_sweep.hip (order 0) at summarize time"}
    want = {"17k_craft_propagateILi13ELb0ELb0ELi2EE": "k_craft_propagate<13,false,false,2>",
            "17k_craft_propagateILi13ELb0ELb0ELi1EE": "k_craft_propagate<13,false,false,1>",
            "13k_craft_queueILi13ELb0ELb0EE": "k_craft_queue<13,false,false>"}
    cur = None
    for ln in text.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = next((v for k, v in want.items() if k in m.group(1)), None)
            if cur:
                out[cur] = {}
            continue
        if cur:
            for key, name in (("VGPRs:", "vgpr_count"), ("VGPRs Spill:", "vgpr_spill_count"), ("TotalSGPRs:", "sgpr_count"),
                              ("SGPRs Spill:", "sgpr_spill_count"), ("ScratchSize [bytes/lane]:", "scratch_bytes_per_lane"),
                              ("Occupancy [waves/SIMD]:", "waves_per_simd"), ("LDS Size [bytes/block]:", "lds_bytes_per_workgroup")):
                m = re.search(re.escape(key) + r"\s+(\d+)", ln)
                if m and " " + key in " " + ln.split("remark:")[-1].strip()[: len(key) + 1]:
                    out[cur][name] = int(m.group(1))
    return out

tag = sys.argv[1]
src = ROOT / "gpurun_out" / tag
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)

# the massless sweep's counters (scripts/prof_craft.sh <tag> -> gpurun_out/prof_<tag>/summary.json), if taken this round
craft = ROOT / "gpurun_out" / f"prof_{tag}" / "summary.json"
if craft.exists():
    c = json.loads(craft.read_text())
    b = c.pop("bench", {})
    att = float(b.get("attempts", 0)) or None
    ns = float(c["avg_ns"])
    q = 4.0                                            # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
    f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    flop = (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + 2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0)) * 64.0
    simd_cycles = 1024 * ns * 1e-9 * 2.4e9             # 256 CUs x 4 SIMDs at the 2.4 GHz shader clock
    derived = {
        "active_inst_any_over_wave_cycles": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "active_inst_valu_over_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
        "wait_inst_any_over_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "wait_any_over_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "mean_resident_waves_per_simd": c["SQ_WAVE_CYCLES"] * q / simd_cycles,
        "f64_wave_insts": f64, "f64_share_of_valu": f64 / c["SQ_INSTS_VALU"],
        "f64_issue_fraction_of_simd_cycles": f64 * 4.0 / simd_cycles,      # a wave64 f64 instruction occupies its SIMD for 4 cycles
        "valu_lane_ops_per_s": c["SQ_INSTS_VALU"] * 64.0 / (ns * 1e-9),
        "counted_tflops": flop / (ns * 1e-9) / 1e12,
        "salu_per_valu": c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"],
        "fetch_bytes_raw": c.get("FETCH_SIZE", 0.0) * 1024.0, "write_bytes_raw": c.get("WRITE_SIZE", 0.0) * 1024.0,
    }
    if att:
        terms = att * 13.0 * 32.0 / 64.0                # wave-level (craft, stage, body) terms
        derived.update({"attempts": att, "valu_wave_insts_per_body_term": c["SQ_INSTS_VALU"] / terms,
                        "f64_wave_insts_per_body_term": f64 / terms,
                        "write_bytes_per_accepted_step": derived["write_bytes_raw"] / float(b.get("accepted_steps", att))})
    code_object = craft_code_object()
    co2 = code_object.get("k_craft_propagate<13,false,false,2>", {})
    (dst / f"{tag}_craft_pmc.json").write_text(json.dumps({
        "command": f"scripts/prof_craft.sh {tag}: scripts/bench_craft.py {b.get('n_craft')} {b.get('days')} under rocprofv3 --kernel-include-regex k_craft, "
                   "one --pmc pass per line of the script (FETCH_SIZE and WRITE_SIZE in passes of their own)",
        "note": "sums over the chip for the ONE launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB, "
                "uncalibrated for this access pattern (scalar loads of coefficient rows, 8-byte knot stores): raw figures",
        "counters": c, "derived": derived, "code_object": code_object, "bench": b, **profile_stamp("craft")}, indent=1) + "\n")
    (dst / "traffic_craft.json").write_text(json.dumps(keep_measurement_commit(dst / "traffic_craft.json", {
        "source": f"profiles/{tag}_craft_pmc.json", "kernel": c["kernel"], "attempts": att,
        "traffic_bytes_per_attempt": (derived["fetch_bytes_raw"] + derived["write_bytes_raw"]) / att if att else None,
        "valu_wave_insts_per_attempt": c["SQ_INSTS_VALU"] / att if att else None,
        "f64_wave_insts_per_attempt": f64 / att if att else None,
        "active_inst_valu_over_wave_cycles": derived["active_inst_valu_over_wave_cycles"],
        "waves_per_simd": co2.get("waves_per_simd", 2), "vgpr_count": co2.get("vgpr_count"), "vgpr_spill_count": co2.get("vgpr_spill_count"),
        "scratch_bytes_per_lane": co2.get("scratch_bytes_per_lane"), "lds_bytes_per_workgroup": co2.get("lds_bytes_per_workgroup"),
        "avg_ns_kernel_trace": ns, **profile_stamp("craft")}), indent=1) + "\n")
    print(json.dumps(derived, indent=1))

if not (src / "stats_kernel_stats.csv").exists():          # only the sweep was profiled so far this round
    sys.exit(0)
shutil.copy(src / "stats_kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")
bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
(dst / f"{tag}_bench.json").write_text(json.dumps(bench, indent=1) + "\n")


for extra in ("bench_fast", "bench_fast_rsq", "bench_craft", "bench_sharded", "bench_f32pairs", "bench_f32pairs_4096",
              "bench_craft_mixed", "bench_craft_mixed_queue", "bench_craft_mixed_static", "bench_craft_1m", "bench_f32pairs_sharded_gpus2_shared_device", "bench_gpus2_shared_device", "bench_gpus4_shared_device",
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
(dst / "traffic.json").write_text(json.dumps(keep_measurement_commit(dst / "traffic.json", {
    "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes)",
    "kernel": main[0], "traffic_bytes_per_launch": main[1].get("traffic_bytes_per_launch"),
    "valu_wave_insts_per_launch": main[1].get("SQ_INSTS_VALU"),
    "avg_ns_kernel_trace": main[1].get("avg_ns_kernel_trace"),
    **profile_stamp("nbody")}), indent=1) + "\n")
print(json.dumps(out, indent=1)[:3000])
print("bench:", bench["value"], bench["roofline"]["launch_us"])

