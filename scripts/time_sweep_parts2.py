"""Where a sweep's wall time goes, in bench.py's own loop shape (batches created before the timed loop, then propagate +
summary per batch), per call: python wall of propagate / summary, the kernel's event time, and (EPH_TRACE_SUMMARY=1|2 on
stderr) the host timers inside eph_craft_batch_summary.
usage (GPU box): [EPH_CRAFT_SORT=0|1|2] [EPH_TRACE_SUMMARY=1|2] python scripts/time_sweep_parts2.py [n_craft] [fresh|reuse]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_ship, load_system
from ephemeris_explorer_amd.workloads import craft_population
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reuse = len(sys.argv) > 2 and sys.argv[2] == "reuse"
sysdir = ROOT / "tests/golden/systems/full_solar_system_2433282.5"
s = load_system(sysdir); ship = load_ship(sysdir / "ships" / "Mars Transfer Ship.json")
sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + 41 * 86400.0)
eph = ea.Ephemeris(sol, s.mu)
pos, vel, fam = craft_population("transfer", n, s, ship)
t_end = ship.start + 0.25 * 86400.0
def evicted():
    """KFD's per-process eviction clock (ms the process's queues spent evicted), summed over the GPUs"""
    import glob, os
    tot = 0
    for f in glob.glob(f"/sys/class/kfd/kfd/proc/{os.getpid()}/stats_*/evicted_ms"):
        try:
            tot += int(open(f).read().strip())
        except OSError:
            pass
    return tot
def make():
    return ea.SpacecraftBatch(eph, ship.start, pos, vel, "Verner87", max_knots=364)
for _ in range(2):                       # warm-up sweeps, batches freed (as bench.py does)
    b = make(); b.propagate(t_end); b.summary(); del b
out = np.zeros(n, dtype=ea.SpacecraftBatch.RECORD) if reuse else None
t0 = time.perf_counter()
batches = [make() for _ in range(5)]
print(f"5 x create {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
import os
pre = os.environ.get("PRE", "")
if "summary" in pre:
    for b in batches: b.summary()
if "sleep" in pre:
    time.sleep(0.2)
if "step" in pre:
    for b in batches: b.step_n(1)
tl0 = time.perf_counter()
for b in batches:
    t1 = time.perf_counter()
    b.propagate(t_end)
    t2 = time.perf_counter()
    st = b.summary(out)
    t3 = time.perf_counter()
    ok = bool((st["status"] == 0).all()); steps = int(st["steps"].sum())
    t4 = time.perf_counter()
    print(f"propagate {1e3 * (t2 - t1):.2f} ms (kernel so far {b.kernel_ms():.2f}), summary {1e3 * (t3 - t2):.2f} ms, numpy {1e3 * (t4 - t3):.2f} ms, ok {ok}, evicted_ms so far {evicted()}", flush=True)
print(f"loop {1e3 * (time.perf_counter() - tl0) / 5:.2f} ms per sweep ({'reused' if reuse else 'fresh'} record array)", flush=True)
