"""k_lm_small gangs: us per step of the gang against the number of systems, every system the same (forward) -- and with
EPH_DEBUG_SMALL=4 workgroup 0's shader-clock ticks per step and the clock itself (a throttled chip shows here), with
EPH_DEBUG_PLACEMENT=1 where the dispatcher put the workgroups (stderr), with EPH_SMALL_LDS_PAD=<bytes> one workgroup per CU.
usage (GPU box): python scripts/time_gang2.py [K ...]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); import hooks     # eph_debug_wg_cycles: a tuning build exports it (tests/hooks.py)
from ephemeris_explorer_amd.systems import load_system
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
Ks = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 16, 64, 128, 192, 256, 512]
sign = -1 if os.environ.get("GANG_MIXED") else 1
for K in Ks:
    gs = [ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt * (1 if i % 2 == 0 else sign)) for i in range(K)]
    ea.advance_many(gs, 12); ea.advance_many(gs, 1000)
    for g in gs: g.sync()
    n = 100000
    best = 1e9
    for rep in range(3):
        t = time.time(); ea.advance_many(gs, n)
        for g in gs: g.sync()
        best = min(best, time.time() - t)
    line = f"gang of {K:4d} x 32 bodies: {best / n * 1e6:.3f} us per step of the gang, {K * s.n * n / best:.3e} body-steps/s"
    if int(os.environ.get("EPH_DEBUG_SMALL", "0")) & 4:
        c = hooks.load(ea.LIB_PATH).debug_wg_cycles()
        line += f" | wg 0: {c[0] / c[2]:.0f} sclk ticks/step, {c[1] / c[2] / 100.0:.3f} us/step, clock {c[0] / (c[1] / 100.0):.0f} MHz"
        if c[3] or c[4]:
            line += " | ticks/step: wait A %.0f | sum1+pair %.0f | wait B %.0f | row sums %.0f | sum2+handover %.0f" % tuple(x / c[2] for x in c[3:8])
    print(line, flush=True)
    del gs
