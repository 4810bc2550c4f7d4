import sys, time, os
sys.path.insert(0, os.getcwd())
import ephemeris_explorer_amd as ea
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); import hooks     # eph_debug_wg_cycles: a tuning build exports it (tests/hooks.py)
from ephemeris_explorer_amd.systems import load_system
for name in ("full_solar_system_2433282.5", "sun_earth_moon_2433282.5"):
    s = load_system("tests/golden/systems/" + name)
    g = ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt)
    g.advance(12); g.advance(200000); g.sync()
    for rep in range(3):
        t = time.time(); g.advance(500000); g.sync(); w = time.time() - t
        c = hooks.load(ea.LIB_PATH).debug_wg_cycles()
        print(name, "us/step %.3f" % (w / 5e5 * 1e6), "sclk ticks/step %.0f" % (c[0] / c[2]), "realtime us/step %.3f" % (c[1] / c[2] / 100.0), "=> clock %.0f MHz" % (c[0] / (c[1] / 100.0)))
    if c[3] or c[4]:
        n = c[2]
        print("   ticks/step: wait A %.0f | sum1+pair %.0f | wait B %.0f | row sums %.0f | sum2+handover %.0f" % tuple(x / n for x in c[3:8]))
