import sys, time, os
sys.path.insert(0, os.getcwd())
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system
s = load_system("tests/golden/systems/full_solar_system_2433282.5")
for K in (1, 2, 64, 128, 256):
    gs = [ea.NBodyIntegration(s.pos, s.vel, s.mu, s.epoch, s.dt) for i in range(K)]
    ea.advance_many(gs, 12); ea.advance_many(gs, 100000)
    for g in gs: g.sync()
    t = time.time(); ea.advance_many(gs, 300000)
    for g in gs: g.sync()
    w = time.time() - t
    c = ea.debug_wg_cycles()
    print(K, "us/step %.3f" % (w / 3e5 * 1e6), "sclk ticks/step %.0f" % (c[0] / c[2]), "=> clock %.0f MHz" % (c[0] / (c[1] / 100.0)))
