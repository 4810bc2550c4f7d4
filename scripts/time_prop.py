import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
p = ea.NBodyPropagator.from_system(s)
p.step_n(12)
t = time.time(); p.step_n(1000000); wall = time.time() - t
sol = p.take_solution()
print(f"full_solar_system (32 bodies, dt 10 min): 1e6 steps incl. solout sampling + {sum(sol.info(b)[2] for b in range(s.n))} polynomial fits: {wall:.3f} s = {wall:.3f} us/step -> {32e6/wall:.3e} body-steps/s")
from oracle import orc
o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree, native=True)
for _ in range(12): o.step()
t = time.time()
for _ in range(100000): o.step()
cpu = time.time() - t
print(f"CPU oracle (1 thread): {cpu/1e5*1e6:.3f} us/step -> GPU/CPU {cpu*10/wall:.1f}x")
