"""The app's calling pattern: step() + has_reached() per step (prediction.rs:422-443), 32-body system: host time per call and
the end-to-end rate, against the oracle (one CPU thread) driven the same way."""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea  # noqa: E402
from ephemeris_explorer_amd.systems import load_system  # noqa: E402
from oracle import orc  # noqa: E402
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
L = ea._lib()
p = ea.NBodyPropagator.from_system(s)
p.step_n(12)
end = s.epoch + 1e12
h = p._h
import ctypes as C  # noqa: E402
flag = C.c_int()
N = 200000
t = time.perf_counter()
for _ in range(N):
    L.eph_prop_step(h)
    L.eph_prop_has_reached(h, end, C.byref(flag))
sol = p.take_solution()
dt = time.perf_counter() - t
print(f"GPU propagator, {N} x (step + has_reached) + take_solution: {dt / N * 1e6:.3f} us per step (ctypes call overhead included)")
o = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree, native=True)
for _ in range(12):
    o.step()
t = time.perf_counter()
for _ in range(N // 4):
    o.step()
    o.has_reached(end)
dt = time.perf_counter() - t
print(f"CPU oracle driven the same way: {dt / (N // 4) * 1e6:.3f} us per step")
