"""Ad hoc: one ship for a year with and without the SpacecraftSolout event search (kernel ms)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.systems import load_system, load_ship, soi_radii, parse_epoch
s = load_system(ROOT / "tests/golden/systems/full_solar_system_2433282.5")
ship = load_ship(ROOT / "tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
sol = ea.NBodyPropagator.from_system(s).propagate(s.epoch + 400 * 86400.0)
eph = ea.Ephemeris(sol, s.mu)
burns = [(b.start, b.start + b.duration, b.acceleration, s.names.index(b.reference) if b.reference else -1) for b in ship.burns]
end = parse_epoch("1951-01-01 00:00:00")
for ev in (False, True):
    b = ea.SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], "Verner87", ea.AdaptiveParams.default(ship.tolerance),
                           [burns], max_knots=20000)
    if ev:
        b.enable_events(soi_radii(s), 64, 8192)
    b.propagate(end)
    st = b.status()
    print(f"events={ev}: {b.kernel_ms():.1f} ms for {st['steps'][0]} steps, status {st['status'][0]}", flush=True)
