"""Headless driver for the reference's on-disk formats: load a system directory -> +-N years of
ephemeris (forward and backward propagators, as `compute_ephemerides_bodies` does,
ephemeris_explorer/src/load/mod.rs:673-687) -> every ship under ships/ -> summary JSON; optional state.json export
at an epoch (ephemeris_explorer/src/ui/windows/export.rs:222-256).

    python -m ephemeris_explorer_amd.cli tests/golden/systems/full_solar_system_2433282.5 --years 2
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

from . import BACKWARD, FORWARD, AdaptiveParams, Ephemeris, NBodyPropagator, SpacecraftBatch
from .systems import format_epoch, load_ship, load_system, soi_radii

SEC_PER_YEAR = 365.0 * 86400.0      # Duration::from_days(365.0 * 2.0) for two years (load/mod.rs:674)


def export_state(system, solution, at, path):
    """state.json of the massive bodies at `at` (export.rs:222-256): name, mu, position, velocity per body."""
    bodies = []
    for b, name in enumerate(system.names):
        pos, vel, inside = solution.eval(b, [at])
        if not inside[0]:
            raise ValueError(f"{name}: epoch outside the computed ephemeris")
        bodies.append({"name": name, "mu": float(system.mu[b]), "position": [float(x) for x in pos[0]],
                       "velocity": [float(x) for x in vel[0]]})
    Path(path).write_text(json.dumps({"name": system.name, "epoch": format_epoch(at), "bodies": bodies}, indent=4))


class Run:
    """What the headless flow produced: the system, the forward / backward Vec<UniformSpline> and per ship the batch (knot
    slabs, events) -- `main` prints a summary of it, tests compare it with the CPU restatement."""

    def __init__(self, system):
        self.system, self.forward, self.backward = system, None, None
        self.fwd_prop, self.bwd_prop = None, None
        self.ships = []          # (ShipFile, burns or None, SpacecraftBatch or None, reason skipped or None)
        self.ephemeris_seconds = 0.0


def run_live(system_dir, years=2.0, chunk_days=30.0, margin_days=40.0):
    """The app's OWN flow instead of "bodies first, ships afterwards": the forward N-body task sends a snapshot every `chunk_days`
    (prediction.rs:422-443: step, take_solution, send), every snapshot is merged into the bodies' LIVE table
    (PredictionTarget::merge, dynamics/celestial.rs:198-204), and one task per ship (load/mod.rs:673-687: one async task per
    propagator) restarts its stored propagator whenever the context has become valid far enough (flight_plan.rs:363-395:
    `propagator.context().is_valid_at(..)`) -- all concurrently, one host thread and one HIP stream each, the table shared. A ship
    only asks for a leg the table already covers by `margin_days` (more than any step of these plans), so no evaluation runs off the
    table's end while it is still growing and the knots equal
    those of run() (ships against the finished table) bit for bit, whatever the interleaving (tests/test_gpu_cli.py)."""
    import threading
    system = load_system(system_dir)
    r = Run(system)
    t0 = time.time()
    day = 86400.0
    r.fwd_prop = NBodyPropagator.from_system(system, FORWARD)
    end = system.epoch + years * SEC_PER_YEAR
    r.fwd_prop.step_to(system.epoch + chunk_days * day)
    first = r.fwd_prop.take_solution()
    eph = Ephemeris(first, system.mu)
    r.forward = first                                   # (grown below: what the N-body task has sent so far, joined)
    errors, done = [], threading.Event()

    def bodies_task():
        try:
            t = system.epoch + chunk_days * day
            while t < end:
                t = min(t + chunk_days * day, end)
                r.fwd_prop.step_to(t)
                piece = r.fwd_prop.take_solution()
                eph.merge(piece)
                r.forward.append(piece)
        except Exception as e:          # noqa: BLE001 -- reported by the caller
            errors.append(("bodies", repr(e)))
        finally:
            done.set()

    system_dir = Path(system_dir)
    ships = sorted((system_dir / "ships").glob("*.json")) if (system_dir / "ships").is_dir() else []
    soi = soi_radii(system)
    slots = [None] * len(ships)

    def ship_task(k, path):
        try:
            ship = load_ship(path)
            try:
                burns = [(b.start, b.start + b.duration, b.acceleration,
                          system.names.index(b.reference) if b.reference else -1) for b in ship.burns]
            except ValueError as e:
                slots[k] = (ship, None, None, f"burn reference not in this system: {e}")
                return
            while not eph.is_valid_at(ship.start + margin_days * day) and not done.is_set():
                time.sleep(0.001)
            batch = SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], ship.integrator,
                                    AdaptiveParams.default(ship.tolerance), [burns], max_knots=1 << 18)
            batch.enable_events(soi, max_transitions=256, max_apsides=1 << 16)
            leg = ship.start
            while leg < ship.end and not errors:
                leg = min(leg + chunk_days * day, ship.end)
                while not eph.is_valid_at(leg + margin_days * day):
                    if done.is_set():               # the bodies have reached their end: what the table covers is all there will be
                        break
                    time.sleep(0.001)
                batch.propagate(leg)
            slots[k] = (ship, burns, batch, None)
        except Exception as e:          # noqa: BLE001
            errors.append((str(path), repr(e)))

    threads = [threading.Thread(target=bodies_task)] + [threading.Thread(target=ship_task, args=(k, p)) for k, p in enumerate(ships)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        raise RuntimeError(f"live run failed: {errors}")
    r.ships = [s for s in slots if s is not None]
    r.ephemeris_seconds = time.time() - t0
    r.live_revision = eph.revision
    return r


def run(system_dir, years=2.0, backward=True):
    """load a system directory -> +-`years` of ephemeris (forward and backward propagators concurrently, as
    compute_ephemerides_bodies does, load/mod.rs:673-687) -> every ship under ships/ with the app's SpacecraftSolout."""
    import threading
    system = load_system(system_dir)
    r = Run(system)
    t0 = time.time()
    # forward and backward propagators run concurrently, one host thread and one HIP stream each -- the reference
    # runs them as two async tasks (prediction.rs:422-443, load/mod.rs:673-687)
    r.fwd_prop = NBodyPropagator.from_system(system, FORWARD)
    r.bwd_prop = NBodyPropagator.from_system(system, BACKWARD) if backward else None
    sols = {}

    def go(key, prop, until):
        sols[key] = prop.propagate(until)

    threads = [threading.Thread(target=go, args=("f", r.fwd_prop, system.epoch + years * SEC_PER_YEAR))]
    if r.bwd_prop is not None:
        threads.append(threading.Thread(target=go, args=("b", r.bwd_prop, system.epoch - years * SEC_PER_YEAR)))
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if "f" not in sols or (r.bwd_prop is not None and "b" not in sols):
        raise RuntimeError("ephemeris propagation failed")
    r.forward, r.backward = sols["f"], sols.get("b")
    r.ephemeris_seconds = time.time() - t0

    system_dir = Path(system_dir)
    ships = sorted((system_dir / "ships").glob("*.json")) if (system_dir / "ships").is_dir() else []
    eph = Ephemeris(r.forward, system.mu) if ships else None
    soi = soi_radii(system)
    for path in ships:
        ship = load_ship(path)
        try:
            burns = [(b.start, b.start + b.duration, b.acceleration,
                      system.names.index(b.reference) if b.reference else -1) for b in ship.burns]
        except ValueError as e:
            r.ships.append((ship, None, None, f"burn reference not in this system: {e}"))
            continue
        batch = SpacecraftBatch(eph, ship.start, [ship.pos], [ship.vel], ship.integrator,
                                AdaptiveParams.default(ship.tolerance), [burns], max_knots=1 << 18)
        batch.enable_events(soi, max_transitions=256, max_apsides=1 << 16)      # the app's SpacecraftSolout
        batch.propagate(ship.end)
        r.ships.append((ship, burns, batch, None))
    return r


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("system", type=Path)
    ap.add_argument("--years", type=float, default=2.0)
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--live", type=float, metavar="CHUNK_DAYS", default=None,
                    help="the app's flow: the bodies' task sends a snapshot every CHUNK_DAYS, each is merged into the LIVE device table, "
                         "the ships' tasks chase it concurrently (forward only)")
    ap.add_argument("--export-state", nargs=2, metavar=("EPOCH", "PATH"),
                    help='write a state.json at "YYYY-MM-DD HH:MM:SS" from the forward ephemeris')
    args = ap.parse_args(argv)

    r = run_live(args.system, args.years, args.live) if args.live else run(args.system, args.years, not args.no_backward)
    system, sol_f, fwd, bwd = r.system, r.forward, r.fwd_prop, r.bwd_prop
    out = {"system": system.name, "bodies": system.n, "dt_s": system.dt, "epoch": format_epoch(system.epoch)}
    out["forward"] = {"reached": format_epoch(fwd.time()), "steps": fwd.state()[3],
                      "polynomials": int(sum(sol_f.info(b)[2] for b in range(system.n)))}
    if bwd is not None:
        sol_b = r.backward
        out["backward"] = {"reached": format_epoch(bwd.time()), "steps": bwd.state()[3],
                           "polynomials": int(sum(sol_b.info(b)[2] for b in range(system.n)))}
    out["ephemeris_seconds"] = r.ephemeris_seconds
    if args.live:
        out["live"] = {"chunk_days": args.live, "snapshots_merged": int(r.live_revision), "seconds_bodies_and_ships_together": r.ephemeris_seconds}
    out["ships"] = []
    for ship, burns, batch, skipped in r.ships:
        entry = {"name": ship.name, "integrator": ship.integrator}
        if skipped:
            entry["skipped"] = skipped
            out["ships"].append(entry)
            continue
        st = batch.status()
        fin = batch.state()
        entry.update({"status": int(st["status"][0]), "knots": int(st["nknots"][0]), "steps": int(st["steps"][0]),
                      "final_epoch": format_epoch(fin["t"][0]), "final_position_km": [float(x) for x in fin["pos"][0]]})
        (tt, tb), (at, ad, ab, ak) = batch.events(0)
        entry["soi_transitions"] = [{"epoch": format_epoch(t), "body": system.names[b]} for t, b in zip(tt, tb)]
        entry["apsides"] = {"count": int(len(at)),
                            "first": [{"epoch": format_epoch(t), "body": system.names[b], "distance_km": float(d),
                                       "kind": "periapsis" if k == 0 else "apoapsis"}
                                      for t, d, b, k in list(zip(at, ad, ab, ak))[:4]]}
        out["ships"].append(entry)
    if args.export_state:
        from .systems import parse_epoch
        export_state(system, sol_f, parse_epoch(args.export_state[0]), args.export_state[1])
        out["exported"] = args.export_state[1]
    json.dump(out, sys.stdout, indent=1)
    print()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
