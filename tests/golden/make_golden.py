#!/usr/bin/env python3
"""Generates tests/golden/nbody_golden.json from the C oracle (oracle/eph_oracle.c), after cross-checking it bit for
bit against the independent Python restatement (oracle/pyoracle.py) on the first steps. Inputs are the committed
systems fixtures. f64 values are stored as hex strings (exact)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from ephemeris_explorer_amd.systems import load_system  # noqa: E402
from oracle import orc, pyoracle as po  # noqa: E402


def hexes(a):
    return [float(x).hex() for x in np.asarray(a, dtype=np.float64).ravel()]


def main():
    out = {"comment": "oracle states (pos, vel as f64 hex, AoS) after the listed step counts; QuinlanTremaine12",
           "systems": {}}
    plan = {
        "sun_earth_moon_2433282.5": [0, 1, 12, 13, 100, 1000, 10000, 100000],
        "simple_solar_system_2433282.5": [0, 1, 12, 13, 100, 1000],
        "full_solar_system_2433282.5": [0, 1, 12, 13, 100, 1000, 10000],
    }
    for name, marks in plan.items():
        s = load_system(ROOT / "tests/golden/systems" / name)
        entry = {"dt": s.dt, "epoch": s.epoch, "forward": {}, "backward": {}, "splines": {}}
        for sign, key in ((1, "forward"), (-1, "backward")):
            nb = orc.NBody(s.pos, s.vel, s.mu, s.epoch, sign * s.dt)
            pr = po.Problem(s.pos, s.vel, s.mu, s.epoch)
            lm = po.LinearMultistep2("QuinlanTremaine12", sign * s.dt, pr)
            done = 0
            for m in (marks if sign == 1 else marks[:6]):
                while done < m:
                    nb.advance(1)
                    if done < 40:
                        lm.advance()
                    done += 1
                    if done <= 40:
                        p, v, t, _ = nb.state()
                        assert np.array_equal(p, np.array(pr.y)) and np.array_equal(v, np.array(pr.dy)) and t == pr.time
                p, v, t, sc = nb.state()
                entry[key][str(m)] = {"t": float(t).hex(), "step_count": sc, "pos": hexes(p), "vel": hexes(v)}
        # splines: first polynomials of every body, forward and backward
        for d, key in ((1, "forward"), (-1, "backward")):
            pr = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, d, s.count, s.degree)
            nsteps = int(8 * s.count.max() * 2)
            for _ in range(nsteps):
                assert pr.step() == 0
            sol = pr.take_solution()
            bodies = []
            for b in range(s.n):
                st, iv, n = sol.info(b)
                co, nc = sol.coeffs(b)
                keep = min(n, 3)
                bodies.append({"start": float(st).hex(), "interval": float(iv).hex(), "npoly": n,
                               "ncoef": [int(x) for x in nc[:keep]], "coeffs": hexes(co[:keep])})
            entry["splines"][key] = {"steps": nsteps, "time": float(pr.time()).hex(), "bodies": bodies}
        out["systems"][name] = entry
    (ROOT / "tests/golden/nbody_golden.json").write_text(json.dumps(out) + "\n")
    print("written", (ROOT / "tests/golden/nbody_golden.json").stat().st_size, "bytes")
    craft_golden()


def craft_golden():
    """tests/golden/craft_golden.json: the reference's spacecraft scenario (ephemeris/tests/spacecraft_propagation.rs:
    357-449 = ships/Mars Transfer Ship.json) on the committed 10-body 1950 system, from the C oracle after a
    cross-check of its first 300 steps against the Python restatement: knot count, step / attempt counters, every
    500th knot and the last one (f64 hex), SOI transitions and the first apsides, for three of the UI's methods."""
    from ephemeris_explorer_amd.systems import load_ship, parse_epoch, soi_radii
    s = load_system(ROOT / "tests/golden/systems/simple_solar_system_2433282.5")
    ship = load_ship(ROOT / "tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
    burns = [(b.start, b.start + b.duration, b.acceleration, s.names.index(b.reference) if b.reference else -1)
             for b in ship.burns]
    end = parse_epoch("1951-01-01 00:00:00")
    pr = orc.Propagator(s.pos, s.vel, s.mu, s.epoch, s.dt, 1, s.count, s.degree)
    assert pr.step_to(parse_epoch("1952-01-01 00:00:00")) == 0
    eph = pr.take_solution()
    soi = soi_radii(s)
    pe = []
    for b in range(s.n):
        st, iv, n = eph.info(b)
        co, nc = eph.coeffs(b)
        pe.append({"start": st, "interval": iv, "polys": [[po.Vec(*co[q, k]) for k in range(nc[q])] for q in range(n)]})
    out = {"comment": "oracle knots of the Mars Transfer Ship on simple_solar_system_2433282.5 (QT12 ephemeris, dt 6 h) "
                      "to 1951-01-01; tolerance and burns from the ship file; f64 as hex",
           "soi_radius": hexes(soi), "methods": {}}
    for method in ("Verner87", "Fine45", "DormandPrince54"):
        c = orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, method, tol_pos=ship.tolerance, tol_vel=ship.tolerance,
                      burns=burns, soi_radius=soi)
        # cross-check against the Python restatement (libm pow on both sides for this part)
        orc.set_pow_mode(1)
        try:
            c2 = orc.Craft(eph, s.mu, ship.start, ship.pos, ship.vel, method, tol_pos=ship.tolerance,
                           tol_vel=ship.tolerance, burns=burns)
            p = po.Craft(pe, s.mu, ship.start, ship.pos, ship.vel, method, ship.tolerance, burns)
            for _ in range(300):
                assert c2.step() == 0 and p.step() == 0
            kt, kp, kv = c2.knots()
            assert all(kt[i] == p.knots[i][0] and tuple(kp[i]) == p.knots[i][1][:3] for i in range(len(kt)))
        finally:
            orc.set_pow_mode(0)
        assert c.step_to(end) == 0
        kt, kp, kv = c.knots()
        st = c.state()
        idx = sorted(set(range(0, len(kt), 500)) | {len(kt) - 1})
        tt, tb = c.transitions()
        at, ad, ab, ak = c.apsides()
        out["methods"][method] = {
            "knots": len(kt), "steps": st["steps"], "attempts": st["attempts"], "next_h": float(st["next_h"]).hex(),
            "sample": [{"i": i, "t": float(kt[i]).hex(), "pos": hexes(kp[i]), "vel": hexes(kv[i])} for i in idx],
            "transitions": [[float(t).hex(), int(b)] for t, b in zip(tt, tb)],
            "apsides": len(at),
            "first_apsides": [[float(t).hex(), float(d).hex(), int(b), int(k)] for t, d, b, k in
                              list(zip(at, ad, ab, ak))[:8]],
        }
    (ROOT / "tests/golden/craft_golden.json").write_text(json.dumps(out) + "\n")
    print("written", (ROOT / "tests/golden/craft_golden.json").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
