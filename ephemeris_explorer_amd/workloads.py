"""Synthetic workloads named by BASELINE.json / SURVEY.md §8(d): Plummer spheres in N-body units."""
import hashlib

import numpy as np


def plummer(n, seed=20260926, min_sep=1e-3):
    """Plummer sphere, G = 1, total mass 1, scale radius 1, equal masses (mu = 1/n), positions and velocities by
    the Aarseth-Henon-Wielen rejection method with numpy's PCG64 `default_rng(seed)`, recentred to zero
    barycentre and momentum. The reference uses softening 0, so a sample closer than `min_sep` to an already
    accepted one is rejected (keeps the fixed step sane). Returns (pos [n,3], vel [n,3], mu [n])."""
    rng = np.random.default_rng(seed)
    pos = np.zeros((n, 3))
    vel = np.zeros((n, 3))
    cell = {}

    def too_close(p):
        k = tuple(np.floor(p / min_sep).astype(np.int64))
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for q in cell.get((k[0] + dx, k[1] + dy, k[2] + dz), ()):
                        if np.linalg.norm(pos[q] - p) < min_sep:
                            return True
        return False

    i = 0
    while i < n:
        m = rng.uniform(0.0, 1.0)
        if m <= 0.0 or m >= 0.999:        # drop the far tail (r < ~38.7)
            continue
        r = 1.0 / np.sqrt(m ** (-2.0 / 3.0) - 1.0)
        z = rng.uniform(-1.0, 1.0)
        phi = rng.uniform(0.0, 2.0 * np.pi)
        s = np.sqrt(1.0 - z * z)
        p = r * np.array([s * np.cos(phi), s * np.sin(phi), z])
        if too_close(p):
            continue
        while True:                       # q in [0,1] with pdf ~ q^2 (1-q^2)^(7/2)
            q = rng.uniform(0.0, 1.0)
            g = rng.uniform(0.0, 0.1)
            if g < q * q * (1.0 - q * q) ** 3.5:
                break
        v = q * np.sqrt(2.0) * (1.0 + r * r) ** -0.25
        z = rng.uniform(-1.0, 1.0)
        phi = rng.uniform(0.0, 2.0 * np.pi)
        s = np.sqrt(1.0 - z * z)
        pos[i] = p
        vel[i] = v * np.array([s * np.cos(phi), s * np.sin(phi), z])
        cell.setdefault(tuple(np.floor(p / min_sep).astype(np.int64)), []).append(i)
        i += 1
    mu = np.full(n, 1.0 / n)
    pos -= pos.mean(axis=0)
    vel -= vel.mean(axis=0)
    return pos, vel, mu


def sha256_of(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def craft_population(kind, n, system, ship, seed=20260926, order="interleaved"):
    """Initial states of n spacecraft for the massless sweep, at `ship.start` (the epoch of `system`, barycentric km, km/s).
    kind "transfer": SURVEY 8(d)4 -- the Mars Transfer Ship's state perturbed by normal(0, 100 km / 0.01 km/s) per
    component (every craft on the same heliocentric arc: equal work per craft).
    kind "mixed": four orbit families around / away from the Earth in equal numbers, whose adaptive step sizes differ by
    more than an order of magnitude -- low Earth orbit (6678 km circular), geostationary transfer (6678 x 42 164 km), lunar
    transfer (6678 x 384 400 km), and a heliocentric cruise (the Earth's own orbit 60 +- 5 degrees ahead of it: days
    per step); random orbital planes, perigee states.
    order "interleaved": craft i belongs to family i % 4 (every wave of 64 holds all four: the worst case for a static
    craft -> lane assignment); "blocked": families in four contiguous blocks. Returns (pos [n,3], vel [n,3], family [n])."""
    rng = np.random.default_rng(seed)
    tpos = ship.pos + rng.normal(0.0, 100.0, size=(n, 3))
    tvel = ship.vel + rng.normal(0.0, 0.01, size=(n, 3))
    if kind == "transfer":
        return tpos, tvel, np.full(n, 3, dtype=np.int32)
    if kind != "mixed":
        raise ValueError(kind)
    earth = system.names.index("Earth")
    mu = system.mu[earth]
    rp = 6678.0
    apo = np.array([6678.0, 42164.0, 384400.0])
    fam = (np.arange(n) % 4) if order == "interleaved" else (np.arange(n) * 4 // n)
    fam = fam.astype(np.int32)
    pos, vel = tpos.copy(), tvel.copy()
    for f in range(3):
        idx = np.nonzero(fam == f)[0]
        m = len(idx)
        a = 0.5 * (rp + apo[f])
        vp = np.sqrt(mu * (2.0 / rp - 1.0 / a))
        # random orthonormal pair (perigee direction, velocity direction)
        u = rng.normal(size=(m, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        w = rng.normal(size=(m, 3))
        w -= (w * u).sum(axis=1, keepdims=True) * u
        w /= np.linalg.norm(w, axis=1, keepdims=True)
        pos[idx] = system.pos[earth] + rp * u
        vel[idx] = system.vel[earth] + vp * w
    idx = np.nonzero(fam == 3)[0]
    sun = system.names.index("Sun")
    r, v = system.pos[earth] - system.pos[sun], system.vel[earth] - system.vel[sun]
    axis = np.cross(r, v)
    axis /= np.linalg.norm(axis)
    ang = np.radians(rng.uniform(55.0, 65.0, len(idx)))[:, None]

    def rot(x):                                            # Rodrigues rotation about the orbit normal
        return x * np.cos(ang) + np.cross(axis, x) * np.sin(ang) + axis * (x @ axis) * (1.0 - np.cos(ang))

    pos[idx] = system.pos[sun] + rot(r)
    vel[idx] = system.vel[sun] + rot(v)
    return pos, vel, fam


def wave_divergence(attempts, wave=64):
    """How unevenly the attempts of a sweep fall on the lanes of a wave when craft i runs on lane i % 64 of wave i // 64
    for the whole sweep: per wave max / mean of the per-craft attempt counts, averaged over the waves weighted by their
    cost (a wave runs as long as its slowest lane). 1.0 = no lane ever idles."""
    a = np.asarray(attempts, dtype=np.float64)
    pad = (-len(a)) % wave
    a = np.concatenate([a, np.zeros(pad)]).reshape(-1, wave)
    mx, mean = a.max(axis=1), a.sum(axis=1) / wave
    return float(mx.sum() / max(mean.sum(), 1e-300))


# ---- which kernel sources a committed counter file (profiles/traffic*.json) was taken with -------------------------------------------
# bench.py prints HBM traffic and VALU counts out of committed rocprofv3 passes (they cannot be collected live). A counter file carries
# the sha256 of the sources of the kernel it describes; bench.py recomputes them and DROPS the figures when a source has changed since.
PROFILE_SOURCES = {
    "nbody": ["step_wg.hip", "step_wave.hip", "pair_term.h", "ieee_seq.h", "force_common.h", "chain_tile.inc"],
    "craft": ["craft_sweep.hip", "craft_device.h", "craft_attempt.inc", "pair_term.h", "ieee_seq.h"],
    "fast": ["fast.hip", "pair_term.h", "ieee_seq.h", "force_common.h"],
}


def source_hashes(kind):
    import hashlib
    from pathlib import Path
    csrc = Path(__file__).resolve().parent / "csrc"
    return {f: hashlib.sha256((csrc / f).read_bytes()).hexdigest()[:16] for f in PROFILE_SOURCES[kind]}


def profile_stamp(kind):
    """what scripts/summarize_profile.py writes into a counter file: the sources' hashes and the commit they were read at"""
    import subprocess
    from pathlib import Path
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=Path(__file__).resolve().parent, capture_output=True,
                              text=True).stdout.strip() or None
    except OSError:
        head = None
    return {"profile_commit": head, "source_sha256_16": source_hashes(kind)}


def profile_is_current(info, kind):
    """(True, None) when the counter file `info` was taken with today's kernel sources, else (False, which source changed)"""
    want = info.get("source_sha256_16")
    if not want:
        return False, "the counter file carries no source hashes"
    have = source_hashes(kind)
    changed = [f for f in have if want.get(f) != have[f]]
    return (not changed), (", ".join(changed) + " changed since the profile" if changed else None)
