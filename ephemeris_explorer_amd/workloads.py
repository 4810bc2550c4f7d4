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
