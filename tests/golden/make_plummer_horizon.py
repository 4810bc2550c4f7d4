"""Generates tests/golden/plummer4096_horizon.npz: the CPU oracle's positions of the 4096-body Plummer sphere
(BASELINE.json configs[2]; QuinlanTremaine12, h = 1/1024) after 10, 100, 1e3, 1e4 and 1e5 integrator advances
(start-up included: the count is IntegratorState::step_count), plus SHA-256 digests of positions and velocities.

The north star's sentence is "positions within 1e-9 AU of the reference over 1e5 steps" at this size. The oracle's
target-partitioned OpenMP form (orc.set_gravity_threads) performs the same f64 additions in the same order as the
triangular single-thread loop (tests/test_oracle.py::test_target_partitioned_openmp_gravity_has_the_same_bits), so this file holds what the
reference's algorithm produces; it takes ~10 minutes on 8 cores, which is why it is a committed fixture and not a
live run inside the GPU test. Usage: python tests/golden/make_plummer_horizon.py [threads]
"""
import hashlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from ephemeris_explorer_amd.workloads import plummer   # noqa: E402
from oracle import orc   # noqa: E402

CHECKPOINTS = (10, 100, 1_000, 10_000, 100_000)


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    last = int(sys.argv[2]) if len(sys.argv) > 2 else CHECKPOINTS[-1]
    pos, vel, mu = plummer(4096)
    orc.set_gravity_threads(threads, native=True)
    o = orc.NBody(pos, vel, mu, 0.0, 1.0 / 1024.0, native=True)
    out = {"checkpoints": np.array([c for c in CHECKPOINTS if c <= last], dtype=np.int64)}
    done, t0 = 0, time.time()
    for c in out["checkpoints"]:
        while done < c:
            k = min(int(c) - done, 500)
            assert o.advance(k) == 0
            done += k
            print(f"{done} steps, {time.time() - t0:.0f} s", flush=True)
        p, v, t, sc = o.state()
        assert sc == c
        out[f"pos_{c}"] = p
        out[f"time_{c}"] = np.float64(t)
        out[f"sha_pos_{c}"] = np.array(hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest())
        out[f"sha_vel_{c}"] = np.array(hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest())
        np.savez(ROOT / "tests/golden/plummer4096_horizon.npz", **out)   # keep partial progress


if __name__ == "__main__":
    main()
