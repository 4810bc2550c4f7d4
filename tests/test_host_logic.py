"""CPU-only checks of host logic added around the hot path: bench.py's block timing, the synthetic spacecraft populations and the
per-wave divergence figure (no device, no oracle)."""
import importlib.util
import sys

import numpy as np

from conftest import ROOT, load_system


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_timed_blocks_times_the_block_repeatedly_and_agrees_on_the_count():
    b = _bench()
    calls, barriers = [], []
    times = b.timed_blocks(lambda: calls.append(1), lambda: barriers.append(1), lambda t: 0.005, min_region_s=0.05)
    assert len(times) == len(calls) == 11 and times == sorted(times)       # 0.05 / 0.005 + 1 blocks, agreed through `agree`
    assert len(barriers) == 2 * len(calls)                                  # a barrier on both sides of every block
    assert len(b.timed_blocks(lambda: None, lambda: None, lambda t: 10.0)) == 3          # never fewer than three
    assert len(b.timed_blocks(lambda: None, lambda: None, lambda t: 1e-9)) == 64         # capped
    assert len(b.timed_blocks(lambda: None, lambda: None, lambda t: 1.0, blocks=5)) == 5  # --blocks overrides


def test_wave_divergence():
    from ephemeris_explorer_amd.workloads import wave_divergence
    assert wave_divergence([7] * 640) == 1.0
    assert abs(wave_divergence([10] * 32 + [100] * 32) - 100 / 55) < 1e-12
    a = np.tile([400, 100, 20, 5], 64)                  # four families interleaved: every wave waits for its slowest lanes
    assert abs(wave_divergence(a) - 400 / 131.25) < 1e-12
    assert abs(wave_divergence(np.sort(a)) - 1.0) < 1e-12   # the same craft in blocks: waves are uniform


def test_craft_populations():
    from ephemeris_explorer_amd.systems import load_ship
    from ephemeris_explorer_amd.workloads import craft_population
    s = load_system("full_solar_system_2433282.5")
    ship = load_ship(ROOT / "tests/golden/systems/full_solar_system_2433282.5/ships/Mars Transfer Ship.json")
    pos, vel, fam = craft_population("transfer", 1000, s, ship)
    assert np.abs(pos - ship.pos).max() < 600.0 and (fam == 3).all()
    p2, v2, f2 = craft_population("transfer", 1000, s, ship)
    assert np.array_equal(pos, p2) and np.array_equal(vel, v2)            # seeded
    pos, vel, fam = craft_population("mixed", 4096, s, ship)
    assert (fam == np.arange(4096) % 4).all()
    earth, sun = s.names.index("Earth"), s.names.index("Sun")
    r = np.linalg.norm(pos - s.pos[earth], axis=1)
    assert np.allclose(r[fam < 3], 6678.0)                                  # perigee states of the three Earth families
    speed = np.linalg.norm(vel - s.vel[earth], axis=1)
    for f, apo in ((0, 6678.0), (1, 42164.0), (2, 384400.0)):             # vis-viva at perigee
        a = 0.5 * (6678.0 + apo)
        assert np.allclose(speed[fam == f], np.sqrt(s.mu[earth] * (2 / 6678.0 - 1 / a)))
    d_sun = np.linalg.norm(pos[fam == 3] - s.pos[sun], axis=1)
    assert np.allclose(d_sun, np.linalg.norm(s.pos[earth] - s.pos[sun]), rtol=1e-12) and r[fam == 3].min() > 1e8
    _, _, fb = craft_population("mixed", 4096, s, ship, order="blocked")
    assert (np.diff(fb) >= 0).all() and np.bincount(fb).tolist() == [1024] * 4


def test_counter_files_are_tied_to_their_kernel_sources():
    """bench.py prints HBM traffic and VALU counts out of committed rocprofv3 passes; each counter file carries the sha256 of the
    kernel sources it was counted on (workloads.profile_stamp, written by scripts/summarize_profile.py) and is used only while
    they match (workloads.profile_is_current)."""
    import json
    from conftest import ROOT
    from ephemeris_explorer_amd.workloads import PROFILE_SOURCES, profile_is_current, profile_stamp, source_hashes
    for kind in PROFILE_SOURCES:
        stamp = profile_stamp(kind)
        assert set(stamp["source_sha256_16"]) == set(PROFILE_SOURCES[kind]) and stamp["source_sha256_16"] == source_hashes(kind)
        assert profile_is_current(stamp, kind) == (True, None)
        stale = {"source_sha256_16": dict(stamp["source_sha256_16"])}
        first = PROFILE_SOURCES[kind][0]
        stale["source_sha256_16"][first] = "0" * 16
        ok, why = profile_is_current(stale, kind)
        assert not ok and first in why
    assert profile_is_current({}, "nbody") == (False, "the counter file carries no source hashes")
    # the committed counter files of the step kernel and of the sweep kernel carry stamps (whether they are CURRENT is for the
    # bench line to say: it drops the figures and names the file that changed)
    for name, kind in (("traffic.json", "nbody"), ("traffic_craft.json", "craft")):
        info = json.loads((ROOT / "profiles" / name).read_text())
        assert set(info["source_sha256_16"]) == set(PROFILE_SOURCES[kind]) and info["profile_commit"]
