import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
SYSTEMS = GOLDEN / "systems"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def product_lib():
    """Builds (if needed) and loads libephemeris_amd.so. hipcc cross-compiles gfx950 without a GPU."""
    from ephemeris_explorer_amd import build as b
    b.build()
    import ephemeris_explorer_amd as ea
    ea._lib()
    return ea


@pytest.fixture(scope="session")
def gpu(product_lib):
    """The product on a live device; fails (never skips silently to a fallback) if no device is visible."""
    ea = product_lib
    if ea.device_count() < 1:
        pytest.fail("-m gpu tests need a HIP device; the product has no CPU path")
    return ea


@pytest.fixture(scope="session")
def hooks(gpu):
    """the eph_debug_* test hooks: a library of their own (tests/hooks.py), never the product's"""
    import hooks as h
    return h.load()


def load_system(name):
    from ephemeris_explorer_amd.systems import load_system as ls
    return ls(SYSTEMS / name)
