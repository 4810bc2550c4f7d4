"""Builds libephemeris_amd.so for gfx950 with hipcc (cross-compiles without a GPU). In-tree output."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libephemeris_amd.so"
N_PAIR_VARIANTS = 7      # csrc/device_math.h: 0 = the product, 1..6 = other orders of the unpinned point-mass term
SOURCES = ["kernels.hip", "craft.hip", "peer.hip", "mem.cpp", "coeffs.cpp", "nbody.cpp", "propagator.cpp", "shard.cpp", "api.cpp"]
HEADERS = ["eph_internal.h", "host.h", "device_math.h", "coeff_tables.inc", "cr_pow_tables.inc", "chain_tile.inc", "craft_attempt.inc", "../../include/ephemeris_amd.h"]
# -ffp-contract=off is REQUIRED for parity (HIP's default is fast contraction): the reference never fuses a*b+c.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
FLAGS += os.environ.get("EPH_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def lib_path(pair_variant=0):
    return LIB if pair_variant == 0 else HERE / f"libephemeris_amd_pv{pair_variant}.so"


def needs_build(pair_variant=0):
    lib = lib_path(pair_variant)
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS) or Path(__file__).stat().st_mtime > t


def build(force=False, verbose=False, pair_variant=0):
    """pair_variant 0 = the product library; 1..6 = the same sources with -DEPH_PAIR_VARIANT=k (another evaluation
    order of the point-mass term whose reference source is absent, csrc/device_math.h) -> libephemeris_amd_pv<k>.so."""
    LIB = lib_path(pair_variant)
    if not force and not needs_build(pair_variant):
        return LIB
    objs = []
    procs = []
    suffix = "" if pair_variant == 0 else f".pv{pair_variant}"
    flags = FLAGS + ([] if pair_variant == 0 else [f"-DEPH_PAIR_VARIANT={pair_variant}"])
    for src in SOURCES:
        obj = CSRC / (src.rsplit(".", 1)[0] + suffix + ".o")
        cmd = [hipcc(), *flags, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs]
    subprocess.check_call(cmd)
    return LIB


def build_all(force=False, verbose=False):
    """The product library and the six pair-variant builds, compiled concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        return list(ex.map(lambda k: build(force, verbose, k), range(N_PAIR_VARIANTS)))


if __name__ == "__main__":
    pv = int(sys.argv[sys.argv.index("--pair-variant") + 1]) if "--pair-variant" in sys.argv else 0
    if "--all" in sys.argv:
        print(build_all(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, pair_variant=pv))
