"""Builds libephemeris_amd.so for gfx950 with hipcc (cross-compiles without a GPU). In-tree output."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libephemeris_amd.so"
SOURCES = ["kernels.hip", "craft.hip", "coeffs.cpp", "nbody.cpp", "propagator.cpp", "shard.cpp", "api.cpp"]
HEADERS = ["eph_internal.h", "host.h", "device_math.h", "coeff_tables.inc", "cr_pow_tables.inc", "../../include/ephemeris_amd.h"]
# -ffp-contract=off is REQUIRED for parity (HIP's default is fast contraction): the reference never fuses a*b+c.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
FLAGS += os.environ.get("EPH_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS) or Path(__file__).stat().st_mtime > t


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = CSRC / (src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc(), *FLAGS, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
