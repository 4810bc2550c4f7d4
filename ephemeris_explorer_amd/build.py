"""Builds libephemeris_amd.so for gfx950 with hipcc (cross-compiles without a GPU). In-tree output."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libephemeris_amd.so"
# the product's objects + debug_api.o: the eph_debug_* test hooks (csrc/eph_debug.h), for tests/ only -- the product never loads it
HOOKS_LIB = HERE / "libephemeris_amd_testhooks.so"
HOOKS_SOURCES = ["debug_api.cpp"]
N_PAIR_VARIANTS = 7      # csrc/pair_term.h: the evaluation orders of the unpinned point-mass term, all in the one library
# compiled once per evaluation order (-DEPH_PAIR_VARIANT=k, every symbol in namespace eph::pv<k>; csrc/pair_ns.h)
PAIR_SOURCES = ["step_wg.hip", "step_wave.hip", "step_small.hip", "fast.hip", "craft_sweep.hip"]
# compiled once
SOURCES = ["solout.hip", "craft.hip", "peer.hip", "dispatch.cpp", "mem.cpp", "coeffs.cpp", "nbody.cpp", "propagator.cpp", "shard.cpp", "api.cpp"]
EXPORTS = CSRC / "exports.map"      # linker version script: only eph_* is a dynamic symbol
HEADERS = ["exports.map", "eph_internal.h", "eph_debug.h", "host.h", "ieee_seq.h", "pair_term.h", "pair_ns.h", "pair_launchers.h", "force_common.h", "craft_device.h",
           "coeff_tables.inc", "cr_pow_tables.inc", "craft_attempt.inc", "../../include/ephemeris_amd.h"]
# -ffp-contract=off is REQUIRED for parity (HIP's default is fast contraction): the reference never fuses a*b+c.
# -fvisibility=hidden: the library exports the extern "C" boundary of include/ephemeris_amd.h and nothing else
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
FLAGS += os.environ.get("EPH_EXTRA_FLAGS", "").split()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def units():
    """(source, object, extra flags) of every translation unit of the library"""
    out = [(src, CSRC / (src.rsplit(".", 1)[0] + ".o"), []) for src in SOURCES + HOOKS_SOURCES]
    for k in range(N_PAIR_VARIANTS):
        out += [(src, CSRC / (src.rsplit(".", 1)[0] + f".pv{k}.o"), [f"-DEPH_PAIR_VARIANT={k}"]) for src in PAIR_SOURCES]
    return out


def needs_build(lib=LIB):
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HOOKS_SOURCES + PAIR_SOURCES + HEADERS) or Path(__file__).stat().st_mtime > t


def build(force=False, verbose=False, lib=LIB, extra_flags=(), obj_dir=None, jobs=None):
    """Compiles every translation unit (objects whose source and headers are older than the object are kept) and links
    libephemeris_amd.so. `lib` / `extra_flags` / `obj_dir`: an experimental build beside the product one (scripts/build_exp.sh)."""
    hooks = lib.with_name(lib.name.replace("libephemeris_amd", "libephemeris_amd_testhooks")) if lib == LIB else None
    if not force and not extra_flags and not needs_build(lib) and (hooks is None or not needs_build(hooks)):
        return lib
    from concurrent.futures import ThreadPoolExecutor
    newest_header = max((CSRC / h).stat().st_mtime for h in HEADERS)
    newest_header = max(newest_header, Path(__file__).stat().st_mtime)
    todo, objs = [], []
    for src, obj, fl in units():
        if obj_dir is not None:
            obj = Path(obj_dir) / obj.name
        objs.append(str(obj))
        if force or extra_flags or not obj.exists() or obj.stat().st_mtime < max((CSRC / src).stat().st_mtime, newest_header):
            todo.append((src, obj, fl))

    def compile_one(job):
        src, obj, fl = job
        cmd = [hipcc(), *FLAGS, *fl, *extra_flags, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {src} {' '.join(fl)}:\n{p.stdout.decode()}")
        if verbose and p.stdout:
            print(p.stdout.decode(), file=sys.stderr)

    # the largest units first, so that the tail of the build is short
    todo.sort(key=lambda j: -(CSRC / j[0]).stat().st_size)
    with ThreadPoolExecutor(jobs or max(2, (os.cpu_count() or 4))) as ex:
        list(ex.map(compile_one, todo))
    hook_objs = [o for o in objs if Path(o).name in {h.rsplit(".", 1)[0] + ".o" for h in HOOKS_SOURCES}]
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS}"]
    if hooks is not None:                              # the product: the boundary only; the hooks in a library of their own
        subprocess.check_call([*link, "-o", str(lib), *[o for o in objs if o not in hook_objs]])
        subprocess.check_call([*link, "-o", str(hooks), *objs])
    else:                                              # a tuning build carries its hooks itself
        subprocess.check_call([*link, "-o", str(lib), *objs])
    return lib


def build_all(force=False, verbose=False):
    """(round 3 built one library per evaluation order; they are all in the one library now)"""
    return [build(force, verbose), HOOKS_LIB]


if __name__ == "__main__":
    if "--exp" in sys.argv:          # python -m ephemeris_explorer_amd.build --exp NAME [-DFLAG ...]
        i = sys.argv.index("--exp")
        name, flags = sys.argv[i + 1], sys.argv[i + 2:]
        od = Path("/tmp") / f"eph_exp_{name}"
        od.mkdir(parents=True, exist_ok=True)
        print(build(force=True, verbose=False, lib=HERE / f"libephemeris_amd_exp_{name}.so", extra_flags=flags, obj_dir=od))
    else:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
