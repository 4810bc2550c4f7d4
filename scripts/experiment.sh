#!/bin/bash
# One parameterised driver for timing experiments on the GPU box (replaces round 2's 27 one-shot scripts/r02_<ordinal>.sh;
# those are in the history: git show d7efbfa:scripts/r02_first.sh ...).
#
#   scripts/experiment.sh TAG [-l LIBNAME]... [-e VAR=VALUE]... WHAT [ARGS...]
#
#   TAG        output goes to gpurun_out/TAG/WHAT[_LIBNAME].log
#   -l NAME    run once per experimental build ephemeris_explorer_amd/libephemeris_amd_exp_NAME.so (scripts/build_exp.sh NAME
#              -DFLAG...; "product" = the product library). Default: product only.
#   -e K=V     environment for every run (EPH_WG_BODIES, EPH_FORCE, EPH_PAIR_VARIANT, EPH_CRAFT_SORT, EPH_CRAFT_QUEUE, ...)
#   WHAT       sizes [N...]      steady QT12 step per size            (scripts/time_sizes.py)
#              path N STEPS P    one size, path P (0 exact, 4 fast..) (scripts/time_path.py)
#              ab N              minimal ctypes timing, any ABI       (scripts/ab_step.py <library>)
#              small             k_lm_small: single, gangs, configs[1] (scripts/time_small.py)
#              clock-small       shader clock + per-phase ticks       (scripts/clock_small.py; -l smallacct, -e EPH_DEBUG_SMALL=4)
#              craft ARGS...     bench.py --workload craft ARGS       (--population mixed --craft 524288 --craft-days 2 --steps 2)
#              clocks CMD...     CMD under rocm-smi clock / power sampling (scripts/sample_clocks.sh)
#              pytest ARGS...    python -m pytest ARGS
# Example (round 3's step-kernel A/B):  scripts/build_exp.sh chainasm -DEPH_CHAIN_ASM=1 &&
#                                       gpurun -- scripts/experiment.sh r03x -l product -l chainasm sizes 1024 2048 4096
set -u
TAG=$1; shift
LIBS=(); ENVS=()
while [ $# -gt 0 ]; do
  case "$1" in
    -l) LIBS+=("$2"); shift 2;;
    -e) ENVS+=("$2"); shift 2;;
    *) break;;
  esac
done
WHAT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
[ ${#LIBS[@]} -eq 0 ] && LIBS=(product)
for lib in "${LIBS[@]}"; do
  path=$ROOT/ephemeris_explorer_amd/libephemeris_amd.so
  [ "$lib" != product ] && path=$ROOT/ephemeris_explorer_amd/libephemeris_amd_exp_$lib.so
  log=$OUT/${WHAT}_$lib.log
  run() { env EPH_AMD_LIBRARY="$path" "${ENVS[@]}" "$@" >> "$log" 2>&1; }
  echo "== $WHAT [$lib] ${ENVS[*]:-} $*" | tee -a "$log"
  case "$WHAT" in
    sizes) run python scripts/time_sizes.py "$@";;
    path) run python scripts/time_path.py "$@";;
    ab) for n in "${@:-4096}"; do run python scripts/ab_step.py "$path"; done;;
    small) run python scripts/time_small.py;;
    clock-small) run python scripts/clock_small.py;;
    craft) run python bench.py --workload craft --no-cpu-baseline "$@";;
    clocks) run scripts/sample_clocks.sh "$OUT/clocks_$lib.csv" "$@";;
    pytest) run python -m pytest "$@";;
    *) echo "unknown experiment $WHAT"; exit 2;;
  esac
  tail -n 12 "$log"
done
