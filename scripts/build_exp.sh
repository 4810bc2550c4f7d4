#!/bin/bash
# scripts/build_exp.sh NAME [extra hipcc flags]: an experimental build of the library beside the product one,
# ephemeris_explorer_amd/libephemeris_amd_exp_NAME.so (selected with EPH_AMD_LIBRARY=<path>); objects under /tmp/eph_exp_NAME.
# Tuning switches need -DEPH_EXPERIMENTS=1 (e.g. -DEPH_EXPERIMENTS=1 -DEPH_SMALL_ACCOUNT=1, -DEPH_EXPERIMENTS=1 -DEPH_WG_SIDE=1).
set -e
name=$1; shift
cd "$(dirname "$0")/.."
python -m ephemeris_explorer_amd.build --exp "$name" "$@"
