#!/bin/bash
# scripts/build_exp.sh NAME [extra hipcc flags]: an experimental build of the library beside the product one,
# ephemeris_explorer_amd/libephemeris_amd_exp_NAME.so (selected with EPH_AMD_LIBRARY=<path>); objects under /tmp.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/ephemeris_explorer_amd/csrc
out=/tmp/eph_exp_$name
mkdir -p $out
for f in kernels.hip craft.hip peer.hip mem.cpp coeffs.cpp nbody.cpp propagator.cpp shard.cpp api.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -x hip -c $src/$f -o $out/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ephemeris_explorer_amd/libephemeris_amd_exp_$name.so $out/*.o
echo $root/ephemeris_explorer_amd/libephemeris_amd_exp_$name.so
