#!/bin/bash
# scripts/ab_craft.sh LIB...: alternate the product library and the named experimental builds on ONE box (two rounds), sweep kernel time
set -u
cd $GRAFT_REPO_ROOT
L=$PWD/ephemeris_explorer_amd
N=${N:-262144}; DAYS=${DAYS:-0.25}; METHOD=${METHOD:-Verner87}
for rep in 1 2; do
  for lib in libephemeris_amd.so "$@"; do
    EPH_AMD_LIBRARY=$L/$lib python scripts/ab_craft.py $N $DAYS 5 $METHOD
  done
done
