#!/bin/bash
# the product against one experimental build over the sweep's other shapes (one box): scripts/ab_craft_matrix.sh LIB
set -u
cd $GRAFT_REPO_ROOT
L=$PWD/ephemeris_explorer_amd
for cfg in "65536 0.25 Verner87" "262144 0.25 DormandPrince54" "262144 0.25 Fine45" "262144 0.25 Tsitouras75" "262144 0.1 Verner98" "1048576 0.25 Verner87" "16384 1.0 Verner87"; do
  set -- $cfg
  for lib in libephemeris_amd.so "$LIB"; do
    EPH_AMD_LIBRARY=$L/$lib python scripts/ab_craft.py $1 $2 3 $3 2>&1 | tail -1
  done
done
