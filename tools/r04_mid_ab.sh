#!/bin/bash
# ON THE GPU BOX: A/B of builds of the direct form's list kernel (build/ab/<name>.so against the tree's library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in tree "$@"; do
  if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
  echo "== $lib (round $round)"
  tools/r04_raster_prof.sh ab_$lib direct 2>&1 | grep -E "shadow|k_raster_mid"
done; done
