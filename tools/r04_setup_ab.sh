#!/bin/bash
# ON THE GPU BOX: A/B of builds of the per-triangle set-up kernels (build/ab/<name>.so against the tree's library), direct and binned form
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in tree "$@"; do
  if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
  echo "== $lib (round $round)"
  tools/r04_raster_prof.sh ab_$lib direct 2>&1 | grep -E "shadow|k_raster_vis"
  tools/r04_raster_prof.sh ab_$lib 2>&1 | grep -E "k_bin_setup"
done; done
