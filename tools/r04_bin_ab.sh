#!/bin/bash
# ON THE GPU BOX: A/B of binned-raster builds (build/ab/<name>.so against the tree's library): per-kernel times, binned form
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in tree "$@"; do
  if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
  echo "== $lib (round $round)"
  tools/r04_raster_prof.sh ab_$lib 2>&1 | grep -E "gbuffer|k_bin_raster<false>"
done; done
