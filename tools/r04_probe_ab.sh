#!/bin/bash
# ON THE GPU BOX: timing-only probe builds (build/ab/<name>.so: results are WRONG by construction) against the tree's library:
# per-kernel times of the binned G-buffer pass.  Usage: tools/r04_probe_ab.sh <name> ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in tree "$@"; do
  if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
  echo "== $lib"
  tools/r04_raster_prof.sh probe_$lib 2>&1 | grep -E "gbuffer|k_bin_raster<false>|k_gbuffer_shade"
done
