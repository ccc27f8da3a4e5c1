#!/bin/bash
# ON THE GPU BOX: counters of the instrumented binned raster (build/ab/stats.so, -DVCT_BIN_STATS=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "atrium 1920 1080" "bistro 1920 1080" "bistro 3840 2160"; do
  VCT_RASTER_PATH=binned VCT_AMD_LIB=$PWD/build/ab/stats.so VCT_BIN_STATS_DUMP=1 python3 tools/raster_prof.py $cfg 2 2>&1 | grep -E "binstats|shadow" | tail -2
done
