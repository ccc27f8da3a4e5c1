#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== parity (tree)" 
timeout 1500 python -m pytest tests/test_gpu_binned.py tests/test_gpu_textures.py tests/test_gpu_raster.py tests/test_gpu_ref.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== stats"
VCT_AMD_LIB=$PWD/build/ab/stats.so VCT_BIN_STATS_DUMP=1 VCT_RASTER_PATH=binned timeout 300 python tools/raster_prof.py bistro 3840 2160 2 2>&1 | grep -E "binstats|shadow" | tail -3
VCT_AMD_LIB=$PWD/build/ab/stats.so VCT_BIN_STATS_DUMP=1 VCT_RASTER_PATH=binned timeout 300 python tools/raster_prof.py bistro 1920 1080 2 2>&1 | grep -E "binstats|shadow" | tail -2
for r in 1 2; do
  for lib in noclass tree; do
    if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
    echo "== $lib round $r"
    tools/raster_prof.sh ab_${lib}_$r 2>&1 | grep -E "shadow|k_bin_raster<false>|k_gbuffer"
  done
done
