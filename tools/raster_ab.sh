#!/bin/bash
# ON THE GPU BOX: parity of the raster tests on the in-tree library, then interleaved per-kernel times of the raster stages
# for the libraries named (tree = in-tree, other words = build/ab/<word>.so).  tools/raster_ab.sh "old tree" [rounds] [kernel regex]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=${1:-"tree"}; ROUNDS=${2:-2}; PAT=${3:-"k_bin_raster<false>|k_gbuffer"}
mkdir -p gpurun_out
echo "== parity (tree): $(timeout 1500 python -m pytest tests/test_gpu_binned.py tests/test_gpu_textures.py tests/test_gpu_raster.py tests/test_gpu_ref.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' | tail -1)"
for r in $(seq 1 $ROUNDS); do
  for lib in $LIBS; do
    if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
    echo "== $lib round $r"
    tools/raster_prof.sh ab_${lib}_$r 2>&1 | grep -E "shadow|$PAT"
  done
done
