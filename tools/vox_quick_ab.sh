#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
use() { if [ "$1" = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB="$PWD/build/ab/$1.so"; fi; }
use tree
echo "== tree parity: $(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_textures.py -m gpu -x -q -k 'not trace' 2>&1 | grep -E 'passed|failed' | tail -1)"
for round in 1 2 3; do for lib in ${1:-nowin tree}; do use $lib
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024" "--scene bistro --voxel-dim 512" "--scene bistro"; do
    echo "r$round $lib [$args] $(python tools/vox_bench.py $args 2>/dev/null | grep 'texture_mipmaps=1 shadow=True' | sed 's/  inject.*items/ items/')"
  done
done; done
