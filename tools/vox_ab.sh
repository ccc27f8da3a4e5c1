#!/bin/bash
# ON THE GPU BOX: A/B of the voxelizer / raster input stages.  tools/vox_ab.sh "lib1 lib2 ..."  (tree = in-tree library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
LIBS=${1:-"base tree"}
use() { if [ "$1" = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB="$PWD/build/ab/$1.so"; fi; }
for lib in $LIBS; do use $lib
  echo "== $lib parity: $(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_textures.py tests/test_gpu_raster.py tests/test_gpu_binned.py -m gpu -x -q -k 'not trace' 2>&1 | grep -E 'passed|failed' | tail -1)"
done
for round in 1 2; do for lib in $LIBS; do use $lib
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024" "--scene bistro" "--scene atrium-textured"; do
    echo "r$round $lib [$args] $(python tools/vox_bench.py $args 2>/dev/null | grep 'texture_mipmaps=1 shadow=True' | sed 's/  inject.*//')"
  done
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
    python bench.py $args --steps 5 --warmup 2 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['gi_pass_ms']
print('r$round $lib [$args] gi:', ' '.join(f'{k}={v}' for k, v in g.items()), 'total', d['gi_pass_total_ms'], 'one_call', d['gi_pass_one_call_ms'])"
  done
done; done
