#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for lib in tree nt; do
  if [ $lib = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB=$PWD/build/ab/$lib.so; fi
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
    python bench.py $args --steps 10 --warmup 3 --cpu-seconds 0 --no-sweep --no-hbm-stress 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['gi_pass_ms']
print('r$r $lib [$args] gbuffer', g['gbuffer_raster'], 'trace', g['trace'], 'total', d['gi_pass_total_ms'], 'one_call', d['gi_pass_one_call_ms'], 'ms_per_step', d['ms_per_step'])"
  done
done; done
