#!/bin/bash
# ON THE GPU BOX: vct_gi_pass with its G-buffer raster on a second stream (default) against the six stages in sequence on one
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for one in 0 1; do
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160" "--scene bistro" "--scene atrium-textured"; do
    VCT_GI_ONE_STREAM=$one python bench.py $args --steps 10 --warmup 3 --cpu-seconds 0 --no-sweep --no-hbm-stress 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('r$r one_stream=$one [$args] stage sum', d['gi_pass_total_ms'], 'one call', d['gi_pass_one_call_ms'])"
  done
done; done
