#!/bin/bash
# ON THE GPU BOX: the exact trace (variant 0) against the loose one (variant 3), whole bench runs interleaved, two rounds:
# steady-state kernel time of the timed region (what `value` is made of), atrium and the street at 1024^3 / 4K
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for v in 0 3; do
  for args in "--scene atrium" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
    timeout 300 python bench.py $args --variant $v --steps 20 --warmup 5 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', '$args'.split()[1], 'kernel_ms', d['trace_kernel_ms'], 'value', d['value'])"
  done
done; done
