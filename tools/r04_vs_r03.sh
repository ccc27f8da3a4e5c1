#!/bin/bash
# ON THE GPU BOX: the round-3 tree (build/r03tree) against this tree, interleaved, same box: trace kernel and stage times
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for tree in build/r03tree .; do
  (cd $tree && timeout 600 python bench.py --scene atrium --steps 30 --warmup 5 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tree', 'trace', d['trace_kernel_ms'], 'value', d['value'], d['gi_pass_ms'])")
done; done
