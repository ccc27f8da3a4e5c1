#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do for tag in "atrium:--scene atrium" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 600 python bench.py $args --steps 10 --warmup 3 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['gi_pass_ms'], 'one_call', d['gi_pass_one_call_ms'])"
done; done
