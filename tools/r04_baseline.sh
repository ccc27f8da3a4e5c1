#!/bin/bash
# ON THE GPU BOX: state at the start of round 4 -- GPU tests, then the bench on the atrium (configs[1]) and on the
# Bistro-class street (configs[4], and at 1080p)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04_base
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r04_base/pytest_gpu.log
for tag in "atrium:--scene atrium" "bistro1080:--scene bistro" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 600 python bench.py $args --steps 10 --warmup 3 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" > gpurun_out/r04_base/$name.json
  python -c "import sys,json; d=json.load(open('gpurun_out/r04_base/$name.json')); print('$name', d['value'], d['gi_pass_ms'], d['gi_pass_one_call_ms'])"
done
