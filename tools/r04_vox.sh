#!/bin/bash
# ON THE GPU BOX: voxelizer parity tests + stage times on the atrium and the street
cd ${GRAFT_REPO_ROOT:-/root/repo}
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_textures.py tests/test_golden.py -m gpu -x -q -k "vox or inject or golden or texture" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8)
for tag in "atrium:--scene atrium" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 600 python bench.py $args --steps 5 --warmup 2 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['gi_pass_ms'], 'one_call', d['gi_pass_one_call_ms'])"
done
