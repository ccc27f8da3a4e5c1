#!/bin/bash
# ON THE GPU BOX: per-lane gather from fp32-decoded levels >= 1 (VCT_F32_LEVELS=1, experiment) against the default, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "parity with the decoded levels: $(VCT_F32_LEVELS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k 'trace or golden' 2>&1 | grep -E 'passed|failed' | tail -1)"
for round in 1 2; do for v in 0 1; do
  for args in "--scene atrium" "--scene bistro" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
    if [ $v = 1 ]; then export VCT_F32_LEVELS=1; else unset VCT_F32_LEVELS; fi
    timeout 300 python bench.py $args --steps 20 --warmup 5 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32_levels $v', '$args', 'kernel_ms', d['trace_kernel_ms'])"
  done
done; done
