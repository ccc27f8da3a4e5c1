#!/bin/bash
# ON THE GPU BOX: live-pixel compaction (trace_variant 4) against the tile trace: parity, kernel times, wave statistics
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04_compact; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "trace or loose or compact" 2>&1 | grep -E "passed|failed|error|Error|assert|loose variant" | tail -8)
for round in 1 2; do for v in 0 4; do
  for tag in "atrium:--scene atrium" "bistro1080:--scene bistro" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
    name=${tag%%:*}; args=${tag#*:}
    timeout 300 python bench.py $args --variant $v --steps 20 --warmup 5 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v $name kernel_ms', d['trace_kernel_ms'], 'steps', d['cone_steps_per_frame'])"
  done
done; done
for v in 0 4; do for tag in "bistro1080:--scene bistro" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
  name=${tag%%:*}; args=${tag#*:}
  VCT_AMD_LIB=$PWD/build/ab/tstats.so timeout 600 python tools/trace_stats.py $args --variant $v > $OUT/stats_${name}_v$v.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/stats_${name}_v$v.json')); l=d['launches'][0]; print('stats variant $v $name', {k: l[k] for k in l if k in ('wave_steps','lane_steps','mean_live_lane_fraction','coop_zero','coop_hit','fallback','per_lane_fraction')})"
done; done
