#!/bin/bash
# ON THE GPU BOX: the whole GPU suite, then the bench on the three raster-relevant workloads (automatic raster form)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04_full; mkdir -p $OUT
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15) > $OUT/pytest.log
cat $OUT/pytest.log
for tag in "atrium:--scene atrium" "bistro1080:--scene bistro" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 600 python bench.py $args --steps 10 --warmup 3 --cpu-seconds 0 --no-sweep 2>$OUT/$name.err | grep "^{" > $OUT/$name.json
  python -c "import sys,json; d=json.load(open('$OUT/$name.json')); g=d['gi_pass_ms']; print('$name', d['value'], g, 'one_call', d['gi_pass_one_call_ms'])" 2>&1 | tail -1
done
