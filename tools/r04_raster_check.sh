#!/bin/bash
# ON THE GPU BOX: raster parity tests on the binned path, then A/B of the raster stages (binned vs direct)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r04_raster; mkdir -p $OUT
(timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_textures.py tests/test_gpu_configs.py tests/test_golden.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15) > $OUT/pytest.log
cat $OUT/pytest.log
for path in binned direct; do
for tag in "atrium:--scene atrium" "bistro1080:--scene bistro" "c5:--scene bistro --voxel-dim 1024 --width 3840 --height 2160"; do
  name=${tag%%:*}; args=${tag#*:}
  VCT_RASTER_PATH=$path timeout 600 python bench.py $args --steps 5 --warmup 2 --cpu-seconds 0 --no-sweep 2>$OUT/$name.$path.err | grep "^{" > $OUT/$name.$path.json
  python -c "import sys,json; d=json.load(open('$OUT/$name.$path.json')); g=d['gi_pass_ms']; print('$path $name shadow', g['shadow_map_raster'], 'gbuffer', g['gbuffer_raster'], 'one_call', d['gi_pass_one_call_ms'])" 2>&1 | tail -1
done; done
