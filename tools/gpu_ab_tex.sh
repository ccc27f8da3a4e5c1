#!/bin/bash
# ON THE GPU BOX: interleaved A/B of build/ab/*.so on the textured scenes (textured atrium at the default configuration,
# the Bistro-class street at 1024^3 / 4K and at 256^3 / 1080p): every stage that samples material textures.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do for lib in build/ab/*.so; do
for args in "--scene atrium-textured" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160" "--scene bistro"; do
VCT_AMD_LIB=$PWD/$lib python bench.py $args --steps 5 --warmup 2 --cpu-seconds 0 --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['gi_pass_ms']; print('$lib','$args'.replace('--scene ','').replace(' --voxel-dim 1024 --width 3840 --height 2160','@4K'),'voxelize',g['voxelize'],'gbuffer',g['gbuffer_raster'],'one_call',d['gi_pass_one_call_ms'])"
done; done; done
