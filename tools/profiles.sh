#!/bin/bash
# ON THE GPU BOX: one round's profile set into gpurun_out/ (copy what is to be judged into profiles/<tag>_*):
# the default workload with its PMC passes (profile_gpu.sh), the other configurations (profile_configs.sh), clean bench
# lines, the raster stages per kernel in both visibility forms, the voxelizer's counters.   Usage: tools/profiles.sh r06
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/profile_gpu.sh $TAG
tools/profile_configs.sh $TAG
L=gpurun_out/${TAG}_lines; mkdir -p $L
python bench.py --steps 20 --warmup 5 > $L/bench_default.json 2> $L/bench_default.err
python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 > $L/bench_c5.json 2> $L/bench_c5.err
tools/raster_prof.sh $TAG > $L/raster_binned.txt 2>&1
tools/raster_prof.sh ${TAG}d direct > $L/raster_direct.txt 2>&1
tools/vox_pmc.sh ${TAG}_atrium > $L/vox_atrium.txt 2>&1
tools/vox_pmc.sh ${TAG}_c5 --scene bistro --voxel-dim 1024 > $L/vox_c5.txt 2>&1
ls $L
