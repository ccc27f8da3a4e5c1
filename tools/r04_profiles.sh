#!/bin/bash
# ON THE GPU BOX: the profiles of round 4 (final kernels): default workload with PMC passes, configs[4] with PMC passes
# (bench.py replays both), the other configurations kernel-trace only, the bench lines, instrumented counters.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
tools/profile_gpu.sh r04f
tools/profile_configs.sh r04f
mkdir -p gpurun_out/r04_lines
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_lines/bench_default.json 2> gpurun_out/r04_lines/bench_default.err
python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 > gpurun_out/r04_lines/bench_c5.json 2> gpurun_out/r04_lines/bench_c5.err
VCT_RASTER_PATH=direct python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 --no-sweep > gpurun_out/r04_lines/bench_c5_direct.json 2>/dev/null
VCT_AMD_LIB=$PWD/build/ab/tstats.so python tools/trace_stats.py > gpurun_out/r04_lines/trace_stats.json 2>/dev/null
VCT_AMD_LIB=$PWD/build/ab/tstats.so python tools/trace_stats.py --scene bistro --voxel-dim 1024 --width 3840 --height 2160 > gpurun_out/r04_lines/bistro_trace_stats.json 2>/dev/null
tools/r04_binstats.sh > gpurun_out/r04_lines/binstats.txt 2>&1
tools/r04_raster_prof.sh r04f > gpurun_out/r04_lines/raster_binned.txt 2>&1
tools/r04_raster_prof.sh r04fd direct > gpurun_out/r04_lines/raster_direct.txt 2>&1
ls gpurun_out/r04_lines
