#!/bin/bash
# ON THE GPU BOX: the clean bench lines of the round (run AFTER profiles/trace_traffic*.json were regenerated for the
# sources being benchmarked: the lines replay the PMC counters only on a sha match)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04_lines
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_lines/bench_default.json 2> gpurun_out/r04_lines/bench_default.err
python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 > gpurun_out/r04_lines/bench_c5.json 2> gpurun_out/r04_lines/bench_c5.err
VCT_RASTER_PATH=direct python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 --no-sweep > gpurun_out/r04_lines/bench_c5_direct.json 2>/dev/null
VCT_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-sweep --slabs interleaved > gpurun_out/r04_lines/bench_1rank_interleaved.json 2>/dev/null
ls -la gpurun_out/r04_lines
