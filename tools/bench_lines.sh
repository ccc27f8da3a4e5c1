#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
( time python bench.py ) > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 > gpurun_out/r06f_c5_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --voxel-dim 512 --width 3840 --height 2160 --bounces 2 --cpu-seconds 0 --no-sweep > gpurun_out/r06f_c3_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --scene bistro --cpu-seconds 0 --no-sweep > gpurun_out/r06f_bistro1080_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --scene atrium-textured --cpu-seconds 0 --no-sweep > gpurun_out/r06f_tex_bench.json 2>/dev/null
tail -4 gpurun_out/r06f_bench.err
