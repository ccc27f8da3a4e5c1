#!/bin/bash
# ON THE GPU BOX: what is run on the final sources of a round -- the whole GPU test suite, the round's profile set
# (tools/profiles.sh), clean bench lines of every configuration, the randomized parity sweep.  Results land in gpurun_out/;
# summarise locally with tools/summarize_prof.py and copy what is to be judged into profiles/.
cd ${GRAFT_REPO_ROOT:-/root/repo}
(timeout 3000 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "passed|failed|error|config5 chain" | tail -6) > gpurun_out/r06_pytest_gpu.txt
tools/profiles.sh r06 > gpurun_out/r06_profiles.log 2>&1
( time python bench.py ) > gpurun_out/r06f_bench.json 2> gpurun_out/r06f_bench.err
python bench.py --steps 10 --warmup 3 --scene bistro --voxel-dim 1024 --width 3840 --height 2160 --cpu-seconds 0 > gpurun_out/r06f_c5_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --voxel-dim 512 --width 3840 --height 2160 --bounces 2 --cpu-seconds 0 --no-sweep > gpurun_out/r06f_c3_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --scene bistro --cpu-seconds 0 --no-sweep > gpurun_out/r06f_bistro1080_bench.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --scene atrium-textured --cpu-seconds 0 --no-sweep > gpurun_out/r06f_tex_bench.json 2>/dev/null
if [ -z "$SKIP_FUZZ" ]; then
  ( FUZZ_SEED0=400000 python tools/fuzz_gpu.py 600 ; FUZZ_SEED0=400000 python tools/fuzz_gpu.py 900 big ) > gpurun_out/r06_fuzz.txt 2>&1
  tail -3 gpurun_out/r06_fuzz.txt
fi
tail -4 gpurun_out/r06f_bench.err
