#!/bin/bash
# ON THE GPU BOX: the randomized parity sweep on the final kernels of the round.  Usage: tools/r04_fuzz.sh <small seconds> <big seconds> [seed0]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04_fuzz
[ -n "$3" ] && export FUZZ_SEED0=$3
timeout $(( $1 + 120 )) python tools/fuzz_gpu.py $1 2>&1 | tail -3 | tee gpurun_out/r04_fuzz/small.txt
timeout $(( $2 + 300 )) python tools/fuzz_gpu.py $2 big 2>&1 | tail -3 | tee gpurun_out/r04_fuzz/big.txt
