#!/bin/bash
# Build ablation variants of the trace kernel (timing only, results wrong by construction):
#   tools/ablate.sh "1 2 4 8 16 32 3 ..."  -> build/ab/abl<mask>.so ; run tools/gpu_ab.sh on the GPU box
set -e
cd "$(dirname "$0")/.."
args=()
for m in ${1:-0 1 2 4 8 16 32}; do args+=("abl$m" "-fno-slp-vectorize -DVCT_ABLATE=$m"); done
tools/build_ab.sh "${args[@]}"
