#!/bin/bash
# ON THE GPU BOX: raster parity tests, then the per-kernel profile.  Usage: tools/r04_quick.sh <tag> [pytest-args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-q}
export VCT_RASTER_PATH=${RPATH:-binned}
(timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_textures.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8)
tools/r04_raster_prof.sh $TAG
