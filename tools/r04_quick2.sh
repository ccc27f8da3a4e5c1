#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
(timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_textures.py tests/test_gpu_binned.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8)
VCT_RASTER_PATH=direct tools/r04_raster_prof.sh d1 direct 2>&1 | grep -E "shadow|k_raster_vis|k_gbuffer"
VCT_RASTER_PATH=binned tools/r04_raster_prof.sh b11 2>&1 | grep -E "shadow|k_bin_setup"
