#!/bin/bash
# Build A/B variants of libvct_amd.so into build/ab/<name>.so:  tools/build_ab.sh name "extra hipcc flags" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
C=voxel-cone-tracing_amd/csrc
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function $flags \
     -shared -o build/ab/$name.so $C/vct_capi.hip $C/vct_trace.hip $C/vct_volume.hip $C/vct_voxelize.hip $C/vct_raster.hip $C/vct_multi.hip -ldl &
done
wait
ls -la build/ab
