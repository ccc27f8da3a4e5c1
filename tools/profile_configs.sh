#!/bin/bash
# ON THE GPU BOX: kernel-trace/stats (+ HBM byte counters) of the non-default BASELINE configurations.
# Usage: tools/profile_configs.sh <tag>   -> gpurun_out/prof_<tag>_{c3,c5,bistro1080,noise,noise_rec,tex}/   (c5 = the Bistro-class street, BASELINE configs[4])
set -u
# --frames-in-flight 1: one trace launch in flight at a time, so that a launch's duration in the kernel trace is its own (the
# default bench line overlaps consecutive steps on two frame slots; the counters per launch are the same either way)
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
run() {   # name, pmc-passes ("yes"/"no"), bench args...
  local name=$1 pmc=$2; shift 2
  local OUT=$ROOT/gpurun_out/prof_${TAG}_$name
  mkdir -p "$OUT"
  local BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --no-sweep --frames-in-flight 1 --no-two-slots $*"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH --steps 10 --warmup 2 > "$OUT/trace.log" 2>&1
  if [ "$pmc" = yes ]; then
    i=0
    for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 900 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc$i.log" 2>&1
      echo "pmc$i [$PMC] rc=$?" >> "$OUT/passes.txt"
    done
  fi
  grep "^{" "$OUT/trace.log" | tail -1 > "$OUT/bench.json"
}
run c3 no --voxel-dim 512 --width 3840 --height 2160 --bounces 2
run c5 yes --scene bistro --voxel-dim 1024 --width 3840 --height 2160
run bistro1080 no --scene bistro
run tex no --scene atrium-textured
run noise yes --scene noise --noise-dense --gbuffer random --voxel-dim 1024
run noise_rec yes --scene noise --noise-dense --gbuffer random --voxel-dim 1024 --footprint-records
