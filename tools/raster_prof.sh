#!/bin/bash
# ON THE GPU BOX: per-kernel times of the raster stages.  Usage: tools/raster_prof.sh <tag> [direct]
TAG=${1:-a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
[ "$2" = direct ] && export VCT_RASTER_PATH=direct
[ -z "$VCT_RASTER_PATH" ] && export VCT_RASTER_PATH=binned
cd /tmp; export TMPDIR=/tmp
for cfg in "atrium 1920 1080" "bistro 1920 1080" "bistro 3840 2160"; do
  set -- $cfg
  OUT=$ROOT/gpurun_out/rprof_${TAG}_$1_$3
  mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $ROOT/tools/raster_prof.py $cfg 12 > $OUT/log.txt 2>&1
  grep -E "shadow .* ms" $OUT/log.txt
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)
if not f: print("no trace"); sys.exit()
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if not any(s in k for s in ("k_bin", "k_raster", "k_gbuffer")): continue
    v2 = sorted(v)
    print(f"  {k[:70]:70s} n={len(v):3d} median {v2[len(v2)//2]/1000:9.1f} us  min {v2[0]/1000:9.1f}  max {v2[-1]/1000:9.1f}")
PY
done
