#!/bin/bash
# ON THE GPU BOX: SQ counters of the binned raster kernels, two PMC passes.  Usage: tools/raster_pmc.sh <tag> <scene> <w> <h>
export VCT_RASTER_PATH=${RPATH:-binned}
TAG=${1:-p}; SCENE=${2:-atrium}; W=${3:-1920}; H=${4:-1080}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/rpmc_${TAG}_${SCENE}_$H
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/a" -o pmc -- python3 $ROOT/tools/raster_prof.py $SCENE $W $H 4 > "$OUT/a.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/b" -o pmc -- python3 $ROOT/tools/raster_prof.py $SCENE $W $H 4 > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("k_bin", "k_raster", "k_gbuffer")):
            name = k.replace("(anonymous namespace)::", "").replace("void ", "")[:28]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:26s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
