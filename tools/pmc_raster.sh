#!/bin/bash
# ON THE GPU BOX: SQ counters of the raster / voxelize kernels (tools/raster_bench.py), one PMC pass.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_raster
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/a" -o pmc -- python3 $ROOT/tools/raster_bench.py > "$OUT/a.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_FLAT \
  --output-format csv -d "$OUT/b" -o pmc -- python3 $ROOT/tools/raster_bench.py > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY' | tee $ROOT/gpurun_out/pmc_raster.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("k_raster", "k_gbuffer", "k_voxelize_list", "k_vis32", "k_resolve")):
            acc[k.split("(")[1][:40] if k.startswith("(") else k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:26s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
