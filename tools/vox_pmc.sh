#!/bin/bash
# ON THE GPU BOX: SQ counters of k_voxelize_bricks (tools/vox_bench.py: texture_mipmaps 1 / 0 x shadow on / off, 8 launches each).
# Usage: tools/vox_pmc.sh <tag> [bench.py scene flags]
TAG=${1:-v}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/voxpmc_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
python3 $ROOT/tools/vox_bench.py "$@" 2>/dev/null | tee $OUT/times.txt
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/a" -o pmc -- python3 $ROOT/tools/vox_bench.py "$@" > "$OUT/a.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM \
  --output-format csv -d "$OUT/b" -o pmc -- python3 $ROOT/tools/vox_bench.py "$@" > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
# dispatches of the kernel in launch order, groups of 8 = one (mipmaps, shadow) combination of vox_bench.py
for sub in ("a", "b"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_voxelize_bricks" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    names = ["mips shadow", "mips noshadow", "nomips shadow", "nomips noshadow"]
    for g in range(0, len(ids), 8):
        grp = ids[g:g + 8]
        acc = collections.defaultdict(list)
        for i in grp:
            for c, v in rows[i].items():
                acc[c].append(v)
        print(names[(g // 8) % 4] if g // 8 < 4 else f"group {g // 8}", " ".join(f"{c}={sum(v)/len(v):.5g}" for c, v in sorted(acc.items())))
PY
