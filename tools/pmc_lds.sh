#!/bin/bash
# ON THE GPU BOX: LDS bank-conflict counters of the trace kernel (one PMC pass, no other trace domain).
# Usage: tools/pmc_lds.sh <tag> [bench args]  -> gpurun_out/pmc_lds_<tag>.txt
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_lds_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  --output-format csv -d "$OUT" -o pmc -- python3 $ROOT/bench.py --cpu-seconds 0 --no-sweep --steps 3 --warmup 1 $* > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY' | tee $ROOT/gpurun_out/pmc_lds_$(basename $1 2>/dev/null).txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_tile" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
tail -3 "$OUT/log.txt"
