#!/bin/bash
# ON THE GPU BOX: PMC counter passes of bench.py for each build/ab/<name>.so given (A/B of kernel variants).
# Usage: tools/pmc_ab.sh <tag> name1 name2 ...  -> gpurun_out/pmc_<tag>/<name>.txt (per-kernel means of the trace kernel)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VMEM_WR"
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum"
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES")
for name in "$@"; do
  export VCT_AMD_LIB=$ROOT/build/ab/$name.so
  : > "$OUT/$name.txt"
  i=0
  for PMC in "${GROUPS_[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/$name.p$i" -o pmc -- python3 $ROOT/bench.py --cpu-seconds 0 --no-sweep --steps 3 --warmup 1 > "$OUT/$name.p$i.log" 2>&1
    f=$(find "$OUT/$name.p$i" -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then
      python3 - "$f" >> "$OUT/$name.txt" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_trace_tile" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in acc.items():
    print(f"{c:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
    else
      echo "group $i failed: $(tail -2 $OUT/$name.p$i.log)" >> "$OUT/$name.txt"
    fi
    rm -rf "$OUT/$name.p$i"
  done
done
