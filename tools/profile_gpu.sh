#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace/stats pass plus separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
# PMC passes never combine with other trace domains (pool rule) and each carries few counters
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
# --frames-in-flight 1: one trace launch in flight at a time, so that a launch's duration in the kernel trace is its own (the
# default bench line overlaps consecutive steps on two frame slots; the counters per launch are the same either way)
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --no-sweep --frames-in-flight 1 --no-two-slots $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH --steps 20 --warmup 3 > "$OUT/trace.log" 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -o pmc -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i [$PMC] rc=$?" >> "$OUT/passes.txt"
done
find "$OUT" -name '*.csv' | head -50
