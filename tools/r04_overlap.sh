#!/bin/bash
# ON THE GPU BOX: kernel timeline of the native 1-rank step loop on an 8-way-slab-sized frame: does the gather's kernel
# run beside the next trace, or after it?  With and without reserved compute units.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
export VCT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29651
for k in 0 8; do
  OUT=$ROOT/gpurun_out/r04_overlap_$k; mkdir -p $OUT
  export VCT_COMM_RESERVED_CUS=$k
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --width 1920 --height 136 --steps 60 --warmup 10 --cpu-seconds 0 --no-sweep --slabs equal > $OUT/log.txt 2>&1
  python3 - "$OUT" $k <<'PY'
import csv, glob, sys
out, k = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-260:]          # the timed loop's tail
t0 = int(rows[0]["Start_Timestamp"])
tr = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in rows if "k_trace_tile_split" in r["Kernel_Name"]]
oth = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"][:50]) for r in rows if "k_trace_tile_split" not in r["Kernel_Name"]]
names = sorted({o[2] for o in oth})
print(f"reserved {k}: {len(tr)} trace kernels, other kernels: {names}")
gaps = [tr[i + 1][0] - tr[i][1] for i in range(len(tr) - 1)]
durs = [b - a for a, b in tr]
print(f"  trace duration median {sorted(durs)[len(durs)//2]/1000:.1f} us, gap between consecutive traces median {sorted(gaps)[len(gaps)//2]/1000:.1f} us")
inside = 0
for a, b, n in oth:
    if any(s <= a and b <= e for s, e in tr): inside += 1
print(f"  other kernels entirely inside a trace kernel's interval: {inside} of {len(oth)}; their median duration {sorted(b - a for a, b, _ in oth)[len(oth)//2]/1000:.1f} us")
PY
done
