"""Timeline of the last vct_gi_pass calls from a rocprofv3 kernel trace of bench.py:  tools/gi_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the gi_pass timing loop: 25 calls; take the last few resolves before the pre-roll of plain traces
idx = [i for i, n in enumerate(names) if "k_resolve_sparse" in n]
i0 = idx[-3] - 9
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0:i0 + 28]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q={r['Queue_Id']} {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:56]}")
