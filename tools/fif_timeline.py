"""Do consecutive trace launches of the two frame slots really overlap?   tools/fif_timeline.py <kernel_trace.csv> [steps]
From a rocprofv3 --kernel-trace of `bench.py --frames-in-flight 2`: the timed region's trace kernels (the run of `steps`
launches that alternate between two queues), their durations, how much of each one's interval the next one shares, and the
span per launch -- beside the same figures for the one-stream run that follows it in the same process."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_trace_tile" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows]


def describe(run, label):
    if len(run) < 2:
        return
    dur = [(e - s) / 1e3 for s, e, _ in run]
    span = (run[-1][1] - run[0][0]) / 1e3 / len(run)
    ov = [max(0, min(run[i][1], run[i + 1][1]) - run[i + 1][0]) / 1e3 for i in range(len(run) - 1)]
    gaps = [(run[i + 1][0] - run[i][1]) / 1e3 for i in range(len(run) - 1)]
    queues = sorted(set(q for _, _, q in run))
    print(f"{label}: {len(run)} launches on queue(s) {queues}: span per launch {span:.1f} us, kernel duration mean {sum(dur) / len(dur):.1f} us, "
          f"overlap with the next launch mean {sum(ov) / len(ov):.1f} us (max {max(ov):.1f}), launches that overlap the next: "
          f"{sum(1 for o in ov if o > 0)} of {len(ov)}, start-to-previous-end gap mean {sum(gaps) / len(gaps):+.1f} us")


# runs of consecutive launches that alternate queues (two frames in flight) / stay on one queue
i = 0
best_alt, best_one = [], []
while i < len(iv):
    j = i
    while j + 1 < len(iv) and iv[j + 1][2] != iv[j][2]:
        j += 1
    if j - i + 1 > len(best_alt):
        best_alt = iv[i:j + 1]
    i = j + 1
describe(best_alt[-steps:], "two frames in flight (longest alternating run, its last launches)")
# the one-stream run right behind it: the next `steps` launches on a single queue
end = best_alt[-1][1] if best_alt else 0
after = [x for x in iv if x[0] > end]
describe(after[:steps], "one stream (the launches that follow)")
