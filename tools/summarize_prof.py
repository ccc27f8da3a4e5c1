#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory (gpurun_out/prof_<tag>) into the small text
summary committed under profiles/: per-kernel stats of the kernel-trace pass and per-kernel means
of every PMC counter collected."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = []
    for f in sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"== kernel stats ({os.path.relpath(f, src)}) ==")
        with open(f) as fh:
            lines += [ln.rstrip() for ln in fh]
    # per-dispatch durations of the trace kernel from the raw kernel trace
    for f in sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)):
        d = defaultdict(list)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        lines.append("== per-kernel durations from kernel_trace.csv (us): n / mean / min / max ==")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{k[:90]:90s} {len(v):5d} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f}")
        # Steady state of the dominant kernel: the profiled bench.py run ends with its timed region (--steps launches
        # back to back) followed by min(steps, 20) event-timed launches; the mean over ALL launches also holds the cold
        # pre-roll launches and those that follow other kernels of the GI-pass timing, so it can exceed ms_per_step.
        # The last 20 launches of the run are what bench.py's clocks see.
        for k, v in d.items():
            if "k_trace_tile" in k and len(v) >= 20:
                tail = sorted(v[-20:])
                lines.append(f"== steady state of {k[:60]}: last 20 launches: mean {sum(tail)/20:.2f} us, median "
                             f"{(tail[9]+tail[10])/2:.2f} us, min {tail[0]:.2f}, max {tail[-1]:.2f} ==")
    for p in sorted(glob.glob(os.path.join(src, "pmc*"))):
        if not os.path.isdir(p):
            continue
        for f in sorted(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            lines.append(f"== PMC {os.path.basename(p)}: per-kernel mean counter value per dispatch ==")
            for k, cs in acc.items():
                for c, v in cs.items():
                    lines.append(f"{k[:70]:70s} {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
    # HBM traffic of the trace kernel per launch, corrected as calibrated on this box
    # (profiles/r01_fetch_write_calibration.txt): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024
    traffic = {}
    trace_name = None
    for p in sorted(glob.glob(os.path.join(src, "pmc*"))):
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if "k_trace_tile" in r["Kernel_Name"]:
                        trace_name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                    if "k_trace_tile" in r["Kernel_Name"] and r["Counter_Name"] in (
                            "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                            "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"):
                        traffic.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic and len(sys.argv) > 3:
        import json
        fetch = sum(traffic["FETCH_SIZE"]) / len(traffic["FETCH_SIZE"])
        write = sum(traffic["WRITE_SIZE"]) / len(traffic["WRITE_SIZE"])
        with open(sys.argv[3], "w") as fh:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
            import bench
            # cone steps of one launch, from the bench line of the kernel-trace pass of the same command: lets bench.py
            # scale the instruction counts to a slab of the frame (N > 1 lines)
            steps = None
            try:
                with open(os.path.join(src, "trace.log")) as tl:
                    for ln in tl:
                        if ln.startswith("{"):
                            steps = json.loads(ln).get("cone_steps_per_frame")
            except (OSError, ValueError):
                pass
            json.dump({"kernel": trace_name, "source": os.path.basename(dst), "cone_steps_per_launch": steps,
                       "kernel_source_sha16": bench.kernel_source_sha(),
                       "fetch_size_kib": fetch, "write_size_kib": write,
                       "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                       "wave_instructions_per_launch": {k[9:].lower(): sum(v) / len(v) for k, v in traffic.items()
                                                        if k.startswith("SQ_INSTS_")},
                       "gpu_cycles_per_launch": (sum(traffic["GRBM_GUI_ACTIVE"]) / len(traffic["GRBM_GUI_ACTIVE"]) / 8.0
                                                 if "GRBM_GUI_ACTIVE" in traffic else None),
                       # of the kernel's wave-cycles: parked on s_waitcnt (memory) / stalled at issue (SQ_WAIT_ANY, SQ_WAIT_INST_ANY)
                       "wave_cycles_waiting_on_memory": (sum(traffic["SQ_WAIT_ANY"]) / max(sum(traffic["SQ_WAVE_CYCLES"]), 1.0)
                                                         if "SQ_WAIT_ANY" in traffic and "SQ_WAVE_CYCLES" in traffic and
                                                         len(traffic["SQ_WAIT_ANY"]) == len(traffic["SQ_WAVE_CYCLES"]) else None),
                       "wave_cycles_waiting_to_issue": (sum(traffic["SQ_WAIT_INST_ANY"]) / max(sum(traffic["SQ_WAVE_CYCLES"]), 1.0)
                                                        if "SQ_WAIT_INST_ANY" in traffic and "SQ_WAVE_CYCLES" in traffic and
                                                        len(traffic["SQ_WAIT_INST_ANY"]) == len(traffic["SQ_WAVE_CYCLES"]) else None),
                       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, calibrated with tools/fetch_calib.hip"}, fh, indent=1)
        lines.append(f"== trace kernel HBM bytes per launch (corrected): {(2.0 * fetch + write) * 1024.0:.4g} ==")
    # HBM traffic of EVERY kernel per dispatch (same correction), for bench.py's stage_roofline
    per_kernel = {}
    for p in sorted(glob.glob(os.path.join(src, "pmc*"))):
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                        per_kernel.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if per_kernel and len(sys.argv) > 3:
        import json
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        import bench
        out = {}
        for k, c in per_kernel.items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c and "anonymous namespace" in k:
                name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
                write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
                out[name] = {"fetch_bytes": int(2.0 * fetch * 1024), "write_bytes": int(write * 1024),
                             "dispatches_profiled": len(c["FETCH_SIZE"])}
        # (argv[4]: another name for the per-kernel file -- non-default workloads keep their own)
        with open(os.path.join(os.path.dirname(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "stage_traffic.json"), "w") as fh:
            json.dump({"source": os.path.basename(dst), "source_sha16": bench.all_sources_sha(),
                       "correction": "fetch = 2 * FETCH_SIZE KiB, write = WRITE_SIZE KiB (profiles/r01_fetch_write_calibration.txt)",
                       "kernels": out}, fh, indent=1)
        lines.append("== per-kernel HBM bytes per dispatch (corrected) -> stage_traffic.json ==")
        for k, v in out.items():
            lines.append(f"{k[:60]:60s} fetch {v['fetch_bytes']:>12d} write {v['write_bytes']:>12d}")
    if os.path.exists(os.path.join(src, "passes.txt")):
        lines.append("== passes ==")
        lines += [ln.rstrip() for ln in open(os.path.join(src, "passes.txt"))]
    with open(dst, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
