"""Host-side logic of bench.py that needs no GPU: the roofline block (VALU-issue bound, useful-FMA floor, PMC replay and
its scaling to a slab at N > 1), the sha gates of the PMC replay files, the argument -> replay-file selection."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def args(**kw):
    d = dict(variant=0, obj=None, bounces=1, anisotropic=False, scene_detail=1.0, voxel_dim=256, width=1920, height=1080,
             scene="atrium", shadow_size=4096)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_roofline_block_without_a_profile_still_reports_the_useful_floor():
    r = bench.roofline_block({"ok": False, "why": "no PMC profile for this configuration"}, 0.6, 132_243_039, 8.67e9, 14000.0)
    assert r["bound"] == "valu_issue" and r["frac"] is None and r["achieved"] is None and r["traffic"] is None
    # 72 useful wave-instructions per 64 cone steps against the 2-cycle issue peak
    want = 132_243_039 / 64.0 * 72 / 0.6e-3 / 1e9 / bench.VALU_PEAK_GINSTR
    assert abs(r["useful_frac"] - want) < 1e-3 and "why" not in r and "no PMC profile" in r["note"]


def test_roofline_block_replays_counters_and_scales_them_to_a_slab():
    prof = {"ok": True, "source": "x.txt", "kernel_source_sha16": "0" * 16, "cone_steps_per_launch": 1000 * 64,
            "wave_instructions_per_launch": {"valu": 263_000.0, "salu": 72_000.0}, "hbm_bytes_per_launch": 2.0e6,
            "gpu_cycles_per_launch": 1000.0, "model_issue_cycles_per_valu_instr": 2.6, "salu_issue_cycles_per_instr": 4.0}
    whole = bench.roofline_block(dict(prof), 1.0e-3, 1000 * 64, 1.0, 1.0)
    assert whole["valu_wave_instructions_per_64_cone_steps"] == 263.0
    assert abs(whole["frac"] - 263_000.0 / 1.0e-6 / 1e9 / bench.VALU_PEAK_GINSTR) < 1e-4
    assert whole["traffic"] == round(2.0e6 / 1.0e-6 / 1e9, 1) and "scaled" not in whole
    # N > 1: rank 0 traced a quarter of the steps in a quarter of the time -> the same rate, labelled, no byte / cycle figures
    slab = dict(prof, scale_by_steps=True, hbm_bytes_per_launch=None)
    r = bench.roofline_block(slab, 0.25e-3, 250 * 64, 1.0, 1.0)
    assert abs(r["frac"] - whole["frac"]) < 1e-4 and r["traffic"] is None and "scaled" in r
    assert "valu_pipe_busy_model" not in r and r["valu_wave_instructions_per_64_cone_steps"] == 263.0


def test_pmc_replay_files_are_chosen_by_workload_and_gated_by_the_source_sha():
    a = bench.pmc_profile(args(), 1)
    c5 = bench.pmc_profile(args(scene="bistro", voxel_dim=1024, width=3840, height=2160), 1)
    other = bench.pmc_profile(args(voxel_dim=512), 1)
    assert other == {"ok": False, "why": "no PMC profile for this configuration"}
    sha = bench.kernel_source_sha()
    for prof, name in ((a, "trace_traffic.json"), (c5, "trace_traffic_c5.json")):
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            rec = json.load(fh)
        if rec["kernel_source_sha16"] == sha:
            assert prof["ok"] and prof["source"] == rec["source"] and prof["cone_steps_per_launch"] > 0
        else:       # a stale file is refused, with the reason in the line
            assert not prof["ok"] and sha in prof["why"]
    assert a.get("source") != c5.get("source") or not (a["ok"] and c5["ok"])
    # N > 1 scales the whole-frame counters instead of refusing them
    n2 = bench.pmc_profile(args(), 2)
    assert n2["ok"] == a["ok"] and (not n2["ok"] or (n2["scale_by_steps"] and n2["hbm_bytes_per_launch"] is None))
    # another variant / a bounce / a user mesh never replay
    assert not bench.pmc_profile(args(variant=3), 1)["ok"] and not bench.pmc_profile(args(bounces=2), 1)["ok"]
