#!/usr/bin/env python3
"""Wave-level statistics of the trace kernel on the bench workload (run ON THE GPU BOX).

Needs an instrumented build of the library:  tools/build_ab.sh stats "-DVCT_STATS=1 -fno-slp-vectorize"
then  VCT_AMD_LIB=$PWD/build/ab/stats.so python tools/trace_stats.py [bench.py scene flags] > profiles/rNN_trace_stats.json

Reports, for one launch of the trace kernel: march-loop iterations executed by waves, the mean fraction
of live lanes per executed iteration (the lane utilisation of the lane-per-pixel mapping), and how the
level samples were served (cooperative block all zero / cooperative gather / per-lane gather).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import bench  # noqa: E402


def main():
    args = bench.parse()
    import vctpkg
    vct = vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    w, h, V = args.width, args.height, args.voxel_dim
    inp = bench.build_inputs(args, vct, sc)
    ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=args.shadow_size,
                                         trace_variant=args.variant))
    ctx.set_camera_position(inp["cam"])
    ctx.set_light_direction(inp["light"])
    if inp["scene"] is not None:
        s = inp["scene"]
        ctx.upload_scene(s)
        ctx.render_shadow_map(inp["light_vp"])
        ctx.render_gbuffer(inp["view_proj"])
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        ctx.trace_gbuffer_rows(0, (h + 7) // 8)
    else:
        ctx.upload_volume(inp["volume"])
        ctx.build_mips()
        ctx.trace(inp["planes"])
    out = {"workload": inp["label"], "voxel_dim": V, "width": w, "height": h, "launches": []}
    if inp["scene"] is not None:
        sc_ = ctx.stage_counts()
        nbricks = (V // 8) ** 3
        out["brick_occupancy"] = {"touched_8x8x8_bricks": sc_["touched_bricks"], "of": nbricks,
                                  "fraction": round(sc_["touched_bricks"] / nbricks, 5),
                                  "voxelizer_candidates": sc_["vox_candidates"], "triangles": sc_["triangles"]}
    for ts in (0.07, 0.105, 0.2):
        ctx.set_cone_apertures(0.577, ts)
        ctx.trace_resident()
        ctx.synchronize()
        st = ctx.last_trace_stats()
        samples = st["coop_zero"] + st["coop_hit"] + st["fallback"]
        st.update({
            "tan_specular": ts,
            "cone_steps": ctx.last_step_count(),
            "kernel_ms_instrumented": round(ctx.last_trace_ms(), 4),
            "mean_live_lane_fraction": round(st["lane_steps"] / max(st["wave_steps"] * 64, 1), 4),
            "level_samples": samples,
            "frac_coop_zero": round(st["coop_zero"] / max(samples, 1), 4),
            "frac_coop_gather": round(st["coop_hit"] / max(samples, 1), 4),
            "frac_per_lane": round(st["fallback"] / max(samples, 1), 4),
            "per_lane_mean_live_lanes": round(st["fallback_lanes"] / max(st["fallback"], 1), 2),
            # of the per-lane samples' live 2x2 quads / 4x4 quadrants: how many could share one small block
            "frac_quads_fit_3x3x3": round(st["quads_fit333"] / max(st["quads_live"], 1), 4),
            "frac_quads_same_cell": round(st["quads_same"] / max(st["quads_live"], 1), 4),
            "frac_quadrants_fit_4x4x4": round(st["quadrants_fit444"] / max(st["quadrants_live"], 1), 4),
        })
        out["launches"].append(st)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
