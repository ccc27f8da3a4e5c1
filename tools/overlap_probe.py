"""ON THE GPU BOX: how much of a frame's input stages (shadow map, voxelize, inject, mips, G-buffer raster) hides behind
the trace of the PREVIOUS frame when both are in flight?  Two contexts on one GPU stand in for the two buffer sets of a
cross-frame pipeline: context A traces its resident frame while context B runs the input stages, and vice versa.
Prints per-frame times: stages alone, trace alone, serial sum, overlapped."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import vctpkg
import bench

vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc

args = bench.parse()
w, h, V = args.width, args.height, args.voxel_dim
inp = bench.build_inputs(args, vct, sc)


def make():
    ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=args.shadow_size))
    ctx.set_camera_position(inp["cam"]); ctx.set_light_direction(inp["light"])
    ctx.upload_scene(inp["scene"])
    ctx.render_shadow_map(inp["light_vp"]); ctx.render_gbuffer(inp["view_proj"])
    ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
    ctx.trace_gbuffer_rows(0, (h + 7) // 8); ctx.synchronize()
    return ctx


os.environ["VCT_STREAM_PRIORITY"] = os.environ.get("PROBE_TRACE_PRIO", "low")
A = make()
os.environ["VCT_STREAM_PRIORITY"] = os.environ.get("PROBE_STAGE_PRIO", "high")
B = make()


def stages(c):
    c.render_shadow_map(inp["light_vp"]); c.voxelize(); c.inject_light(); c.build_mips(); c.render_gbuffer(inp["view_proj"])


def timed(fn, n=40):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


t_trace = timed(lambda: A.trace_resident())
t_stages = timed(lambda: stages(B))
t_serial = timed(lambda: (stages(A), A.trace_resident()))
t_gi = timed(lambda: A.gi_pass(inp["light_vp"], inp["view_proj"]))


def overlapped():
    # frame k: A traces while B prepares; frame k+1: B traces while A prepares
    A.trace_resident(); stages(B)
    B.trace_resident(); stages(A)


t_ov = timed(overlapped) / 2
t_fixed = timed(lambda: (A.trace_resident(), stages(B)))       # fixed roles: A (its stream priority) traces, B prepares
print(f"fixed roles (trace on {os.environ.get('PROBE_TRACE_PRIO', 'low')}-priority stream, stages on "
      f"{os.environ.get('PROBE_STAGE_PRIO', 'high')}): {t_fixed:.4f} ms per frame")
print(f"trace alone {t_trace:.4f} ms, input stages alone {t_stages:.4f} ms, serial on one stream {t_serial:.4f} ms, "
      f"vct_gi_pass {t_gi:.4f} ms, two contexts overlapped {t_ov:.4f} ms per frame")
