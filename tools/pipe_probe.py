"""ON THE GPU BOX: what would two frames in flight buy?  Two contexts on one GPU stand in for the two buffer sets
(G-buffer, frame, stream) of a renderer that starts frame k + 1 while frame k drains.
  trace only:     A.trace x 2n on one stream      vs   A.trace, B.trace alternating (two streams, two kernels in flight)
  Render():       (raster + trace) x 2n on A      vs   alternating A / B
Prints ms per step / per frame (wall, n steps issued back to back, one synchronize at the end)."""
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


A, B = make(), make()


def timed(fn, n=100):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


def render(c):
    c.render_gbuffer(inp["view_proj"]); c.trace_resident()


# the product form: ONE context, two frame slots (vct_set_frames_in_flight)
S = make()
S.set_frames_in_flight(2)
S.select_frame_slot(1); S.render_gbuffer(inp["view_proj"]); S.trace_resident(); S.select_frame_slot(0); S.synchronize()
kk = [0]


def slot_trace():
    S.select_frame_slot(kk[0] & 1); kk[0] += 1
    S.trace_resident()


def slot_render():
    S.select_frame_slot(kk[0] & 1); kk[0] += 1
    render(S)


def gi(c):
    c.gi_pass(inp["light_vp"], inp["view_proj"])


def slot_gi():
    S.select_frame_slot(kk[0] & 1); kk[0] += 1
    gi(S)


for rnd in range(2):
    g1 = timed(lambda: (gi(A), gi(A)), 50) / 2
    g2 = timed(lambda: (gi(A), gi(B)), 50) / 2
    g3 = timed(lambda: (slot_gi(), slot_gi()), 50) / 2
    print(f"round {rnd}: vct_gi_pass one context {g1:.4f} ms/pass, two CONTEXTS alternating {g2:.4f} ({100 * (g1 / g2 - 1):+.1f} %), "
          f"one context with two frame slots {g3:.4f}", flush=True)
    t1 = timed(lambda: (A.trace_resident(), A.trace_resident())) / 2
    t2 = timed(lambda: (A.trace_resident(), B.trace_resident())) / 2
    r1 = timed(lambda: (render(A), render(A))) / 2
    r2 = timed(lambda: (render(A), render(B))) / 2
    s1 = timed(lambda: (slot_trace(), slot_trace())) / 2
    s2 = timed(lambda: (slot_render(), slot_render())) / 2
    print(f"round {rnd}: one context, two frame slots: trace {s1:.4f} ms/step, Render() {s2:.4f} ms/frame")
    print(f"round {rnd}: trace one stream {t1:.4f} ms/step, two in flight {t2:.4f} ({100 * (t1 / t2 - 1):+.1f} %);  "
          f"Render() one stream {r1:.4f} ms/frame, two in flight {r2:.4f} ({100 * (r1 / r2 - 1):+.1f} %)", flush=True)
