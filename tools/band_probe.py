#!/usr/bin/env python3
"""ON THE GPU BOX: does the G-buffer pass evict the chain from the Infinity Cache between frames?  Times the trace kernel
(HIP events) (a) repeated alone, (b) alternating with the raster pass of the same frame, (c) the frame cut into B bands
of tile rows -- raster band, trace band, next band -- so that a band's G-buffer and the chain fit the 256 MiB cache
together.  Usage: tools/band_probe.py [bench.py scene flags]"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import vctpkg
import bench

vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc

args = bench.parse()
inp = bench.build_inputs(args, vct, sc)
w, h, V = args.width, args.height, args.voxel_dim
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=args.shadow_size))
ctx.set_camera_position(inp["cam"]); ctx.set_light_direction(inp["light"])
ctx.upload_scene(inp["scene"])
ctx.render_shadow_map(inp["light_vp"]); ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
ty = (h + 7) // 8
for _ in range(12):
    ctx.render_gbuffer(inp["view_proj"]); ctx.trace_gbuffer_rows(0, ty)
ctx.synchronize()


def wall(fn, n=40):
    for _ in range(5):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def trace_ms(pre, n=20):
    ms = []
    for _ in range(n):
        pre()
        ctx.trace_gbuffer_rows(0, ty)
        ms.append(ctx.last_trace_ms())
    return float(np.median(ms))


print(f"{inp['label']}  {w}x{h} V={V}")
print(f"(a) trace alone, repeated:            kernel {trace_ms(lambda: None):.4f} ms   wall/frame {wall(lambda: ctx.trace_resident()):.4f} ms")
print(f"(b) raster pass then trace:           kernel {trace_ms(lambda: ctx.render_gbuffer(inp['view_proj'])):.4f} ms   "
      f"wall/frame {wall(lambda: (ctx.render_gbuffer(inp['view_proj']), ctx.trace_resident())):.4f} ms")
for B in (2, 3, 4, 6):
    cuts = [round(i * ty / B) for i in range(B + 1)]

    def frame():
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.render_gbuffer_rows(inp["view_proj"], a, b)
            ctx.trace_resident_rows(a, b) if hasattr(ctx, "trace_resident_rows") else ctx.trace_gbuffer_rows(a, b)
    print(f"(c) {B} bands (raster band, trace band): wall/frame {wall(frame):.4f} ms")
