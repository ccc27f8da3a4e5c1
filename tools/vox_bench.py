"""ON THE GPU BOX: time the voxel stages (voxelize / inject / mips) with and without the shadow map (PCF),
HIP events on the context stream.  Usage: tools/vox_bench.py [bench.py scene flags]"""
import os
import sys

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
V, S = args.voxel_dim, args.shadow_size
def ev():
    return torch.cuda.Event(enable_timing=True)


for mips in (1, 0):
  ctx = vct.Context(vct.default_config(voxel_dim=V, width=64, height=64, shadow_map_size=S, texture_mipmaps=mips))
  ctx.upload_scene(inp["scene"])
  st = torch.cuda.ExternalStream(ctx.stream())
  for shadow in (True, False):
    if shadow:
        ctx.render_shadow_map(inp["light_vp"])
    else:
        ctx.upload_shadow_map(None, None)
    best = None
    with torch.cuda.stream(st):
        for _ in range(8):
            e = [ev() for _ in range(4)]
            e[0].record(); ctx.voxelize(); e[1].record(); ctx.inject_light(); e[2].record(); ctx.build_mips(); e[3].record()
            ctx.synchronize()
            t = [e[i].elapsed_time(e[i + 1]) for i in range(3)]
            best = t if best is None else [min(a, b) for a, b in zip(best, t)]
    c = ctx.stage_counts()
    print(f"V={V} texture_mipmaps={mips} shadow={shadow}: voxelize {best[0]:.4f} ms  inject {best[1]:.4f} ms  mips {best[2]:.4f} ms   "
          f"fragments {c['vox_candidates']}  bricks {c['touched_bricks']}  triangles {c['triangles']}  items {c.get('vox_items')}")
  del ctx
raise SystemExit(0)


def ev():
    return torch.cuda.Event(enable_timing=True)


for shadow in (True, False):
    if shadow:
        ctx.render_shadow_map(inp["light_vp"])
    else:
        ctx.upload_shadow_map(None, None)
    best = None
    with torch.cuda.stream(st):
        for _ in range(8):
            e = [ev() for _ in range(4)]
            e[0].record(); ctx.voxelize(); e[1].record(); ctx.inject_light(); e[2].record(); ctx.build_mips(); e[3].record()
            ctx.synchronize()
            t = [e[i].elapsed_time(e[i + 1]) for i in range(3)]
            best = t if best is None else [min(a, b) for a, b in zip(best, t)]
    c = ctx.stage_counts()
    print(f"V={V} shadow={shadow}: voxelize {best[0]:.4f} ms  inject {best[1]:.4f} ms  mips {best[2]:.4f} ms   "
          f"fragments {c['vox_candidates']}  bricks {c['touched_bricks']}  triangles {c['triangles']}")
