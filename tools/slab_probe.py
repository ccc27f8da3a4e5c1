import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, vctpkg, bench
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc
args = bench.parse()
w, h, V = 1920, 1080, 256
inp = bench.build_inputs(args, vct, sc)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=4096))
ctx.set_camera_position(inp["cam"]); ctx.set_light_direction(inp["light"])
ctx.upload_scene(inp["scene"])
ctx.render_shadow_map(inp["light_vp"]); ctx.render_gbuffer(inp["view_proj"])
ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
for world in (2, 4, 8):
    res = []
    for rank in range(world):
        r0, r1, per = vct.slab_partition(h, world, rank)
        for _ in range(30): ctx.trace_gbuffer_rows(r0, r1)
        ctx.synchronize()
        ms = []
        for _ in range(10):
            ctx.trace_gbuffer_rows(r0, r1); ctx.synchronize(); ms.append(ctx.last_trace_ms())
        res.append((r0, r1, ctx.last_step_count(), round(float(np.median(ms)), 4)))
    steps = [r[2] for r in res]; t = [r[3] for r in res]
    print(world, "slabs: steps max/mean", round(max(steps) / np.mean(steps), 3), "kernel ms max", max(t), "mean", round(float(np.mean(t)), 4), "sum", round(sum(t), 4), t)
