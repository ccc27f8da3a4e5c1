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
ctx.trace_gbuffer_rows(0, (h + 7) // 8); ctx.synchronize()
row_steps = ctx.last_row_steps()
for world, balanced in ((2, False), (4, False), (8, False), (2, True), (4, True), (8, True)):
    res = []
    starts = vct.slab_partition_weighted(row_steps, world)
    for rank in range(world):
        r0, r1, per = vct.slab_partition(h, world, rank)
        if balanced:
            r0, r1 = int(starts[rank]), int(starts[rank + 1])
        for _ in range(30): ctx.trace_gbuffer_rows(r0, r1)
        ctx.synchronize()
        ms = []
        for _ in range(10):
            ctx.trace_gbuffer_rows(r0, r1); ctx.synchronize(); ms.append(ctx.last_trace_ms())
        res.append((r0, r1, ctx.last_step_count(), round(float(np.median(ms)), 4)))
    steps = [r[2] for r in res]; t = [r[3] for r in res]
    print(world, "slabs" + (" balanced" if balanced else " equal") + ": steps max/mean", round(max(steps) / np.mean(steps), 3), "kernel ms max", max(t), "mean", round(float(np.mean(t)), 4), "sum", round(sum(t), 4), t)


# interleaved slabs (tile row r -> rank r % world): balanced by construction, no histogram and no feedback
ty = (h + 7) // 8
for world in (2, 4, 8):
    t, steps = [], []
    for rank in range(world):
        for _ in range(30): ctx.trace_gbuffer_strided(rank, ty, world)
        ctx.synchronize()
        ms = []
        for _ in range(10):
            ctx.trace_gbuffer_strided(rank, ty, world); ctx.synchronize(); ms.append(ctx.last_trace_ms())
        t.append(round(float(np.median(ms)), 4)); steps.append(ctx.last_step_count())
    print(world, "slabs interleaved: steps max/mean", round(max(steps) / np.mean(steps), 3), "kernel ms max", max(t), "mean", round(float(np.mean(t)), 4), "sum", round(sum(t), 4), t)


# time-feedback balancing (what bench.py --slabs balanced does over the control plane): rows are re-weighted by the
# measured kernel time per executed step of the slab they were in, the boundaries re-cut, a few rounds
def measure(r0, r1):
    for _ in range(20): ctx.trace_gbuffer_rows(r0, r1)
    ctx.synchronize()
    ms = []
    for _ in range(10):
        ctx.trace_gbuffer_rows(r0, r1); ctx.synchronize(); ms.append(ctx.last_trace_ms())
    return float(np.median(ms))


for world in (4, 8):
    cost = row_steps.astype(np.float64) + 1.0
    for it in range(4):
        starts = vct.slab_partition_weighted(np.maximum(cost, 1.0).astype(np.uint64), world)
        t = [measure(int(starts[r]), int(starts[r + 1])) for r in range(world)]
        print(world, "slabs, feedback round", it, "rows", list(np.diff(starts)), "kernel ms max", round(max(t), 4), "mean", round(float(np.mean(t)), 4))
        for r in range(world):
            a, b = int(starts[r]), int(starts[r + 1])
            if b > a:
                cost[a:b] *= t[r] / max(float(cost[a:b].sum()), 1.0) * 1e6       # -> microseconds-equivalents per row
