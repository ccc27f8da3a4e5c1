"""ON THE GPU BOX: how much of the trace kernel's time is ramp-up + tail?  T(full frame) against the
sum over k equal slabs (each launch pays its own ramp/tail once)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, vctpkg
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc, slabs
V, w, h, S = 256, 1920, 1080, 4096
s = sc.Scene(sc.ATRIUM, 1.0, 1234)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=S))
ctx.upload_triangles(s.pos, s.material, s.albedo)
ctx.upload_mesh_attributes(*s.frames(), s.specular)
light = (0.0, 1.0, 0.25)
cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
ctx.set_camera_position(tuple(cam.position)); ctx.set_light_direction(light)
ctx.render_shadow_map(sc.light_view_proj(light))
ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
ctx.render_gbuffer(sc.camera_view_proj(cam, w, h))
def t(rows):
    best = 1e9
    for _ in range(8):
        ctx.trace_gbuffer_rows(*rows); best = min(best, ctx.last_trace_ms())
    return best, ctx.last_step_count()
full, steps = t((0, 135))
print(f"full frame: {full:.4f} ms  {steps} steps")
for k in (2, 4, 8):
    parts = [t(r) for r in slabs.partition(h, k)]
    tot = sum(p[0] for p in parts)
    print(f"{k} slabs: sum {tot:.4f} ms (+{(tot/full-1)*100:.1f}%)  per-slab ms {[round(p[0],3) for p in parts]}  steps {[p[1] for p in parts]}")
