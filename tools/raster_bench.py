"""ON THE GPU BOX: time the GPU raster input stages (shadow map 4096^2, 1080p G-buffer) and a full
resident frame (G-buffer + trace)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, vctpkg
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc
V, w, h, S = 256, 1920, 1080, 4096
s = sc.Scene(sc.ATRIUM, 1.0, 1234)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=S))
ctx.upload_triangles(s.pos, s.material, s.albedo)
ctx.upload_mesh_attributes(*s.frames(), s.specular)
light = (0.0, 1.0, 0.25)
cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
ctx.set_camera_position(tuple(cam.position)); ctx.set_light_direction(light)
lvp, vp = sc.light_view_proj(light), sc.camera_view_proj(cam, w, h)
st = torch.cuda.ExternalStream(ctx.stream())
def ev(): return torch.cuda.Event(enable_timing=True)
best = None
with torch.cuda.stream(st):
    for _ in range(5):
        e = [ev() for _ in range(7)]
        e[0].record(); ctx.render_shadow_map(lvp)
        e[1].record(); ctx.voxelize()
        e[2].record(); ctx.inject_light()
        e[3].record(); ctx.build_mips()
        e[4].record(); ctx.render_gbuffer(vp)
        e[5].record(); ctx.trace_resident()
        e[6].record(); ctx.synchronize()
        t = [e[i].elapsed_time(e[i + 1]) for i in range(6)]
        best = t if best is None else [min(a, b) for a, b in zip(best, t)]
names = ["shadow_map", "voxelize", "inject_resolve", "build_mips", "gbuffer", "trace"]
print("resident frame, ms:", {n: round(v, 4) for n, v in zip(names, best)}, "total", round(sum(best), 3))
