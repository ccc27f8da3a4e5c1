"""ON THE GPU BOX: the two raster input stages alone (shadow map, G-buffer), N passes -- meant to run under
`rocprofv3 --kernel-trace --stats` (tools/raster_prof.sh).  Usage: raster_prof.py atrium|bistro W H [passes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (device runtime initialisation order as in bench.py)
import vctpkg  # noqa: E402

vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 10
if name == "bistro":
    s = sc.Scene(sc.BISTRO, 1.0, 1234)
    cam = sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)
elif name == "atrium-textured":
    s = sc.Scene(sc.ATRIUM_TEXTURED, 1.0, 1234)
    cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
else:
    s = sc.Scene(sc.ATRIUM, 1.0, 1234)
    cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
ctx = vct.Context(vct.default_config(voxel_dim=64, width=w, height=h, shadow_map_size=4096))
ctx.upload_scene(s)
light = (0.0, 1.0, 0.25)
ctx.set_camera_position(tuple(cam.position))
ctx.set_light_direction(light)
lvp, vp = sc.light_view_proj(light), sc.camera_view_proj(cam, w, h)
st = torch.cuda.ExternalStream(ctx.stream())
best = [1e9, 1e9]
with torch.cuda.stream(st):
    for _ in range(passes):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); ctx.render_shadow_map(lvp)
        e[1].record(); ctx.render_gbuffer(vp)
        e[2].record(); ctx.synchronize()
        best = [min(best[i], e[i].elapsed_time(e[i + 1])) for i in range(2)]
print(f"{name} {w}x{h}: shadow {best[0]:.4f} ms, gbuffer {best[1]:.4f} ms")
