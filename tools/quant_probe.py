"""ON THE GPU BOX: is the trace launch quantised in generations of workgroups?  Kernel time of the first r tile rows of the
1080p atrium frame, r chosen around whole multiples of the 2,304 resident workgroups."""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, vctpkg
vct = vctpkg.load()
from voxel_cone_tracing_amd import scene as sc
V, w, h, S = 256, 1920, 1080, 4096
s = sc.Scene(sc.ATRIUM, 1.0, 1234)
ctx = vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=S))
ctx.upload_scene(s)
light = (0.0, 1.0, 0.25)
cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
ctx.set_camera_position(tuple(cam.position)); ctx.set_light_direction(light)
ctx.render_shadow_map(sc.light_view_proj(light))
ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
ctx.render_gbuffer(sc.camera_view_proj(cam, w, h))
def t(rows, n=12):
    ts = []
    for _ in range(n):
        ctx.trace_gbuffer_rows(*rows); ts.append(ctx.last_trace_ms())
    return min(ts), float(np.median(ts)), ctx.last_step_count()
for _ in range(10): t((0, 135), 8)     # clocks up before anything is compared
base = None
for r1 in (135, 134, 135, 133, 130, 125, 120, 115, 106, 96, 77, 58, 48, 39, 29, 19, 10):
    mn, md, steps = t((0, r1))
    if base is None: base = (mn, steps)
    print(f"rows 0..{r1:3d}: tiles {r1*240:6d} ({r1*240/2304:5.2f} gens of 2304)  min {mn:.4f} ms  median {md:.4f}  steps {steps}  ms per Msteps {mn/steps*1e6:.4f}")
