"""ON THE GPU BOX: the library on the scene and size tools/time_ref_gl.py times the reference's own GLSL on
(tests/golden/ref_pipeline_v128.npz scene arrays, 128^3, 1280 x 720 = the fixture's 16:9 camera at the reference's
window size): shadow map, voxelize (the shaders-as-written mode) + inject + mips, Render = G-buffer + trace.
Also the PCIe-inclusive form of the boundary: a frame downloaded to host memory every Render."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import vctpkg  # noqa: E402

vct = vctpkg.load()
with np.load(os.path.join(ROOT, "tests", "golden", "ref_pipeline_v128.npz")) as z:
    f = {k: z[k] for k in z.files}
W, H = 1280, 720
ctx = vct.Context(vct.default_config(voxel_dim=int(f["V"]), width=W, height=H, shadow_map_size=int(f["S"]),
                                     ambient_factor=float(f["ambient"])))
ctx.upload_triangles(f["pos"], f["material"], f["albedo"])
ctx.upload_mesh_attributes(f["nrm"], f["tan"], f["bit"], f["specular"])
ctx.upload_mesh_uvs(f["uv"])
ctx.upload_textures([f[f"texture_{i}"] for i in range(9)], f["mat_tex"])
ctx.set_camera_position(tuple(float(x) for x in f["eye"]))
ctx.set_light_direction(tuple(float(x) for x in f["light_dir"]))
view_proj = (f["proj"].reshape(4, 4).T @ f["view"].reshape(4, 4).T).T.astype(np.float32).reshape(16)
st = torch.cuda.ExternalStream(ctx.stream())
best = [1e9] * 4
with torch.cuda.stream(st):
    for _ in range(20):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        e[0].record(); ctx.render_shadow_map(f["depth_vp"])
        e[1].record(); ctx.voxelize(vct.VOX_REFERENCE); ctx.inject_light(); ctx.build_mips()
        e[2].record(); ctx.render_gbuffer(view_proj)
        e[3].record(); ctx.trace_resident()
        e[4].record(); ctx.synchronize()
        best = [min(best[i], e[i].elapsed_time(e[i + 1])) for i in range(4)]
    planes = ctx.download_gbuffer()
    shaded = int((planes[18] >= 0.5).sum())
    # the boundary handing the frame to host memory every Render (what the facade did before round 5)
    host = 1e9
    for _ in range(10):
        t0 = time.perf_counter()
        ctx.render_gbuffer(view_proj)
        ctx.trace_current()
        host = min(host, time.perf_counter() - t0)
render_ms = best[2] + best[3]
print(f"HIP, same scene / sizes as tools/time_ref_gl.py: V={int(f['V'])}, shadow map {int(f['S'])}^2, frame {W}x{H}, {shaded} shaded pixels")
print(f"shadow map {best[0]:.4f} ms, voxelize (reference mode) + inject + mips {best[1]:.4f} ms")
print(f"Render = G-buffer {best[2]:.4f} + trace {best[3]:.4f} = {render_ms:.4f} ms = {shaded * 7 / render_ms / 1e3:.0f} Mcones/s over shaded pixels")
print(f"Render with the RGBA16F frame ({W * H * 8 / 1e6:.1f} MB) downloaded to host memory each call (PCIe-inclusive, wall): "
      f"{host * 1e3:.4f} ms = {shaded * 7 / host / 1e6:.0f} Mcones/s")
# the other host-buffer boundary: vct_trace with the G-buffer planes in host memory (23 fp32 planes up, the frame down)
with torch.cuda.stream(st):
    up = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.trace(planes)
        up = min(up, time.perf_counter() - t0)
print(f"vct_trace from a host G-buffer ({planes.nbytes / 1e6:.1f} MB up, {W * H * 8 / 1e6:.1f} MB down; PCIe-inclusive, wall): "
      f"{up * 1e3:.3f} ms = {shaded * 7 / up / 1e6:.0f} Mcones/s, against {best[3]:.4f} ms resident")
