"""BUILD CONTAINER ONLY: the reference's own Render() (VCT.h:146-190, its unmodified VoxelConeTracing.vs/.fs) timed on
Mesa llvmpipe -- the reference's CPU path -- at the reference's own size (128^3 grid, VCT.h:16; 1280 x 720 window,
main.cpp), on the small textured test scene (tests/refscene.py), with the oracle's whole-frame trace on the same
machine beside it.  Usage: python tools/time_ref_gl.py [threads] [W H]      Output -> profiles/r05_ref_gl_timing.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
W = int(sys.argv[2]) if len(sys.argv) > 3 else 1280
H = int(sys.argv[3]) if len(sys.argv) > 3 else 720
from oracle import pyrefgl as rg  # noqa: E402
rg.lib(precise=False, win=max(W, H, 1024), threads=threads)      # llvmpipe as shipped: its default filtering paths
import make_ref_golden as mg  # noqa: E402
import refscene  # noqa: E402

c = dict(mg.PIPES["ref_pipeline_v128"], W=W, H=H)
sc = refscene.build(c["scene_seed"])
m = mg.pipeline_matrices(c)
tex = [rg.texture_create(t) for t in sc["textures"]]
meshes = []
for mat in range(len(sc["mat_tex"])):
    v = refscene.gl_vertices(sc, mat)
    d, s, h = sc["mat_tex"][mat]
    meshes.append(rg.mesh_create(v, np.arange(len(v), dtype=np.uint32),
                                 [(tex[d], rg.TEX_DIFFUSE), (tex[s], rg.TEX_SPECULAR), (tex[h], rg.TEX_HEIGHT)]))
depth_mvp = rg.mul(m["depth_vp"], m["model"])
rg.shadow_create(c["S"])
t0 = time.perf_counter(); rg.draw_depth_texture(depth_mvp, meshes); t_shadow = time.perf_counter() - t0
rg.volume_create(c["V"])
t0 = time.perf_counter(); rg.draw_voxel_texture(mg.G, m["model"], depth_mvp, meshes, generate_mipmap=True); t_vox = time.perf_counter() - t0
fp = rg.frame_params(c["V"], G=mg.G, camera_pos=c["eye"], light_dir=c.get("light", mg.LIGHT), ambient=c.get("ambient", 0.1),
                     model=m["model"], view=m["view"], projection=m["proj"], depth_vp=m["depth_vp"])
times = []
for _ in range(3):
    t0 = time.perf_counter()
    frame = rg.render(W, H, fp, meshes)        # glFinish + read-back of the RGBA32F frame inside
    times.append(time.perf_counter() - t0)
clear = np.array([1.0, 1.0, 1.0, 1.0], np.float32) if c.get("ambient", 0.1) >= 0.5 else np.array([0.5, 0.5, 0.5, 1.0], np.float32)
shaded = int((~np.all(frame.reshape(-1, 4) == clear, axis=1)).sum())
best = min(times)
s = rg.gl_strings()
print(f"{s['renderer']} / {s['version']}, LP_NUM_THREADS={threads}, host cores {os.cpu_count()}")
print(f"scene: tests/refscene.py seed {c['scene_seed']} ({sum(len(refscene.gl_vertices(sc, k)) for k in range(len(sc['mat_tex']))) // 3} triangles), "
      f"V={c['V']}, shadow map {c['S']}^2, frame {W}x{H}, {shaded} shaded pixels")
print(f"reference DrawDepthTexture {t_shadow * 1e3:.1f} ms, DrawVoxelTexture + glGenerateMipmap {t_vox * 1e3:.1f} ms")
print(f"reference Render(): {best * 1e3:.1f} ms best of 3 ({', '.join(f'{t * 1e3:.0f}' for t in times)}) "
      f"= {shaded * 7 / best / 1e6:.2f} Mcones/s over shaded pixels (raster + 7-cone trace + composite in one pass)")
