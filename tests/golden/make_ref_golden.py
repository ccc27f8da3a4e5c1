#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz: outputs of the REFERENCE'S OWN GLSL, run unmodified.

Build container only.  The seven shader files are read at run time from
/root/reference/Voxel_Cone_Tracing_Final/Shader by oracle/ref_gl.c (oracle/_ref/libvct_refgl.so) and executed by Mesa
llvmpipe (OpenGL 4.5 core); nothing of the reference is stored here -- the fixtures hold inputs (seeded synthetic
G-buffers, volumes, a small textured scene) and the numbers the reference's shaders produced for them.

    python tests/golden/make_ref_golden.py            # rewrites every ref_*.npz
    python tests/golden/make_ref_golden.py --check    # regenerates in memory and compares with the committed files

llvmpipe reads GALLIVM_PERF when the driver is loaded, so each precision setting runs in its own process:
`--worker precise|default <case> <out.npz>` is that process (spawned by this script).
  precise  GALLIVM_PERF=no_aos_sampling,no_rho_approx,no_brilinear,no_quad_lod -- float filter weights, exact rho
  default  llvmpipe's defaults (8-bit fixed-point filter weights ...) -- recorded to show what "real GL" precision
           variance looks like next to the 1e-3 bar of BASELINE.json's north_star

Cases: TRACE_CASES (small point traces, everything stored), PIPES (the whole pipeline on the small textured scene),
ref_mips3d, FULL_CASES (point traces at BASELINE configs[1]'s / configs[2]'s sizes: inputs regenerate from seeds under
stored checksums, of the frame 65,536 sample pixels and the block means are kept), PIPES_HIRES (Render() at the
reference's own 1280 x 720 on the stages of ref_pipeline_v128: coverage mask, samples, block means).  1024^3 does not
run here (llvmpipe refuses a 4 GiB 3-D texture).

The oracle is used here for ONE thing: the shadow term of the point fixtures' G-buffers (plane 22 = PCF x 0.111 of the
stored shadow map at the stored light matrix), so that the GPU test can feed vct_trace a complete G-buffer without the
oracle; tests/test_ref_gl.py re-derives it.  Every `ref_*` array in the fixtures comes from GL.
"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import refscene                   # noqa: E402
import synth                      # noqa: E402

LIGHT = (0.0, 1.0, 0.25)          # VCT.h:14
G = 150.0                         # VCT.h:17

# ---- the point-trace cases: (name, V, W, H, volume seed, occupancy, G-buffer kind, G-buffer seed, clamp) ------------
TRACE_CASES = {
    "ref_trace_v32_random": dict(V=32, W=32, H=32, vol_seed=7, occ=0.05, gb="random", gb_seed=42, clamp=0),
    "ref_trace_v64_coherent": dict(V=64, W=48, H=32, vol_seed=3, occ=0.12, gb="coherent", gb_seed=3, clamp=0),
    "ref_trace_v256_random": dict(V=256, W=32, H=32, vol_seed=7, occ=0.05, gb="random", gb_seed=5, clamp=0),
    "ref_trace_v32_clamp": dict(V=32, W=24, H=16, vol_seed=12, occ=0.2, gb="random", gb_seed=7, clamp=1),
}
# BASELINE.json configs[1]'s size through the reference's GLSL: 256^3 chain, 1920 x 1080 coherent G-buffer.  The frame is
# 33 MB of fp32 -- the fixture keeps 65,536 seeded sample pixels exactly and the frame averaged over 8 x 8 blocks; the
# inputs are regenerated from their seeds (checksums stored).
FULL_CASES = {
    "ref_trace_c2_1080p": dict(V=256, W=1920, H=1080, vol_seed=7, occ=0.05, gb="coherent", gb_seed=3, clamp=0,
                               samples=65536, sample_seed=99, block=8),
    # configs[2]'s size (the screen trace; its second bounce has no reference code): 512^3 chain (10 levels), 3840 x 2160
    "ref_trace_c3_4k": dict(V=512, W=3840, H=2160, vol_seed=11, occ=0.05, gb="coherent", gb_seed=5, clamp=0,
                            samples=65536, sample_seed=101, block=16),
}
SHADOW_S = 64


def trace_inputs(c):
    """Seeded inputs of a point-trace case (pure numpy: the same on every host)."""
    l0 = synth.noise_volume(c["V"], seed=c["vol_seed"], occupancy=c["occ"])
    n = c["W"] * c["H"]
    if c["gb"] == "random":
        planes = synth.random_gbuffer(n, seed=c["gb_seed"], discard_frac=0.06)
    else:
        planes = synth.coherent_gbuffer(c["W"], c["H"], seed=c["gb_seed"])
        r = np.random.default_rng(c["gb_seed"])
        planes[18, r.uniform(size=n) < 0.04] = 0.0
    r = np.random.default_rng(1000 + c["gb_seed"])
    # a smooth shadow map with both lit and occluded regions along the light direction (24-bit quantised like the
    # DEPTH_COMPONENT24 texture it is uploaded to, VCT.h:90)
    ys, xs = np.meshgrid(np.arange(SHADOW_S), np.arange(SHADOW_S), indexing="ij")
    d = 0.5 + 0.22 * np.sin(xs * 0.21 + 0.3) * np.cos(ys * 0.17) + r.uniform(-0.02, 0.02, (SHADOW_S, SHADOW_S))
    depth = (np.floor(np.clip(d, 0, 1) * 16777215.0 + 0.5) / 16777215.0).astype(np.float32)
    cam = np.array([3.0, 4.0, -2.0], np.float32)
    return l0, planes, depth, cam


def shadow_plane(planes, depth):
    """Plane 22 (shadow_value, trace.fs:186) of a shadow map for the G-buffer's positions -- the one oracle-derived input."""
    from oracle import pyoracle, pyrefgl as rg
    dvp = rg.depth_view_proj(LIGHT)
    P = planes[0:3].T
    M = dvp.reshape(4, 4).T
    clip = P @ M[:3, :3].T.astype(np.float32) + M[:3, 3]
    coord = (clip * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
    return pyoracle.pcf25_batch(depth, coord).astype(np.float32) * np.float32(0.111)


def full_inputs(c):
    """Inputs of a FULL_CASES case, the same arrays on every host (numpy only + the oracle's PCF for plane 22)."""
    l0, planes, depth, cam = trace_inputs(c)
    planes[22] = shadow_plane(planes, depth)
    return l0, planes, depth, cam


def block_mean(frame, W, H, b=8):
    """[H*W, 4] fp32 -> [H/b, W/b, 4]: mean over b x b pixel blocks in float64, stored fp32."""
    f = frame.reshape(H // b, b, W // b, b, 4).astype(np.float64)
    return f.mean(axis=(1, 3)).astype(np.float32)


def gb_to_vertices(planes):
    n = planes.shape[1]
    v = np.zeros((n, 14), np.float32)
    v[:, 0:3], v[:, 3:6], v[:, 8:11], v[:, 11:14] = planes[0:3].T, planes[3:6].T, planes[6:9].T, planes[9:12].T
    return v


def gl_trace_points(c):
    """The reference's VoxelConeTracing.vs + .fs on one GL_POINT per G-buffer pixel (oracle/ref_gl.c)."""
    from oracle import pyoracle, pyrefgl as rg
    l0, planes, depth, cam = trace_inputs(c)      # (plane 22 does not reach GL: the shader computes it from the shadow map)
    V, W, H = c["V"], c["W"], c["H"]
    chain = pyoracle.build_mips(l0)        # every level uploaded: isolates the sampler from glGenerateMipmap
    rg.volume_upload_chain(V, [pyoracle.level_view(chain, V, k) for k in range(pyoracle.num_levels(V))])
    rg.volume_set_wrap(c["clamp"])
    rg.shadow_create(SHADOW_S)
    rg.shadow_set(depth)
    fp = rg.frame_params(V, G=G, camera_pos=cam, light_dir=LIGHT)
    alb = np.ascontiguousarray(planes[15:19].T.reshape(H, W, 4))
    spec = np.concatenate([planes[19:22].T, np.ones((W * H, 1), np.float32)], 1).reshape(H, W, 4)
    hgt = np.zeros((H, W, 4), np.float32)                       # flat height map: bump normal = geometric normal
    ta, ts, th = (rg.texture_create_f32(a) for a in (alb, spec, hgt))
    return rg.trace_points(W, H, fp, gb_to_vertices(planes), ta, ts, th).reshape(-1, 4)


# ---- the pipeline case: DrawDepthTexture -> DrawVoxelTexture (+ glGenerateMipmap) -> Render ---------------------------
PIPES = {
    "ref_pipeline_v32": dict(V=32, S=256, W=96, H=64, eye=(10.0, -5.0, 52.0), center=(-5.0, -25.0, 0.0), fov_deg=45.0,
                             scene_seed=11),
    # a finer grid, a larger map, another camera (from the left, looking down along the cut-out sheet), other textures
    "ref_pipeline_v64": dict(V=64, S=512, W=128, H=80, eye=(-48.0, 8.0, 30.0), center=(10.0, -35.0, -20.0), fov_deg=45.0,
                             scene_seed=23),
    # the reference's own grid size (VCT.h:16), 16:9 like its 1280 x 720 window, another light, and AmbientFactor >= 0.5:
    # the white clear colour branch of VCT.h:156-159 and a strong ambient term (trace.fs:225)
    "ref_pipeline_v128": dict(V=128, S=1024, W=160, H=90, eye=(30.0, 20.0, 55.0), center=(-10.0, -30.0, -10.0), fov_deg=45.0,
                              scene_seed=37, light=(0.35, 1.0, -0.2), ambient=0.6),
}
PIPE = PIPES["ref_pipeline_v32"]
# The reference's OWN configuration -- 128^3 (VCT.h:16), a 1280 x 720 window (main.cpp) -- through Render(): the scene,
# light, shadow map and voxel chain of ref_pipeline_v128 (same 16:9 camera), of whose 921,600-pixel frame the fixture
# keeps the coverage mask, 65,536 sample pixels exactly and the 8 x 8 block means.
PIPES_HIRES = {
    "ref_pipeline_v128_720p": dict(base="ref_pipeline_v128", W=1280, H=720, samples=65536, sample_seed=7, block=8),
}


def pipeline_matrices(c=None):
    from oracle import pyrefgl as rg
    c = c or PIPE
    return dict(model=rg.scale(0.05), depth_vp=rg.depth_view_proj(c.get("light", LIGHT)),
                view=rg.look_at(c["eye"], c["center"], (0, 1, 0)),
                proj=rg.perspective(np.deg2rad(c["fov_deg"]), c["W"] / c["H"], 0.1, 1000.0))


def gl_pipeline(c=None):
    from oracle import pyrefgl as rg
    c = c or PIPE
    sc = refscene.build(c["scene_seed"])
    m = pipeline_matrices(c)
    tex = [rg.texture_create(t) for t in sc["textures"]]
    tex_chains = []
    for h, t in zip(tex, sc["textures"]):                      # the 2-D chains glGenerateMipmap made (Model.h:169)
        lv, k = [t.reshape(-1, 4)], 1
        while True:
            g = rg.texture_get_level(h, k)
            if g is None:
                break
            lv.append(g.reshape(-1, 4))
            k += 1
        tex_chains.append(np.concatenate(lv))
    meshes = []
    for mat in range(len(sc["mat_tex"])):
        v = refscene.gl_vertices(sc, mat)
        d, s, h = sc["mat_tex"][mat]
        meshes.append(rg.mesh_create(v, np.arange(len(v), dtype=np.uint32),
                                     [(tex[d], rg.TEX_DIFFUSE), (tex[s], rg.TEX_SPECULAR), (tex[h], rg.TEX_HEIGHT)]))
    depth_mvp = rg.mul(m["depth_vp"], m["model"])
    rg.shadow_create(c["S"])
    rg.draw_depth_texture(depth_mvp, meshes)
    shadow = rg.shadow_get(c["S"])
    rg.volume_create(c["V"])
    rg.draw_voxel_texture(G, m["model"], depth_mvp, meshes, generate_mipmap=True)
    nlev = int(np.log2(c["V"])) + 1
    levels = [rg.volume_get_level(c["V"], k) for k in range(nlev)]
    fp = rg.frame_params(c["V"], G=G, camera_pos=c["eye"], light_dir=c.get("light", LIGHT), ambient=c.get("ambient", 0.1),
                         model=m["model"], view=m["view"], projection=m["proj"], depth_vp=m["depth_vp"])
    frame, zbuf = rg.render(c["W"], c["H"], fp, meshes, want_depth=True)
    out = dict(ref_shadow=shadow, ref_frame=frame, ref_zbuf=zbuf,
               ref_chain=np.concatenate([l.reshape(-1, 4) for l in levels]))
    for i, ch in enumerate(tex_chains):
        out[f"ref_tex_chain_{i}"] = ch
    return out


def gl_mips3d():
    """glGenerateMipmap(GL_TEXTURE_3D) (VCT.h:248) on seeded volumes: the chain Mesa builds."""
    from oracle import pyrefgl as rg
    out = {}
    for V, seed, occ in ((32, 7, 0.3), (64, 9, 0.1)):
        l0 = synth.noise_volume(V, seed=seed, occupancy=occ)
        rg.volume_create(V)
        rg.volume_set_level(0, l0)
        rg.volume_generate_mipmap()
        out[f"ref_chain_v{V}"] = np.concatenate([rg.volume_get_level(V, k).reshape(-1, 4)
                                                 for k in range(int(np.log2(V)) + 1)])
        out[f"args_v{V}"] = np.array([V, seed, occ], np.float64)
    return out


def worker(mode, case, out_path):
    from oracle import pyrefgl as rg
    rg.lib(precise=(mode == "precise"))
    if case in TRACE_CASES:
        res = dict(rgba=gl_trace_points(TRACE_CASES[case]))
    elif case in FULL_CASES:
        res = dict(rgba=gl_trace_points(FULL_CASES[case]))
    elif case in PIPES:
        res = gl_pipeline(PIPES[case])
    elif case in PIPES_HIRES:
        h = PIPES_HIRES[case]
        full = gl_pipeline(dict(PIPES[h["base"]], W=h["W"], H=h["H"]))
        res = dict(ref_frame=full["ref_frame"], ref_shadow=full["ref_shadow"], ref_chain=full["ref_chain"])
    elif case == "ref_mips3d":
        res = gl_mips3d()
    else:
        raise SystemExit("unknown case " + case)
    s = rg.gl_strings()
    res["gl"] = np.array([s["version"], s["renderer"], s["glsl"], os.environ.get("GALLIVM_PERF", "")])
    np.savez(out_path, **res)


def run_worker(mode, case):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", mode, case, out])
        with np.load(out) as z:
            return {k: z[k] for k in z.files}


def build_fixture(name):
    from oracle import pyoracle
    if name in TRACE_CASES:
        c = TRACE_CASES[name]
        l0, planes, depth, cam = trace_inputs(c)
        from oracle import pyrefgl as rg
        dvp = rg.depth_view_proj(LIGHT)
        # plane 22 (shadow_value, trace.fs:186) of the stored shadow map -- the one oracle-derived input (see docstring)
        P = planes[0:3].T
        M = dvp.reshape(4, 4).T
        clip = P @ M[:3, :3].T.astype(np.float32) + M[:3, 3]
        coord = (clip * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
        planes[22] = [pyoracle.pcf25(depth, coord[i]) * np.float32(0.111) for i in range(P.shape[0])]
        precise = run_worker("precise", name)
        default = run_worker("default", name)
        f = dict(V=c["V"], W=c["W"], H=c["H"], clamp=c["clamp"], planes=planes, shadow_map=depth, camera_pos=cam,
                 light_dir=np.array(LIGHT, np.float32), depth_vp=dvp, ref_rgba=precise["rgba"],
                 ref_rgba_default_precision=default["rgba"], gl=precise["gl"])
        if c["V"] <= 64:
            f["level0"] = l0
        else:       # 64 MiB of texels: the seeded generator (tests/synth.py) + a checksum stand in for the array
            f["level0_args"] = np.array([c["V"], c["vol_seed"], c["occ"]], np.float64)
            f["level0_crc32"] = np.uint32(zlib.crc32(l0.tobytes()))
        return f
    if name in FULL_CASES:
        c = FULL_CASES[name]
        l0, planes, depth, cam = full_inputs(c)
        from oracle import pyrefgl as rg
        pw = run_worker("precise", name)
        precise, default = pw["rgba"], run_worker("default", name)["rgba"]
        idx = np.sort(np.random.default_rng(c["sample_seed"]).choice(c["W"] * c["H"], c["samples"], replace=False))
        return dict(V=c["V"], W=c["W"], H=c["H"], clamp=c["clamp"], shadow_map=depth, camera_pos=cam,
                    light_dir=np.array(LIGHT, np.float32), depth_vp=rg.depth_view_proj(LIGHT),
                    case=np.array([c["vol_seed"], c["occ"], c["gb_seed"], c["samples"], c["sample_seed"], c["block"]], np.float64),
                    level0_crc32=np.uint32(zlib.crc32(l0.tobytes())), planes_crc32=np.uint32(zlib.crc32(planes.tobytes())),
                    sample_idx=idx.astype(np.int64), ref_sample=precise[idx], ref_sample_default_precision=default[idx],
                    ref_block_mean=block_mean(precise, c["W"], c["H"], c["block"]),
                    discards=np.int64((planes[18] < 0.5).sum()), gl=pw["gl"])
    if name in PIPES_HIRES:
        h = PIPES_HIRES[name]
        r = run_worker("precise", name)
        with np.load(os.path.join(HERE, h["base"] + ".npz")) as z:       # the stages in front of Render are the base fixture's
            assert np.array_equal(z["ref_shadow"], r["ref_shadow"]) and np.array_equal(z["ref_chain"], r["ref_chain"])
        W, H = h["W"], h["H"]
        frame = r["ref_frame"].reshape(-1, 4)
        amb = PIPES[h["base"]].get("ambient", 0.1)
        clear = np.array([1.0, 1.0, 1.0, 1.0] if amb >= 0.5 else [0.5, 0.5, 0.5, 1.0], np.float32)
        cov = ~np.all(frame == clear, axis=1)
        idx = np.sort(np.random.default_rng(h["sample_seed"]).choice(W * H, h["samples"], replace=False))
        return dict(base=np.array(h["base"]), W=W, H=H, block=h["block"], coverage_bits=np.packbits(cov), sample_idx=idx.astype(np.int64),
                    ref_sample=frame[idx], ref_block_mean=block_mean(frame, W, H, h["block"]), gl=r["gl"])
    if name in PIPES:
        sc = refscene.build(PIPES[name]["scene_seed"])
        m = pipeline_matrices(PIPES[name])
        f = run_worker("precise", name)
        # how many triangles store into each voxel in the reference's voxelization (geometry only; the oracle's raster, one
        # triangle at a time): where it is > 1 the reference's result depends on an order GL does not define
        V = PIPES[name]["V"]
        count = np.zeros((V, V, V), np.uint8)
        p = pyoracle.default_params(V)
        for t in range(len(sc["pos"])):
            one = pyoracle.make_scene(sc["pos"][t:t + 1], np.zeros(1, np.int32), sc["albedo"][:1])
            count += (pyoracle.voxelize_reference(p, one)[..., 3] > 0).astype(np.uint8)
        f["writers"] = count
        f.update({k: np.asarray(v) for k, v in PIPES[name].items()})
        f.update(light_dir=np.array(PIPES[name].get("light", LIGHT), np.float32),
                 ambient=np.float32(PIPES[name].get("ambient", 0.1)), **m)
        for k in ("pos", "uv", "material", "nrm", "tan", "bit", "albedo", "specular", "mat_tex"):
            f[k] = sc[k]
        for i, t in enumerate(sc["textures"]):
            f[f"texture_{i}"] = t
        return f
    if name == "ref_mips3d":
        return run_worker("precise", name)
    raise KeyError(name)


ALL = list(TRACE_CASES) + list(PIPES) + ["ref_mips3d"] + list(FULL_CASES) + list(PIPES_HIRES)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], sys.argv[3], sys.argv[4])
    check = "--check" in sys.argv
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ALL
    bad = 0
    for name in names:
        f = build_fixture(name)
        path = os.path.join(HERE, name + ".npz")
        if check:
            with np.load(path) as z:
                for k in f:
                    same = np.array_equal(np.asarray(f[k]), z[k])
                    if not same:
                        bad += 1
                        print("DIFFERS", name, k)
            print("checked", name)
        else:
            np.savez_compressed(path, **f)
            print(name + ".npz", os.path.getsize(path), "bytes")
    if bad:
        raise SystemExit(f"{bad} arrays differ from the committed fixtures")


if __name__ == "__main__":
    main()
