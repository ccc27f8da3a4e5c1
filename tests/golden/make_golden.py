#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

These vectors are produced by the build's own scalar oracle (oracle/vct_oracle.cpp) on seeded inputs.  They pin
(1) the oracle against regressions / compiler drift and (2) the HIP path against fixed expected outputs without the
oracle library being present at run time.  The vectors produced by the REFERENCE'S OWN shaders (which pin the oracle
itself) are the ref_*.npz files made by make_ref_golden.py next to this script.

    python tests/golden/make_golden.py        # rewrites the .npz files
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import synth                      # noqa: E402
from oracle import pyoracle       # noqa: E402


def small_scene(seed=5, ntri=60):
    r = np.random.default_rng(seed)
    c = r.uniform(-1200, 1200, (ntri, 1, 3))
    pos = (c + r.normal(scale=120.0, size=(ntri, 3, 3))).astype(np.float32)
    pos[0] = [[-1400, -1000, -1400], [1400, -1000, -1400], [1400, -1000, 1400]]    # a floor triangle
    mat = r.integers(0, 3, ntri).astype(np.int32)
    alb = np.array([[0.8, 0.2, 0.2, 1], [0.2, 0.8, 0.2, 1], [0.6, 0.6, 0.9, 1]], np.float32)
    return pos, mat, alb


def trace_case(V, w, h, vol_seed, gb_seed, occupancy, coherent, **params):
    l0 = synth.noise_volume(V, seed=vol_seed, occupancy=occupancy)
    chain = pyoracle.build_mips(l0)
    planes = synth.coherent_gbuffer(w, h, seed=gb_seed) if coherent else \
        synth.random_gbuffer(w * h, seed=gb_seed, discard_frac=0.1)
    p = pyoracle.default_params(V, **params)
    ref = pyoracle.trace(p, chain, planes, nthreads=1, want_cones=True)
    return dict(V=V, w=w, h=h, level0=l0, planes=planes, chain=chain, rgba32f=ref["rgba32f"],
                rgba16f=ref["rgba16f"], steps=ref["steps"], cones=ref["cones"],
                total_steps=np.int64(ref["total_steps"]),
                params=np.array([params.get("tan_diffuse", 0.577), params.get("tan_specular", 0.07),
                                 params.get("ambient_factor", 0.1), params.get("wrap_repeat", 1)],
                                np.float32))


def fnv1a(u16):
    h = 1469598103934665603
    for v in np.asarray(u16, np.uint16).reshape(-1).tolist():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def config1_cornell():
    """Procedural Cornell box -> CPU shadow map + G-buffer raster (oracle/vct_oracle_raster.cpp) -> oracle
    voxelization (conservative, shadowed) -> oracle mips -> oracle trace.  Small outputs only."""
    import vctpkg
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    V, w, h, S = 64, 128, 128, 512
    light, cam_pos = (0.0, 1.0, 0.25), (0.0, 0.0, 58.0)
    import raster_oracle
    scene = sc.Scene(sc.CORNELL)
    depth, lvp_row = raster_oracle.shadow_map(sc, scene, light, S)
    planes = raster_oracle.gbuffer(sc, scene, sc.default_camera(position=cam_pos), w, h, depth, lvp_row)
    p = pyoracle.default_params(V, camera_pos=cam_pos, light_dir=light)
    l0 = pyoracle.voxelize_conservative(p, pyoracle.make_scene(scene.pos, scene.material, scene.albedo,
                                                               shadow_depth=depth, light_vp=lvp_row))
    chain = pyoracle.build_mips(l0)
    r = pyoracle.trace(p, chain, planes, nthreads=8)
    img = r["rgba32f"].reshape(h, w, 4)
    small = img.reshape(16, 8, 16, 8, 4).mean((1, 3)).astype(np.float32)
    return dict(V=V, w=w, h=h, S=S, total_steps=np.int64(r["total_steps"]),
                frame_fnv1a=np.uint64(fnv1a(r["rgba16f"])), chain_fnv1a=np.uint64(fnv1a(chain.view(np.uint16))),
                covered=np.float32((planes[18] >= 0.5).mean()), image16=small,
                steps_hist=np.bincount(r["steps"].reshape(-1), minlength=32).astype(np.int64))


def textured_raster(mipmaps=False):
    """The procedurally TEXTURED atrium (diffuse / specular / height maps, alpha cut-outs) through the CPU raster
    stages: shadow map + G-buffer with matColor / alpha test / CalcBumpNormal / specColor (.rrra) / PCF.
    mipmaps: textures sampled mip-mapped with implicit derivatives (round 3) or level 0 only (the round-2 fixture)."""
    import vctpkg
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    import raster_oracle
    w, h, S = 48, 32, 64
    light = (0.0, 1.0, 0.25)
    scene = sc.Scene(sc.ATRIUM_TEXTURED, 0.1, 7)
    cam = sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)
    depth, lvp_row = raster_oracle.shadow_map(sc, scene, light, S)
    planes = raster_oracle.gbuffer(sc, scene, cam, w, h, depth, lvp_row, mipmaps=mipmaps)
    return dict(w=w, h=h, S=S, detail=np.float32(0.1), seed=np.int32(7), shadow=depth, planes=planes,
                ntri=np.int32(scene.ntri), covered=np.float32((planes[18] >= 0.5).mean()))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "textured":      # only the fixture added in round 2
        np.savez_compressed(os.path.join(HERE, "raster_textured_48x32.npz"), **textured_raster())
        print("raster_textured_48x32.npz", os.path.getsize(os.path.join(HERE, "raster_textured_48x32.npz")), "bytes")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "textured-mips":      # only the fixture added in round 3
        np.savez_compressed(os.path.join(HERE, "raster_textured_mips_48x32.npz"), **textured_raster(mipmaps=True))
        print("raster_textured_mips_48x32.npz", os.path.getsize(os.path.join(HERE, "raster_textured_mips_48x32.npz")), "bytes")
        return
    # 1. SURVEY.md 8c suggestion: 16^3 volume + 8x8 G-buffer -> 64 RGBA fp32 + 64x7 step counts
    np.savez_compressed(os.path.join(HERE, "trace_v16_8x8_random.npz"),
                        **trace_case(16, 8, 8, 11, 42, 0.15, False))
    # 2. a screen-coherent tile set (cooperative sampler path), ragged frame
    np.savez_compressed(os.path.join(HERE, "trace_v32_20x12_coherent.npz"),
                        **trace_case(32, 20, 12, 3, 3, 0.25, True))
    # 3. clamp-to-edge wrap mode + a wider specular aperture (config 5 roughness sweep)
    np.savez_compressed(os.path.join(HERE, "trace_v16_8x8_clamp_glossy.npz"),
                        **trace_case(16, 8, 8, 12, 7, 0.2, False, wrap_repeat=0, tan_specular=0.2))
    # 3b. anisotropic option: directional chains + the trace through them
    c = trace_case(16, 8, 8, 13, 5, 0.2, True)
    an = pyoracle.build_mips_aniso(c["level0"])
    ra = pyoracle.trace_aniso(pyoracle.default_params(16), c["chain"], an, c["planes"], want_cones=True)
    np.savez_compressed(os.path.join(HERE, "aniso_v16_8x8.npz"), V=16, w=8, h=8, level0=c["level0"],
                        planes=c["planes"], aniso=an, rgba16f=ra["rgba16f"], steps=ra["steps"], cones=ra["cones"])
    # 4. voxelization (conservative + integer average) + mip chain of a small triangle soup
    V = 32
    pos, mat, alb = small_scene()
    p = pyoracle.default_params(V)
    sc = pyoracle.make_scene(pos, mat, alb)
    l0, acc = pyoracle.voxelize_conservative(p, sc, want_acc=True)
    chain0 = pyoracle.build_mips(l0)
    # ... and the second bounce on top of it (BASELINE.json config 3; oracle/vct_oracle.h vcto_bounce)
    l0b, attr_alb, attr_nrm = pyoracle.voxelize_conservative_attr(p, sc)
    assert np.array_equal(l0b, l0)
    l1, bounce_steps = pyoracle.bounce(p, chain0, attr_alb, attr_nrm)
    np.savez_compressed(os.path.join(HERE, "voxelize_v32.npz"), V=V, pos=pos, material=mat, albedo=alb,
                        level0=l0, count=acc[..., 3].astype(np.uint16), chain=chain0,
                        attr_albedo=attr_alb, attr_normal=attr_nrm, bounce_level0=l1,
                        bounce_steps=np.int64(bounce_steps), bounce_chain=pyoracle.build_mips(l1))
    # 5. BASELINE.json configs[0]: Cornell box, 64^3, 128x128, the scalar CPU path end to end
    np.savez_compressed(os.path.join(HERE, "config1_cornell_v64_128.npz"), **config1_cornell())
    # 6. the textured raster stages (SURVEY.md 8 f1 with materials)
    np.savez_compressed(os.path.join(HERE, "raster_textured_48x32.npz"), **textured_raster())
    np.savez_compressed(os.path.join(HERE, "raster_textured_mips_48x32.npz"), **textured_raster(mipmaps=True))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
