"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must reproduce them bit for bit (regression / compiler-drift guard).
GPU: the HIP path must reproduce them through the C ABI WITHOUT the oracle being involved."""
import glob
import os

import numpy as np
import pytest

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACE_CASES = sorted(glob.glob(os.path.join(GOLDEN, "trace_*.npz")))
REL_L2_TOL = 1e-3      # BASELINE.json north_star: within 1e-3 relative L2 (fp32)


def _params_kw(g):
    td, ts, amb, wrap = g["params"]
    return dict(tan_diffuse=float(td), tan_specular=float(ts), ambient_factor=float(amb),
                wrap_repeat=int(wrap))


def test_fixture_set_is_complete():
    assert len(TRACE_CASES) == 3 and os.path.exists(os.path.join(GOLDEN, "aniso_v16_8x8.npz"))
    assert os.path.exists(os.path.join(GOLDEN, "voxelize_v32.npz"))
    assert os.path.exists(os.path.join(GOLDEN, "raster_textured_48x32.npz"))


@pytest.mark.parametrize("path", TRACE_CASES, ids=os.path.basename)
def test_oracle_reproduces_golden_trace(oracle, path):
    g = np.load(path)
    V = int(g["V"])
    chain = oracle.build_mips(g["level0"])
    assert np.array_equal(chain, g["chain"])
    ref = oracle.trace(oracle.default_params(V, **_params_kw(g)), chain, g["planes"], nthreads=2,
                       want_cones=True)
    assert np.array_equal(ref["steps"], g["steps"])
    assert np.array_equal(ref["cones"].view(np.uint32), g["cones"].view(np.uint32))
    assert np.array_equal(ref["rgba32f"].view(np.uint32), g["rgba32f"].view(np.uint32))
    assert np.array_equal(ref["rgba16f"], g["rgba16f"])
    assert ref["total_steps"] == int(g["total_steps"])


def test_oracle_reproduces_golden_voxelization(oracle):
    g = np.load(os.path.join(GOLDEN, "voxelize_v32.npz"))
    V = int(g["V"])
    sc = oracle.make_scene(g["pos"], g["material"], g["albedo"])
    l0, acc = oracle.voxelize_conservative(oracle.default_params(V), sc, want_acc=True)
    assert np.array_equal(l0, g["level0"])
    assert np.array_equal(acc[..., 3], g["count"])
    assert np.array_equal(oracle.build_mips(l0), g["chain"])
    assert 0.01 < (l0[..., 3] > 0).mean() < 0.6
    l0b, alb, nrm = oracle.voxelize_conservative_attr(oracle.default_params(V), sc)
    assert np.array_equal(l0b, l0) and np.array_equal(alb, g["attr_albedo"]) and np.array_equal(nrm, g["attr_normal"])
    l1, steps = oracle.bounce(oracle.default_params(V), g["chain"], alb, nrm, nthreads=2)
    assert np.array_equal(l1, g["bounce_level0"]) and steps == int(g["bounce_steps"])
    assert (l1 != l0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("path", TRACE_CASES, ids=os.path.basename)
def test_hip_reproduces_golden_trace(path, variant):
    import vctpkg
    vct = vctpkg.load()
    g = np.load(path)
    V, w, h = int(g["V"]), int(g["w"]), int(g["h"])
    kw = _params_kw(g)
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, debug_outputs=1,
                                        trace_variant=variant, **kw)) as ctx:
        ctx.upload_volume(g["level0"])
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), g["chain"])
        out = ctx.trace(g["planes"])
        assert np.array_equal(ctx.steps(), g["steps"])
        assert np.array_equal(ctx.cones().view(np.uint32), g["cones"].view(np.uint32))
        assert ctx.last_step_count() == int(g["total_steps"])
        err = synth.rel_l2(vct.half_to_float(out.reshape(-1, 4)), g["rgba32f"])
        assert err <= REL_L2_TOL, err
        assert (out.reshape(-1, 4) == g["rgba16f"]).mean() > 0.995


@pytest.mark.gpu
def test_hip_reproduces_golden_voxelization():
    import vctpkg
    vct = vctpkg.load()
    g = np.load(os.path.join(GOLDEN, "voxelize_v32.npz"))
    with vct.Context(vct.default_config(voxel_dim=int(g["V"]), width=8, height=8,
                                        voxel_attributes=1)) as ctx:
        ctx.upload_triangles(g["pos"], g["material"], g["albedo"])
        ctx.voxelize()
        ctx.inject_light()
        ctx.build_mips()
        assert np.array_equal(ctx.download_chain(), g["chain"])
        alb, nrm = ctx.voxel_attributes()
        assert np.array_equal(alb, g["attr_albedo"]) and np.array_equal(nrm, g["attr_normal"])
        ctx.bounce()
        assert ctx.last_step_count() == int(g["bounce_steps"])
        assert np.array_equal(ctx.download_chain(), g["bounce_chain"])


def test_oracle_reproduces_golden_aniso(oracle):
    g = np.load(os.path.join(GOLDEN, "aniso_v16_8x8.npz"))
    an = oracle.build_mips_aniso(g["level0"])
    assert np.array_equal(an, g["aniso"])
    r = oracle.trace_aniso(oracle.default_params(16), oracle.build_mips(g["level0"]), an, g["planes"], want_cones=True)
    assert np.array_equal(r["steps"], g["steps"]) and np.array_equal(r["rgba16f"], g["rgba16f"])
    assert np.array_equal(r["cones"].view(np.uint32), g["cones"].view(np.uint32))


@pytest.mark.gpu
def test_hip_reproduces_golden_aniso():
    import vctpkg
    vct = vctpkg.load()
    g = np.load(os.path.join(GOLDEN, "aniso_v16_8x8.npz"))
    with vct.Context(vct.default_config(voxel_dim=16, width=8, height=8, debug_outputs=1,
                                        anisotropic_mips=1)) as ctx:
        ctx.upload_volume(g["level0"])
        ctx.build_mips()
        assert np.array_equal(ctx.download_aniso(), g["aniso"])
        out = ctx.trace(g["planes"])
        assert np.array_equal(ctx.steps(), g["steps"])
        assert np.array_equal(ctx.cones().view(np.uint32), g["cones"].view(np.uint32))
        assert (out.reshape(-1, 4) == g["rgba16f"]).mean() > 0.995


def _config1_inputs():
    import vctpkg
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    light, cam_pos = (0.0, 1.0, 0.25), (0.0, 0.0, 58.0)
    scene = sc.Scene(sc.CORNELL)
    return sc, scene, light, cam_pos


def test_config1_cornell_scalar_path(oracle):
    """BASELINE.json configs[0]: Cornell box, 64^3, 128x128 through the scalar CPU path (host
    rasterisers + oracle), pinned by the committed hashes."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    g = np.load(os.path.join(GOLDEN, "config1_cornell_v64_128.npz"))
    got = make_golden.config1_cornell()
    assert int(got["total_steps"]) == int(g["total_steps"])
    assert int(got["frame_fnv1a"]) == int(g["frame_fnv1a"]) and int(got["chain_fnv1a"]) == int(g["chain_fnv1a"])
    assert np.array_equal(got["steps_hist"], g["steps_hist"])
    assert np.allclose(got["image16"], g["image16"], atol=0) and 0.5 < float(g["covered"]) <= 1.0
    img = g["image16"]
    assert img[..., :3].max() < 4.0 and img[..., :3].min() >= 0.0         # composite stays bounded
    # colour bleeding: the part of the floor next to the red (left) wall is redder than the part next
    # to the green (right) wall
    floor = img[2:5]
    left, right = floor[:, 2:5].mean((0, 1)), floor[:, 11:14].mean((0, 1))
    assert left[0] - left[1] > right[0] - right[1]


@pytest.mark.gpu
def test_config1_cornell_on_the_gpu_matches_the_scalar_path():
    import sys
    import vctpkg
    sys.path.insert(0, GOLDEN)
    import make_golden
    vct = vctpkg.load()
    g = np.load(os.path.join(GOLDEN, "config1_cornell_v64_128.npz"))
    sc, scene, light, cam_pos = _config1_inputs()
    V, w, h, S = int(g["V"]), int(g["w"]), int(g["h"]), int(g["S"])
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=S)) as ctx:
        ctx.set_camera_position(cam_pos); ctx.set_light_direction(light)
        ctx.upload_triangles(scene.pos, scene.material, scene.albedo)
        ctx.upload_mesh_attributes(*scene.frames(), scene.specular)
        ctx.render_shadow_map(sc.light_view_proj(light))
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        assert make_golden.fnv1a(ctx.download_chain().view(np.uint16)) == int(g["chain_fnv1a"])
        ctx.render_gbuffer(sc.camera_view_proj(sc.default_camera(position=cam_pos), w, h))
        frame = ctx.trace_current()
        assert ctx.last_step_count() == int(g["total_steps"])
        img = vct.half_to_float(frame.reshape(-1, 4)).reshape(h, w, 4)
        small = img.reshape(16, 8, 16, 8, 4).mean((1, 3))
        assert synth.rel_l2(small, g["image16"]) <= REL_L2_TOL


def _textured_inputs(g):
    import vctpkg
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(sc.ATRIUM_TEXTURED, float(g["detail"]), int(g["seed"]))
    assert scene.ntri == int(g["ntri"])
    return sc, scene, (0.0, 1.0, 0.25), sc.default_camera(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0)


def test_oracle_reproduces_golden_textured_raster(oracle):
    """Shadow map + G-buffer of the procedurally textured atrium (alpha test, bump normal, specular map, PCF):
    the CPU rasterisers against the committed planes, bit for bit."""
    import raster_oracle
    g = np.load(os.path.join(GOLDEN, "raster_textured_48x32.npz"))
    sc, scene, light, cam = _textured_inputs(g)
    depth, lvp_row = raster_oracle.shadow_map(sc, scene, light, int(g["S"]))
    assert np.array_equal(depth.view(np.uint32), g["shadow"].view(np.uint32))
    planes = raster_oracle.gbuffer(sc, scene, cam, int(g["w"]), int(g["h"]), depth, lvp_row, mipmaps=False)
    assert np.array_equal(planes.view(np.uint32), g["planes"].view(np.uint32))     # the round-2 fixture: level-0 sampling
    assert 0.9 < float(g["covered"]) <= 1.0
    # round 3: the same scene with the reference's sampler state (mip chains, implicit derivatives)
    gm = np.load(os.path.join(GOLDEN, "raster_textured_mips_48x32.npz"))
    assert np.array_equal(gm["shadow"].view(np.uint32), g["shadow"].view(np.uint32))
    planes_m = raster_oracle.gbuffer(sc, scene, cam, int(g["w"]), int(g["h"]), depth, lvp_row, mipmaps=True)
    assert np.array_equal(planes_m.view(np.uint32), gm["planes"].view(np.uint32))
    assert not np.array_equal(planes_m[15:22], planes[15:22])                      # minified maps differ from level 0
    moved = (planes_m[:12].view(np.uint32) != planes[:12].view(np.uint32)).any(0).mean()
    assert moved < 0.05          # geometry only changes where the mip-mapped alpha test cuts a different silhouette


@pytest.mark.gpu
def test_hip_reproduces_golden_textured_raster():
    import vctpkg
    vct = vctpkg.load()
    g = np.load(os.path.join(GOLDEN, "raster_textured_48x32.npz"))
    sc, scene, light, cam = _textured_inputs(g)
    w, h, S = int(g["w"]), int(g["h"]), int(g["S"])
    for mips, fixture in ((0, g), (1, np.load(os.path.join(GOLDEN, "raster_textured_mips_48x32.npz")))):
        with vct.Context(vct.default_config(voxel_dim=16, width=w, height=h, shadow_map_size=S, texture_mipmaps=mips)) as ctx:
            ctx.upload_scene(scene)
            ctx.render_shadow_map(sc.light_view_proj(light))
            assert np.array_equal(ctx.download_shadow_map().view(np.uint32), fixture["shadow"].view(np.uint32))
            ctx.render_gbuffer(sc.camera_view_proj(cam, w, h))
            assert np.array_equal(ctx.download_gbuffer().view(np.uint32), fixture["planes"].view(np.uint32))
