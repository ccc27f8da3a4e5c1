"""The tile-binned visibility stage (csrc/vct_raster.hip k_bin_*, round 4) against the direct one: the same shadow-map
words and the same G-buffer bit for bit, on every scene class -- Cornell walls (records of the huge list), the atrium
(opaque, small triangles), the textured atrium and the Bistro-class street (alpha-tested foliage: sort, alpha queue,
quad derivatives by DPP), ragged frame sizes, scissored tile rows, several passes on one context (nothing is cleared
between passes), and the automatic choice between the two forms.  The direct form itself is checked against the CPU
rasteriser in test_gpu_raster.py / test_gpu_textures.py / test_gpu_configs.py."""
import os

import numpy as np
import pytest

import vctpkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vct():
    import torch
    assert torch.cuda.is_available()
    return vctpkg.load()


def render(vct, path, scene, w, h, S, cams, rows=None, mips=1):
    """G-buffers (one per camera, same context) + the shadow map, with VCT_RASTER_PATH = path (None: automatic)."""
    from voxel_cone_tracing_amd import scene as sc
    old = os.environ.pop("VCT_RASTER_PATH", None)
    if path:
        os.environ["VCT_RASTER_PATH"] = path
    try:
        ctx = vct.Context(vct.default_config(voxel_dim=32, width=w, height=h, shadow_map_size=S, texture_mipmaps=mips))
    finally:
        os.environ.pop("VCT_RASTER_PATH", None)
        if old is not None:
            os.environ["VCT_RASTER_PATH"] = old
    ctx.upload_scene(scene)
    light = (0.0, 1.0, 0.25)
    ctx.render_shadow_map(sc.light_view_proj(light))
    out = [ctx.download_shadow_map().copy()]
    for cam in cams:
        vp = sc.camera_view_proj(cam, w, h)
        if rows:
            ctx.render_gbuffer_rows(vp, *rows)
        else:
            ctx.render_gbuffer(vp)
        out.append(ctx.download_gbuffer().copy())
    ctx.close()
    return out


CASES = [
    # kind, detail, w, h, S, cameras
    (0, 1.0, 200, 120, 256, [dict(position=(0.0, 0.0, 58.0)), dict(position=(0.0, 0.0, 20.0), yaw=-60.0, pitch=-20.0)]),
    (1, 0.15, 333, 187, 512, [dict(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0), dict(position=(-20.0, 5.0, 10.0), yaw=30.0, pitch=-10.0)]),
    (2, 0.15, 320, 200, 512, [dict(position=(-56.0, -9.0, 2.0), yaw=0.0, pitch=8.0), dict(position=(-30.0, -5.0, -8.0), yaw=-20.0, pitch=5.0)]),
    (3, 0.03, 480, 270, 1024, [dict(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0), dict(position=(-30.0, -15.0, 4.0), yaw=15.0, pitch=20.0)]),
]


@pytest.mark.parametrize("kind,detail,w,h,S,cams", CASES)
@pytest.mark.parametrize("mips", [1, 0])
def test_binned_visibility_equals_direct(vct, kind, detail, w, h, S, cams, mips):
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(kind, detail, 1234)
    cameras = [sc.default_camera(**kw) for kw in cams]
    a = render(vct, "direct", scene, w, h, S, cameras, mips=mips)
    b = render(vct, "binned", scene, w, h, S, cameras, mips=mips)
    assert (a[0] < 1.0).mean() > 0.02
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert (a[1][18] >= 0.5).mean() > 0.2            # the frame shows the scene


def test_binned_visibility_on_scissored_rows_and_repeated_passes(vct):
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(3, 0.03, 1234)
    w, h = 400, 300                                   # 38 tile rows: the slab's edges cut 16x16 bins in half
    cams = [sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)] * 3
    a = render(vct, "direct", scene, w, h, 512, cams, rows=(5, 22))
    b = render(vct, "binned", scene, w, h, 512, cams, rows=(5, 22))
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    full = render(vct, "direct", scene, w, h, 512, cams[:1])[1].reshape(23, h, w)
    slab = b[1].reshape(23, h, w)
    assert np.array_equal(slab[:, 40:176].view(np.uint32), full[:, 40:176].view(np.uint32))


def test_automatic_choice_of_the_form_keeps_the_result(vct):
    """Without VCT_RASTER_PATH a scene with alpha-tested textures renders its first whole frame with the direct form,
    then samples both forms (warm-up + two timed passes each, alternating) and keeps the faster: nine passes, nine identical
    G-buffers."""
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(3, 0.03, 1234)
    cams = [sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)] * 9
    a = render(vct, None, scene, 320, 180, 512, cams)
    want = render(vct, "direct", scene, 320, 180, 512, cams[:1])
    for x in a[1:]:
        assert np.array_equal(x.view(np.uint32), want[1].view(np.uint32))


@pytest.mark.parametrize("caps", ["40,1000000", "1000000,300", "64,64"])
def test_binned_capacity_overflow_falls_back_in_place(vct, caps):
    """Scratch capacities are sizes, not limits: what does not fit is rasterised in place by k_bin_setup (the direct
    form's code) and k_bin_raster merges with atomicMin.  VCT_BIN_TEST_CAPS reports tiny capacities to the kernels."""
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(3, 0.03, 1234)
    cams = [sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0),
            sc.default_camera(position=(-30.0, -15.0, 4.0), yaw=15.0, pitch=20.0)]
    want = render(vct, "direct", scene, 320, 180, 512, cams)
    os.environ["VCT_BIN_TEST_CAPS"] = caps
    try:
        got = render(vct, "binned", scene, 320, 180, 512, cams)
    finally:
        os.environ.pop("VCT_BIN_TEST_CAPS", None)
    for x, y in zip(want, got):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_binned_visibility_at_1080p_on_the_street(vct):
    """Full-size: the Bistro-class street at 1920 x 1080 and a 2048^2 shadow map, binned == direct bit for bit."""
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(3, 0.25, 1234)
    cams = [sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)]
    a = render(vct, "direct", scene, 1920, 1080, 2048, cams)
    b = render(vct, "binned", scene, 1920, 1080, 2048, cams)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert (a[1][18] >= 0.5).mean() > 0.2


def test_automatic_choice_on_a_slab_of_tile_rows(vct):
    """A rank of a multi-GPU frame only rasterises its slab: the sampling sequence (both forms over the same rows) and
    whatever form wins give the slab the direct form gives."""
    from voxel_cone_tracing_amd import scene as sc
    scene = sc.Scene(3, 0.03, 1234)
    cams = [sc.default_camera(position=(-58.0, -19.0, 1.5), yaw=0.0, pitch=12.0)] * 8
    a = render(vct, None, scene, 400, 300, 512, cams, rows=(5, 22))
    want = render(vct, "direct", scene, 400, 300, 512, cams[:1], rows=(5, 22))
    for x in a[1:]:
        assert np.array_equal(x.view(np.uint32), want[1].view(np.uint32))
