"""Wavefront OBJ ingestion (stands where the reference's assimp import stands,
R/Model.h:39-61): triangulation, materials, generated normals / tangent frames."""
import os

import numpy as np
import pytest

import vctpkg

CUBE = """# unit cube, quads + one triangle pair, two materials, no normals
mtllib cube.mtl
v -1 -1 -1
v  1 -1 -1
v  1  1 -1
v -1  1 -1
v -1 -1  1
v  1 -1  1
v  1  1  1
v -1  1  1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
usemtl red
f 4/4 3/3 2/2 1/1
f 5/1 6/2 7/3 8/4
usemtl green
f 1/1 2/2 6/3 5/4
f 2/1 3/2 7/3 6/4
f 3/1 4/2 8/3 7/4
f -5/1 -8/2 -4/3 -1/4
"""
MTL = """newmtl red
Kd 0.8 0.1 0.1
Ks 0.5 0.5 0.5
newmtl green
Kd 0.1 0.7 0.2
d 1.0
"""


@pytest.fixture()
def cube(tmp_path):
    (tmp_path / "cube.obj").write_text(CUBE)
    (tmp_path / "cube.mtl").write_text(MTL)
    return str(tmp_path / "cube.obj")


def test_obj_loader_geometry_and_materials(cube):
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    s = sc.Scene(cube)
    assert s.ntri == 12 and s.nmat == 2
    assert np.allclose(s.albedo[0], [0.8, 0.1, 0.1, 1.0]) and np.allclose(s.specular[0], [0.5, 0.5, 0.5])
    assert np.allclose(s.albedo[1], [0.1, 0.7, 0.2, 1.0])
    assert list(s.material) == [0] * 4 + [1] * 8
    pos = s.pos.reshape(12, 3, 3)
    assert np.abs(pos).max() == 1.0
    # outward-facing CCW faces: the face normal points away from the centre
    fn = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
    assert (np.einsum("ij,ij->i", fn, pos.mean(1)) > 0).all()
    nrm, tan, bit = (a.reshape(12, 3, 3) for a in s.frames())
    for a in (nrm, tan, bit):
        assert np.allclose(np.linalg.norm(a, axis=-1), 1.0, atol=1e-5)
    assert np.abs(np.einsum("ijk,ijk->ij", nrm, tan)).max() < 1e-5          # orthonormal frames
    assert np.allclose(np.cross(nrm, tan), bit, atol=1e-5)
    # generated smooth normals (area-weighted over the fan triangles) point out of the corner octant
    assert (np.sign(nrm) == np.sign(pos)).all() and np.abs(nrm).min() > 0.3


def test_obj_loader_errors(tmp_path):
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    with pytest.raises(ValueError, match="cannot open"):
        sc.Scene(str(tmp_path / "missing.obj"))
    (tmp_path / "empty.obj").write_text("v 0 0 0\n")
    with pytest.raises(ValueError, match="no faces"):
        sc.Scene(str(tmp_path / "empty.obj"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError, match="missing vertex"):
        sc.Scene(str(tmp_path / "bad.obj"))


@pytest.mark.gpu
def test_obj_scene_through_the_resident_pipeline(cube, oracle):
    vct = vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    s = sc.Scene(cube)
    pos = s.pos * 600.0            # model units: +-600 -> +-30 world under the 0.05 scale
    V, w, h = 32, 64, 48
    with vct.Context(vct.default_config(voxel_dim=V, width=w, height=h, shadow_map_size=256)) as ctx:
        ctx.upload_triangles(pos, s.material, s.albedo)
        ctx.upload_mesh_attributes(*s.frames(), s.specular)
        light = (0.0, 1.0, 0.25)
        ctx.render_shadow_map(sc.light_view_proj(light))
        ctx.voxelize(); ctx.inject_light(); ctx.build_mips()
        depth = ctx.download_shadow_map()
        lvp = sc.light_view_proj(light).reshape(4, 4).T
        want = oracle.voxelize_conservative(oracle.default_params(V),
                                            oracle.make_scene(pos, s.material, s.albedo, shadow_depth=depth, light_vp=lvp))
        chain = ctx.download_chain()
        assert np.array_equal(chain, oracle.build_mips(want))
        cam = sc.default_camera(position=(0.0, 10.0, 200.0))
        ctx.set_camera_position((0.0, 10.0, 200.0)); ctx.set_light_direction(light)
        ctx.render_gbuffer(sc.camera_view_proj(cam, w, h))
        planes = ctx.download_gbuffer()
        assert 0.05 < (planes[18] >= 0.5).mean() < 0.9
        frame = ctx.trace_current()
        ref = oracle.trace(oracle.default_params(V, camera_pos=(0.0, 10.0, 200.0), light_dir=light), chain, planes, nthreads=4)
        assert np.array_equal(ctx.last_step_count(), ref["total_steps"])
        assert (frame.reshape(-1, 4) == ref["rgba16f"]).mean() > 0.999


def test_scene_cache_round_trip(tmp_path):
    """the on-disk cache holds everything a scene holds (triangles, frames, texture coordinates,
    materials, decoded maps); corrupt or foreign files are refused."""
    import vctpkg
    vctpkg.load()
    from voxel_cone_tracing_amd import scene as sc
    s = sc.Scene(sc.ATRIUM_TEXTURED, 0.05, 7)
    path = str(tmp_path / "atrium.vctscene")
    s.save(path)
    t = sc.Scene(path)
    assert t.ntri == s.ntri and t.nmat == s.nmat and len(t.textures) == len(s.textures) > 0
    for a, b in ((s.pos, t.pos), (s.uv, t.uv), (s.material, t.material), (s.albedo, t.albedo),
                 (s.specular, t.specular), (s.mat_tex, t.mat_tex)):
        assert np.array_equal(a, b)
    assert all(np.array_equal(a, b) for a, b in zip(s.frames(), t.frames()))
    assert all(np.array_equal(a, b) for a, b in zip(s.textures, t.textures))
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        sc.Scene(path)
    open(path, "wb").write(b"NOTACACHE" + raw[9:])
    with pytest.raises(ValueError):
        sc.Scene(path)


@pytest.mark.gpu
def test_presenter_writes_the_tonemapped_frame(tmp_path):
    """vct_demo --ppm (the headless stand-in for swap-buffers, R/main.cpp:92) writes the frame of
    its last Render() as a P6 image, bottom row last, tonemapped from the RGBA16F halves."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    demo = os.path.join(root, "voxel-cone-tracing_amd", "vct_demo")
    out = str(tmp_path / "frame.ppm")
    r = subprocess.run([demo, "--scene", "procedural:cornell", "--voxels", "32", "--size", "96x64", "--shadow", "256",
                        "--frames", "1", "--ppm", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    assert raw.startswith(b"P6\n96 64\n255\n") and len(raw) == len(b"P6\n96 64\n255\n") + 96 * 64 * 3
    img = np.frombuffer(raw[-96 * 64 * 3:], np.uint8).reshape(64, 96, 3)
    assert 10 < img.mean() < 245 and img.std() > 5            # a lit scene, not a constant image
    left, right = img[20:44, 4:24].mean((0, 1)), img[20:44, 72:92].mean((0, 1))
    assert left[0] > left[1] and right[1] > right[0]          # red wall on the left, green wall on the right
