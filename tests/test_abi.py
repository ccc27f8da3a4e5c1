"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/vct.h declares, validates arguments, and fails loudly (no CPU fallback) without a GPU.
No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

import vctpkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vct():
    return vctpkg.load()


def test_header_symbols_all_exported(vct):
    hdr = open(os.path.join(ROOT, "include", "vct.h")).read()
    declared = set(re.findall(r"\b(vct_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vct_status"}
    assert declared == set(vct.ABI_SYMBOLS)
    L = vct.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} not exported by libvct_amd.so"


def test_default_config_matches_reference_constants(vct):
    cfg = vct.default_config()
    assert cfg.abi_version == vct.ABI_VERSION
    assert cfg.voxel_dim == 128 and cfg.grid_world_size == 150.0          # VCT.h:16-17
    assert (cfg.width, cfg.height) == (1280, 720)                          # VCT.h:24-25
    assert cfg.shadow_map_size == 4096                                     # VCT.h:35
    assert abs(cfg.model_scale - 0.05) < 1e-8 and abs(cfg.ambient_factor - 0.1) < 1e-8
    assert cfg.shininess == 20.0 and cfg.max_distance == 75.0              # Mesh.h:86, trace.fs:43
    assert abs(cfg.max_alpha - 0.95) < 1e-7 and abs(cfg.tan_diffuse - 0.577) < 1e-7
    assert abs(cfg.tan_specular - 0.07) < 1e-8 and cfg.wrap_repeat == 1


def test_config_struct_layout_matches_header(vct):
    hdr = open(os.path.join(ROOT, "include", "vct.h")).read()
    body = re.search(r"typedef struct vct_config \{(.*?)\} vct_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        _, rest = decl.split(None, 1)
        names += [n.strip() for n in rest.split(",")]
    assert names == [f[0] for f in vct.Config._fields_]
    assert C.sizeof(vct.Config) == 4 * len(names)


def test_chain_texels(vct):
    assert vct.chain_texels(8) == 512 + 64 + 8 + 1
    assert vct.chain_texels(256) == sum((256 >> l) ** 3 for l in range(9))
    assert vct.chain_texels(100) == 0


def test_create_rejects_bad_config(vct):
    for kw in (dict(voxel_dim=100), dict(voxel_dim=4), dict(voxel_dim=2048), dict(width=0),
               dict(abi_version=99)):
        with pytest.raises(vct.VctError):
            vct.Context(vct.default_config(**kw))


def test_no_cpu_fallback_without_gpu(vct):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vct.VctError) as e:
        vct.Context(vct.default_config(voxel_dim=16, width=8, height=8))
    assert "no HIP device" in str(e.value) or "no CPU path" in str(e.value)


def test_null_handles_are_rejected(vct):
    L = vct.lib()
    assert L.vct_build_mips(None) != 0
    assert L.vct_voxelize(None, 0) != 0
    assert L.vct_trace(None, None, None, 0) != 0
    assert L.vct_default_config(None) != 0


def test_host_library_exports_every_declared_symbol(vct):
    import ctypes
    hdr = open(os.path.join(ROOT, "voxel-cone-tracing_amd", "host", "vct_host.h")).read()
    declared = set(re.findall(r"\b(vcth_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 11
    lib = ctypes.CDLL(os.path.join(ROOT, "voxel-cone-tracing_amd", "libvct_host.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} not exported by libvct_host.so"


def test_facade_header_keeps_the_reference_surface():
    """The names main.cpp uses (R/main.cpp:64-68,90,117-142) and the public fields of the reference's
    orchestrator (VCT.h:14-53) exist in the facade header."""
    hdr = open(os.path.join(ROOT, "voxel-cone-tracing_amd", "host", "Voxel_Cone_Tracing.h")).read()
    for name in ("struct Voxel_Cone_Tracing", "void init_voxel_cone_tracing()", "void Render()",
                 "void DrawDepthTexture()", "void DrawVoxelTexture()", "lightDirection", "VoxelDimensions",
                 "VoxelGridWorldSize", "screen_width", "screen_height", "ShadowMapSize",
                 "DepthViewProjectionMatrix", "ProjX", "ProjY", "ProjZ", "AmbientFactor", "ShowDiffuse",
                 "Camera camera(vec3(0.0f, 4.0f, 0.0f))", "MovementSpeed", "MouseSensitivity",
                 "ProcessKeyBoard", "ProcessMouseMovement", "ProcessMouseScroll", "GetViewMatrix"):
        assert name in hdr, name
