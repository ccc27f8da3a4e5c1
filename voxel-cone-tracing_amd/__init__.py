"""Thin ctypes binding of libvct_amd.so (the C ABI in include/vct.h).

The directory name carries a hyphen, so import it through `vctpkg.load()` (repo root) or
importlib; the module registers itself as `voxel_cone_tracing_amd`.  There is no Python or CPU
implementation of any kernel here: if the HIP library is missing, import fails.
"""
import ctypes as C
import os

import numpy as np

try:
    # torch bundles its own libamdhip64.so.7 (same soname as /opt/rocm's): whichever is loaded
    # first serves the whole process, and torch breaks on the other one.  Load torch's first so
    # that torch tensors, streams and this library share one HIP runtime.
    import torch  # noqa: F401
except ImportError:  # pure ctypes use without torch is fine
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
# VCT_AMD_LIB: path of an alternative build of the same library (A/B experiments only)
LIB_PATH = os.environ.get("VCT_AMD_LIB") or os.path.join(_HERE, "libvct_amd.so")

GB_PLANES = 23
GB_LINEAR, GB_TILED = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
VOX_CONSERVATIVE_AVG, VOX_REFERENCE = 0, 1
ABI_VERSION = 7

# every symbol include/vct.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "vct_default_config", "vct_create", "vct_destroy", "vct_last_error", "vct_get_config",
    "vct_set_camera_position", "vct_set_light_direction", "vct_set_ambient_factor",
    "vct_set_cone_apertures", "vct_set_trace_variant", "vct_trace_resident_strided", "vct_comm_set_interleaved",
    "vct_selftest_interleaved", "vct_upload_triangles", "vct_upload_shadow_map", "vct_voxelize",
    "vct_inject_light", "vct_build_mips", "vct_upload_volume_rgba8", "vct_upload_chain_rgba8",
    "vct_download_chain_rgba8", "vct_chain_texels", "vct_trace", "vct_trace_slab",
    "vct_trace_resident", "vct_synchronize", "vct_download_steps", "vct_download_cones",
    "vct_last_step_count", "vct_last_trace_ms", "vct_get_stream", "vct_get_frame_device",
    "vct_selftest_const_divide", "vct_selftest_area_divide", "vct_set_frame_target", "vct_bounce",
    "vct_download_voxel_attributes", "vct_download_aniso_rgba8", "vct_upload_mesh_attributes", "vct_render_shadow_map",
    "vct_download_shadow_map", "vct_render_gbuffer", "vct_download_gbuffer", "vct_trace_current", "vct_trace_resident_rows",
    "vct_last_trace_stats", "vct_download_frame", "vct_render_gbuffer_rows",
    "vct_slab_partition", "vct_comm_get_unique_id", "vct_comm_init", "vct_comm_destroy", "vct_comm_slab",
    "vct_frame_step", "vct_comm_sync", "vct_comm_frame", "vct_comm_download_frame",
    "vct_upload_mesh_uvs", "vct_upload_textures", "vct_gi_pass",
    "vct_comm_set_timeout_ms", "vct_last_row_steps", "vct_slab_partition_weighted", "vct_comm_set_slab_rows",
    "vct_get_stage_counts", "vct_comm_info", "vct_comm_last_gather_ms", "vct_set_footprint_records",
    "vct_set_frames_in_flight", "vct_get_frames_in_flight", "vct_select_frame_slot", "vct_selftest_texel_buffer",
    "vct_set_trace_timing",
]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("voxel_dim", C.c_int32),
        ("grid_world_size", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
        ("shadow_map_size", C.c_int32), ("model_scale", C.c_float),
        ("ambient_factor", C.c_float), ("shininess", C.c_float), ("max_distance", C.c_float),
        ("max_alpha", C.c_float), ("tan_diffuse", C.c_float), ("tan_specular", C.c_float),
        ("wrap_repeat", C.c_int32), ("debug_outputs", C.c_int32), ("trace_variant", C.c_int32),
        ("voxel_attributes", C.c_int32), ("anisotropic_mips", C.c_int32), ("texture_mipmaps", C.c_int32),
    ]


class GBuffer(C.Structure):
    _fields_ = [("planes", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("layout", C.c_int32), ("location", C.c_int32)]


class VctError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
        "There is no CPU fallback for the voxel-cone-tracing path.")

_lib = C.CDLL(LIB_PATH)
_lib.vct_last_error.restype = C.c_char_p
_lib.vct_last_error.argtypes = [C.c_void_p]
_lib.vct_chain_texels.restype = C.c_size_t
_lib.vct_chain_texels.argtypes = [C.c_int32]
_lib.vct_destroy.restype = None
_lib.vct_destroy.argtypes = [C.c_void_p]
_lib.vct_create.argtypes = [C.c_void_p, C.c_void_p]
for _n in ("vct_set_camera_position", "vct_set_light_direction", "vct_upload_volume_rgba8",
           "vct_upload_chain_rgba8", "vct_download_chain_rgba8", "vct_download_aniso_rgba8", "vct_download_steps",
           "vct_download_cones", "vct_last_step_count", "vct_last_trace_ms", "vct_last_trace_stats", "vct_get_stream",
           "vct_get_config"):
    getattr(_lib, _n).argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_set_ambient_factor.argtypes = [C.c_void_p, C.c_float]
_lib.vct_set_cone_apertures.argtypes = [C.c_void_p, C.c_float, C.c_float]
_lib.vct_set_trace_variant.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_trace_resident_strided.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
_lib.vct_comm_set_interleaved.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_selftest_interleaved.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
_lib.vct_upload_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.c_int32]
_lib.vct_upload_shadow_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
_lib.vct_voxelize.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_download_voxel_attributes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
_lib.vct_upload_mesh_attributes.argtypes = [C.c_void_p] * 5
for _n in ("vct_render_shadow_map", "vct_download_shadow_map", "vct_render_gbuffer", "vct_download_gbuffer",
           "vct_download_frame"):
    getattr(_lib, _n).argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_trace_current.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
_lib.vct_trace_resident_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
_lib.vct_gi_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
_lib.vct_gi_pass.restype = C.c_int
for _n in ("vct_inject_light", "vct_build_mips", "vct_trace_resident", "vct_synchronize", "vct_bounce"):
    getattr(_lib, _n).argtypes = [C.c_void_p]
_lib.vct_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
_lib.vct_trace_slab.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
_lib.vct_get_frame_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
_lib.vct_set_frame_target.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_selftest_const_divide.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
_lib.vct_selftest_area_divide.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
_lib.vct_render_gbuffer_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
_lib.vct_slab_partition.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.vct_comm_get_unique_id.argtypes = [C.c_void_p]
_lib.vct_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
_lib.vct_comm_slab.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
_lib.vct_comm_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
_lib.vct_comm_download_frame.argtypes = [C.c_void_p, C.c_void_p]
for _n in ("vct_comm_destroy", "vct_frame_step", "vct_comm_sync"):
    getattr(_lib, _n).argtypes = [C.c_void_p]
COMM_ID_BYTES = 128
_lib.vct_get_stage_counts.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_comm_set_timeout_ms.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_set_footprint_records.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_comm_info.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_comm_last_gather_ms.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_last_row_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
_lib.vct_slab_partition_weighted.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
_lib.vct_comm_set_slab_rows.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_upload_mesh_uvs.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_set_frames_in_flight.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_get_frames_in_flight.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_lib.vct_select_frame_slot.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_set_trace_timing.argtypes = [C.c_void_p, C.c_int32]
_lib.vct_selftest_texel_buffer.argtypes = [C.c_void_p, C.c_void_p]
_lib.vct_upload_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]


def lib():
    return _lib


def default_config(**kw):
    cfg = Config()
    _lib.vct_default_config(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def slab_partition(height, world, rank):
    """(tile_row0, tile_row1, rows_per_rank) of rank's slab: equal padded slabs of ceil(tile_rows / world)."""
    r0, r1, per = C.c_int32(), C.c_int32(), C.c_int32()
    if _lib.vct_slab_partition(height, world, rank, C.byref(r0), C.byref(r1), C.byref(per)) != 0:
        raise VctError("vct_slab_partition: bad arguments")
    return r0.value, r1.value, per.value


def slab_partition_weighted(row_cost, world):
    """Tile-row boundaries [world + 1] of `world` contiguous slabs of near-equal cost (row_cost: per tile row)."""
    cost = np.ascontiguousarray(row_cost, np.uint64)
    starts = np.zeros(world + 1, np.int32)
    if _lib.vct_slab_partition_weighted(_ptr(cost), cost.shape[0], world, _ptr(starts)) != 0:
        raise VctError("vct_slab_partition_weighted: bad arguments")
    return starts


def comm_unique_id():
    """ncclGetUniqueId as 128 bytes (rank 0 creates it, every rank passes it to Context.comm_init)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = _lib.vct_comm_get_unique_id(buf)
    if rc != 0:
        raise VctError(f"vct_comm_get_unique_id failed ({rc}): {_lib.vct_last_error(None).decode()}")
    return bytes(buf)


def chain_texels(V):
    return _lib.vct_chain_texels(V)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One context per GPU; mirrors the call order of the reference's orchestrator
    (ctor -> init: voxelize, inject, mips -> Render per frame; VCT.h:57-190)."""

    def __init__(self, cfg=None, **kw):
        cfg = cfg or default_config(**kw)
        self._h = C.c_void_p()
        rc = _lib.vct_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise VctError(f"vct_create failed ({rc}): {_lib.vct_last_error(None).decode()}")
        self.cfg = Config()
        _lib.vct_get_config(self._h, C.byref(self.cfg))

    def close(self):
        if self._h:
            _lib.vct_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc, what):
        if rc != 0:
            raise VctError(f"{what} failed ({rc}): {_lib.vct_last_error(self._h).decode()}")

    # --- uniforms
    def set_camera_position(self, pos):
        a = np.ascontiguousarray(pos, np.float32)
        self._ck(_lib.vct_set_camera_position(self._h, _ptr(a)), "vct_set_camera_position")

    def set_light_direction(self, d):
        a = np.ascontiguousarray(d, np.float32)
        self._ck(_lib.vct_set_light_direction(self._h, _ptr(a)), "vct_set_light_direction")

    def set_ambient_factor(self, a):
        self._ck(_lib.vct_set_ambient_factor(self._h, float(a)), "vct_set_ambient_factor")

    def set_cone_apertures(self, td, ts):
        self._ck(_lib.vct_set_cone_apertures(self._h, float(td), float(ts)), "vct_set_cone_apertures")

    def set_footprint_records(self, on=True):
        """One 32-byte footprint record per texel of the levels >= 1 (include/vct.h): for HBM-bound volumes."""
        self._ck(_lib.vct_set_footprint_records(self._h, int(bool(on))), "vct_set_footprint_records")

    def set_trace_variant(self, variant):
        self._ck(_lib.vct_set_trace_variant(self._h, int(variant)), "vct_set_trace_variant")

    # --- scene / volume
    def upload_triangles(self, pos, material, albedo):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
        material = np.ascontiguousarray(material, np.int32)
        albedo = np.ascontiguousarray(albedo, np.float32).reshape(-1, 4)
        self._ck(_lib.vct_upload_triangles(self._h, _ptr(pos), _ptr(material), pos.shape[0],
                                           _ptr(albedo), albedo.shape[0]), "vct_upload_triangles")

    def upload_shadow_map(self, depth, light_vp_rowmajor):
        if depth is None:
            self._ck(_lib.vct_upload_shadow_map(self._h, None, 0, None), "vct_upload_shadow_map")
            return
        depth = np.ascontiguousarray(depth, np.float32)
        m = np.ascontiguousarray(np.asarray(light_vp_rowmajor, np.float32).T)   # -> column-major
        self._ck(_lib.vct_upload_shadow_map(self._h, _ptr(depth), depth.shape[0], _ptr(m)),
                 "vct_upload_shadow_map")

    def voxelize(self, mode=VOX_CONSERVATIVE_AVG):
        self._ck(_lib.vct_voxelize(self._h, mode), "vct_voxelize")

    def inject_light(self):
        self._ck(_lib.vct_inject_light(self._h), "vct_inject_light")

    def build_mips(self):
        self._ck(_lib.vct_build_mips(self._h), "vct_build_mips")

    # --- raster input stages on the GPU
    def upload_mesh_attributes(self, normal, tangent, bitangent, specular):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (normal, tangent, bitangent, specular)]
        self._ck(_lib.vct_upload_mesh_attributes(self._h, *(_ptr(a) for a in arrs)),
                 "vct_upload_mesh_attributes")

    def upload_mesh_uvs(self, uv):
        uv = np.ascontiguousarray(uv, np.float32)
        self._ck(_lib.vct_upload_mesh_uvs(self._h, _ptr(uv)), "vct_upload_mesh_uvs")

    def upload_textures(self, textures, mat_tex):
        """textures: list of uint8 [h, w, 4] (row 0 at v = 0); mat_tex: int32 [nmat, 3] diffuse / specular /
        height texture index or -1.  An empty list detaches the textures."""
        texs = [np.ascontiguousarray(t, np.uint8) for t in textures]
        n = len(texs)
        ptrs = (C.c_void_p * max(n, 1))(*[t.ctypes.data for t in texs])
        w = np.array([t.shape[1] for t in texs], np.int32)
        h = np.array([t.shape[0] for t in texs], np.int32)
        mt = np.ascontiguousarray(mat_tex, np.int32)
        self._ck(_lib.vct_upload_textures(self._h, ptrs, _ptr(w), _ptr(h), n, _ptr(mt)), "vct_upload_textures")

    def upload_scene(self, scene):
        """Everything of a voxel_cone_tracing_amd.scene.Scene: triangles, frames, texture coordinates, maps."""
        self.upload_triangles(scene.pos, scene.material, scene.albedo)
        self.upload_mesh_attributes(*scene.frames(), scene.specular)
        if scene.textures:
            self.upload_mesh_uvs(scene.uv)
            self.upload_textures(scene.textures, scene.mat_tex)

    def render_shadow_map(self, light_vp_colmajor):
        m = np.ascontiguousarray(light_vp_colmajor, np.float32).reshape(16)
        self._ck(_lib.vct_render_shadow_map(self._h, _ptr(m)), "vct_render_shadow_map")

    def download_shadow_map(self):
        S = self.cfg.shadow_map_size
        out = np.zeros((S, S), np.float32)
        self._ck(_lib.vct_download_shadow_map(self._h, _ptr(out)), "vct_download_shadow_map")
        return out

    def render_gbuffer(self, view_proj_colmajor):
        m = np.ascontiguousarray(view_proj_colmajor, np.float32).reshape(16)
        self._ck(_lib.vct_render_gbuffer(self._h, _ptr(m)), "vct_render_gbuffer")

    def gi_pass(self, light_vp_colmajor, view_proj_colmajor, mode=VOX_CONSERVATIVE_AVG):
        """Shadow map -> {voxelize, inject, mips} beside {G-buffer raster} -> trace, one call (include/vct.h)."""
        lv = np.ascontiguousarray(light_vp_colmajor, np.float32).reshape(16)
        m = np.ascontiguousarray(view_proj_colmajor, np.float32).reshape(16)
        self._ck(_lib.vct_gi_pass(self._h, _ptr(lv), _ptr(m), mode), "vct_gi_pass")

    def render_gbuffer_rows(self, view_proj_colmajor, row0, row1):
        m = np.ascontiguousarray(view_proj_colmajor, np.float32).reshape(16)
        self._ck(_lib.vct_render_gbuffer_rows(self._h, _ptr(m), row0, row1), "vct_render_gbuffer_rows")

    # --- multi-GPU slabs + one RCCL gather per frame
    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(_lib.vct_comm_init(self._h, buf, rank, world), "vct_comm_init")

    def comm_destroy(self):
        self._ck(_lib.vct_comm_destroy(self._h), "vct_comm_destroy")

    def comm_slab(self):
        r0, r1 = C.c_int32(), C.c_int32()
        self._ck(_lib.vct_comm_slab(self._h, C.byref(r0), C.byref(r1)), "vct_comm_slab")
        return r0.value, r1.value

    def comm_info(self):
        """What RCCL says about the communicator: dict(nranks, rank, device, rccl_version)."""
        v = (C.c_int32 * 4)()
        self._ck(_lib.vct_comm_info(self._h, v), "vct_comm_info")
        return dict(nranks=v[0], rank=v[1], device=v[2], rccl_version=v[3])

    def comm_last_gather_ms(self):
        v = C.c_float()
        self._ck(_lib.vct_comm_last_gather_ms(self._h, C.byref(v)), "vct_comm_last_gather_ms")
        return v.value

    def comm_set_timeout_ms(self, ms):
        self._ck(_lib.vct_comm_set_timeout_ms(self._h, int(ms)), "vct_comm_set_timeout_ms")

    def comm_set_slab_rows(self, starts):
        """Collective: load-aware slab boundaries [world + 1] in tile rows (None: the equal partition)."""
        a = None if starts is None else np.ascontiguousarray(starts, np.int32)
        self._ck(_lib.vct_comm_set_slab_rows(self._h, _ptr(a)), "vct_comm_set_slab_rows")

    def comm_set_interleaved(self, on=True):
        """Collective: tile row r belongs to rank r % world (every rank needs the whole G-buffer resident)."""
        self._ck(_lib.vct_comm_set_interleaved(self._h, 1 if on else 0), "vct_comm_set_interleaved")

    def selftest_interleaved(self, world):
        """Pixels that differ between the de-interleaved frame of `world` emulated ranks and the frame of one launch."""
        bad = C.c_uint64(0)
        self._ck(_lib.vct_selftest_interleaved(self._h, int(world), C.byref(bad)), "vct_selftest_interleaved")
        return bad.value

    def trace_gbuffer_strided(self, row0, row1, stride):
        """Asynchronous trace of every stride-th tile row of [row0, row1) of the resident G-buffer."""
        self._ck(_lib.vct_trace_resident_strided(self._h, row0, row1, stride), "vct_trace_resident_strided")

    def last_row_steps(self):
        """Executed cone steps per 8-pixel tile row of the last screen trace (uint64 [ceil(height / 8)])."""
        n = (self.cfg.height + 7) // 8
        out = np.zeros(n, np.uint64)
        self._ck(_lib.vct_last_row_steps(self._h, _ptr(out), n), "vct_last_row_steps")
        return out

    def frame_step(self):
        self._ck(_lib.vct_frame_step(self._h), "vct_frame_step")

    def comm_sync(self):
        self._ck(_lib.vct_comm_sync(self._h), "vct_comm_sync")

    def comm_download_frame(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), np.uint16)
        self._ck(_lib.vct_comm_download_frame(self._h, _ptr(out)), "vct_comm_download_frame")
        return out

    def download_gbuffer(self):
        out = np.zeros((GB_PLANES, self.cfg.width * self.cfg.height), np.float32)
        self._ck(_lib.vct_download_gbuffer(self._h, _ptr(out)), "vct_download_gbuffer")
        return out

    def trace_gbuffer_rows(self, row0, row1):
        """Asynchronous trace of tile rows [row0, row1) of the resident G-buffer."""
        self._ck(_lib.vct_trace_resident_rows(self._h, row0, row1), "vct_trace_resident_rows")

    def trace_current(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), np.uint16)
        self._ck(_lib.vct_trace_current(self._h, _ptr(out), MEM_HOST), "vct_trace_current")
        return out

    def bounce(self):
        self._ck(_lib.vct_bounce(self._h), "vct_bounce")

    def voxel_attributes(self):
        V = self.cfg.voxel_dim
        alb = np.zeros((V, V, V, 4), np.uint8)
        nrm = np.zeros((V, V, V, 4), np.uint8)
        self._ck(_lib.vct_download_voxel_attributes(self._h, _ptr(alb), _ptr(nrm)),
                 "vct_download_voxel_attributes")
        return alb, nrm

    def upload_volume(self, l0):
        l0 = np.ascontiguousarray(l0, np.uint8)
        assert l0.size == self.cfg.voxel_dim ** 3 * 4
        self._ck(_lib.vct_upload_volume_rgba8(self._h, _ptr(l0)), "vct_upload_volume_rgba8")

    def upload_chain(self, chain):
        chain = np.ascontiguousarray(chain, np.uint8)
        assert chain.size == chain_texels(self.cfg.voxel_dim) * 4
        self._ck(_lib.vct_upload_chain_rgba8(self._h, _ptr(chain)), "vct_upload_chain_rgba8")

    def download_aniso(self):
        V = self.cfg.voxel_dim
        out = np.zeros((6, chain_texels(V) - V ** 3, 4), np.uint8)
        self._ck(_lib.vct_download_aniso_rgba8(self._h, _ptr(out)), "vct_download_aniso_rgba8")
        return out

    def download_chain(self):
        out = np.zeros((chain_texels(self.cfg.voxel_dim), 4), np.uint8)
        self._ck(_lib.vct_download_chain_rgba8(self._h, _ptr(out)), "vct_download_chain_rgba8")
        return out

    # --- trace
    def _gb(self, planes, layout, location):
        gb = GBuffer()
        gb.planes = planes if isinstance(planes, int) else planes.ctypes.data
        gb.width, gb.height = self.cfg.width, self.cfg.height
        gb.layout, gb.location = layout, location
        return gb

    def trace(self, planes, rows=None, layout=GB_LINEAR, out_device_ptr=None):
        """planes: float32 [23, h*w] numpy (host) or an int device pointer (location=device).
        Returns the RGBA16F frame as uint16 [h, w, 4] (host) unless out_device_ptr is given."""
        location = MEM_DEVICE if isinstance(planes, int) else MEM_HOST
        if location == MEM_HOST:
            planes = np.ascontiguousarray(planes, np.float32)
        gb = self._gb(planes, layout, location)
        h, w = self.cfg.height, self.cfg.width
        if out_device_ptr is not None:
            out, optr, oloc = None, C.c_void_p(out_device_ptr), MEM_DEVICE
        else:
            out = np.zeros((h, w, 4), np.uint16)
            optr, oloc = _ptr(out), MEM_HOST
        if rows is None:
            self._ck(_lib.vct_trace(self._h, C.byref(gb), optr, oloc), "vct_trace")
        else:
            self._ck(_lib.vct_trace_slab(self._h, C.byref(gb), rows[0], rows[1], optr, oloc),
                     "vct_trace_slab")
        return out

    def download_frame(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), np.uint16)
        self._ck(_lib.vct_download_frame(self._h, _ptr(out)), "vct_download_frame")
        return out

    def set_frame_target(self, dev_ptr):
        """Kernel output goes to caller-owned HBM (full-frame addressing); None restores the default."""
        self._ck(_lib.vct_set_frame_target(self._h, C.c_void_p(dev_ptr) if dev_ptr else None),
                 "vct_set_frame_target")

    def trace_resident(self):
        self._ck(_lib.vct_trace_resident(self._h), "vct_trace_resident")

    def synchronize(self):
        self._ck(_lib.vct_synchronize(self._h), "vct_synchronize")

    def selftest_texel_buffer(self):
        """Channel values (of 4096) that the typed-buffer texel load converts differently from (float)c / 255.0f: 0 expected."""
        v = C.c_uint64()
        self._ck(_lib.vct_selftest_texel_buffer(self._h, C.byref(v)), "vct_selftest_texel_buffer")
        return v.value

    def set_frames_in_flight(self, n):
        """2: a second frame slot (stream, G-buffer, frame) so that frame k + 1 starts while frame k drains; 1: default."""
        self._ck(_lib.vct_set_frames_in_flight(self._h, n), "vct_set_frames_in_flight")

    def frames_in_flight(self):
        """(frames in flight, selected slot)"""
        n, s = C.c_int32(), C.c_int32()
        self._ck(_lib.vct_get_frames_in_flight(self._h, C.byref(n), C.byref(s), None), "vct_get_frames_in_flight")
        return n.value, s.value

    def frame_slot_streams_overlap(self):
        """True when the second slot's stream was seen to run beside the first (include/vct.h)."""
        v = C.c_int32()
        self._ck(_lib.vct_get_frames_in_flight(self._h, None, None, C.byref(v)), "vct_get_frames_in_flight")
        return bool(v.value)

    def select_frame_slot(self, slot):
        """Every later call works on this slot's G-buffer / frame / stream (call with k & 1 before frame k)."""
        self._ck(_lib.vct_select_frame_slot(self._h, slot), "vct_select_frame_slot")

    def steps(self):
        out = np.zeros((self.cfg.height * self.cfg.width, 7), np.uint8)
        self._ck(_lib.vct_download_steps(self._h, _ptr(out)), "vct_download_steps")
        return out

    def cones(self):
        out = np.zeros((self.cfg.height * self.cfg.width, 7, 4), np.float32)
        self._ck(_lib.vct_download_cones(self._h, _ptr(out)), "vct_download_cones")
        return out

    def last_step_count(self):
        v = C.c_uint64()
        self._ck(_lib.vct_last_step_count(self._h, C.byref(v)), "vct_last_step_count")
        return v.value

    def last_trace_stats(self):
        """Instrumented builds (-DVCT_STATS=1) only: dict of wave-level march counters."""
        v = (C.c_uint64 * 16)()
        self._ck(_lib.vct_last_trace_stats(self._h, v), "vct_last_trace_stats")
        keys = ("wave_steps", "lane_steps", "coop_zero", "coop_hit", "fallback", "fallback_lanes", "fallback_fits",
                "greedy_blocks", "greedy_le2", "greedy_le3", "greedy_le4", "quads_live", "quads_fit333", "quads_same",
                "quadrants_live", "quadrants_fit444")
        return dict(zip(keys, (int(x) for x in v)))

    def stage_counts(self):
        v = (C.c_uint64 * 8)()
        self._ck(_lib.vct_get_stage_counts(self._h, v), "vct_get_stage_counts")
        return dict(zip(("triangles", "vox_candidates", "reserved", "accumulator_bricks", "touched_bricks",
                         "comm_reserved_cus", "raster_form", "vox_items"), (int(x) for x in v)))

    def set_trace_timing(self, on=True):
        """Bracket march launches with the timing events last_trace_ms() reads (default on; ~7 us per launch: include/vct.h)."""
        self._ck(_lib.vct_set_trace_timing(self._h, int(bool(on))), "vct_set_trace_timing")

    def last_trace_ms(self):
        v = C.c_float()
        self._ck(_lib.vct_last_trace_ms(self._h, C.byref(v)), "vct_last_trace_ms")
        return v.value

    def selftest_const_divide(self, d):
        v = C.c_uint64()
        self._ck(_lib.vct_selftest_const_divide(self._h, float(d), C.byref(v)), "vct_selftest_const_divide")
        return v.value

    def selftest_area_divide(self, seed, count):
        v = C.c_uint64()
        self._ck(_lib.vct_selftest_area_divide(self._h, int(seed), int(count), C.byref(v)), "vct_selftest_area_divide")
        return v.value

    def stream(self):
        v = C.c_void_p()
        self._ck(_lib.vct_get_stream(self._h, C.byref(v)), "vct_get_stream")
        return v.value

    def frame_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(_lib.vct_get_frame_device(self._h, C.byref(p), C.byref(n)), "vct_get_frame_device")
        return p.value, n.value


def half_to_float(u16):
    return np.asarray(u16, np.uint16).view(np.float16).astype(np.float32)
