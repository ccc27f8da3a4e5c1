"""ctypes binding of libvct_host.so: procedural scenes + the CPU input stages (shadow-map raster,
G-buffer raster) that feed vct_voxelize / vct_trace.  See host/vct_host.h for what each stands in
for in the reference."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvct_host.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make host`")
_lib = C.CDLL(LIB_PATH)

CORNELL, ATRIUM = 0, 1


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("yaw", C.c_float), ("pitch", C.c_float),
                ("zoom", C.c_float), ("z_near", C.c_float), ("z_far", C.c_float)]


_lib.vcth_scene_create.restype = C.c_void_p
_lib.vcth_scene_create.argtypes = [C.c_int, C.c_float, C.c_uint32]
_lib.vcth_scene_destroy.argtypes = [C.c_void_p]
_lib.vcth_scene_load_obj.restype = C.c_void_p
_lib.vcth_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p]
_lib.vcth_scene_num_triangles.argtypes = [C.c_void_p]
_lib.vcth_scene_num_materials.argtypes = [C.c_void_p]
_lib.vcth_scene_get.argtypes = [C.c_void_p] * 5
_lib.vcth_light_view_proj.argtypes = [C.c_void_p, C.c_void_p]
_lib.vcth_scene_get_frames.argtypes = [C.c_void_p] * 4
_lib.vcth_camera_view_proj.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
_lib.vcth_render_shadow_map.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]
_lib.vcth_render_gbuffer.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]


def default_camera(position=None, yaw=None, pitch=None, zoom=None):
    cam = Camera()
    _lib.vcth_default_camera(C.byref(cam))
    if position is not None:
        cam.position[:] = list(position)
    if yaw is not None:
        cam.yaw = yaw
    if pitch is not None:
        cam.pitch = pitch
    if zoom is not None:
        cam.zoom = zoom
    return cam


def camera_view_proj(cam, w, h):
    """Column-major view-projection (float32[16]) of VCT.h:161-163 for this camera and frame size."""
    vp = np.zeros(16, np.float32)
    _lib.vcth_camera_view_proj(C.byref(cam), w, h, vp.ctypes.data)
    return vp


def light_view_proj(light_dir):
    """Column-major DepthViewProjectionMatrix (float32[16]) of VCT.h:84-86."""
    L = np.ascontiguousarray(light_dir, np.float32)
    vp = np.zeros(16, np.float32)
    _lib.vcth_light_view_proj(L.ctypes.data, vp.ctypes.data)
    return vp


class Scene:
    def __init__(self, kind, detail=1.0, seed=1234):
        """kind: CORNELL / ATRIUM (procedural) or the path of a Wavefront .obj file."""
        if isinstance(kind, (str, bytes, os.PathLike)):
            err = C.create_string_buffer(256)
            self._h = _lib.vcth_scene_load_obj(os.fsencode(kind), err)
            if not self._h:
                raise ValueError(f"cannot load {kind}: {err.value.decode()}")
        else:
            self._h = _lib.vcth_scene_create(kind, float(detail), int(seed))
            if not self._h:
                raise ValueError("unknown scene kind")
        self.ntri = _lib.vcth_scene_num_triangles(self._h)
        self.nmat = _lib.vcth_scene_num_materials(self._h)
        self.pos = np.zeros((self.ntri, 9), np.float32)
        self.material = np.zeros(self.ntri, np.int32)
        self.albedo = np.zeros((self.nmat, 4), np.float32)
        self.specular = np.zeros((self.nmat, 3), np.float32)
        _lib.vcth_scene_get(self._h, self.pos.ctypes.data, self.material.ctypes.data,
                            self.albedo.ctypes.data, self.specular.ctypes.data)

    def frames(self):
        """Per-vertex (normal, tangent, bitangent), each float32 [ntri, 9]."""
        out = [np.zeros((self.ntri, 9), np.float32) for _ in range(3)]
        _lib.vcth_scene_get_frames(self._h, *(a.ctypes.data for a in out))
        return out

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.vcth_scene_destroy(self._h)
            self._h = None

    def shadow_map(self, light_dir, size, model_scale=0.05):
        """Returns (depth [size,size] fp32, light_vp row-major 4x4)."""
        L = np.ascontiguousarray(light_dir, np.float32)
        vp = np.zeros(16, np.float32)
        _lib.vcth_light_view_proj(L.ctypes.data, vp.ctypes.data)
        depth = np.zeros((size, size), np.float32)
        _lib.vcth_render_shadow_map(self._h, model_scale, vp.ctypes.data, size, depth.ctypes.data)
        return depth, vp.reshape(4, 4).T.copy()

    def gbuffer(self, cam, w, h, shadow=None, light_vp=None, model_scale=0.05):
        planes = np.zeros((23, w * h), np.float32)
        if shadow is not None:
            sd = np.ascontiguousarray(shadow, np.float32)
            vp = np.ascontiguousarray(np.asarray(light_vp, np.float32).T)   # column-major
            _lib.vcth_render_gbuffer(self._h, model_scale, C.byref(cam), w, h, sd.ctypes.data,
                                     sd.shape[0], vp.ctypes.data, planes.ctypes.data)
        else:
            vp = np.eye(4, dtype=np.float32)
            _lib.vcth_render_gbuffer(self._h, model_scale, C.byref(cam), w, h, None, 0,
                                     vp.ctypes.data, planes.ctypes.data)
        return planes
