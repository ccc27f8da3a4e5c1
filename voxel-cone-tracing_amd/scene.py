"""ctypes binding of libvct_host.so: scenes (procedural, or a Wavefront OBJ + MTL with texture maps),
camera and light matrices.  See host/vct_host.h for what each stands in for in the reference."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvct_host.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make host`")
_lib = C.CDLL(LIB_PATH)

CORNELL, ATRIUM, ATRIUM_TEXTURED, BISTRO = 0, 1, 2, 3


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
_lib.vcth_scene_save.argtypes = [C.c_void_p, C.c_char_p]
_lib.vcth_scene_load_cache.restype = C.c_void_p
_lib.vcth_scene_load_cache.argtypes = [C.c_char_p, C.c_char_p]
_lib.vcth_scene_get_uvs.argtypes = [C.c_void_p, C.c_void_p]
_lib.vcth_scene_num_textures.argtypes = [C.c_void_p]
_lib.vcth_scene_texture_info.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
_lib.vcth_scene_get_texture.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
_lib.vcth_scene_get_material_textures.argtypes = [C.c_void_p, C.c_void_p]


_lib.vcth_image_load.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]


def load_image(path):
    """Decode a PNG / JPEG / BMP / TGA / PPM file with the OBJ reader's decoders: uint8 [h, w, 4], row 0 = bottom row.
    Raises ValueError for an unreadable, unsupported or corrupt file."""
    w, h = C.c_int32(), C.c_int32()
    if _lib.vcth_image_load(os.fsencode(path), C.byref(w), C.byref(h), None, 0) != 0:
        raise ValueError(f"cannot decode {path}")
    out = np.zeros((h.value, w.value, 4), np.uint8)
    w2, h2 = C.c_int32(), C.c_int32()
    if _lib.vcth_image_load(os.fsencode(path), C.byref(w2), C.byref(h2), out.ctypes.data, out.nbytes) != 0 or \
            (w2.value, h2.value) != (w.value, h.value):
        raise ValueError(f"cannot decode {path} (or it changed while it was being read)")
    return out


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
            if os.fsdecode(kind).endswith(".vctscene"):           # on-disk cache written by Scene.save()
                self._h = _lib.vcth_scene_load_cache(os.fsencode(kind), err)
            else:
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
        self.uv = np.zeros((self.ntri, 6), np.float32)
        _lib.vcth_scene_get_uvs(self._h, self.uv.ctypes.data)
        self.mat_tex = np.full((self.nmat, 3), -1, np.int32)       # diffuse / specular / height texture or -1
        _lib.vcth_scene_get_material_textures(self._h, self.mat_tex.ctypes.data)
        self.textures = []                                         # uint8 [h, w, 4], row 0 at v = 0
        for i in range(_lib.vcth_scene_num_textures(self._h)):
            w, h = C.c_int32(), C.c_int32()
            _lib.vcth_scene_texture_info(self._h, i, C.byref(w), C.byref(h))
            t = np.zeros((h.value, w.value, 4), np.uint8)
            _lib.vcth_scene_get_texture(self._h, i, t.ctypes.data)
            self.textures.append(t)

    def save(self, path):
        """Write the on-disk cache of this scene (load it back with Scene(path); use the suffix .vctscene)."""
        if _lib.vcth_scene_save(self._h, os.fsencode(path)) != 0:
            raise OSError(f"cannot write {path}")

    def frames(self):
        """Per-vertex (normal, tangent, bitangent), each float32 [ntri, 9]."""
        out = [np.zeros((self.ntri, 9), np.float32) for _ in range(3)]
        _lib.vcth_scene_get_frames(self._h, *(a.ctypes.data for a in out))
        return out

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.vcth_scene_destroy(self._h)
            self._h = None
