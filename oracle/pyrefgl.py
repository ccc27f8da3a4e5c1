"""ctypes binding of oracle/_ref/libvct_refgl.so -- the REFERENCE'S OWN GLSL executed on Mesa llvmpipe.

TEST INFRASTRUCTURE ONLY, and build-container only: the shaders are read at run time from
/root/reference/Voxel_Cone_Tracing_Final/Shader (unmodified; nothing is copied), which does not exist on the GPU
box.  Only tests/golden/make_ref_golden.py and the `not gpu` tests that re-check the committed fixtures import this
module; `available()` says whether both the reference tree and the Mesa software driver are present.

The matrix helpers restate the glm calls the reference's host makes (glm itself is not vendored in the reference and
not installed here): glm::lookAt / ortho / perspective / scale as published (right-handed, depth -1..1), in fp32.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SHADER_DIR = os.environ.get("VCT_REFERENCE_SHADERS", "/root/reference/Voxel_Cone_Tracing_Final/Shader")
DRI_DRIVER = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"
SO = os.path.join(_HERE, "_ref", "libvct_refgl.so")
# llvmpipe's default texture filtering uses 8-bit fixed-point weights and approximated rho / lod; these switches
# select its full-precision float paths (what a hardware GL implementation is required to be at least as good as).
PRECISE = "no_aos_sampling,no_rho_approx,no_brilinear,no_quad_lod"

_LIB = None


def available():
    return os.path.isdir(SHADER_DIR) and os.path.exists(DRI_DRIVER)


class FrameParams(C.Structure):
    _fields_ = [("camera_pos", C.c_float * 3), ("light_dir", C.c_float * 3), ("grid_world_size", C.c_float),
                ("voxel_dim", C.c_int32), ("ambient_factor", C.c_float), ("model", C.c_float * 16),
                ("model_view", C.c_float * 16), ("projection", C.c_float * 16), ("depth_mvp", C.c_float * 16)]


def build():
    src = os.path.join(_HERE, "ref_gl.c")
    if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
        subprocess.check_call(["make", "-C", _HERE, "ref"])
    return SO


def lib(precise=True, win=1024, threads=1):
    """Loads the harness and creates the context (once per process: `precise` is fixed by the first call, because
    llvmpipe reads GALLIVM_PERF when the driver is loaded)."""
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError("reference shaders or Mesa swrast driver not present (build container only)")
        if precise:
            os.environ["GALLIVM_PERF"] = PRECISE
        else:
            os.environ.pop("GALLIVM_PERF", None)
        # One rasteriser thread: llvmpipe then draws bin by bin in submission order, so the reference's unordered
        # imageStore (vox.fs:88: overlapping fragments race) comes out the same on every run.  Which of several writers
        # of a voxel wins is still GL's choice (bins, not triangles, are the outer loop) -- the fixtures record how many
        # triangles write each voxel and the tests compare such voxels against the set of candidates.
        # (`threads` != 1: timing runs only -- tools/time_ref_gl.py)
        os.environ["LP_NUM_THREADS"] = str(threads)
        L = C.CDLL(build())
        L.refgl_log.restype = C.c_char_p
        L.refgl_string.restype = C.c_char_p
        if L.refgl_init(SHADER_DIR.encode(), win, win) != 0:
            raise RuntimeError("refgl_init: " + L.refgl_log().decode())
        _LIB = L
    return _LIB


def _chk(rc):
    if rc < 0:
        raise RuntimeError(_LIB.refgl_log().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def gl_strings():
    L = lib()
    return dict(version=L.refgl_string(0).decode(), renderer=L.refgl_string(1).decode(),
                glsl=L.refgl_string(2).decode())


# ---- glm restated (column-major float32[16], element [col*4 + row]) -----------------------------------------------
def _norm(v):
    v = np.asarray(v, np.float32)
    return (v * np.float32(1.0 / np.sqrt(np.float32(np.dot(v, v))))).astype(np.float32)


def look_at(eye, center, up):
    eye, center, up = (np.asarray(a, np.float32) for a in (eye, center, up))
    f = _norm(center - eye)
    s = _norm(np.cross(f, up).astype(np.float32))
    u = np.cross(s, f).astype(np.float32)
    m = np.eye(4, dtype=np.float32)          # m[row, col]
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    return np.ascontiguousarray(m.T).reshape(16)


def ortho(l, r, b, t, n, f):
    l, r, b, t, n, f = (np.float32(x) for x in (l, r, b, t, n, f))
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = np.float32(2) / (r - l), np.float32(2) / (t - b), -np.float32(2) / (f - n)
    m[0, 3], m[1, 3], m[2, 3] = -(r + l) / (r - l), -(t + b) / (t - b), -(f + n) / (f - n)
    return np.ascontiguousarray(m.T).reshape(16)


def perspective(fovy_rad, aspect, n, f):
    fovy_rad, aspect, n, f = (np.float32(x) for x in (fovy_rad, aspect, n, f))
    th = np.float32(np.tan(fovy_rad / np.float32(2)))
    m = np.zeros((4, 4), np.float32)
    m[0, 0] = np.float32(1) / (aspect * th)
    m[1, 1] = np.float32(1) / th
    m[2, 2] = -(f + n) / (f - n)
    m[3, 2] = -np.float32(1)
    m[2, 3] = -(np.float32(2) * f * n) / (f - n)
    return np.ascontiguousarray(m.T).reshape(16)


def scale(s):
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = m[1, 1] = m[2, 2] = np.float32(s)
    return m.reshape(16)


def mul(a, b):
    """a * b for column-major float32[16] (fp32 accumulation like glm's operator*)."""
    A = np.asarray(a, np.float32).reshape(4, 4).T
    B = np.asarray(b, np.float32).reshape(4, 4).T
    out = np.zeros((4, 4), np.float32)
    for c in range(4):
        acc = A[:, 0] * B[0, c]
        for k in range(1, 4):
            acc = (acc + A[:, k] * B[k, c]).astype(np.float32)
        out[:, c] = acc
    return np.ascontiguousarray(out.T).reshape(16)


def identity():
    return np.eye(4, dtype=np.float32).reshape(16)


def depth_view_proj(light_dir):
    """VCT.h:84-86"""
    v = look_at(light_dir, (0, 0, 0), (0, 1, 0))
    p = ortho(-120, 120, -120, 120, -100, 100)
    return mul(p, v)


def voxel_projections(G):
    """VCT.h:128-134 -> (ProjX, ProjY, ProjZ)"""
    G = np.float32(G)
    h = G * np.float32(0.5)
    p = ortho(-h, h, -h, h, h, G * np.float32(1.5))
    return (mul(p, look_at((G, 0, 0), (0, 0, 0), (0, 1, 0))), mul(p, look_at((0, G, 0), (0, 0, 0), (0, 0, -1))),
            mul(p, look_at((0, 0, G), (0, 0, 0), (0, 1, 0))))


def frame_params(V, G=150.0, camera_pos=(0, 4, 0), light_dir=(0, 1, 0.25), ambient=0.1, model=None, view=None,
                 projection=None, depth_vp=None):
    fp = FrameParams()
    fp.camera_pos[:] = [float(x) for x in camera_pos]
    fp.light_dir[:] = [float(x) for x in light_dir]
    fp.grid_world_size, fp.voxel_dim, fp.ambient_factor = float(G), int(V), float(ambient)
    model = identity() if model is None else _f32(model)
    view = identity() if view is None else _f32(view)
    projection = identity() if projection is None else _f32(projection)
    depth_vp = depth_view_proj(light_dir) if depth_vp is None else _f32(depth_vp)
    fp.model[:] = model.tolist()
    fp.model_view[:] = mul(view, model).tolist()                 # VCT.h:185
    fp.projection[:] = projection.tolist()
    fp.depth_mvp[:] = mul(depth_vp, model).tolist()              # VCT.h:187
    return fp


# ---- volume -------------------------------------------------------------------------------------------------------
def volume_create(V):
    _chk(lib().refgl_volume_create(int(V)))


def volume_set_level(level, texels):
    t = np.ascontiguousarray(texels, np.uint8)
    _chk(lib().refgl_volume_set_level(int(level), _p(t)))


def volume_generate_mipmap():
    _chk(lib().refgl_volume_generate_mipmap())


def volume_get_level(V, level):
    N = V >> level
    out = np.zeros((N, N, N, 4), np.uint8)
    _chk(lib().refgl_volume_get_level(int(level), _p(out)))
    return out


def volume_set_wrap(clamp):
    _chk(lib().refgl_volume_set_wrap(int(bool(clamp))))


def volume_upload_chain(V, level_arrays):
    """level_arrays[k]: uint8 [N,N,N,4] (z,y,x) -- every level uploaded as given (no glGenerateMipmap)."""
    volume_create(V)
    for k, a in enumerate(level_arrays):
        volume_set_level(k, a)


# ---- textures / meshes --------------------------------------------------------------------------------------------
def texture_create(img):
    """img: uint8 [h, w, c], c in (1, 3, 4); row 0 = v 0 (what glTexImage2D makes of stb_image's rows)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    return _chk(lib().refgl_texture_create(w, h, c, _p(img)))


def texture_create_f32(img):
    img = _f32(img)
    h, w, c = img.shape
    assert c == 4
    return _chk(lib().refgl_texture_create_f32(w, h, _p(img)))


def texture_get_level(tex, level):
    w, h = C.c_int(), C.c_int()
    _chk(lib().refgl_texture_get_level(tex, level, None, C.byref(w), C.byref(h)))
    if w.value == 0:
        return None
    out = np.zeros((h.value, w.value, 4), np.uint8)
    _chk(lib().refgl_texture_get_level(tex, level, _p(out), C.byref(w), C.byref(h)))
    return out


TEX_DIFFUSE, TEX_SPECULAR, TEX_NORMAL, TEX_HEIGHT = 0, 1, 2, 3


def mesh_create(verts, indices, textures=()):
    """verts [n,14] (Position, Normal, TexCoords, Tangents, Bi_Tangents); textures: [(handle, TEX_*)]."""
    verts = _f32(verts)
    idx = np.ascontiguousarray(indices, np.uint32)
    th = np.array([t[0] for t in textures], np.int32)
    tt = np.array([t[1] for t in textures], np.int32)
    return _chk(lib().refgl_mesh_create(_p(verts), len(verts), _p(idx), idx.size, _p(th), _p(tt), len(textures)))


# ---- passes -------------------------------------------------------------------------------------------------------
def shadow_create(S):
    _chk(lib().refgl_shadow_create(int(S)))


def draw_depth_texture(depth_mvp, meshes):
    m = np.asarray(meshes, np.int32)
    d = _f32(depth_mvp)
    _chk(lib().refgl_draw_depth_texture(_p(d), _p(m), len(m)))


def shadow_get(S):
    out = np.zeros((S, S), np.float32)
    _chk(lib().refgl_shadow_get(_p(out)))
    return out


def shadow_set(depth):
    d = _f32(depth)
    _chk(lib().refgl_shadow_set(_p(d)))


def draw_voxel_texture(G, model, depth_mvp, meshes, generate_mipmap=True):
    px, py, pz = voxel_projections(G)
    m = np.asarray(meshes, np.int32)
    model, depth_mvp = _f32(model), _f32(depth_mvp)
    _chk(lib().refgl_draw_voxel_texture(_p(model), _p(depth_mvp), _p(px), _p(py), _p(pz), _p(m), len(m),
                                        int(generate_mipmap)))


def render(W, H, fp, meshes, want_depth=False):
    m = np.asarray(meshes, np.int32)
    out = np.zeros((H, W, 4), np.float32)
    dep = np.zeros((H, W), np.float32) if want_depth else None
    _chk(lib().refgl_render(W, H, C.byref(fp), _p(m), len(m), _p(out), _p(dep)))
    return (out, dep) if want_depth else out


def trace_points(W, H, fp, verts, tex_albedo, tex_spec, tex_height):
    verts = _f32(verts)
    assert verts.shape == (W * H, 14)
    out = np.zeros((H, W, 4), np.float32)
    _chk(lib().refgl_trace_points(W, H, C.byref(fp), _p(verts), tex_albedo, tex_spec, tex_height, _p(out)))
    return out
