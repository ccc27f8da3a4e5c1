"""ctypes binding of the CPU oracle (oracle/libvct_oracle.so).

TEST INFRASTRUCTURE ONLY -- pinned against the reference's own GLSL (oracle/vct_oracle.h, tests/test_ref_gl.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GB_PLANES = 23
GB_P, GB_NW, GB_TW, GB_BW, GB_BUMPN, GB_ALBEDO, GB_SPEC, GB_SHADOW = 0, 3, 6, 9, 12, 15, 19, 22


class Params(C.Structure):
    _fields_ = [
        ("V", C.c_int32),
        ("G", C.c_float),
        ("camera_pos", C.c_float * 3),
        ("light_dir", C.c_float * 3),
        ("ambient_factor", C.c_float),
        ("shininess", C.c_float),
        ("max_distance", C.c_float),
        ("max_alpha", C.c_float),
        ("tan_diffuse", C.c_float),
        ("tan_specular", C.c_float),
        ("wrap_repeat", C.c_int32),
    ]


class Texture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("mips", C.c_void_p), ("nlev", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [
        ("pos", C.c_void_p),
        ("material", C.c_void_p),
        ("albedo", C.c_void_p),
        ("ntri", C.c_int32),
        ("nmat", C.c_int32),
        ("model_scale", C.c_float),
        ("shadow_depth", C.c_void_p),
        ("shadow_size", C.c_int32),
        ("light_vp", C.c_float * 16),
        ("uv", C.c_void_p),
        ("mat_tex", C.c_void_p),
        ("textures", C.c_void_p),
        ("ntex", C.c_int32),
    ]


class Mesh(C.Structure):
    _fields_ = [
        ("pos", C.c_void_p), ("nrm", C.c_void_p), ("tan", C.c_void_p), ("bit", C.c_void_p),
        ("uv", C.c_void_p), ("material", C.c_void_p), ("albedo", C.c_void_p), ("specular", C.c_void_p),
        ("mat_tex", C.c_void_p), ("textures", C.c_void_p),
        ("ntri", C.c_int32), ("nmat", C.c_int32), ("ntex", C.c_int32), ("model_scale", C.c_float),
    ]


def build():
    """Compile the oracle with gcc if the .so is missing or stale."""
    so = os.path.join(_HERE, "libvct_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("vct_oracle.cpp", "vct_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libvct_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.vcto_level_offset_texels.restype = C.c_size_t
        L.vcto_chain_texels.restype = C.c_size_t
        L.vcto_trace.restype = C.c_uint64
        L.vcto_f32_to_f16.restype = C.c_uint16
        L.vcto_f32_to_f16.argtypes = [C.c_float]
        L.vcto_f16_to_f32.restype = C.c_float
        L.vcto_f16_to_f32.argtypes = [C.c_uint16]
        L.vcto_shadow_tex.restype = C.c_float
        L.vcto_shadow_tex.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.vcto_pcf25.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float]
        L.vcto_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.vcto_cone.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_float, C.c_void_p]
        L.vcto_max_steps.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.vcto_voxel_proj.argtypes = [C.c_float, C.c_int, C.c_void_p]
        L.vcto_frag_to_voxel.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.c_void_p]
        L.vcto_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vcto_trace_aniso.restype = C.c_uint64
        L.vcto_trace_aniso.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vcto_build_mips_aniso.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vcto_bounce.restype = C.c_uint64
        L.vcto_bounce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vcto_voxelize_conservative_attr.argtypes = [C.c_void_p] * 6
        L.vcto_tex_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.vcto_render_shadow_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.vcto_render_gbuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def default_params(V=128, **kw):
    p = Params()
    lib().vcto_default_params(C.byref(p))
    p.V = V
    for k, v in kw.items():
        if k in ("camera_pos", "light_dir"):
            getattr(p, k)[:] = list(v)
        else:
            setattr(p, k, v)
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def num_levels(V):
    return lib().vcto_num_levels(V)


def level_offset(V, level):
    return lib().vcto_level_offset_texels(V, level)


def chain_texels(V):
    return lib().vcto_chain_texels(V)


def build_mips(l0):
    """l0: uint8 [V,V,V,4] (z,y,x order).  Returns the linear chain uint8 [chain_texels,4]."""
    V = l0.shape[0]
    chain = np.zeros((chain_texels(V), 4), np.uint8)
    chain[: V ** 3] = l0.reshape(-1, 4)
    lib().vcto_build_mips(_ptr(chain), V)
    return chain


def level_view(chain, V, level):
    N = V >> level
    o = level_offset(V, level)
    return chain[o:o + N ** 3].reshape(N, N, N, 4)


def sample(p, chain, pos, lod):
    pos = np.asarray(pos, np.float32)
    out = np.zeros(4, np.float32)
    lib().vcto_sample(C.byref(p), _ptr(chain), _ptr(pos), float(lod), _ptr(out))
    return out


def cone(p, chain, P, Nw, d, tan_half):
    P, Nw, d = (np.ascontiguousarray(a, np.float32) for a in (P, Nw, d))
    out = np.zeros(4, np.float32)
    steps = lib().vcto_cone(C.byref(p), _ptr(chain), _ptr(P), _ptr(Nw), _ptr(d),
                            float(tan_half), _ptr(out))
    return out, steps


def cone_constants():
    d = np.zeros(18, np.float32)
    w = np.zeros(6, np.float32)
    lib().vcto_cone_constants(_ptr(d), _ptr(w))
    return d.reshape(6, 3), w


def max_steps(p, tan_half):
    lod = C.c_float()
    n = lib().vcto_max_steps(C.byref(p), float(tan_half), C.byref(lod))
    return n, lod.value


def trace(p, chain, planes, nthreads=1, want_cones=False):
    """planes: float32 [23, npix].  Returns dict(rgba32f, rgba16f, steps, cones, total_steps)."""
    planes = np.ascontiguousarray(planes, np.float32)
    assert planes.shape[0] == GB_PLANES
    npix = planes.shape[1]
    o32 = np.zeros((npix, 4), np.float32)
    o16 = np.zeros((npix, 4), np.uint16)
    steps = np.zeros((npix, 7), np.uint8)
    cones = np.zeros((npix, 7, 4), np.float32) if want_cones else None
    total = lib().vcto_trace(C.byref(p), _ptr(chain), _ptr(planes), npix, _ptr(o32), _ptr(o16),
                             _ptr(steps), _ptr(cones), int(nthreads))
    return dict(rgba32f=o32, rgba16f=o16, steps=steps, cones=cones, total_steps=int(total))


def f32_to_f16(x):
    return lib().vcto_f32_to_f16(float(x))


def f16_to_f32(h):
    return lib().vcto_f16_to_f32(int(h))


def voxel_proj(G, axis):
    m = np.zeros(16, np.float32)
    lib().vcto_voxel_proj(float(G), int(axis), _ptr(m))
    return m.reshape(4, 4).T.copy()   # row-major 4x4 (m was column-major)


def dominant_axis(v0, v1, v2):
    a, b, c = (np.ascontiguousarray(v, np.float32) for v in (v0, v1, v2))
    return lib().vcto_dominant_axis(_ptr(a), _ptr(b), _ptr(c))


def frag_to_voxel(V, axis, fx, fy, fz):
    out = np.zeros(3, np.int32)
    lib().vcto_frag_to_voxel(int(V), int(axis), float(fx), float(fy), float(fz), _ptr(out))
    return out


def shadow_tex(depth, u, v):
    depth = np.ascontiguousarray(depth, np.float32)
    return lib().vcto_shadow_tex(_ptr(depth), depth.shape[0], float(u), float(v))


def pcf25(depth, coord, bias=0.002):
    depth = np.ascontiguousarray(depth, np.float32)
    c = np.ascontiguousarray(coord, np.float32)
    return lib().vcto_pcf25(_ptr(depth), depth.shape[0], _ptr(c), float(bias))


def pcf25_batch(depth, coords, bias=0.002):
    """pcf25 for [n, 3] shadow coordinates -> int32 [n]."""
    depth = np.ascontiguousarray(depth, np.float32)
    c = np.ascontiguousarray(coords, np.float32).reshape(-1, 3)
    out = np.zeros(c.shape[0], np.int32)
    lib().vcto_pcf25_batch(_ptr(depth), depth.shape[0], _ptr(c), C.c_size_t(c.shape[0]), C.c_float(bias), _ptr(out))
    return out


def tex_build_mips(texture):
    """uint8 [h, w, 4] -> (chain uint8 [texels, 4] with level 0 first, number of levels): glGenerateMipmap restated."""
    t = np.ascontiguousarray(texture, np.uint8)
    h, w = t.shape[:2]
    L = lib()
    L.vcto_tex_level_offset.restype = C.c_size_t
    nlev = L.vcto_tex_num_levels(w, h)
    chain = np.zeros((L.vcto_tex_level_offset(w, h, nlev), 4), np.uint8)
    L.vcto_tex_build_mips(_ptr(t), w, h, _ptr(chain))
    return chain, nlev


def tex_level(texture, k):
    """Level k of a texture's mip chain as uint8 [hk, wk, 4]."""
    t = np.ascontiguousarray(texture, np.uint8)
    h, w = t.shape[:2]
    chain, nlev = tex_build_mips(t)
    L = lib()
    off = L.vcto_tex_level_offset(w, h, k)
    wk, hk = max(1, w >> k), max(1, h >> k)
    return chain[off:off + wk * hk].reshape(hk, wk, 4)


def _texture_table(textures, keep, mipmaps=False, chains=None):
    """textures: list of uint8 [h, w, 4] arrays -> (ctypes array of Texture, count).  mipmaps: attach the mip chain
    (mip-mapped sampling with implicit derivatives, the reference's sampler state); else level-0 sampling.
    chains: optional list of ready-made chains (uint8 [texels, 4], level 0 first, the layout of tex_build_mips) used
    instead of the oracle's own box filter -- tests feed the chain a GL driver generated (tests/test_ref_gl.py)."""
    if not textures:
        return None, 0
    arr = (Texture * len(textures))()
    for i, t in enumerate(textures):
        t = np.ascontiguousarray(t, np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        keep.append(t)
        arr[i].rgba, arr[i].height, arr[i].width = _ptr(t), t.shape[0], t.shape[1]
        arr[i].mips, arr[i].nlev = None, 1
        if mipmaps:
            chain, nlev = tex_build_mips(t)
            if chains is not None:
                assert chains[i].shape == chain.shape
                chain = np.ascontiguousarray(chains[i], np.uint8)
            keep.append(chain)
            arr[i].mips, arr[i].nlev = _ptr(chain), nlev
    keep.append(arr)
    return arr, len(textures)


def tex_sample(texture, u, v, duv=None):
    """texture(sampler, (u, v)).  duv = (ds_dx, dt_dx, ds_dy, dt_dy): mip-mapped with those quad differences."""
    keep = []
    arr, _ = _texture_table([texture], keep, mipmaps=duv is not None)
    out = np.zeros(4, np.float32)
    L = lib()
    if duv is None:
        L.vcto_tex_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.vcto_tex_sample(C.byref(arr[0]), float(u), float(v), _ptr(out))
    else:
        L.vcto_tex_sample_lod.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p]
        L.vcto_tex_sample_lod(C.byref(arr[0]), float(u), float(v), *[float(x) for x in duv], _ptr(out))
    return out


def log2_det(x):
    L = lib()
    L.vcto_log2_det.restype = C.c_float
    L.vcto_log2_det.argtypes = [C.c_float]
    return L.vcto_log2_det(float(x))


def make_mesh(pos, material, albedo, specular=None, frames=None, uv=None, mat_tex=None, textures=None,
              model_scale=0.05, mipmaps=False, tex_chains=None):
    """Input of the raster oracles (render_shadow_map / render_gbuffer).  frames = (normal, tangent, bitangent)."""
    m = Mesh()
    k = m._keep = []

    def arr(a, dt, shape):
        a = np.ascontiguousarray(a, dt).reshape(shape)
        k.append(a)
        return _ptr(a)
    m.pos = arr(pos, np.float32, (-1, 9))
    m.ntri = k[-1].shape[0]
    m.material = arr(material, np.int32, (-1,))
    m.albedo = arr(albedo, np.float32, (-1, 4))
    m.nmat = k[-1].shape[0]
    m.specular = arr(specular if specular is not None else np.zeros((m.nmat, 3)), np.float32, (-1, 3))
    if frames is not None:
        m.nrm, m.tan, m.bit = (arr(f, np.float32, (-1, 9)) for f in frames)
    m.uv = arr(uv, np.float32, (-1, 6)) if uv is not None else None
    m.mat_tex = arr(mat_tex, np.int32, (-1, 3)) if mat_tex is not None else None
    tab, n = _texture_table(textures, k, mipmaps, tex_chains)
    m.textures = C.cast(tab, C.c_void_p) if tab is not None else None
    m.ntex = n
    m.model_scale = model_scale
    return m


def render_shadow_map(mesh, light_vp_colmajor, size):
    """DrawDepthTexture on the CPU: depth [size, size] fp32."""
    vp = np.ascontiguousarray(light_vp_colmajor, np.float32).reshape(16)
    depth = np.zeros((size, size), np.float32)
    lib().vcto_render_shadow_map(C.byref(mesh), _ptr(vp), size, _ptr(depth))
    return depth


def render_gbuffer(mesh, view_proj_colmajor, w, h, shadow_depth=None, light_vp_colmajor=None):
    """The raster + non-cone fragment work of Render() on the CPU: planes float32 [23, w*h]."""
    vp = np.ascontiguousarray(view_proj_colmajor, np.float32).reshape(16)
    planes = np.zeros((GB_PLANES, w * h), np.float32)
    if shadow_depth is not None:
        sd = np.ascontiguousarray(shadow_depth, np.float32)
        lvp = np.ascontiguousarray(light_vp_colmajor, np.float32).reshape(16)
        lib().vcto_render_gbuffer(C.byref(mesh), _ptr(vp), w, h, _ptr(sd), sd.shape[0], _ptr(lvp), _ptr(planes))
    else:
        lvp = np.eye(4, dtype=np.float32).reshape(16)
        lib().vcto_render_gbuffer(C.byref(mesh), _ptr(vp), w, h, None, 0, _ptr(lvp), _ptr(planes))
    return planes


def make_scene(pos, material, albedo, model_scale=0.05, shadow_depth=None, light_vp=None, uv=None, mat_tex=None,
               textures=None, mipmaps=False, tex_chains=None):
    """Keeps references to the numpy arrays alive on the returned struct."""
    s = Scene()
    s._keep = [np.ascontiguousarray(pos, np.float32).reshape(-1, 9),
               np.ascontiguousarray(material, np.int32),
               np.ascontiguousarray(albedo, np.float32).reshape(-1, 4)]
    s.pos, s.material, s.albedo = (_ptr(a) for a in s._keep)
    s.ntri = s._keep[0].shape[0]
    s.nmat = s._keep[2].shape[0]
    s.model_scale = model_scale
    if shadow_depth is not None:
        sd = np.ascontiguousarray(shadow_depth, np.float32)
        s._keep.append(sd)
        s.shadow_depth = _ptr(sd)
        s.shadow_size = sd.shape[0]
    else:
        s.shadow_depth = None
        s.shadow_size = 0
    lv = np.eye(4, dtype=np.float32) if light_vp is None else np.asarray(light_vp, np.float32)
    s.light_vp[:] = list(lv.T.reshape(-1))   # row-major in -> column-major struct
    s.uv, s.mat_tex, s.textures, s.ntex = None, None, None, 0
    if uv is not None and mat_tex is not None and textures:
        a = np.ascontiguousarray(uv, np.float32).reshape(-1, 6)
        b = np.ascontiguousarray(mat_tex, np.int32).reshape(-1, 3)
        s._keep += [a, b]
        s.uv, s.mat_tex = _ptr(a), _ptr(b)
        tab, n = _texture_table(textures, s._keep, mipmaps, tex_chains)
        s.textures, s.ntex = C.cast(tab, C.c_void_p), n
    return s


def voxelize_reference(p, scene):
    V = p.V
    l0 = np.zeros((V, V, V, 4), np.uint8)
    lib().vcto_voxelize_reference(C.byref(p), C.byref(scene), _ptr(l0))
    return l0


def voxelize_conservative(p, scene, want_acc=False):
    V = p.V
    l0 = np.zeros((V, V, V, 4), np.uint8)
    acc = np.zeros((V, V, V, 4), np.uint32) if want_acc else None
    lib().vcto_voxelize_conservative(C.byref(p), C.byref(scene), _ptr(l0), _ptr(acc))
    return (l0, acc) if want_acc else l0


def voxelize_conservative_zslab(p, scene, z0, z1):
    """z-slices [z0, z1) of voxelize_conservative's level 0: uint8 [z1 - z0, V, V, 4]."""
    V = p.V
    l0 = np.zeros((z1 - z0, V, V, 4), np.uint8)
    lib().vcto_voxelize_conservative_zslab(C.byref(p), C.byref(scene), int(z0), int(z1), _ptr(l0))
    return l0


def voxelize_conservative_attr(p, scene):
    """Returns (l0, attr_albedo, attr_normal), each uint8 [V,V,V,4]."""
    V = p.V
    l0 = np.zeros((V, V, V, 4), np.uint8)
    alb = np.zeros((V, V, V, 4), np.uint8)
    nrm = np.zeros((V, V, V, 4), np.uint8)
    lib().vcto_voxelize_conservative_attr(C.byref(p), C.byref(scene), _ptr(l0), None, _ptr(alb), _ptr(nrm))
    return l0, alb, nrm


def bounce(p, chain0, attr_albedo, attr_normal, nthreads=1):
    """One re-injection bounce.  Returns (level0' uint8 [V,V,V,4], cone steps executed)."""
    V = p.V
    out = np.zeros((V, V, V, 4), np.uint8)
    a = np.ascontiguousarray(attr_albedo, np.uint8)
    n = np.ascontiguousarray(attr_normal, np.uint8)
    steps = lib().vcto_bounce(C.byref(p), _ptr(chain0), _ptr(a), _ptr(n), _ptr(out), int(nthreads))
    return out, int(steps)


def build_mips_aniso(l0):
    """Directional chains of level 0: uint8 [6, chain_texels - V^3, 4] (levels 1.. of each direction)."""
    l0 = np.ascontiguousarray(l0, np.uint8)
    V = l0.shape[0]
    out = np.zeros((6, chain_texels(V) - V ** 3, 4), np.uint8)
    lib().vcto_build_mips_aniso(_ptr(l0), V, _ptr(out))
    return out


def trace_aniso(p, chain, aniso, planes, nthreads=1, want_cones=False):
    planes = np.ascontiguousarray(planes, np.float32)
    npix = planes.shape[1]
    o32 = np.zeros((npix, 4), np.float32)
    o16 = np.zeros((npix, 4), np.uint16)
    steps = np.zeros((npix, 7), np.uint8)
    cones = np.zeros((npix, 7, 4), np.float32) if want_cones else None
    total = lib().vcto_trace_aniso(C.byref(p), _ptr(chain), _ptr(aniso), _ptr(planes), npix, _ptr(o32),
                                   _ptr(o16), _ptr(steps), _ptr(cones), int(nthreads))
    return dict(rgba32f=o32, rgba16f=o16, steps=steps, cones=cones, total_steps=int(total))
