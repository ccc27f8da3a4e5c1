"""A small self-contained textured scene for the reference-GLSL fixtures (tests/golden/ref_*.npz).

Pure numpy, seeded; no oracle, no GPU, no reference.  Every material carries a diffuse, a specular and a height map
(the reference's Draw_Mesh leaves stale sampler bindings for missing maps, R/Mesh.h:91-108 -- SURVEY.md Appendix B says
not to reproduce that, so the fixtures avoid it).  Triangles are grouped by material, in material order: one GL mesh per
material then draws them in the order the oracle walks the triangle list.
"""
import numpy as np


def _quad(p0, du, dv, nu=1, nv=1, uv_scale=(1.0, 1.0)):
    """Grid of nu x nv quads spanning p0 + s*du + t*dv, CCW seen from cross(du, dv).  Returns (tris [n,3,3], uvs [n,3,2])."""
    p0, du, dv = (np.asarray(a, np.float64) for a in (p0, du, dv))
    tris, uvs = [], []
    for j in range(nv):
        for i in range(nu):
            s0, s1, t0, t1 = i / nu, (i + 1) / nu, j / nv, (j + 1) / nv
            c = [(s0, t0), (s1, t0), (s1, t1), (s0, t1)]
            P = [p0 + s * du + t * dv for s, t in c]
            U = [(s * uv_scale[0], t * uv_scale[1]) for s, t in c]
            for a, b, d in ((0, 1, 2), (0, 2, 3)):
                tris.append([P[a], P[b], P[d]])
                uvs.append([U[a], U[b], U[d]])
    return np.array(tris), np.array(uvs)


def _box(center, half, rot_y_deg, uv_scale):
    c = np.asarray(center, np.float64)
    th = np.deg2rad(rot_y_deg)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    hx, hy, hz = half
    faces = [((-hx, -hy, hz), (2 * hx, 0, 0), (0, 2 * hy, 0)), ((hx, -hy, -hz), (-2 * hx, 0, 0), (0, 2 * hy, 0)),
             ((hx, -hy, hz), (0, 0, -2 * hz), (0, 2 * hy, 0)), ((-hx, -hy, -hz), (0, 0, 2 * hz), (0, 2 * hy, 0)),
             ((-hx, hy, hz), (2 * hx, 0, 0), (0, 0, -2 * hz)), ((-hx, -hy, -hz), (2 * hx, 0, 0), (0, 0, 2 * hz))]
    T, U = [], []
    for p0, du, dv in faces:
        t, u = _quad(R @ np.array(p0) + c, R @ np.array(du), R @ np.array(dv), 1, 1, uv_scale)
        T.append(t)
        U.append(u)
    return np.concatenate(T), np.concatenate(U)


def _smooth(r, h, w, cells):
    lat = r.uniform(size=(cells + 1, cells + 1))
    ys, xs = (np.arange(h) + 0.5) * cells / h, (np.arange(w) + 0.5) * cells / w
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = lat[y0][:, x0] * (1 - fx) + lat[y0][:, x0 + 1] * fx
    b = lat[y0 + 1][:, x0] * (1 - fx) + lat[y0 + 1][:, x0 + 1] * fx
    return a * (1 - fy) + b * fy


def _textures(r, size, cutout):
    """(diffuse, specular, height) uint8 [size, size, 4]."""
    base = r.uniform(0.25, 0.95, 3)
    n = _smooth(r, size, size, 4)
    checker = ((np.arange(size)[:, None] // max(size // 8, 1) + np.arange(size)[None, :] // max(size // 8, 1)) & 1)
    d = np.zeros((size, size, 4), np.float64)
    d[..., :3] = base * (0.55 + 0.45 * n[..., None]) * (0.8 + 0.2 * checker[..., None])
    d[..., 3] = 1.0
    if cutout:      # alpha holes: the discard at trace.fs:171
        d[..., 3] = np.where(_smooth(r, size, size, 3) > 0.55, 0.0, 1.0)
    s = np.zeros((size, size, 4), np.float64)
    g = _smooth(r, size, size, 2)
    s[..., 0] = 0.2 + 0.6 * g
    if not cutout:
        s[..., 1], s[..., 2] = 0.15 + 0.5 * g, 0.1 + 0.7 * (1 - g)
    # cut-out materials keep g = b = 0: the .rrra rule of trace.fs:210
    s[..., 3] = 1.0
    hmap = np.zeros((size, size, 4), np.float64)
    hmap[..., 0] = hmap[..., 1] = hmap[..., 2] = _smooth(r, size, size, 6)
    hmap[..., 3] = 1.0
    q = lambda a: np.clip(np.floor(a * 255.0 + 0.5), 0, 255).astype(np.uint8)
    return q(d), q(s), q(hmap)


def build(seed=11, tex_size=32):
    """Returns a dict: pos [n,9], nrm/tan/bit [n,9], uv [n,6], material [n], albedo [m,4], specular [m,3],
    mat_tex [m,3], textures (list of uint8 [h,w,4]).  Model units (world = 0.05 * model, VCT.h:183)."""
    r = np.random.default_rng(seed)
    groups = []
    # material 0: floor + back wall + left wall (large surfaces: magnified and minified texels, shadow receivers)
    t0, u0 = _quad((-1200, -900, 1200), (2400, 0, 0), (0, 0, -2400), 3, 3, (6.0, 6.0))
    t1, u1 = _quad((-1200, -900, -1200), (2400, 0, 0), (0, 1500, 0), 2, 2, (3.0, 2.0))
    t2, u2 = _quad((-1200, -900, 1200), (0, 0, -2400), (0, 1500, 0), 2, 2, (9.0, 5.0))
    groups.append((np.concatenate([t0, t1, t2]), np.concatenate([u0, u1, u2])))
    # material 1: two boxes (casters; oblique faces for every dominant axis of the voxelizer)
    b0, ub0 = _box((-350, -500, -300), (260, 400, 220), 27.0, (2.0, 3.0))
    b1, ub1 = _box((480, -640, 250), (230, 260, 230), -18.0, (1.0, 1.0))
    ramp, ur = _quad((-200, -880, 700), (900, 300, 0), (0, 420, -500), 2, 2, (2.0, 2.0))
    groups.append((np.concatenate([b0, b1, ramp]), np.concatenate([ub0, ub1, ur])))
    # material 2: an alpha cut-out sheet standing in front of the back wall, drawn double sided as two opposite quads
    s0, us0 = _quad((-700, -900, -600), (1300, 0, 250), (0, 1100, 0), 2, 2, (2.0, 2.0))
    s1, us1 = _quad((600, -900, -350), (-1300, 0, -250), (0, 1100, 0), 2, 2, (2.0, 2.0))
    groups.append((np.concatenate([s0, s1]), np.concatenate([us0, us1])))

    pos, uv, mat, nrm, tan, bit = [], [], [], [], [], []
    for m, (t, u) in enumerate(groups):
        for tri, tuv in zip(t, u):
            e1, e2 = tri[1] - tri[0], tri[2] - tri[0]
            n = np.cross(e1, e2)
            n /= np.linalg.norm(n)
            d1, d2 = tuv[1] - tuv[0], tuv[2] - tuv[0]
            det = d1[0] * d2[1] - d2[0] * d1[1]
            tg = (e1 * d2[1] - e2 * d1[1]) / det
            tg -= n * np.dot(n, tg)
            tg /= np.linalg.norm(tg)
            bt = np.cross(n, tg)
            pos.append(tri.reshape(9))
            uv.append(tuv.reshape(6))
            mat.append(m)
            nrm.append(np.tile(n, 3))
            tan.append(np.tile(tg, 3))
            bit.append(np.tile(bt, 3))
    textures, mat_tex = [], []
    for m in range(len(groups)):
        d, s, h = _textures(r, tex_size if m != 0 else tex_size * 2, cutout=(m == 2))
        mat_tex.append([len(textures), len(textures) + 1, len(textures) + 2])
        textures += [d, s, h]
    f32 = lambda a: np.ascontiguousarray(np.array(a), np.float32)
    return dict(pos=f32(pos), uv=f32(uv), material=np.array(mat, np.int32), nrm=f32(nrm), tan=f32(tan), bit=f32(bit),
                albedo=np.ones((len(groups), 4), np.float32), specular=np.zeros((len(groups), 3), np.float32),
                mat_tex=np.array(mat_tex, np.int32), textures=textures)


def gl_vertices(scene, material):
    """struct Vertex rows (R/Mesh.h:12-19: Position, Normal, TexCoords, Tangents, Bi_Tangents) of one material's
    triangles, unindexed: float32 [3n, 14]."""
    sel = np.nonzero(scene["material"] == material)[0]
    n = len(sel)
    v = np.zeros((n, 3, 14), np.float32)
    v[:, :, 0:3] = scene["pos"][sel].reshape(n, 3, 3)
    v[:, :, 3:6] = scene["nrm"][sel].reshape(n, 3, 3)
    v[:, :, 6:8] = scene["uv"][sel].reshape(n, 3, 2)
    v[:, :, 8:11] = scene["tan"][sel].reshape(n, 3, 3)
    v[:, :, 11:14] = scene["bit"][sel].reshape(n, 3, 3)
    return v.reshape(3 * n, 14)
