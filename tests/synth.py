"""Seeded synthetic micro-inputs: random G-buffer + procedural noise volume.

Pure numpy; shared by tests/, bench.py and __graft_entry__.smoke().  No oracle, no GPU.
"""
import numpy as np

GB_PLANES = 23


def random_gbuffer(npix, seed=42, discard_frac=0.0, model_scale=0.05, extent=70.0):
    """fp32 [23, npix]: P U[-extent,extent]^3; orthonormal (T,B,N) scaled by model_scale
    (mimics ModelMatrix = scale(0.05), VCT.h:183); bump N = unit N; albedo U[.2,.9], a=1;
    spec U[0,1]; shadow U[0,1]."""
    r = np.random.default_rng(seed)
    g = np.zeros((GB_PLANES, npix), np.float32)
    g[0:3] = r.uniform(-extent, extent, (3, npix))
    n = r.normal(size=(3, npix))
    n /= np.linalg.norm(n, axis=0, keepdims=True)
    h = np.where(np.abs(n[0]) < 0.9, 1.0, 0.0)
    helper = np.stack([h, 1.0 - h, np.zeros_like(h)])
    t = np.cross(helper.T, n.T).T
    t /= np.linalg.norm(t, axis=0, keepdims=True)
    b = np.cross(n.T, t.T).T
    g[3:6] = n * model_scale
    g[6:9] = t * model_scale
    g[9:12] = b * model_scale
    g[12:15] = n
    g[15:18] = r.uniform(0.2, 0.9, (3, npix))
    g[18] = 1.0
    g[19:22] = r.uniform(0.0, 1.0, (3, npix))
    g[22] = r.uniform(0.0, 1.0, npix)
    if discard_frac > 0:
        g[18, r.uniform(size=npix) < discard_frac] = 0.0
    return np.ascontiguousarray(g, np.float32)


def coherent_gbuffer(w, h, seed=3, model_scale=0.05, plane_y=-20.0, extent=60.0):
    """A screen-coherent G-buffer: pixels lie on a gently curved floor patch, so neighbouring
    pixels trace neighbouring cones (what a rasterised frame looks like to the kernel)."""
    r = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32),
                         indexing="ij")
    u = (xs / max(w - 1, 1) - 0.5) * 2 * extent
    v = (ys / max(h - 1, 1) - 0.5) * 2 * extent
    g = np.zeros((GB_PLANES, h * w), np.float32)
    yy = plane_y + 3.0 * np.sin(u * 0.05) * np.cos(v * 0.04)
    g[0], g[1], g[2] = u.ravel(), yy.ravel(), v.ravel()
    dydu = 3.0 * 0.05 * np.cos(u * 0.05) * np.cos(v * 0.04)
    dydv = -3.0 * 0.04 * np.sin(u * 0.05) * np.sin(v * 0.04)
    n = np.stack([-dydu.ravel(), np.ones(h * w, np.float32), -dydv.ravel()])
    n /= np.linalg.norm(n, axis=0, keepdims=True)
    t = np.stack([np.ones(h * w), dydu.ravel(), np.zeros(h * w)])
    t -= n * (t * n).sum(0, keepdims=True)
    t /= np.linalg.norm(t, axis=0, keepdims=True)
    b = np.cross(n.T, t.T).T
    g[3:6], g[6:9], g[9:12] = n * model_scale, t * model_scale, b * model_scale
    g[12:15] = n
    g[15:18] = r.uniform(0.2, 0.9, (3, 1)) * np.ones((1, h * w))
    g[18] = 1.0
    g[19:22] = 0.5
    g[22] = 0.8
    return np.ascontiguousarray(g, np.float32)


def _value_noise(V, cells, r):
    lat = r.uniform(size=(cells + 1,) * 3).astype(np.float32)
    c = (np.arange(V, dtype=np.float32) + 0.5) * (cells / V)
    i0 = np.minimum(c.astype(np.int64), cells - 1)
    f = (c - i0).astype(np.float32)
    f = f * f * (3 - 2 * f)

    def lerp_axis(a, axis):
        sl0 = [slice(None)] * 3
        sl1 = [slice(None)] * 3
        sl0[axis] = i0
        sl1[axis] = i0 + 1
        shape = [1, 1, 1]
        shape[axis] = V
        ff = f.reshape(shape)
        return a[tuple(sl0)] * (1 - ff) + a[tuple(sl1)] * ff

    out = lerp_axis(lat, 0)
    out = lerp_axis(out, 1)
    out = lerp_axis(out, 2)
    return out


def noise_volume(V, seed=7, occupancy=0.05):
    """uint8 [V,V,V,4] (z,y,x): thin-shell occupancy from thresholded smooth 3-D value noise,
    ~occupancy of the voxels with a=255 and rgb U[0,255]; everything else 0."""
    r = np.random.default_rng(seed)
    cells = max(V // 16, 2)
    nz = _value_noise(V, cells, r)
    d = np.abs(nz - np.float32(0.5))
    thr = np.quantile(d.ravel()[:: max(1, d.size // 2000000)], occupancy)
    occ = d < thr
    vol = np.zeros((V, V, V, 4), np.uint8)
    rgb = r.integers(0, 256, size=(V, V, V, 3), dtype=np.uint8)
    vol[..., :3] = rgb * occ[..., None]
    vol[..., 3] = np.where(occ, 255, 0)
    return vol


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
