"""Independent numpy restatement of the sampler + cone march,
written separately from oracle/vct_oracle.cpp (lerp form, vectorised over cones) so that an
indexing / wrap / lod-selection mistake in one shows up against the other.  fp32 without fused
multiply-add, so agreement is to ~1e-6, not bit-exact."""
import numpy as np

f32 = np.float32


def levels_from_chain(chain, V):
    out, off, n = [], 0, V
    while n >= 1:
        out.append(chain[off:off + n ** 3].reshape(n, n, n, 4))
        off += n ** 3
        n //= 2
    return out


def tri(level, uv):
    """level: uint8 [N,N,N,4] (z,y,x); uv: [M,3] in texture space.  GL_REPEAT, texel centres."""
    N = level.shape[0]
    c = uv.astype(f32) * f32(N) - f32(0.5)
    i0 = np.floor(c).astype(np.int64)
    fr = (c - np.floor(c)).astype(f32)
    i1 = i0 + 1
    i0 %= N
    i1 %= N
    lv = level.astype(f32) / f32(255.0)

    def fetch(ix, iy, iz):
        return lv[iz, iy, ix]

    x0, y0, z0 = i0[:, 0], i0[:, 1], i0[:, 2]
    x1, y1, z1 = i1[:, 0], i1[:, 1], i1[:, 2]
    a, b, g = fr[:, 0:1], fr[:, 1:2], fr[:, 2:3]
    c00 = fetch(x0, y0, z0) * (1 - a) + fetch(x1, y0, z0) * a
    c10 = fetch(x0, y1, z0) * (1 - a) + fetch(x1, y1, z0) * a
    c01 = fetch(x0, y0, z1) * (1 - a) + fetch(x1, y0, z1) * a
    c11 = fetch(x0, y1, z1) * (1 - a) + fetch(x1, y1, z1) * a
    c0 = c00 * (1 - b) + c10 * b
    c1 = c01 * (1 - b) + c11 * b
    return (c0 * (1 - g) + c1 * g).astype(f32)


def texture_lod(levels, uv, lod):
    maxl = len(levels) - 1
    lod = np.clip(np.asarray(lod, f32), 0, maxl)
    out = np.zeros((uv.shape[0], 4), f32)
    d1 = np.floor(lod).astype(int)
    d2 = np.minimum(d1 + 1, maxl)
    f = (lod - np.floor(lod)).astype(f32)
    for l in np.unique(d1):
        m = d1 == l
        t1 = tri(levels[l], uv[m])
        t2 = tri(levels[d2[m][0]], uv[m]) if True else None
        # d2 is a function of d1 only
        out[m] = t1 * (1 - f[m])[:, None] + t2 * f[m][:, None]
    return out


def sample(levels, G, pos, lod):
    uv = (pos.astype(f32) / f32(G * 0.5)) * f32(0.5) + f32(0.5)
    return texture_lod(levels, uv, lod)


def cone(levels, V, G, P, Nw, d, tan_half, max_distance=75.0, max_alpha=0.95):
    """Single cone, scalar loop (trace.fs:82-107).  Returns (rgb-occlusion vec4, steps)."""
    vs = f32(G) / f32(V)
    dist = vs
    start = P.astype(f32) + Nw.astype(f32) * vs
    col = np.zeros(3, f32)
    alpha = f32(0)
    occ = f32(0)
    steps = 0
    while dist < f32(max_distance) and alpha < f32(max_alpha):
        diam = max(vs, f32(2.0) * f32(tan_half) * dist)
        lod = np.log2(diam / vs, dtype=f32)
        v = sample(levels, G, (start + dist * d.astype(f32))[None, :], np.array([lod], f32))[0]
        col = col + (f32(1) - alpha) * v[:3]
        occ = occ + ((f32(1) - alpha) * v[3]) / (f32(1) + f32(0.03) * diam)
        alpha = alpha + (f32(1) - alpha) * v[3]
        dist = dist + diam
        steps += 1
    return np.concatenate([col, [occ]]).astype(f32), steps
