"""Known-answer tests for the CPU oracle, derived from the reference's shader text
(SURVEY.md section 4).  The reference has no tests of its own: these and the numpy
restatement pin the oracle from the shader TEXT; since round 5 tests/test_ref_gl.py also pins it against the outputs of the
reference's own GLSL run on Mesa llvmpipe (oracle/ref_gl.c)."""
import numpy as np
import pytest

import np_restatement as npr
import synth


def uniform_chain(o, V, rgba):
    l0 = np.zeros((V, V, V, 4), np.uint8)
    l0[...] = np.asarray(rgba, np.uint8)
    return o.build_mips(l0)


def frame(n, model_scale=0.05):
    """An orthonormal world-aligned frame: N=+Y."""
    g = np.zeros((23, n), np.float32)
    g[3:6] = np.array([[0], [1], [0]]) * model_scale
    g[6:9] = np.array([[1], [0], [0]]) * model_scale
    g[9:12] = np.array([[0], [0], [-1]]) * model_scale    # B = N x T
    g[12:15] = np.array([[0], [1], [0]])
    g[15:19] = np.array([[0.8], [0.6], [0.4], [1.0]])
    g[19:22] = np.array([[0.5], [0.25], [0.125]])
    g[22] = 0.7
    return g


def test_cone_constants(oracle):                       # trace.fs:48-57
    d, w = oracle.cone_constants()
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)
    assert abs(float(w.sum()) - 1.0) < 1e-6
    assert w[0] == np.float32(0.25) and np.all(w[1:] == np.float32(0.15))


@pytest.mark.parametrize("V,nd,ns", [(64, 5, 18), (128, 6, 23), (256, 7, 29), (512, 8, 34),
                                     (1024, 9, 39)])
def test_max_steps(oracle, V, nd, ns):                 # trace.fs:90-104
    p = oracle.default_params(V)
    d, lod_d = oracle.max_steps(p, 0.577)
    s, lod_s = oracle.max_steps(p, 0.07)
    assert (d, s) == (nd, ns)
    assert lod_d < np.log2(V) and lod_s < np.log2(V)
    if V == 256:
        assert abs(lod_d - 6.85) < 0.01 and abs(lod_s - 4.13) < 0.01


def test_empty_volume(oracle):                         # trace.fs:94-107,201-227
    V = 64
    p = oracle.default_params(V)
    chain = uniform_chain(oracle, V, (0, 0, 0, 0))
    g = frame(4)
    g[0:3] = np.array([[1.0, -3.0, 20.0, 5.0], [2.0, 2.5, -7.0, 0.0], [3.0, 9.0, 1.0, -30.0]])
    r = oracle.trace(p, chain, g, want_cones=True)
    assert np.all(r["cones"] == 0.0)
    assert np.all(r["steps"][:, :6] == 5) and np.all(r["steps"][:, 6] == 18)
    L = np.array(p.light_dir[:], np.float64)
    L /= np.linalg.norm(L)
    N = np.array([0.0, 1.0, 0.0])
    cam = np.array(p.camera_pos[:], np.float64)
    for i in range(4):
        P = g[0:3, i].astype(np.float64)
        E = (cam - P) / np.linalg.norm(cam - P)
        R = -L - 2 * np.dot(N, -L) * N
        R /= np.linalg.norm(R)
        spec = max(np.dot(E, R), 0.0) ** 20.0
        alb = g[15:18, i].astype(np.float64)
        sc = g[19:22, i].astype(np.float64)
        sh = 0.7
        want = 0.1 * alb * 1.0 + (sh * max(np.dot(N, L), 0)) * alb + (spec * sh) * sc
        assert np.allclose(r["rgba32f"][i, :3], want, rtol=2e-5, atol=1e-6)
        assert r["rgba32f"][i, 3] == 1.0


@pytest.mark.parametrize("V,t,want", [(256, 0.577, 0.980118), (256, 0.07, 0.982726),
                                      (64, 0.577, 0.924949), (64, 0.07, 0.934307)])
def test_uniform_opaque(oracle, V, t, want):           # trace.fs:96-103
    p = oracle.default_params(V)
    chain = uniform_chain(oracle, V, (51, 102, 204, 255))
    out, steps = oracle.cone(p, chain, [3, 4, 5], [0, 0.05, 0], [0.6, 0.8, 0.0], t)
    assert steps == 1
    assert np.allclose(out[:3], np.array([51, 102, 204]) / 255.0, atol=1e-6)
    assert abs(out[3] - want) < 1e-6


def test_uniform_alpha_geometric(oracle):              # trace.fs:100-102
    V = 64
    p = oracle.default_params(V)
    a8, c8 = 64, 128
    chain = uniform_chain(oracle, V, (c8, c8, c8, a8))
    out, steps = oracle.cone(p, chain, [0, 0, 0], [0, 0.05, 0], [0, 0, 1], 0.07)
    a, c = a8 / 255.0, c8 / 255.0
    n = 0
    alpha = 0.0
    while n < 18 and alpha < 0.95:
        alpha = 1 - (1 - a) ** (n + 1)
        n += 1
    assert steps == n
    assert np.allclose(out[:3], c * sum((1 - a) ** k for k in range(n)), rtol=1e-5)


def test_world_to_voxel_index_all_axes(oracle):        # VCT.h:128-134, vox.fs:58-86
    V, G = 128, 150.0
    w = np.array([10.3, -20.7, 33.1, 1.0], np.float32)
    want = np.floor((w[:3] / G + 0.5) * V).astype(int)
    assert tuple(want) == (72, 46, 92)
    for axis in (1, 2, 3):
        m = oracle.voxel_proj(G, axis)
        ndc = m @ w
        fx, fy, fz = (ndc[0] * 0.5 + 0.5) * V, (ndc[1] * 0.5 + 0.5) * V, ndc[2] * 0.5 + 0.5
        vp = oracle.frag_to_voxel(V, axis, fx, fy, fz)
        assert tuple(vp) == (72, 46, 92), axis


def test_proj_matrices_times_75(oracle):
    G = 150.0
    want = {1: [[0, 0, -1, 0], [0, 1, 0, 0], [-1, 0, 0, 0]],
            2: [[1, 0, 0, 0], [0, 0, -1, 0], [0, -1, 0, 0]],
            3: [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0]]}
    for axis, rows in want.items():
        m = oracle.voxel_proj(G, axis)
        assert np.allclose(m[:3] * 75.0, np.array(rows, np.float32), atol=1e-5), axis
        assert np.allclose(m[3], [0, 0, 0, 1])


def test_dominant_axis(oracle):                        # vox.gs:24-39
    assert oracle.dominant_axis([0, 0, 0], [0, 1, 0], [0, 0, 1]) == 1
    assert oracle.dominant_axis([0, 0, 0], [1, 0, 0], [0, 0, 1]) == 2
    assert oracle.dominant_axis([0, 0, 0], [1, 0, 0], [0, 1, 0]) == 3
    # ties resolve x, then y
    assert oracle.dominant_axis([0, 0, 0], [1, -1, 0], [0, 0, 1]) == 1
    # degenerate -> NaN normal -> z
    assert oracle.dominant_axis([0, 0, 0], [0, 0, 0], [0, 0, 0]) == 3


def test_gl_repeat(oracle):                            # wrap never set: VCT.h:110-113
    V = 32
    p = oracle.default_params(V)
    chain = oracle.build_mips(synth.noise_volume(V, seed=5, occupancy=0.3))
    r = np.random.default_rng(0)
    for _ in range(50):
        pos = r.uniform(-70, 70, 3).astype(np.float32)
        lod = float(r.uniform(0, 4))
        a = oracle.sample(p, chain, pos, lod)
        b = oracle.sample(p, chain, pos + np.array([150, 0, 0], np.float32), lod)
        c = oracle.sample(p, chain, pos - np.array([0, 150, 150], np.float32), lod)
        assert np.allclose(a, b, atol=2e-4) and np.allclose(a, c, atol=2e-4)
    # clamp mode differs at the border
    p2 = oracle.default_params(V, wrap_repeat=0)
    l0 = np.zeros((V, V, V, 4), np.uint8)
    l0[:, :, 0] = 255
    ch = oracle.build_mips(l0)
    edge = np.array([74.9, 0, 0], np.float32)   # right border: REPEAT sees column x=0
    assert oracle.sample(p, ch, edge, 0.0)[3] > 0.4
    assert oracle.sample(p2, ch, edge, 0.0)[3] == 0.0


def test_box_mip(oracle):                              # VCT.h:248 + GL box filter
    V = 8
    l0 = np.zeros((V, V, V, 4), np.uint8)
    l0[2, 4, 6] = 255
    chain = oracle.build_mips(l0)
    l1 = oracle.level_view(chain, V, 1)
    assert tuple(l1[1, 2, 3]) == (32, 32, 32, 32)
    assert l1.sum() == 4 * 32
    l2 = oracle.level_view(chain, V, 2)
    assert l2[0, 1, 1, 3] == 4     # requantised parent: round(32/8)
    assert oracle.num_levels(V) == 4 and oracle.chain_texels(V) == 512 + 64 + 8 + 1
    # full random check against a numpy box filter
    r = np.random.default_rng(1)
    l0 = r.integers(0, 256, (16, 16, 16, 4), dtype=np.uint8)
    chain = oracle.build_mips(l0)
    prev = l0.astype(np.uint32)
    for lvl in range(1, 5):
        n = prev.shape[0] // 2
        s = prev.reshape(n, 2, n, 2, n, 2, 4).sum(axis=(1, 3, 5))
        want = ((s + 4) >> 3).astype(np.uint8)
        assert np.array_equal(oracle.level_view(chain, 16, lvl), want)
        prev = want.astype(np.uint32)


def test_sample_level_selection(oracle):               # [GL] A.2
    V = 16
    p = oracle.default_params(V)
    l0 = np.zeros((V, V, V, 4), np.uint8)
    chain = oracle.build_mips(l0)
    # paint each level a distinct constant alpha
    for lvl in range(5):
        oracle.level_view(chain, V, lvl)[...] = 40 * (lvl + 1)
    pos = np.array([1.0, 2.0, 3.0], np.float32)
    for lod, want in [(-1.0, 40), (0.0, 40), (0.5, 60), (1.0, 80), (2.25, 130), (4.0, 200),
                      (9.0, 200)]:
        got = oracle.sample(p, chain, pos, lod)[3] * 255.0
        assert abs(got - want) < 1e-3, (lod, got, want)


def test_texel_centre_convention(oracle):              # [GL] u*N - 0.5
    V = 8
    p = oracle.default_params(V)
    l0 = np.zeros((V, V, V, 4), np.uint8)
    l0[3, 5, 2] = (10, 20, 30, 255)     # z=3,y=5,x=2
    chain = oracle.build_mips(l0)
    vs = 150.0 / V
    centre = (np.array([2, 5, 3]) + 0.5) * vs - 75.0
    s = oracle.sample(p, chain, centre.astype(np.float32), 0.0)
    assert np.allclose(s, np.array([10, 20, 30, 255]) / 255.0, atol=1e-6)
    half = centre + np.array([vs / 2, 0, 0])
    s = oracle.sample(p, chain, half.astype(np.float32), 0.0)
    assert np.allclose(s, 0.5 * np.array([10, 20, 30, 255]) / 255.0, atol=1e-5)


def test_half_conversion_matches_numpy(oracle):
    r = np.random.default_rng(2)
    xs = np.concatenate([
        r.uniform(-4, 4, 2000), r.uniform(-7e4, 7e4, 500), r.uniform(-1e-4, 1e-4, 500),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 6e-8, 2.98e-8, 2.99e-8, 6.1e-5,
                  np.inf, -np.inf, 1.0009765625, 1.00048828125, 1.00146484375]),
    ]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f32_to_f16(x) for x in xs], np.uint16)
    assert np.array_equal(got, want)
    back = np.array([oracle.f16_to_f32(h) for h in want], np.float32)
    assert np.array_equal(back, want.view(np.float16).astype(np.float32))


def test_against_numpy_restatement(oracle):
    V = 32
    p = oracle.default_params(V)
    chain = oracle.build_mips(synth.noise_volume(V, seed=11, occupancy=0.15))
    levels = npr.levels_from_chain(chain, V)
    r = np.random.default_rng(4)
    pos = r.uniform(-90, 90, (300, 3)).astype(np.float32)     # includes out-of-grid (wrap)
    lod = r.uniform(-0.5, 5.5, 300).astype(np.float32)
    want = npr.sample(levels, 150.0, pos, lod)
    got = np.stack([oracle.sample(p, chain, pos[i], lod[i]) for i in range(300)])
    assert np.abs(got - want).max() < 3e-6
    nsteps_equal = 0
    for i in range(40):
        P = r.uniform(-60, 60, 3).astype(np.float32)
        n = r.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        d = r.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        t = [0.577, 0.07][i % 2]
        a, sa = oracle.cone(p, chain, P, n * np.float32(0.05), d, t)
        b, sb = npr.cone(levels, V, 150.0, P, n * np.float32(0.05), d, t)
        nsteps_equal += sa == sb
        if sa == sb:
            assert np.abs(a - b).max() < 2e-5
    assert nsteps_equal >= 39


def test_discard_and_clear_colour(oracle):             # trace.fs:171, VCT.h:156-159
    V = 16
    chain = uniform_chain(oracle, V, (0, 0, 0, 0))
    g = frame(2)
    g[18, 1] = 0.2
    r = oracle.trace(oracle.default_params(V), chain, g)
    assert np.array_equal(r["rgba32f"][1], np.array([0.5, 0.5, 0.5, 1.0], np.float32))
    assert np.all(r["steps"][1] == 0)
    r = oracle.trace(oracle.default_params(V, ambient_factor=0.6), chain, g)
    assert np.array_equal(r["rgba32f"][1], np.array([1, 1, 1, 1], np.float32))


def test_threads_equal_scalar(oracle):
    V = 32
    p = oracle.default_params(V)
    chain = oracle.build_mips(synth.noise_volume(V))
    g = synth.random_gbuffer(1000, seed=9, discard_frac=0.1)
    a = oracle.trace(p, chain, g, nthreads=1, want_cones=True)
    b = oracle.trace(p, chain, g, nthreads=4, want_cones=True)
    for k in ("rgba32f", "rgba16f", "steps", "cones"):
        assert np.array_equal(a[k], b[k])
    assert a["total_steps"] == b["total_steps"] == int(a["steps"].sum())


def test_tbn_general_frame(oracle):                    # trace.fs:175,198
    """For an orthonormal frame inverse(transpose(M)) = M: cone 0 runs along N."""
    V = 16
    p = oracle.default_params(V)
    l0 = np.zeros((V, V, V, 4), np.uint8)
    l0[:, V - 1, :] = (255, 0, 0, 255)     # a red ceiling at the top of the grid (y max)
    chain = oracle.build_mips(l0)
    g = frame(1)
    g[0:3, 0] = [0.0, 20.0, 0.0]
    r = oracle.trace(p, chain, g, want_cones=True)
    c = r["cones"][0]
    assert c[0, 0] > 0.1 and c[0, 1] == 0.0          # the +N cone sees the red ceiling
    # non-orthogonal frame: result equals normalize(inv(transpose(M)) d)
    T = np.array([1.0, 0.2, 0.0]) * 0.05
    B = np.array([0.1, 0.0, -1.0]) * 0.05
    N = np.array([0.0, 1.0, 0.1]) * 0.05
    M = np.stack([T, B, N], axis=1)
    K = np.linalg.inv(M.T)
    dirs, _ = oracle.cone_constants()
    g2 = frame(1)
    g2[3:6, 0], g2[6:9, 0], g2[9:12, 0] = N, T, B
    g2[0:3, 0] = [0, 0, 0]
    # uniform translucent volume: cone result depends on direction only through wrap -> use
    # a volume with a gradient instead
    l0 = np.zeros((V, V, V, 4), np.uint8)
    l0[..., 0] = (np.arange(V) * 8)[None, None, :]
    l0[..., 3] = 40
    chain = oracle.build_mips(l0)
    r2 = oracle.trace(p, chain, g2, want_cones=True)
    for i in range(6):
        d = K @ dirs[i].astype(np.float64)
        d /= np.linalg.norm(d)
        want, _ = oracle.cone(p, chain, g2[0:3, 0], g2[3:6, 0], d.astype(np.float32), 0.577)
        assert np.allclose(r2["cones"][0, i], want, atol=2e-4)


def test_depth24_dequantisation_needs_no_double():
    """[GL] DEPTH_COMPONENT24 (VCT.h:90): the oracle's shadow map holds float(double(q) / (2^24 - 1)); the HIP raster
    writes float(q) / 16777215.0f (IEEE fp32 divide) straight into the map's words (csrc/vct_internal.h
    vct_depth24_bits).  The two agree for every one of the 2^24 codes."""
    q = np.arange(1 << 24, dtype=np.uint32)
    want = (q.astype(np.float64) / 16777215.0).astype(np.float32)
    got = q.astype(np.float32) / np.float32(16777215.0)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert got.max() == np.float32(1.0) and got.view(np.uint32).max() == 0x3F800000      # fits under the epoch bits


# ---- mip-mapped material textures (R/Model.h:168-173; oracle/vct_oracle.h "Mip-mapped sampling") ----------------

def test_texture_mip_chain_is_the_rounded_box_filter(oracle):
    """glGenerateMipmap restated: level k is max(1, W >> k) x max(1, H >> k); a texel is (a + b + c + d + 2) >> 2 of its
    2x2 parents, parent indices clamped (odd sizes drop the last row / column), down to 1 x 1."""
    t = np.zeros((4, 4, 4), np.uint8)
    t[..., 0] = np.arange(16).reshape(4, 4) * 16
    t[..., 1] = 255
    t[..., 3] = [[0, 255, 0, 255]] * 4
    chain, nlev = oracle.tex_build_mips(t)
    assert nlev == 3 and chain.shape == (16 + 4 + 1, 4)
    l1 = oracle.tex_level(t, 1)
    assert l1.shape == (2, 2, 4)
    assert l1[0, 0, 0] == (0 + 16 + 64 + 80 + 2) >> 2 and l1[1, 1, 0] == (160 + 176 + 224 + 240 + 2) >> 2
    assert (l1[..., 1] == 255).all() and (l1[..., 3] == 128).all()          # (0 + 255 + 0 + 255 + 2) >> 2
    l2 = oracle.tex_level(t, 2)
    assert l2.shape == (1, 1, 4) and l2[0, 0, 0] == (int(l1[..., 0].astype(int).sum()) + 2) >> 2
    odd = np.random.default_rng(4).integers(0, 256, (3, 5, 4), dtype=np.uint8)       # 5 x 3 -> 2 x 1 -> 1 x 1
    chain, nlev = oracle.tex_build_mips(odd)
    assert nlev == 3 and chain.shape[0] == 15 + 2 + 1
    a = oracle.tex_level(odd, 1)
    assert a.shape == (1, 2, 4)
    for x in range(2):
        want = (odd[0, 2 * x].astype(int) + odd[0, 2 * x + 1].astype(int) + odd[1, 2 * x].astype(int) + odd[1, 2 * x + 1].astype(int) + 2) >> 2
        assert np.array_equal(a[0, x], want)
    b = oracle.tex_level(odd, 2)                                                      # 2 x 1 parent: rows clamp
    assert np.array_equal(b[0, 0], (2 * a[0, 0].astype(int) + 2 * a[0, 1].astype(int) + 2) >> 2)


def test_log2_det_is_exact_on_powers_of_two_and_close_elsewhere(oracle):
    for k in range(0, 40):
        assert oracle.log2_det(float(2.0 ** k)) == float(k)
    xs = np.exp(np.random.default_rng(2).uniform(0.0, 28.0, 4000)).astype(np.float32)
    err = max(abs(oracle.log2_det(float(x)) - float(np.log2(np.float64(x)))) for x in xs)
    assert err < 4e-6


def test_minified_checker_takes_the_level_lambda_selects(oracle):
    """A one-texel checkerboard: level 0 alternates 0 / 255, every coarser level is the flat mean 128.  With the
    coordinate moving 4 texels per pixel lambda = log2(4) = 2 exactly: the sample is the flat level; at <= 1 texel per
    pixel (lambda <= 0) it is the level-0 bilinear sample; in between, the linear blend of two levels."""
    n = 64
    t = np.zeros((n, n, 4), np.uint8)
    t[..., :3] = (((np.arange(n)[:, None] + np.arange(n)[None, :]) & 1) * 255)[..., None]
    t[..., 3] = 255
    assert (oracle.tex_level(t, 1)[..., 0] == 128).all() and (oracle.tex_level(t, 5)[..., 0] == 128).all()
    u, v = (10 + 0.5) / n, (20 + 0.5) / n                                   # a texel centre: level 0 returns the texel
    lvl0 = oracle.tex_sample(t, u, v)
    assert lvl0[0] == np.float32(t[20, 10, 0]) / np.float32(255)
    flat = np.float32(128) / np.float32(255)
    for duv in ((4.0 / n, 0, 0, 4.0 / n), (4.0 / n, 0, 0, 0), (0, 0, 0.5 / n, 4.0 / n), (64.0 / n, 0, 0, 0), (1e9, 0, 0, 0)):
        assert oracle.tex_sample(t, u, v, duv)[0] == flat                   # lambda >= 1: only flat levels contribute
    for duv in ((1.0 / n, 0, 0, 1.0 / n), (0.3 / n, 0, 0, 0.2 / n), (0, 0, 0, 0), (float("nan"), 0, 0, 0)):
        assert np.array_equal(oracle.tex_sample(t, u, v, duv), lvl0)        # lambda <= 0 (and NaN): magnification
    # rho = sqrt(2) texels per pixel: lambda = 0.5, an even blend of level 0 and level 1, fma(f, t2, (1 - f) * t1)
    lam = np.float32(0.5) * np.float32(oracle.log2_det(2.0))
    assert lam == np.float32(0.5)
    got = oracle.tex_sample(t, u, v, (np.sqrt(2.0) / n, 0, 0, 0))
    m = np.float32((np.float32(np.sqrt(2.0) / n) * np.float32(n)) ** 2)
    f = np.float32(0.5) * np.float32(oracle.log2_det(float(m)))
    want = np.float32(np.float64(f) * np.float64(flat) + np.float64(np.float32((np.float32(1) - f) * lvl0[0])))   # fma
    assert got[0] == want and abs(float(got[0]) - (0.5 * float(flat) + 0.5 * float(lvl0[0]))) < 1e-6
    # the larger of the two axes' footprints decides (the ideal rho of the GL specification)
    assert np.array_equal(oracle.tex_sample(t, u, v, (3.0 / n, 0, 0, 1.0 / n)), oracle.tex_sample(t, u, v, (0, 3.0 / n, 1.0 / n, 0)))


def test_zslab_voxelization_equals_the_slices_of_the_whole_volume():
    """vcto_voxelize_conservative_zslab (the small-host form of the 1024^3 check in test_gpu_configs.py) returns exactly
    the z-slices of vcto_voxelize_conservative."""
    from oracle import pyoracle
    V = 32
    rng = np.random.default_rng(5)
    c = rng.uniform(-1300, 1300, (300, 1, 3))
    pos = (c + rng.normal(scale=120.0, size=(300, 3, 3))).astype(np.float32).reshape(300, 9)
    mat = rng.integers(0, 3, 300).astype(np.int32)
    alb = rng.uniform(0.2, 0.9, (3, 4)).astype(np.float32)
    sc = pyoracle.make_scene(pos, mat, alb)
    p = pyoracle.default_params(V)
    full = pyoracle.voxelize_conservative(p, sc)
    assert (full[..., 3] > 0).sum() > 500
    for z0, z1 in ((0, 32), (0, 7), (9, 20), (31, 32)):
        assert np.array_equal(pyoracle.voxelize_conservative_zslab(p, sc, z0, z1), full[z0:z1])
