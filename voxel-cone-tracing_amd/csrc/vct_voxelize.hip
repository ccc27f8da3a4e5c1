// vct_voxelize.hip -- triangle voxelization + light injection into level 0 of the brick chain.
//
// Replaces the reference's voxelization draw (VCT.h:213-245): S/Voxelization.vs:15-22 (world
// position, shadow coordinate), S/Voxelization.gs:22-51 (dominant axis), S/Voxelization.fs:18-89
// (PCF, voxel index, imageStore of albedo*shadow).  North-star mode: conservative triangle /
// voxel-box overlap (Schwarz & Seidel 2010) instead of pixel-centre raster, and instead of the
// racy last-writer imageStore every fragment adds its unorm8 value into per-voxel 64-bit integer
// accumulators (sumR|sumG, sumB|count), which is exact and order-independent; vct_launch_resolve
// turns them into the rounded mean.  fp32 operation order mirrors the scalar oracle
// (compile with -ffp-contract=off).
//
// Work distribution (north-star mode): BRICK-parallel, no global atomics.  Which voxels a triangle overlaps depends
// only on geometry, V and G, so the fragment list is built ONCE when the triangles are uploaded (k_vox_plan*: exact
// conservative overlap, entries (triangle, Morton voxel)) and sorted by 8^3 brick (counting sort: k_frag_count /
// k_frag_scatter).  What a fragment contributes apart from the light -- its barycentrics on its triangle and its albedo
// (the mip-mapped diffuse fetch of vox.fs:56, or the material's colour) -- is evaluated once as well (k_frag_geom, round 4)
// and stored per sorted fragment.  A voxelize pass -- e.g. after the light moved -- is then one workgroup per touched
// brick (k_voxelize_bricks): its threads read their fragments' stored values, transform the triangle's vertices by the
// light matrix, run the 25-tap PCF and add albedo * shadow into the brick's 512 accumulators IN LDS (64-bit ds_add: the exact, order-independent "atomic RGBA average"), then the same workgroup
// resolves the rounded means and writes the brick's 2 KiB of texels -- coalesced, once.  Rounds 1-2 accumulated with
// two device-scope 64-bit atomics per fragment into per-brick pools in HBM and resolved in a second kernel: 0.098 +
// 0.018 ms at configs[1], bound by the atomic unit.  All fragments of a brick lie within 8 voxels of each other, so
// their 25-tap shadow windows overlap: coherent fetches.
//
// The texels of a pass go to a staging pool (one 2 KiB slot per brick the mesh can touch); vct_inject_light's sparse
// resolve (k_resolve_sparse) copies the slots of touched bricks into level 0 and clears bricks that are no longer
// covered -- vct_voxelize evaluates the light, vct_inject_light makes it visible, as before.
#include "vct_internal.h"

namespace {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float comp(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

__device__ __forceinline__ F3 xform_point(const float* m, F3 p) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}

// [GL] bilinear, clamp-to-edge depth fetch (VCT.h:93-96)
__device__ __forceinline__ float shadow_tex(const uint32_t* __restrict__ words, uint32_t eb, int S, float u, float v) {
    const float fS = (float)S;
    const float x = u * fS - 0.5f, y = v * fS - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const float top = (float)(S - 1);
    auto cl = [&](float f) -> int {
        if (!(f > 0.0f)) return 0;
        if (f >= top) return S - 1;
        return (int)f;
    };
    const int i0 = cl(fx), i1 = cl(fx + 1.0f), j0 = cl(fy), j1 = cl(fy + 1.0f);
    const float d00 = vct_shadow_depth(words[(size_t)j0 * S + i0], eb), d10 = vct_shadow_depth(words[(size_t)j0 * S + i1], eb);
    const float d01 = vct_shadow_depth(words[(size_t)j1 * S + i0], eb), d11 = vct_shadow_depth(words[(size_t)j1 * S + i1], eb);
    const float a0 = 1.0f - a, b0 = 1.0f - b;
    float acc = (a0 * b0) * d00;
    acc = fmaf(a * b0, d10, acc);
    acc = fmaf(a0 * b, d01, acc);
    acc = fmaf(a * b, d11, acc);
    return acc;
}

// One axis of the 5 PCF taps (offsets -2..2): floor of the texel coordinate and filter fraction, each computed exactly
// as shadow_tex() does for that tap.
struct TapAxis { float f[5], a[5]; };
__device__ __forceinline__ void tap_axis(float coord, float inv, float fS, TapAxis& t) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float u = coord + inv * (float)(k - 2);
        const float x = u * fS - 0.5f;
        t.f[k] = floorf(x);
        t.a[k] = x - t.f[k];
    }
}
// The window of the five taps of an axis is six CONSECUTIVE, UNCLAMPED texels when the floors step by exactly one
// (fp rounding at a texel boundary can make them step by 0 or 2) and the first / last lie inside the map: then
// shadow_tex()'s clamped indices are (int)f[k] and (int)f[k] + 1 for every tap -- without computing the ten clamped
// conversions of each axis (round 4: ~85 of the PCF's instructions).
__device__ __forceinline__ bool tap_axis_regular(const TapAxis& t, float top) {
    bool ok = t.f[0] >= 0.0f && t.f[4] + 1.0f <= top;
#pragma unroll
    for (int k = 0; k < 4; ++k) ok = ok && t.f[k + 1] == t.f[k] + 1.0f;
    return ok;
}

// vox.fs:18-52 (count of passing taps; caller divides by 25).  The 25 bilinear taps sit one texel
// apart, so they normally share a 6x6 texel window: when that window is six consecutive unclamped texels on both axes
// (checked per lane) it is loaded once (6 x 2 wide loads instead of 100 single ones) and each tap is evaluated from
// registers with its own exact weights; lanes whose window is irregular (fp rounding at a texel boundary, the rim of
// the map) take the tap-by-tap path.  Same bits either way.  (A clamped window used to have a path of its own -- 36
// single loads, i.e. 72 address registers, which set the register peak of the whole voxelize kernel: 158 VGPRs, 3 waves
// per SIMD, for a case only the rim of the map ever sees.)
__device__ __forceinline__ int pcf25(const uint32_t* __restrict__ words, uint32_t eb, int S, F3 c, float bias,
                                     const uint2* __restrict__ tiles = nullptr) {
    const float inv = __fdiv_rn(1.0f, (float)S);
    const float fS = (float)S, top = (float)(S - 1);
    TapAxis X, Y;
    tap_axis(c.x, inv, fS, X);
    tap_axis(c.y, inv, fS, Y);
    const float cur = c.z - bias;
    int count = 0;
    if (tap_axis_regular(X, top) && tap_axis_regular(Y, top)) {
        const uint32_t row0 = (uint32_t)(int)Y.f[0];
        const uint32_t col0 = (uint32_t)(int)X.f[0];
        // Round 5: first the depth bounds of the map tile the window starts in (ONE 8-byte load from a table the fragments of
        // a brick share, instead of twelve wide unaligned ones): a fragment away from every shadow boundary never fetches
        // a window.
        if (tiles) {
            const int v = vct_pcf_tile_verdict(tiles, S, col0, row0, cur);
#if defined(VCT_PROBE_TILE_ALWAYS)
            return v >= 0 ? v : 12;          // timing probe only (wrong results): every window decided by the tiles
#endif
            if (v >= 0) return v;
        }
        // One pass over the window: decode in place and take the smallest / largest decoded word on the way (the decode is
        // monotonic, so these are the window's depth bounds: vct_internal.h "PCF short cut").  A window the shadow boundary
        // does not cross ends here; the others evaluate the 25 taps from the same registers.  (The first form of the short
        // cut took the bounds on the raw words and loaded the window a second time: one more round trip in a kernel that
        // waits on memory two thirds of its time.)
        float d[6][6];
        uint32_t wmin = 0xffffffffu, wmax = 0u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const VctWords6 w = *reinterpret_cast<const VctWords6*>(words + ((row0 + (uint32_t)j) * (uint32_t)S + col0));
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const uint32_t v = min(w.v[i] - eb, VCT_SHADOW_ONE);      // = vct_shadow_depth()
                d[j][i] = __uint_as_float(v);
                wmin = min(wmin, v); wmax = max(wmax, v);
            }
        }
        {
            const float lo = __uint_as_float(wmin) * 0.999998f, hi = __uint_as_float(wmax) * 1.000002f;
            if (cur <= lo) return 25;
            if (cur > hi) return 0;
        }
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) {
                const float a = X.a[x], b = Y.a[y];
                const float a0 = 1.0f - a, b0 = 1.0f - b;
                float acc = (a0 * b0) * d[y][x];
                acc = fmaf(a * b0, d[y][x + 1], acc);
                acc = fmaf(a0 * b, d[y + 1][x], acc);
                acc = fmaf(a * b, d[y + 1][x + 1], acc);
                if (cur <= acc) ++count;
            }
    } else {
        for (int x = -2; x <= 2; ++x)
            for (int y = -2; y <= 2; ++y) {
                const float ox = inv * (float)x, oy = inv * (float)y;
                const float closest = shadow_tex(words, eb, S, c.x + ox, c.y + oy);
                if (cur <= closest) ++count;
            }
    }
    return count;
}

__device__ __forceinline__ uint32_t to_unorm8(float f) {
    const float s = f * 255.0f + 0.5f;
    if (!(s > 0.0f)) return 0u;
    if (s >= 255.0f) return 255u;
    return (uint32_t)(int)s;
}

struct TriSetup {
    F3 g[3];     // voxel-space vertices
    F3 dc[3];    // shadow coordinates
    F3 n;
    float d1, d2;
    float ne[3][3][2];
    float de[3][3];
    int lo[3], hi[3];
    int ua, ub;
    float area;
    float alb[3];
    float uv[3][2];     // TexCoord per vertex (vox.vs:17)
    int tex;            // diffuse texture of the triangle's material or -1 (flat colour)
    bool valid;
};

// vox.fs:56: the fragment's albedo -- texture(DiffuseTexture, uv) with uv interpolated by the fragment's
// barycentrics, or the flat material colour when the material has no diffuse texture
// duv: quad differences of the coordinate in the voxelization raster (mip-mapped textures; oracle/vct_oracle.h vcto_scene)
template <class Setup>
__device__ __forceinline__ void frag_albedo(const VctVoxParams& p, const Setup& r, float b0, float b1, float b2,
                                            const float duv[4], float alb[3]) {
    alb[0] = r.alb[0]; alb[1] = r.alb[1]; alb[2] = r.alb[2];
    if (r.tex >= 0) {
        const float u = b0 * r.uv[0][0] + b1 * r.uv[1][0] + b2 * r.uv[2][0];
        const float v = b0 * r.uv[0][1] + b1 * r.uv[1][1] + b2 * r.uv[2][1];
        const float4 c = vct_tex_sample_lod(p.tex, r.tex, u, v, duv[0], duv[1], duv[2], duv[3]);
        alb[0] = c.x; alb[1] = c.y; alb[2] = c.z;
    }
}
template <class Setup>
__device__ __forceinline__ void load_tex_setup(const VctVoxParams& p, int t, Setup& r) {
    r.tex = vct_tex_of(p.tex, p.material[t], 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.uv[k][0] = r.tex >= 0 ? p.tex.uv[(size_t)t * 6 + 2 * k] : 0.0f;
        r.uv[k][1] = r.tex >= 0 ? p.tex.uv[(size_t)t * 6 + 2 * k + 1] : 0.0f;
    }
}

__device__ __forceinline__ void setup_tri(const VctVoxParams& p, int t, TriSetup& r) {
    F3 w[3];
    const float fV = (float)p.V;
    const VctTri9 rec = *reinterpret_cast<const VctTri9*>(p.pos + (size_t)t * 9);     // three wide loads, not nine
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* q = rec.v + 3 * k;
        w[k] = {q[0] * p.model_scale, q[1] * p.model_scale, q[2] * p.model_scale};   // vox.vs:21
        const F3 d = xform_point(p.light_vp, w[k]);                                  // vox.vs:18
        r.dc[k] = {d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f};          // vox.vs:19
        r.g[k] = {(__fdiv_rn(w[k].x, p.G) + 0.5f) * fV, (__fdiv_rn(w[k].y, p.G) + 0.5f) * fV,
                  (__fdiv_rn(w[k].z, p.G) + 0.5f) * fV};
    }
    // vox.gs:24-39 dominant axis
    const F3 e1 = sub3(w[0], w[1]), e2 = sub3(w[2], w[0]);
    F3 nn = cross3(e1, e2);
    const float len = __builtin_sqrtf(dot3(nn, nn));
    const float nx = fabsf(__fdiv_rn(nn.x, len)), ny = fabsf(__fdiv_rn(nn.y, len)),
                nz = fabsf(__fdiv_rn(nn.z, len));
    int axis;
    if (nx >= ny && nx >= nz) axis = 1;
    else if (ny >= nx && ny >= nz) axis = 2;
    else axis = 3;

    const F3 e[3] = {sub3(r.g[1], r.g[0]), sub3(r.g[2], r.g[1]), sub3(r.g[0], r.g[2])};
    r.n = cross3(e[0], e[1]);
    r.valid = !((r.n.x == 0.0f && r.n.y == 0.0f && r.n.z == 0.0f) || r.n.x != r.n.x ||
                r.n.y != r.n.y || r.n.z != r.n.z);
    const F3 cp = {r.n.x > 0.0f ? 1.0f : 0.0f, r.n.y > 0.0f ? 1.0f : 0.0f, r.n.z > 0.0f ? 1.0f : 0.0f};
    r.d1 = dot3(r.n, sub3(cp, r.g[0]));
    r.d2 = dot3(r.n, sub3(sub3(F3{1.0f, 1.0f, 1.0f}, cp), r.g[0]));
    const float nsel[3] = {r.n.z, r.n.x, r.n.y};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const int a0 = pl, a1 = (pl + 1) % 3;    // xy, yz, zx
        const float sg = nsel[pl] >= 0.0f ? 1.0f : -1.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float ea = comp(e[i], a0), eb = comp(e[i], a1);
            const float na = -eb * sg, nb = ea * sg;
            r.ne[pl][i][0] = na;
            r.ne[pl][i][1] = nb;
            const float va = comp(r.g[i], a0), vb = comp(r.g[i], a1);
            r.de[pl][i] = -(na * va + nb * vb) + fmaxf(0.0f, na) + fmaxf(0.0f, nb);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float mn = fminf(fminf(comp(r.g[0], a), comp(r.g[1], a)), comp(r.g[2], a));
        const float mx = fmaxf(fmaxf(comp(r.g[0], a), comp(r.g[1], a)), comp(r.g[2], a));
        r.lo[a] = max((int)floorf(mn), 0);
        r.hi[a] = min((int)floorf(mx), p.V - 1);
    }
    r.ua = axis == 1 ? 1 : 0;
    r.ub = axis == 3 ? 1 : 2;
    const float ax0 = comp(r.g[0], r.ua), ay0 = comp(r.g[0], r.ub);
    const float ax1 = comp(r.g[1], r.ua), ay1 = comp(r.g[1], r.ub);
    const float ax2 = comp(r.g[2], r.ua), ay2 = comp(r.g[2], r.ub);
    r.area = (ax1 - ax0) * (ay2 - ay0) - (ax2 - ax0) * (ay1 - ay0);
    const float* alb = p.albedo + 4 * (size_t)p.material[t];
    r.alb[0] = alb[0]; r.alb[1] = alb[1]; r.alb[2] = alb[2];
    load_tex_setup(p, t, r);
}

__device__ __forceinline__ bool overlap(const TriSetup& c, int i, int j, int k) {
    const F3 pp = {(float)i, (float)j, (float)k};
    const float np = dot3(c.n, pp);
    if ((np + c.d1) * (np + c.d2) > 0.0f) return false;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const float pa = comp(pp, pl), pb = comp(pp, (pl + 1) % 3);
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (c.ne[pl][e][0] * pa + c.ne[pl][e][1] * pb + c.de[pl][e] < 0.0f) return false;
    }
    return true;
}

// The barycentrics of the conservative fragment of triangle set-up `r` at voxel (i, j, k) (vox.fs has none: the oracle's
// north-star mode evaluates the triangle at the voxel centre projected along the dominant axis, clamped into the
// triangle), and -- for mip-mapped textures -- the differences of the texture coordinate one voxel further along each
// in-plane axis.  Pure geometry (triangle, V, G, texture coordinates): evaluated ONCE per mesh by k_frag_geom and stored
// per fragment; the voxelize pass reads them back (round 4: the pass used to re-derive the whole triangle set-up --
// nine IEEE divisions, the dominant axis, the in-plane area -- for every fragment, ~300 of its ~1,200 instructions).
__device__ __forceinline__ void frag_bary(const TriSetup& r, int i, int j, int k, float& b0, float& b1) {
    const F3 ctr = {(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
    const float cx = comp(ctr, r.ua), cy = comp(ctr, r.ub);
    const float ax0 = comp(r.g[0], r.ua), ay0 = comp(r.g[0], r.ub);
    const float ax1 = comp(r.g[1], r.ua), ay1 = comp(r.g[1], r.ub);
    const float ax2 = comp(r.g[2], r.ua), ay2 = comp(r.g[2], r.ub);
    b0 = __fdiv_rn((ax1 - cx) * (ay2 - cy) - (ax2 - cx) * (ay1 - cy), r.area);
    b1 = __fdiv_rn((ax2 - cx) * (ay0 - cy) - (ax0 - cx) * (ay2 - cy), r.area);
    b0 = fminf(fmaxf(b0, 0.0f), 1.0f);
    b1 = fminf(fmaxf(b1, 0.0f), 1.0f);
    const float sum = b0 + b1;
    if (sum > 1.0f) { b0 = __fdiv_rn(b0, sum); b1 = __fdiv_rn(b1, sum); }
}
__device__ __forceinline__ void frag_duv(const TriSetup& r, int i, int j, int k, float duv[4]) {
    const F3 ctr = {(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
    const float cx = comp(ctr, r.ua), cy = comp(ctr, r.ub);
    const float ax0 = comp(r.g[0], r.ua), ay0 = comp(r.g[0], r.ub);
    const float ax1 = comp(r.g[1], r.ua), ay1 = comp(r.g[1], r.ub);
    const float ax2 = comp(r.g[2], r.ua), ay2 = comp(r.g[2], r.ub);
    // (no reference code for this mode) the UNCLAMPED barycentrics one voxel further along each in-plane axis
    auto uv_at = [&](float qx, float qy, float& ou, float& ov) {
        const float c0 = __fdiv_rn((ax1 - qx) * (ay2 - qy) - (ax2 - qx) * (ay1 - qy), r.area);
        const float c1 = __fdiv_rn((ax2 - qx) * (ay0 - qy) - (ax0 - qx) * (ay2 - qy), r.area);
        const float c2 = 1.0f - c0 - c1;
        ou = c0 * r.uv[0][0] + c1 * r.uv[1][0] + c2 * r.uv[2][0];
        ov = c0 * r.uv[0][1] + c1 * r.uv[1][1] + c2 * r.uv[2][1];
    };
    float mu, mv, xu, xv, yu, yv;
    uv_at(cx, cy, mu, mv);
    uv_at(cx + 1.0f, cy, xu, xv);
    uv_at(cx, cy + 1.0f, yu, yv);
    duv[0] = xu - mu; duv[1] = xv - mv; duv[2] = yu - mu; duv[3] = yv - mv;
}

// What a voxelize pass still needs of a fragment's triangle: the shadow coordinates of its vertices (they follow the
// light) and, in a scene without textures, the material's colour (FALB = false; with textures every fragment's albedo
// comes from frag_alb).
struct PassTri {
    F3 dc[3];
    float alb[3];
    uint32_t nrm[3];    // biased quantised face normal (voxel attributes)
};
#ifndef VCT_VOX_PREFETCH
#define VCT_VOX_PREFETCH 0     // EXPERIMENT (round 5): the next fragment's triangle requested one iteration ahead (list entry two ahead)
#endif
template <bool ATTR, bool FALB>
__device__ __forceinline__ void setup_pass(const VctVoxParams& p, int t, PassTri& r, const VctTri9* pre = nullptr) {
    if (p.shadow) {
        const VctTri9 rec = pre ? *pre : *reinterpret_cast<const VctTri9*>(p.pos + (size_t)t * 9);     // three wide loads, not nine
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* q = rec.v + 3 * k;
            const F3 w = {q[0] * p.model_scale, q[1] * p.model_scale, q[2] * p.model_scale};   // vox.vs:21
            const F3 d = xform_point(p.light_vp, w);                                          // vox.vs:18
            r.dc[k] = {d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f};               // vox.vs:19
        }
    }
    if (!FALB) {
        const float* alb = p.albedo + 4 * (size_t)p.material[t];
        r.alb[0] = alb[0]; r.alb[1] = alb[1]; r.alb[2] = alb[2];
    }
    if (ATTR) {
#pragma unroll
        for (int k = 0; k < 3; ++k) r.nrm[k] = p.tri_qnrm[(size_t)t * 3 + k];
    }
}

// One conservative fragment of triangle `r` with barycentrics (b0, b1) and albedo alb: the vox.fs:88 value
// unorm8(albedo * PCF / 25) and, for the second bounce, the fragment's albedo (unorm8) -- vox.fs:18-56.
struct FragValue { uint32_t r, g, b, ar, ag, ab; };
__device__ __forceinline__ FragValue frag_eval(const VctVoxParams& p, const PassTri& r, float b0, float b1, const float alb[3]) {
    const float b2 = fmaxf(1.0f - b0 - b1, 0.0f);
    float sh = 1.0f;
    if (p.shadow) {
        const F3 dc = {b0 * r.dc[0].x + b1 * r.dc[1].x + b2 * r.dc[2].x,
                       b0 * r.dc[0].y + b1 * r.dc[1].y + b2 * r.dc[2].y,
                       b0 * r.dc[0].z + b1 * r.dc[1].z + b2 * r.dc[2].z};
        sh = __fdiv_rn((float)pcf25(p.shadow, p.shadow_ebase, p.shadow_size, dc, 0.002f, p.shadow_tiles), 25.0f);   // vox.fs:46
    }
    FragValue f;
    f.r = to_unorm8(alb[0] * sh); f.g = to_unorm8(alb[1] * sh); f.b = to_unorm8(alb[2] * sh);   // vox.fs:88
    f.ar = to_unorm8(alb[0]); f.ag = to_unorm8(alb[1]); f.ab = to_unorm8(alb[2]);                // the fragment's albedo
    return f;
}

__device__ __forceinline__ uint32_t resolve_voxel(ulonglong2 a) {
    const uint32_t c = (uint32_t)(a.y >> 32);
    if (!c) return 0u;
    const uint32_t h = c >> 1;
    const uint32_t r = ((uint32_t)a.x + h) / c, g = ((uint32_t)(a.x >> 32) + h) / c,
                   b = ((uint32_t)a.y + h) / c;
    return r | (g << 8) | (b << 16) | 0xff000000u;      // vox.fs:88: a = 1
}

#define VCT_VOX_BIG 4096

__device__ __forceinline__ long long tri_candidates(const TriSetup& r) {
    if (!r.valid) return 0;
    const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
    return (long long)nx * ny * nz;
}

// Builds the fragment list: one entry (triangle, Morton index of the voxel) per voxel a triangle OVERLAPS.  The overlap
// test is pure geometry (triangle, V, G), so it runs here, once per uploaded mesh, and a voxelize pass spends its threads
// on fragments only.  (Round 2 listed every voxel of the clipped bounding box and tested in the pass: fine for the
// wall-and-floor atrium, but an oblique triangle overlaps a thin diagonal slice of its box -- the Bistro-class street's
// randomly oriented foliage cards at 1024^3: 753 M candidates for ~1/5 as many fragments.)  WRITE = false: only totals
// (plan[0] = fragments of the small triangles, plan[1] = big triangles, listed in big_list: bounding box > VCT_VOX_BIG
// voxels, enumerated by a workgroup each, k_vox_plan_big); WRITE = true: the entries.  Entry order is
// irrelevant (integer accumulation), so workgroups claim ranges with one atomic each.
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_vox_plan(const VctVoxParams p, uint32_t* plan, uint2* frags, int32_t* big_list) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t block_base;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t chunks = 0;
    bool big = false;
    TriSetup r;
    int nx = 0, ny = 0, nz = 0;
    if (t < p.ntri) {
        setup_tri(p, t, r);
        const long long cnt = tri_candidates(r);
        if (cnt > VCT_VOX_BIG) big = true;
        else if (cnt > 0) {
            nx = r.hi[0] - r.lo[0] + 1; ny = r.hi[1] - r.lo[1] + 1; nz = r.hi[2] - r.lo[2] + 1;
            for (int k = 0; k < nz; ++k)
                for (int j = 0; j < ny; ++j)
                    for (int i = 0; i < nx; ++i)
                        if (overlap(r, r.lo[0] + i, r.lo[1] + j, r.lo[2] + k)) ++chunks;
        }
    }
    // workgroup exclusive scan of `chunks`
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = chunks;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) wave_base += wave_tot[w];
        total += wave_tot[w];
    }
    if (threadIdx.x == 0) {
        block_base = total ? atomicAdd(&plan[0], total) : 0u;
        if (block_base + total < block_base) plan[3] = 1u;      // the 32-bit counter wrapped: the caller rejects the mesh
    }
    __syncthreads();
    if (big && !WRITE) big_list[atomicAdd(&plan[1], 1u)] = t;      // (capacity ntri: the count pass lists them)
    if (WRITE && chunks) {
        uint32_t at = block_base + wave_base + incl - chunks;
        for (int k = 0; k < nz; ++k)
            for (int j = 0; j < ny; ++j)
                for (int i = 0; i < nx; ++i)
                    if (overlap(r, r.lo[0] + i, r.lo[1] + j, r.lo[2] + k))
                        frags[at++] = make_uint2((uint32_t)t, vct_morton3((uint32_t)(r.lo[0] + i), (uint32_t)(r.lo[1] + j),
                                                                          (uint32_t)(r.lo[2] + k)));
    }
}

// fragments of the big triangles, one workgroup per triangle striding its bounding box.  WRITE = false: plan[2] += count;
// WRITE = true: entries appended at plan[2] (the caller starts it behind the small triangles' entries).
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_vox_plan_big(const VctVoxParams p, const int32_t* __restrict__ big_list, int n_big, uint32_t* plan, uint2* frags) {
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x; b < n_big; b += gridDim.x) {
        const int t = big_list[b];
        TriSetup r;
        setup_tri(p, t, r);
        const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
        const long long cnt = (long long)nx * ny * nz;
        for (long long v0 = 0; v0 < cnt; v0 += blockDim.x) {         // wave-uniform trip count: ballots below
            const long long v = v0 + threadIdx.x;
            bool hit = false;
            int i = 0, j = 0, k = 0;
            if (v < cnt) {
                i = r.lo[0] + (int)(v % nx);
                j = r.lo[1] + (int)((v / nx) % ny);
                k = r.lo[2] + (int)(v / ((long long)nx * ny));
                hit = overlap(r, i, j, k);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m == 0ull) continue;
            uint32_t base = 0u;
            const int leader = (int)__ffsll((long long)m) - 1;
            if (lane == leader) {
                base = atomicAdd(&plan[2], (uint32_t)__popcll(m));      // one atomic per wave
                if (base + (uint32_t)__popcll(m) < base) plan[3] = 1u;  // wrapped (see k_vox_plan)
            }
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (WRITE && hit)
                frags[base + __popcll(m & ((1ull << lane) - 1ull))] =
                    make_uint2((uint32_t)t, vct_morton3((uint32_t)i, (uint32_t)j, (uint32_t)k));
        }
    }
}

// counting sort of the fragment list by brick: marks, per-slot counts, scatter
__global__ void __launch_bounds__(256)
k_frag_mark(const uint2* __restrict__ frags, uint32_t n, uint32_t* __restrict__ mark) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) mark[frags[i].y >> 9] = 1u;
}
__global__ void __launch_bounds__(256)
k_frag_count(const uint2* __restrict__ frags, uint32_t n, const uint32_t* __restrict__ brick_slot, uint32_t* __restrict__ count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        atomicAdd(&count[brick_slot[frags[i].y >> 9]], 1u);
}
__global__ void __launch_bounds__(256)
k_frag_scatter(const uint2* __restrict__ frags, uint32_t n, const uint32_t* __restrict__ brick_slot,
               const uint32_t* __restrict__ first, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted,
               uint32_t* __restrict__ slot_brick) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 e = frags[i];
        const uint32_t brick = e.y >> 9, slot = brick_slot[brick];
        const uint32_t at = first[slot] + atomicAdd(&cursor[slot], 1u);
        sorted[at] = (e.x << 9) | (e.y & 511u);          // (triangle, voxel inside the brick): 23 + 9 bits
        slot_brick[slot] = brick;                          // benign race: every writer stores the same value
    }
}

// Per sorted fragment, the values of the pass that do not depend on the light: its barycentrics (BARY: geometry only,
// once per mesh) and its albedo (ALB: vox.fs:56 texture(DiffuseTexture, uv) -- mip-mapped -- or the material's colour;
// once per change of the textures / texture coordinates; only scenes with textures store it).  One workgroup per slot
// (the sorted list does not name a fragment's brick).
struct __attribute__((packed, aligned(4))) VctF3 { float v[3]; };
template <bool BARY, bool ALB>
__global__ void __launch_bounds__(256)
k_frag_geom(const VctVoxParams p, float2* __restrict__ bary, VctF3* __restrict__ falb) {
    for (uint32_t slot = blockIdx.x; slot < p.nslots; slot += gridDim.x) {
        const uint32_t first = p.slot_first[slot], n = p.slot_first[slot + 1] - first;
        const uint32_t bm = p.slot_brick[slot] << 9;
        for (uint32_t f = threadIdx.x; f < n; f += blockDim.x) {
            const uint32_t e = p.frag_sorted[first + f];
            const uint32_t vox = bm | (e & 511u);
            const int t = (int)(e >> 9);
            TriSetup r;
            setup_tri(p, t, r);
            const int i = (int)vct_compact3(vox), j = (int)vct_compact3(vox >> 1), k = (int)vct_compact3(vox >> 2);
            float b0, b1;
            frag_bary(r, i, j, k, b0, b1);
            if (BARY) bary[first + f] = make_float2(b0, b1);
            if (ALB) {
                const float b2 = fmaxf(1.0f - b0 - b1, 0.0f);
                float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (r.tex >= 0 && p.tex.mips) frag_duv(r, i, j, k, d);
                VctF3 o;
                frag_albedo(p, r, b0, b1, b2, d, o.v);
                falb[first + f] = o;
            }
        }
    }
}

// front-face unit normal n = normalize(cross(v1-v0, v2-v0)) of every triangle, quantised floor(n*127+.5)+128 (the voxel
// attribute of the second bounce): geometry only, once per mesh
__global__ void __launch_bounds__(256)
k_tri_nrm(const VctVoxParams p, uint32_t* __restrict__ tri_nrm) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.ntri) return;
    F3 w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* q = p.pos + (size_t)t * 9 + 3 * k;
        w[k] = {q[0] * p.model_scale, q[1] * p.model_scale, q[2] * p.model_scale};
    }
    const F3 fn = cross3(sub3(w[1], w[0]), sub3(w[2], w[0]));
    const float fl = __builtin_sqrtf(dot3(fn, fn));
    const float fc[3] = {__fdiv_rn(fn.x, fl), __fdiv_rn(fn.y, fl), __fdiv_rn(fn.z, fl)};
#pragma unroll
    for (int k = 0; k < 3; ++k) tri_nrm[(size_t)t * 3 + k] = (uint32_t)((int)floorf(fc[k] * 127.0f + 0.5f) + 128);
}

// ---- the voxelize pass: one workgroup per work item = up to VCT_VOX_CHUNK fragments of one brick slot ------------
// acc in LDS: [512][2] u64 (sumR | sumG << 32, sumB | count << 32), + [512][3] for the voxel attributes.
// A slot of one chunk (most) is resolved from LDS by its workgroup.  The chunks of a heavier slot add their non-empty
// LDS sums to the slot's accumulators in HBM (acc2 / acc2_attr; integer sums: any order gives the same bits); a SECOND
// small launch, k_vox_resolve_multi, turns those sums into texels and re-zeroes them for the next pass.  (A "last
// chunk resolves" variant with an arrival counter needs a device-scope release / acquire per workgroup -- a write-back
// and invalidate of the XCD's L2 on this GPU -- and ran 4x slower.)
__device__ __forceinline__ void resolve_attr(uint32_t c, unsigned long long q0, unsigned long long q1, unsigned long long q2,
                                             uint32_t& alb, uint32_t& nrm) {
    alb = 0u; nrm = 0u;
    if (c) {
        const uint32_t h = c >> 1;
        alb = (((uint32_t)q0 + h) / c) | ((((uint32_t)(q0 >> 32) + h) / c) << 8) |
              ((((uint32_t)q1 + h) / c) << 16) | 0xff000000u;
        nrm = (((uint32_t)(q1 >> 32) + h) / c) | ((((uint32_t)q2 + h) / c) << 8) |
              ((((uint32_t)(q2 >> 32) + h) / c) << 16) | 0xff000000u;
    }
}

// (4 waves per SIMD: the kernel waits on memory more than it computes; unbounded, hipcc takes 132-134 VGPRs and loses one)
#ifndef VCT_VOX_MIN_BLOCKS
#define VCT_VOX_MIN_BLOCKS 4
#endif
// Threads per work item: 256, or 128 for meshes of many work items (VCT_VOX_SMALL_BLOCK_ITEMS).  The kernel takes its
// stride from blockDim; with half the threads an item runs twice the iterations but twice as many items are resident
// (8 instead of 4 per CU at the same 4 waves per SIMD), which hides the dependent round trips at the head and tail of an
// item better once there are several generations of items anyway.  Street: 1024^3 (40 k items) 0.459 -> 0.408 ms,
// 512^3 0.221 -> 0.205; with few items it loses (street 256^3 0.127 -> 0.145, atrium 0.035 -> 0.048: 2.5 k items are
// one generation of 128-thread workgroups).  512 threads lose everywhere (0.56 / 0.25 / 0.134 / 0.042).
#ifndef VCT_VOX_SMALL_BLOCK_ITEMS
#define VCT_VOX_SMALL_BLOCK_ITEMS 8192u
#endif
#ifndef VCT_VOX_TINY_BLOCK_ITEMS
#define VCT_VOX_TINY_BLOCK_ITEMS 32768u     // from here on one wave per item
#endif
template <bool ATTR, bool FALB>
__global__ void __launch_bounds__(256, VCT_VOX_MIN_BLOCKS)
k_voxelize_bricks(const VctVoxParams p) {
    __shared__ unsigned long long acc[512 * 2];
    __shared__ unsigned long long acc_attr[ATTR ? 512 * 3 : 1];
    const VctF3* __restrict__ falb = reinterpret_cast<const VctF3*>(p.frag_alb);
    for (uint32_t it = blockIdx.x; it < p.nitems; it += gridDim.x) {
        // (slot, first fragment, fragments, index among the multi-chunk slots or ~0: the slot's only chunk): everything
        // the workgroup needs to start on its fragments in ONE scalar load
        const uint4 w = p.items[it];
        const uint32_t slot = w.x, first = w.y, n = w.z, mi = w.w;
        uint32_t* __restrict__ out = p.stage + (size_t)slot * 512;
        if (n == 0u) {                    // a brick only the reference-mode voxelizer can touch: nothing of this mode
            for (uint32_t v = threadIdx.x; v < 512u; v += blockDim.x) {
                out[v] = 0u;
                if (ATTR) { p.stage_albedo[(size_t)slot * 512 + v] = 0u; p.stage_normal[(size_t)slot * 512 + v] = 0u; }
            }
            continue;
        }
        // the first fragment's list entry and light-independent values are requested before the accumulators are zeroed;
        // inside the loop the next fragment's are requested before the current one is evaluated (the kernel waits on
        // memory two thirds of its time at 4 waves per SIMD: every dependent round trip taken off a fragment counts)
        uint32_t f = threadIdx.x;
        uint32_t e = 0u;
        float2 bb = make_float2(0.0f, 0.0f);
        VctF3 fa = {{0.0f, 0.0f, 0.0f}};
        if (f < n) {
            e = p.frag_sorted[first + f];
            bb = p.frag_bary[first + f];
            if (FALB) fa = falb[first + f];
        }
#if VCT_VOX_PREFETCH
        // two list entries and one triangle ahead: the triangle fetch (dependent on the list entry) leaves the chain of
        // dependent round trips of an iteration, which is then the PCF window alone
        uint32_t e1 = 0u;
        if (f + blockDim.x < n) e1 = p.frag_sorted[first + f + blockDim.x];
        VctTri9 rec = {};
        if (p.shadow && f < n) rec = *reinterpret_cast<const VctTri9*>(p.pos + (size_t)(e >> 9) * 9);
#endif
        for (uint32_t v = threadIdx.x; v < 512u * 2u; v += blockDim.x) acc[v] = 0ull;
        if (ATTR) for (uint32_t v = threadIdx.x; v < 512u * 3u; v += blockDim.x) acc_attr[v] = 0ull;
        __syncthreads();
        while (f < n) {
            const uint32_t fn = f + blockDim.x;
            uint32_t e_next = 0u;
            float2 bb_next = bb;
            VctF3 fa_next = fa;
#if VCT_VOX_PREFETCH
            e_next = e1;
            uint32_t e2 = 0u;
            if (fn + blockDim.x < n) e2 = p.frag_sorted[first + fn + blockDim.x];
            VctTri9 rec_next = rec;
            if (fn < n) {
                if (p.shadow) rec_next = *reinterpret_cast<const VctTri9*>(p.pos + (size_t)(e_next >> 9) * 9);
                bb_next = p.frag_bary[first + fn];
                if (FALB) fa_next = falb[first + fn];
            }
#else
            if (fn < n) {
                e_next = p.frag_sorted[first + fn];
                bb_next = p.frag_bary[first + fn];
                if (FALB) fa_next = falb[first + fn];
            }
#endif
            const uint32_t local = e & 511u;
            PassTri r;
#if VCT_VOX_PREFETCH
            setup_pass<ATTR, FALB>(p, (int)(e >> 9), r, &rec);
#else
            setup_pass<ATTR, FALB>(p, (int)(e >> 9), r);
#endif
            const FragValue fv = frag_eval(p, r, bb.x, bb.y, FALB ? fa.v : r.alb);
            atomicAdd(&acc[2 * local], (unsigned long long)fv.r | ((unsigned long long)fv.g << 32));       // ds_add_u64
            atomicAdd(&acc[2 * local + 1], (unsigned long long)fv.b | (1ull << 32));
            if (ATTR) {
                atomicAdd(&acc_attr[3 * local], (unsigned long long)fv.ar | ((unsigned long long)fv.ag << 32));
                atomicAdd(&acc_attr[3 * local + 1], (unsigned long long)fv.ab | ((unsigned long long)r.nrm[0] << 32));
                atomicAdd(&acc_attr[3 * local + 2], (unsigned long long)r.nrm[1] | ((unsigned long long)r.nrm[2] << 32));
            }
            f = fn; e = e_next; bb = bb_next; fa = fa_next;
#if VCT_VOX_PREFETCH
            e1 = e2; rec = rec_next;
#endif
        }
        __syncthreads();
        if (mi == 0xffffffffu) {
            for (uint32_t v = threadIdx.x; v < 512u; v += blockDim.x) {
                const ulonglong2 a = make_ulonglong2(acc[2 * v], acc[2 * v + 1]);
                out[v] = resolve_voxel(a);
                if (ATTR) {
                    uint32_t alb, nrm;
                    resolve_attr((uint32_t)(a.y >> 32), acc_attr[3 * v], acc_attr[3 * v + 1], acc_attr[3 * v + 2], alb, nrm);
                    p.stage_albedo[(size_t)slot * 512 + v] = alb;
                    p.stage_normal[(size_t)slot * 512 + v] = nrm;
                }
            }
            if (threadIdx.x == 0) p.brick_flags[p.slot_brick[slot]] = 1u;
        } else {
            unsigned long long* __restrict__ g = p.acc2 + (size_t)mi * 1024;
            unsigned long long* __restrict__ ga = ATTR ? p.acc2_attr + (size_t)mi * 1536 : nullptr;
            for (uint32_t v = threadIdx.x; v < 512u; v += blockDim.x) {
                const unsigned long long a1 = acc[2 * v + 1];
                if ((a1 >> 32) == 0ull) continue;                    // no fragment of this chunk in the voxel
                atomicAdd(&g[2 * v], acc[2 * v]);
                atomicAdd(&g[2 * v + 1], a1);
                if (ATTR) {
                    atomicAdd(&ga[3 * v], acc_attr[3 * v]);
                    atomicAdd(&ga[3 * v + 1], acc_attr[3 * v + 1]);
                    atomicAdd(&ga[3 * v + 2], acc_attr[3 * v + 2]);
                }
            }
            // (k_vox_resolve_multi, the next launch, turns the sums into texels: a "last chunk resolves" scheme needs a
            // device-scope release / acquire pair per workgroup, which on this GPU is a write-back and an invalidate of the
            // XCD's L2 -- measured: the pass four times slower)
        }
        __syncthreads();           // the accumulators are re-zeroed for the next work item of this workgroup
    }
}

// the multi-chunk slots of the pass: sums -> texels (+ attributes), accumulators back to zero
template <bool ATTR>
__global__ void __launch_bounds__(256)
k_vox_resolve_multi(const VctVoxParams p, const uint32_t* __restrict__ multi_slot, uint32_t nmulti) {
    for (uint32_t mi = blockIdx.x; mi < nmulti; mi += gridDim.x) {
        const uint32_t slot = multi_slot[mi];
        unsigned long long* __restrict__ g = p.acc2 + (size_t)mi * 1024;
        unsigned long long* __restrict__ ga = ATTR ? p.acc2_attr + (size_t)mi * 1536 : nullptr;
        for (uint32_t v = threadIdx.x; v < 512u; v += blockDim.x) {
            const ulonglong2 a = make_ulonglong2(g[2 * v], g[2 * v + 1]);
            p.stage[(size_t)slot * 512 + v] = resolve_voxel(a);
            const bool any = (a.y >> 32) != 0ull;
            if (any) { g[2 * v] = 0ull; g[2 * v + 1] = 0ull; }
            if (ATTR) {
                uint32_t alb = 0u, nrm = 0u;
                if (any) {
                    resolve_attr((uint32_t)(a.y >> 32), ga[3 * v], ga[3 * v + 1], ga[3 * v + 2], alb, nrm);
                    ga[3 * v] = 0ull; ga[3 * v + 1] = 0ull; ga[3 * v + 2] = 0ull;
                }
                p.stage_albedo[(size_t)slot * 512 + v] = alb;
                p.stage_normal[(size_t)slot * 512 + v] = nrm;
            }
        }
        if (threadIdx.x == 0) p.brick_flags[p.slot_brick[slot]] = 1u;
    }
}

// ---- reference mode (S/Voxelization.vs/.gs/.fs as written): dominant-axis projection, V x V raster
// at pixel centres with the top-left rule, voxel index from (x, y, V*depth) (vox.fs:58-86), value
// = unorm8(albedo * PCF/25), a = 1, and "imageStore: last writer wins".  The reference's store order
// is a race; the deterministic reading the oracle uses -- the last triangle in submission order
// wins -- is an order-independent 64-bit atomicMax on (triangle + 1) << 32 | rgb.
struct RefSetup {
    float wx[3], wy[3], wz[3];
    F3 dc[3];
    float area, sgn;
    int x0, x1, y0, y1, axis;
    float alb[3];
    float uv[3][2];
    int tex;
    bool ok;
};

__device__ __forceinline__ void ref_setup(const VctVoxParams& p, int t, RefSetup& r) {
    F3 w[3];
    const float fV = (float)p.V;
    for (int k = 0; k < 3; ++k) {
        const float* q = p.pos + (size_t)t * 9 + 3 * k;
        w[k] = {q[0] * p.model_scale, q[1] * p.model_scale, q[2] * p.model_scale};   // vox.vs:21
        const F3 d = xform_point(p.light_vp, w[k]);                                  // vox.vs:18
        r.dc[k] = {d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f};          // vox.vs:19
    }
    const F3 e1 = sub3(w[0], w[1]), e2 = sub3(w[2], w[0]);                            // vox.gs:24-25
    const F3 nn = cross3(e1, e2);
    const float len = __builtin_sqrtf(dot3(nn, nn));
    const float nx = fabsf(__fdiv_rn(nn.x, len)), ny = fabsf(__fdiv_rn(nn.y, len)),
                nz = fabsf(__fdiv_rn(nn.z, len));
    if (nx >= ny && nx >= nz) r.axis = 1;                                             // vox.gs:34-39
    else if (ny >= nx && ny >= nz) r.axis = 2;
    else r.axis = 3;
    const float* proj = p.proj + 16 * (r.axis - 1);
    for (int k = 0; k < 3; ++k) {
        const F3 ndc = xform_point(proj, w[k]);                                       // vox.gs:47 (w = 1)
        r.wx[k] = (ndc.x * 0.5f + 0.5f) * fV;
        r.wy[k] = (ndc.y * 0.5f + 0.5f) * fV;
        r.wz[k] = ndc.z * 0.5f + 0.5f;
    }
    const float area = (r.wx[1] - r.wx[0]) * (r.wy[2] - r.wy[0]) - (r.wx[2] - r.wx[0]) * (r.wy[1] - r.wy[0]);
    r.ok = !(area == 0.0f || area != area);
    r.sgn = area > 0.0f ? 1.0f : -1.0f;
    r.area = area * r.sgn;
    r.x0 = max((int)floorf(fminf(fminf(r.wx[0], r.wx[1]), r.wx[2])), 0);
    r.x1 = min((int)floorf(fmaxf(fmaxf(r.wx[0], r.wx[1]), r.wx[2])), p.V - 1);
    r.y0 = max((int)floorf(fminf(fminf(r.wy[0], r.wy[1]), r.wy[2])), 0);
    r.y1 = min((int)floorf(fmaxf(fmaxf(r.wy[0], r.wy[1]), r.wy[2])), p.V - 1);
    const float* alb = p.albedo + 4 * (size_t)p.material[t];
    r.alb[0] = alb[0]; r.alb[1] = alb[1]; r.alb[2] = alb[2];
    load_tex_setup(p, t, r);
}

__device__ __forceinline__ void ref_fragment(const VctVoxParams& p, const RefSetup& r, int t, int px, int py) {
    const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
    float e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;            // edge opposite vertex k
        const float dx = (r.wx[b] - r.wx[a]) * r.sgn, dy = (r.wy[b] - r.wy[a]) * r.sgn;
        e[k] = dx * (cy - r.wy[a]) - dy * (cx - r.wx[a]);
        const bool top_left = (dy > 0.0f) || (dy == 0.0f && dx < 0.0f);              // [GL] fill rule
        if (e[k] < 0.0f || (e[k] == 0.0f && !top_left)) return;
    }
    const float l0 = __fdiv_rn(e[0], r.area), l1 = __fdiv_rn(e[1], r.area), l2 = 1.0f - l0 - l1;
    const float fz = l0 * r.wz[0] + l1 * r.wz[1] + l2 * r.wz[2];
    const int V = p.V;
    const int ix = (int)cx, iy = (int)cy, iz = (int)((float)V * fz);                  // vox.fs:58
    int vx, vy, vz;
    if (r.axis == 1) { vx = V - 1 - iz; vz = V - 1 - ix; vy = iy; }                   // vox.fs:70-75
    else if (r.axis == 2) { vz = V - 1 - iy; vy = V - 1 - iz; vx = ix; }              // :76-81
    else { vx = ix; vy = iy; vz = V - 1 - iz; }                                       // :82-86
    if (vx < 0 || vy < 0 || vz < 0 || vx >= V || vy >= V || vz >= V) return;          // [GL] store dropped
    const uint32_t vox = vct_morton3((uint32_t)vx, (uint32_t)vy, (uint32_t)vz);
    if (p.mark_only) { p.brick_mark[vox >> 9] = 1u; return; }
    float sh = 1.0f;
    if (p.shadow) {
        const F3 dc = {l0 * r.dc[0].x + l1 * r.dc[1].x + l2 * r.dc[2].x,
                       l0 * r.dc[0].y + l1 * r.dc[1].y + l2 * r.dc[2].y,
                       l0 * r.dc[0].z + l1 * r.dc[1].z + l2 * r.dc[2].z};
        sh = __fdiv_rn((float)pcf25(p.shadow, p.shadow_ebase, p.shadow_size, dc, 0.002f, p.shadow_tiles), 25.0f);    // vox.fs:46
    }
    float alb[3];
    float duv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (r.tex >= 0 && p.tex.mips) {
        // texture() derivatives: the neighbouring pixel centres of the fragment's 2x2 quad, on this triangle
        auto uv_at = [&](int qx, int qy, float& ou, float& ov) {
            const float nx = (float)qx + 0.5f, ny = (float)qy + 0.5f;
            float f[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int a = (k + 1) % 3, b = (k + 2) % 3;
                const float dx = (r.wx[b] - r.wx[a]) * r.sgn, dy = (r.wy[b] - r.wy[a]) * r.sgn;
                f[k] = dx * (ny - r.wy[a]) - dy * (nx - r.wx[a]);
            }
            const float c0 = __fdiv_rn(f[0], r.area), c1 = __fdiv_rn(f[1], r.area), c2 = 1.0f - c0 - c1;
            ou = c0 * r.uv[0][0] + c1 * r.uv[1][0] + c2 * r.uv[2][0];
            ov = c0 * r.uv[0][1] + c1 * r.uv[1][1] + c2 * r.uv[2][1];
        };
        float mu, mv, xu, xv, yu, yv;
        uv_at(px, py, mu, mv);
        uv_at(px ^ 1, py, xu, xv);
        uv_at(px, py ^ 1, yu, yv);
        duv[0] = xu - mu; duv[1] = xv - mv; duv[2] = yu - mu; duv[3] = yv - mv;
    }
    frag_albedo(p, r, l0, l1, l2, duv, alb);                                          // vox.fs:56
    const unsigned long long rgb = to_unorm8(alb[0] * sh) | (to_unorm8(alb[1] * sh) << 8) |
                                   (to_unorm8(alb[2] * sh) << 16);                   // vox.fs:88
    atomicMax(p.acc + 2 * ((size_t)p.brick_slot[vox >> 9] * 512 + (vox & 511u)),
              ((unsigned long long)(uint32_t)(t + 1) << 32) | rgb);
    p.brick_flags[vox >> 9] = 1u;
}

#define VCT_REF_SMALL 64

__global__ void __launch_bounds__(256)
k_voxelize_reference(const VctVoxParams p, int32_t* big_list, int32_t* big_count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.ntri) return;
    RefSetup r;
    ref_setup(p, t, r);
    if (!r.ok || r.x1 < r.x0 || r.y1 < r.y0) return;
    if ((long long)(r.x1 - r.x0 + 1) * (r.y1 - r.y0 + 1) > VCT_REF_SMALL) {
        big_list[atomicAdd(big_count, 1)] = t;
        return;
    }
    for (int py = r.y0; py <= r.y1; ++py)
        for (int px = r.x0; px <= r.x1; ++px) ref_fragment(p, r, t, px, py);
}

__global__ void __launch_bounds__(256)
k_voxelize_reference_big(const VctVoxParams p, const int32_t* big_list, const int32_t* big_count) {
    const int nbig = *big_count;
    for (int b = blockIdx.x; b < nbig; b += gridDim.x) {
        const int t = big_list[b];
        RefSetup r;
        ref_setup(p, t, r);
        const int bw = r.x1 - r.x0 + 1;
        const long long box = (long long)bw * (r.y1 - r.y0 + 1);
        for (long long i = threadIdx.x; i < box; i += blockDim.x)
            ref_fragment(p, r, t, r.x0 + (int)(i % bw), r.y0 + (int)(i / bw));
    }
}

// accumulators -> RGBA8 level 0 (Morton), rounded mean, a = 255 where any fragment landed.
// One wave per 8^3 brick (512 voxels, one contiguous 8 KiB run of accumulators); bricks that were
// touched neither in this pass nor in the previous one are skipped unless `dense`.
__global__ void __launch_bounds__(256)
k_resolve_sparse(unsigned long long* __restrict__ acc, const uint32_t* __restrict__ brick_slot,
                 uint32_t* __restrict__ level0,
                 uint32_t* __restrict__ flags, uint32_t* __restrict__ prev, uint32_t nbricks,
                 uint32_t brick_voxels, int dense, unsigned long long* __restrict__ acc_attr,
                 uint32_t* __restrict__ attr_albedo, uint32_t* __restrict__ attr_normal, int reference,
                 const uint32_t* __restrict__ stage, const uint32_t* __restrict__ stage_albedo,
                 const uint32_t* __restrict__ stage_normal, int scan64) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    // The flags are read 64 bricks at a time, one brick per lane, and the wave then serves the flagged ones one after the
    // other (round 3: one brick per wave iteration made the scan of a 1024^3 grid's 2 M bricks -- 36 k of them touched --
    // 64 dependent round trips per wave: 0.20 ms).  A wave's 64 bricks are spread over the whole grid
    // (lane * nchunks + chunk): touched bricks cluster in Morton order and would otherwise all fall to a few waves.
    // Only worth it for the largest grids: with few bricks per wave (256^3: one, 512^3: eight) the per-brick scan is a
    // handful of round trips and keeps every wave busy with at most a brick or two, while 64 bricks per wave would put a
    // dozen touched bricks in a row on a few waves (measured: 0.012 -> 0.046 ms at 256^3, 0.20 -> 0.086 ms at 1024^3).
    // group = bricks a wave looks at per iteration: 64 lanes' worth, or 1.
    const uint32_t group = scan64 ? 64u : 1u;
    const uint32_t nchunks = (nbricks + group - 1u) / group;
    for (uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nchunks; c += waves) {
      const uint32_t mine = scan64 ? (uint32_t)lane * nchunks + c : c;
      const bool in = mine < nbricks && (scan64 || lane == 0);
      const uint32_t now_l = in ? flags[mine] : 0u, before_l = in ? prev[mine] : 0u;
      for (unsigned long long todo = __builtin_amdgcn_ballot_w64(in && (dense || (now_l | before_l) != 0u)); todo != 0ull;
           todo &= todo - 1ull) {
        const int src = (int)__ffsll((long long)todo) - 1;
        const uint32_t b = scan64 ? (uint32_t)src * nchunks + c : c;
        const uint32_t now = (uint32_t)__builtin_amdgcn_readlane((int)now_l, src);
        uint32_t* l0 = level0 + (size_t)b * brick_voxels;
        const uint32_t slot = brick_slot[b];
        if (slot == VCT_NO_SLOT) {        // no fragment of this mesh can land here: level 0 is empty
            for (uint32_t v = lane; v < brick_voxels; v += 64) l0[v] = 0u;
            if (lane == 0) { prev[b] = 0u; flags[b] = 0u; }
            continue;
        }
        if (stage) {        // north-star mode: k_voxelize_bricks already resolved the brick into its staging slot
            for (uint32_t v = lane; v < brick_voxels; v += 64) {
                const size_t vox = (size_t)slot * brick_voxels + v;
                l0[v] = stage[vox];
                if (stage_albedo) { attr_albedo[vox] = stage_albedo[vox]; attr_normal[vox] = stage_normal[vox]; }
            }
            if (lane == 0) { prev[b] = now; flags[b] = 0u; }
            continue;
        }
        ulonglong2* a2 = reinterpret_cast<ulonglong2*>(acc) + (size_t)slot * brick_voxels;
        for (uint32_t v = lane; v < brick_voxels; v += 64) {
            const ulonglong2 a = a2[v];
            if (reference) {        // (triangle + 1) << 32 | rgb of the last triangle that stored here
                l0[v] = a.x ? ((uint32_t)a.x & 0xffffffu) | 0xff000000u : 0u;
                if (a.x) a2[v] = make_ulonglong2(0ull, 0ull);
                continue;
            }
            l0[v] = resolve_voxel(a);
            if (a.y) a2[v] = make_ulonglong2(0ull, 0ull);
            if (acc_attr) {
                const size_t vox = (size_t)slot * brick_voxels + v;       // attributes are pooled like the accumulators
                unsigned long long* q = acc_attr + 3 * vox;
                const uint32_t c = (uint32_t)(a.y >> 32);
                uint32_t alb = 0u, nrm = 0u;
                if (c) {
                    const unsigned long long q0 = q[0], q1 = q[1], q2 = q[2];
                    const uint32_t h = c >> 1;
                    alb = (((uint32_t)q0 + h) / c) | ((((uint32_t)(q0 >> 32) + h) / c) << 8) |
                          ((((uint32_t)q1 + h) / c) << 16) | 0xff000000u;
                    nrm = (((uint32_t)(q1 >> 32) + h) / c) | ((((uint32_t)q2 + h) / c) << 8) |
                          ((((uint32_t)(q2 >> 32) + h) / c) << 16) | 0xff000000u;
                    q[0] = 0ull; q[1] = 0ull; q[2] = 0ull;
                }
                attr_albedo[vox] = alb;
                attr_normal[vox] = nrm;
            }
        }
        if (lane == 0) { prev[b] = now; flags[b] = 0u; }
      }
    }
}

__global__ void __launch_bounds__(256)
k_assign_slots(const uint32_t* __restrict__ mark, uint32_t* __restrict__ slot, uint32_t* __restrict__ count,
               uint32_t nbricks) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nbricks; b += gridDim.x * blockDim.x)
        slot[b] = mark[b] ? atomicAdd(count, 1u) : VCT_NO_SLOT;
}

// slot -> brick for EVERY slot (k_frag_scatter only names the bricks that hold a conservative fragment; a slot the
// mark-only run of the reference-mode voxelizer created would keep pointing at brick 0 -- the slot-driven loops of the
// second bounce then visited brick 0 once per such slot: found by the fuzzer as an over-count of cone steps)
__global__ void __launch_bounds__(256)
k_slot_bricks(const uint32_t* __restrict__ slot, uint32_t nbricks, uint32_t* __restrict__ slot_brick) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nbricks; b += gridDim.x * blockDim.x) {
        const uint32_t s = slot[b];
        if (s != VCT_NO_SLOT) slot_brick[s] = b;
    }
}

__global__ void __launch_bounds__(256)
k_unpool(const uint32_t* __restrict__ pooled, const uint32_t* __restrict__ brick_slot, uint32_t* __restrict__ dense,
         uint32_t nbricks) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; b < nbricks; b += waves) {
        const uint32_t slot = brick_slot[b];
        for (uint32_t v = lane; v < 512u; v += 64)
            dense[(size_t)b * 512 + v] = slot == VCT_NO_SLOT ? 0u : pooled[(size_t)slot * 512 + v];
    }
}

}  // namespace

hipError_t vct_launch_assign_slots(const uint32_t* mark, uint32_t* slot, uint32_t* count, uint32_t nbricks, hipStream_t s) {
    hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_assign_slots, dim3((nbricks + 255) / 256 < 4096 ? (nbricks + 255) / 256 : 4096), dim3(256), 0, s,
                       mark, slot, count, nbricks);
    return hipGetLastError();
}

hipError_t vct_launch_slot_bricks(const uint32_t* slot, uint32_t nbricks, uint32_t* slot_brick, hipStream_t s) {
    hipLaunchKernelGGL(k_slot_bricks, dim3((nbricks + 255) / 256 < 4096 ? (nbricks + 255) / 256 : 4096), dim3(256), 0, s,
                       slot, nbricks, slot_brick);
    return hipGetLastError();
}

hipError_t vct_launch_unpool(const uint32_t* pooled, const uint32_t* brick_slot, uint32_t* dense, uint32_t nbricks, hipStream_t s) {
    size_t blocks = ((size_t)nbricks + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_unpool, dim3((unsigned)blocks), dim3(256), 0, s, pooled, brick_slot, dense, nbricks);
    return hipGetLastError();
}

hipError_t vct_launch_vox_plan(const VctVoxParams& p, uint32_t* plan, uint2* frags,
                               int32_t* big_list, bool write, hipStream_t s) {
    if (p.ntri <= 0) return hipSuccess;
    const dim3 grid((p.ntri + 255) / 256), block(256);
    if (write) hipLaunchKernelGGL(k_vox_plan<true>, grid, block, 0, s, p, plan, frags, big_list);
    else hipLaunchKernelGGL(k_vox_plan<false>, grid, block, 0, s, p, plan, frags, big_list);
    return hipGetLastError();
}

hipError_t vct_launch_vox_plan_big(const VctVoxParams& p, const int32_t* big_list, int n_big, uint32_t* plan, uint2* frags,
                                   bool write, hipStream_t s) {
    if (n_big <= 0) return hipSuccess;
    const dim3 grid(n_big < 256 * 8 ? n_big : 256 * 8), block(256);
    if (write) hipLaunchKernelGGL(k_vox_plan_big<true>, grid, block, 0, s, p, big_list, n_big, plan, frags);
    else hipLaunchKernelGGL(k_vox_plan_big<false>, grid, block, 0, s, p, big_list, n_big, plan, frags);
    return hipGetLastError();
}

static unsigned frag_blocks(uint32_t n) {
    const size_t b = ((size_t)n + 255) / 256;
    return (unsigned)(b < 256 * 32 ? (b ? b : 1) : 256 * 32);
}
hipError_t vct_launch_frag_mark(const uint2* frags, uint32_t n, uint32_t* mark, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_frag_mark, dim3(frag_blocks(n)), dim3(256), 0, s, frags, n, mark);
    return hipGetLastError();
}
hipError_t vct_launch_frag_count(const uint2* frags, uint32_t n, const uint32_t* brick_slot, uint32_t* count, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_frag_count, dim3(frag_blocks(n)), dim3(256), 0, s, frags, n, brick_slot, count);
    return hipGetLastError();
}
hipError_t vct_launch_frag_scatter(const uint2* frags, uint32_t n, const uint32_t* brick_slot, const uint32_t* first,
                                   uint32_t* cursor, uint32_t* sorted, uint32_t* slot_brick, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_frag_scatter, dim3(frag_blocks(n)), dim3(256), 0, s, frags, n, brick_slot, first, cursor, sorted,
                       slot_brick);
    return hipGetLastError();
}

hipError_t vct_launch_frag_geom(const VctVoxParams& p, float2* bary, float* falb, hipStream_t s) {
    if (p.nslots == 0u || (!bary && !falb)) return hipSuccess;
    const dim3 grid(p.nslots < 256u * 64u ? p.nslots : 256u * 64u), block(256);
    VctF3* fa = reinterpret_cast<VctF3*>(falb);
    if (bary && falb) hipLaunchKernelGGL((k_frag_geom<true, true>), grid, block, 0, s, p, bary, fa);
    else if (bary) hipLaunchKernelGGL((k_frag_geom<true, false>), grid, block, 0, s, p, bary, fa);
    else hipLaunchKernelGGL((k_frag_geom<false, true>), grid, block, 0, s, p, bary, fa);
    return hipGetLastError();
}

hipError_t vct_launch_tri_nrm(const VctVoxParams& p, uint32_t* tri_nrm, hipStream_t s) {
    if (p.ntri <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tri_nrm, dim3((p.ntri + 255) / 256), dim3(256), 0, s, p, tri_nrm);
    return hipGetLastError();
}

hipError_t vct_launch_voxelize_reference(const VctVoxParams& p, int32_t* big_list, int32_t* big_count,
                                         hipStream_t s) {
    if (p.ntri <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(big_count, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_voxelize_reference, dim3((p.ntri + 255) / 256), dim3(256), 0, s, p, big_list, big_count);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_voxelize_reference_big, dim3(256 * 8), dim3(256), 0, s, p, big_list, big_count);
    return hipGetLastError();
}

hipError_t vct_launch_voxelize(const VctVoxParams& p, hipStream_t s) {
    if (p.nslots == 0u || p.nitems == 0u) return hipSuccess;
    const unsigned blocks = p.nitems < 256u * 64u ? p.nitems : 256u * 64u;
    const bool attr = p.stage_albedo != nullptr, falb = p.frag_alb != nullptr;
    const unsigned threads = p.nitems >= VCT_VOX_TINY_BLOCK_ITEMS ? 64u : (p.nitems >= VCT_VOX_SMALL_BLOCK_ITEMS ? 128u : 256u);
    if (attr && falb) hipLaunchKernelGGL((k_voxelize_bricks<true, true>), dim3(blocks), dim3(threads), 0, s, p);
    else if (attr) hipLaunchKernelGGL((k_voxelize_bricks<true, false>), dim3(blocks), dim3(threads), 0, s, p);
    else if (falb) hipLaunchKernelGGL((k_voxelize_bricks<false, true>), dim3(blocks), dim3(threads), 0, s, p);
    else hipLaunchKernelGGL((k_voxelize_bricks<false, false>), dim3(blocks), dim3(threads), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.nmulti == 0u) return e;
    const unsigned mblocks = p.nmulti < 256u * 16u ? p.nmulti : 256u * 16u;
    if (p.stage_albedo) hipLaunchKernelGGL(k_vox_resolve_multi<true>, dim3(mblocks), dim3(256), 0, s, p, p.multi_slot, p.nmulti);
    else hipLaunchKernelGGL(k_vox_resolve_multi<false>, dim3(mblocks), dim3(256), 0, s, p, p.multi_slot, p.nmulti);
    return hipGetLastError();
}

hipError_t vct_launch_resolve(unsigned long long* acc, const uint32_t* brick_slot, uint32_t* level0, uint32_t* flags,
                              uint32_t* prev, int V, bool dense, unsigned long long* acc_attr,
                              uint32_t* attr_albedo, uint32_t* attr_normal, bool reference, const uint32_t* stage,
                              const uint32_t* stage_albedo, const uint32_t* stage_normal, hipStream_t s) {
    const uint32_t brick_voxels = V >= 8 ? 512u : (uint32_t)(V * V * V);
    const uint32_t nbricks = (uint32_t)(((size_t)V * V * V) / brick_voxels);
    size_t blocks = ((size_t)nbricks + 3) / 4;      // 4 waves per workgroup, one brick per wave
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_resolve_sparse, dim3((unsigned)blocks), dim3(256), 0, s, acc, brick_slot, level0, flags,
                       prev, nbricks, brick_voxels, dense ? 1 : 0, reference ? nullptr : acc_attr, attr_albedo,
                       attr_normal, reference ? 1 : 0, reference ? nullptr : stage, reference ? nullptr : stage_albedo,
                       reference ? nullptr : stage_normal, nbricks > 524288u ? 1 : 0);
    return hipGetLastError();
}
