// vct_raster.hip -- the two raster input stages of the GI path on the GPU.
//
//   shadow map  DrawDepthTexture (VCT.h:192-211, S/Shadow.vs): depth-only orthographic raster from
//               the light, back faces culled, 24-bit depth.
//   G-buffer    the vertex + fixed-function part of the main draw (S/VoxelConeTracing.vs:23-37,
//               perspective raster, depth test LESS, back-face cull: R/main.cpp:55-58) and the
//               per-fragment inputs of S/VoxelConeTracing.fs that are not cone tracing: bump normal
//               (:110-128 with a flat height map), material colours (:167,:209-210), the 25-tap PCF
//               shadow term with its 0.111 scale (:132-163).  Output: the tiled 23-plane G-buffer
//               k_trace_tile reads, written in place -- a frame never leaves HBM.
//
// Rasterisation rules are those of the CPU rasteriser in oracle/vct_oracle_raster.cpp (the checker of these
// stages): near-plane clip, window coordinates snapped to 1/256 pixel, edge functions in double
// (exact with snapped inputs, so shared edges are watertight), pixel-centre sampling, top-left
// rule, CCW front faces.  Visibility is order-independent here: one 64-bit atomicMin per covered
// pixel on (depth bits << 32 | triangle id * 2 + sub-triangle) -- the nearest fragment wins and ties
// go to the earliest triangle, which is what "depth test LESS in submission order" produces.  The
// shading pass then re-derives the winning fragment from its triangle; nothing per-fragment is
// stored besides the 8-byte visibility word (4 bytes of depth in the shadow pass).  A pass is two dependent launches
// (k_raster_vis, k_raster_mid) and no clear: consumers reset the words they read, the counters alternate.
// Round 4 adds a second, bit-identical form of the visibility stage -- tile-binned, the word in a lane's registers
// instead of an L2 atomic (k_bin_setup / _alloc / _fill / _raster, below) -- which the context picks for the main draw
// of scenes with alpha-tested foliage when it measures faster (vct_ctx.h raster_mode).
#include <stddef.h>
#include <string.h>

#include "vct_internal.h"

namespace {

struct RVert { float c[4]; };

__device__ __forceinline__ void xform4(const float* m, float x, float y, float z, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = m[r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r];
}

// Near-plane clip (z >= -w) of a clip-space triangle.  A polygon vertex is lerp(in[a], in[b], t) (b < 0: a copy
// of in[a]); every varying is interpolated the same way.  Nothing here is indexed at run time: a run-time index
// puts the polygon into scratch memory, and with more than ~120 B of scratch per lane the runtime allocates the
// scratch arena of a kernel per dispatch (+1.2 ms per launch, measured) -- appends and reads go through selects.
struct PolyVert { RVert v; int a, b; float t; };
struct ClipPoly {
    PolyVert p0, p1, p2, p3;        // named slots, not an array: the compiler turns "if (n == k) p[k] = x" back into p[n] = x
    int n;
};

__device__ __forceinline__ PolyVert pv_select(bool c, const PolyVert& x, const PolyVert& y) {
    PolyVert r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.v.c[k] = c ? x.v.c[k] : y.v.c[k];
    r.a = c ? x.a : y.a; r.b = c ? x.b : y.b; r.t = c ? x.t : y.t;
    return r;
}

__device__ __forceinline__ void poly_push(ClipPoly& q, const PolyVert& x) {
    q.p0 = pv_select(q.n == 0, x, q.p0);
    q.p1 = pv_select(q.n == 1, x, q.p1);
    q.p2 = pv_select(q.n == 2, x, q.p2);
    q.p3 = pv_select(q.n == 3, x, q.p3);
    ++q.n;
}

// polygon vertex i (1..3) by value
__device__ __forceinline__ PolyVert poly_vertex(const ClipPoly& q, int i) {
    return pv_select(i == 1, q.p1, pv_select(i == 2, q.p2, q.p3));
}

__device__ __forceinline__ void clip_near(const RVert in[3], ClipPoly& p) {
    p.n = 0;
    p.p0.v = in[0]; p.p0.a = 0; p.p0.b = -1; p.p0.t = 0.0f;
    p.p1 = p.p0; p.p2 = p.p0; p.p3 = p.p0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const RVert& a = in[i];
        const RVert& b = in[(i + 1) % 3];
        const float da = a.c[2] + a.c[3], db = b.c[2] + b.c[3];
        if (da >= 0.0f) {
            PolyVert x;
            x.v = a; x.a = i; x.b = -1; x.t = 0.0f;
            poly_push(p, x);
        }
        if ((da >= 0.0f) != (db >= 0.0f)) {
            const float t = __fdiv_rn(da, da - db);
            PolyVert x;
#pragma unroll
            for (int k = 0; k < 4; ++k) x.v.c[k] = a.c[k] + (b.c[k] - a.c[k]) * t;
            x.a = i; x.b = (i + 1) % 3; x.t = t;
            poly_push(p, x);
        }
    }
}

// the three polygon vertices of fan sub-triangle f (1 or 2): 0, f, f + 1
struct FanTri { PolyVert v[3]; };
__device__ __forceinline__ FanTri fan_tri(const ClipPoly& q, int f) {
    FanTri r;
    r.v[0] = q.p0;
    r.v[1] = poly_vertex(q, f);
    r.v[2] = poly_vertex(q, f + 1);
    return r;
}

// true when no vertex is behind the near plane: the polygon is the triangle itself.  Almost every
// triangle takes this path, which never indexes a vertex array at run time (a run-time index forces
// the whole ClipPoly into scratch memory).
__device__ __forceinline__ bool unclipped(const RVert in[3]) {
    return in[0].c[2] + in[0].c[3] >= 0.0f && in[1].c[2] + in[1].c[3] >= 0.0f && in[2].c[2] + in[2].c[3] >= 0.0f;
}

struct SubTri {
    double sx[3], sy[3], area, sgn;
    float sz[3], iw[3];
    int x0, x1, y0, y1;
    bool ok;
    // alpha test (trace.fs:169-172; a discarded fragment writes neither colour nor depth): 0 = opaque, 1 = every
    // fragment discarded (flat alpha < 0.5), 2 = per fragment from the diffuse texture `tex` at the
    // perspective-correct (u, v)
    int alpha_mode, tex;
    float tu[3], tv[3];
};

// [ys0, ys1): scissor in pixel rows (a multi-GPU rank rasterises only its slab; the whole frame is [0, H))
__device__ __forceinline__ void setup_subtri(const RVert* v0, const RVert* v1, const RVert* v2, int W,
                                             int H, int ys0, int ys1, SubTri& s) {
    const RVert* v[3] = {v0, v1, v2};
    s.ok = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!(v[k]->c[3] > 1e-20f)) return;
        s.iw[k] = __fdiv_rn(1.0f, v[k]->c[3]);
        s.sx[k] = floor((double)((v[k]->c[0] * s.iw[k] * 0.5f + 0.5f) * (float)W) * 256.0 + 0.5) / 256.0;
        s.sy[k] = floor((double)((v[k]->c[1] * s.iw[k] * 0.5f + 0.5f) * (float)H) * 256.0 + 0.5) / 256.0;
        s.sz[k] = v[k]->c[2] * s.iw[k] * 0.5f + 0.5f;
    }
    double area = (s.sx[1] - s.sx[0]) * (s.sy[2] - s.sy[0]) - (s.sx[2] - s.sx[0]) * (s.sy[1] - s.sy[0]);
    if (area == 0.0 || area != area) return;
    if (area < 0.0) return;                             // back face (CCW = front), always culled
    s.sgn = 1.0;
    s.area = area;
    s.x0 = max(0, (int)floor(fmin(fmin(s.sx[0], s.sx[1]), s.sx[2])));
    s.x1 = min(W - 1, (int)floor(fmax(fmax(s.sx[0], s.sx[1]), s.sx[2])));
    s.y0 = max(ys0, (int)floor(fmin(fmin(s.sy[0], s.sy[1]), s.sy[2])));
    s.y1 = min(ys1 - 1, (int)floor(fmax(fmax(s.sy[0], s.sy[1]), s.sy[2])));
    s.ok = s.x1 >= s.x0 && s.y1 >= s.y0;
    s.alpha_mode = 0;
    s.tex = -1;
}

// What k_raster_vis hands to k_raster_mid for every sub-triangle it lists: the finished set-up, so that the list consumers
// do not clip, project and snap the triangle again (that re-derivation -- ~600 instructions, a third of them fp64 -- was
// more than half of what a 16-lane group spent on a typical 64-pixel triangle).  Window coordinates travel as integers
// (x 256: they are snapped to 1/256 pixel, so this is exact); the area is recomputed from them by the same expression.
// A record whose coordinates do not fit (a near-clipped triangle projected far outside the frame) is marked invalid and
// the consumer falls back to rebuild_subtri.  96 bytes; group entries fill the array from the front, wave entries from
// the back (a sub-triangle is in at most one list, and there are at most 2 per triangle).
struct SubTriRec {
    int32_t fx[3], fy[3];
    float sz[3], iw[3];
    int16_t x0, x1, y0, y1;
    int32_t alpha_tex;          // alpha_mode | (tex + 1) << 2 | valid << 31
    float tu[3], tv[3];
    int32_t pad[3];
};
static_assert(sizeof(SubTriRec) == 96, "SubTriRec is written and read as six 16-byte words");

__device__ __forceinline__ void pack_subtri(const SubTri& s, SubTriRec& r) {
    bool fits = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        fits = fits && fabs(s.sx[k]) < 8388607.0 && fabs(s.sy[k]) < 8388607.0;
        r.fx[k] = (int32_t)(s.sx[k] * 256.0);
        r.fy[k] = (int32_t)(s.sy[k] * 256.0);
        r.sz[k] = s.sz[k]; r.iw[k] = s.iw[k];
        r.tu[k] = s.tu[k]; r.tv[k] = s.tv[k];
    }
    fits = fits && s.x1 < 32768 && s.y1 < 32768;
    r.x0 = (int16_t)s.x0; r.x1 = (int16_t)s.x1; r.y0 = (int16_t)s.y0; r.y1 = (int16_t)s.y1;
    r.alpha_tex = s.alpha_mode | ((s.tex + 1) << 2) | (fits ? (int32_t)0x80000000 : 0);
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
}

// false: the record is marked invalid (rebuild the set-up from the mesh)
__device__ __forceinline__ bool unpack_subtri(const SubTriRec& r, SubTri& s) {
    if (r.alpha_tex >= 0) return false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.sx[k] = (double)r.fx[k] / 256.0;
        s.sy[k] = (double)r.fy[k] / 256.0;
        s.sz[k] = r.sz[k]; s.iw[k] = r.iw[k];
        s.tu[k] = r.tu[k]; s.tv[k] = r.tv[k];
    }
    s.area = (s.sx[1] - s.sx[0]) * (s.sy[2] - s.sy[0]) - (s.sx[2] - s.sx[0]) * (s.sy[1] - s.sy[0]);
    s.sgn = 1.0;
    s.x0 = r.x0; s.x1 = r.x1; s.y0 = r.y0; s.y1 = r.y1;
    s.alpha_mode = r.alpha_tex & 3;
    s.tex = ((r.alpha_tex & 0x7fffffff) >> 2) - 1;
    s.ok = true;
    return true;
}

// pixel-centre coverage + barycentrics; returns false when the pixel is not covered
// The three edge functions as plane equations, for the list consumers' inner loops: e_k(cx, cy) = a_k*cx + b_k*cy + c_k
// with the top-left rule folded into c_k (an edge that does not own its boundary gets c_k - 2^-16, so that "inside" is
// e'_k >= 0 for every edge; bias_k restores e_k for the barycentrics).  EXACT -- the same real numbers as the
// difference-of-products form of cover(), which the oracle states -- as long as every window coordinate is below
// 2^16 in magnitude: coordinates are multiples of 2^-8, pixel centres of 2^-1, so every product and partial sum is a
// multiple of 2^-16 below 2^37 and fits a double's 53 bits.  Sub-triangles that reach further (near-clipped ones
// projected far outside the frame) keep the general form, whose roundings the fast form could not reproduce.
#ifndef VCT_SHADE_PREFETCH
#define VCT_SHADE_PREFETCH 1
#endif
#ifndef VCT_FAST_COVER
#define VCT_FAST_COVER 1      // 0: every sub-triangle takes the general form (A/B measurements)
#endif
struct FastEdges {
    double a[3], b[3], c[3], bias[3];
    double rcp;              // RN(1 / area)
    bool ok;
};
// x / area, correctly rounded, from the sub-triangle's correctly rounded reciprocal y = RN(1 / area): two
// Newton-Raphson corrections of q = x * y, each with the exact remainder of an FMA.  The first leaves q within one ulp of
// the quotient (|x*y - x/area| <= |x/area| * 2^-53 before the product's own rounding); Markstein's theorem (IBM J. Res.
// Dev. 34(1), 1990; restated in Muller et al., Handbook of Floating-Point Arithmetic, ch. "Division") then makes
// RN(q + r * y), r = x - area * q, THE correctly rounded quotient -- the value the `/` of the general form and of the
// oracle returns -- for every x, with no case distinction (no overflow or underflow can occur: |x|, area are
// multiples of 2^-16 below 2^37).  Five FMA-rate operations instead of the ~11 (one of them a quarter-rate
// v_rcp_f64) of an IEEE division; the six divisions of an alpha-tested fragment share the reciprocal.
__device__ __forceinline__ double div_area(double x, double area, double y) {
    double q = x * y;
    q = fma(fma(-area, q, x), y, q);
    q = fma(fma(-area, q, x), y, q);
    // x = -0: the corrections turn the quotient into +0 (-0 + +0), the division keeps -0 -- and the sign of a zero
    // barycentric does reach the G-buffer (a sum of zero products keeps it).  area > 0, so the quotient's sign is x's.
    return copysign(q, x);
}
// the general form's companion: only the reciprocal
__device__ __forceinline__ FastEdges slow_edges(const SubTri& s) {
    FastEdges f;
    f.ok = false;
    f.rcp = 1.0 / s.area;
    return f;
}
__device__ __forceinline__ void make_fast(const SubTri& s, FastEdges& f) {
    f.ok = VCT_FAST_COVER != 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) f.ok = f.ok && fabs(s.sx[k]) < 65536.0 && fabs(s.sy[k]) < 65536.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = (k + 1) % 3, j = (k + 2) % 3;
        const double dx = (s.sx[j] - s.sx[i]) * s.sgn, dy = (s.sy[j] - s.sy[i]) * s.sgn;
        const bool top_left = (dy > 0.0) || (dy == 0.0 && dx < 0.0);
        f.bias[k] = top_left ? 0.0 : 0x1p-16;
        f.a[k] = -dy;
        f.b[k] = dx;
        f.c[k] = (dy * s.sx[i] - dx * s.sy[i]) - f.bias[k];
    }
    f.rcp = 1.0 / s.area;
}

// pixel-centre coverage + barycentrics; returns false when the pixel is not covered
template <bool FAST>
__device__ __forceinline__ bool cover(const SubTri& s, const FastEdges& f, int px, int py, float& b0, float& b1,
                                      float& b2, float& z, double& e0, double& e1) {
    const double cx = (double)px + 0.5, cy = (double)py + 0.5;
    double e[3];
    if (FAST) {
#pragma unroll
        for (int k = 0; k < 3; ++k) e[k] = fma(f.a[k], cx, fma(f.b[k], cy, f.c[k]));
        if (e[0] < 0.0 || e[1] < 0.0 || e[2] < 0.0) return false;
        // (an edge function that is exactly zero may come out as +0 here and as -0 in the general form, or the other
        // way round.  The plane-equation form is only used by plot(), whose results are the depth -- normalised below,
        // and a zero product never changes a non-zero sum -- and the alpha test, which a zero's sign cannot change
        // either; k_gbuffer_shade, whose interpolated attributes DO keep the sign of a zero, uses the general form.)
        e[0] += f.bias[0];
        e[1] += f.bias[1];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3, b = (k + 2) % 3;
            const double dx = (s.sx[b] - s.sx[a]) * s.sgn, dy = (s.sy[b] - s.sy[a]) * s.sgn;
            e[k] = dx * (cy - s.sy[a]) - dy * (cx - s.sx[a]);
            const bool top_left = (dy > 0.0) || (dy == 0.0 && dx < 0.0);
            if (e[k] < 0.0 || (e[k] == 0.0 && !top_left)) return false;
        }
    }
    b0 = (float)div_area(e[0], s.area, f.rcp);
    b1 = (float)div_area(e[1], s.area, f.rcp);
    e0 = e[0]; e1 = e[1];
    b2 = 1.0f - b0 - b1;
    z = b0 * s.sz[0] + b1 * s.sz[1] + b2 * s.sz[2];
    z = z + 0.0f;      // -0 -> +0: depth is ordered through its bit pattern below
    // far-plane clip (and NaN); z == 1 can never pass "LESS" against a depth buffer cleared to 1
    return z >= 0.0f && z < 1.0f;
}

// vct_selftest_area_divide: div_area next to the IEEE division on `count` pseudo-random (x, area) pairs -- integers of
// 1..53 bits (every width equally often, so small and large operands and all alignments of the quotient's rounding
// bit occur) scaled by 2^-16, x of either sign, area > 0.  A superset of what the raster feeds it.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256)
k_area_divide_selftest(uint64_t seed, uint64_t count, unsigned long long* out) {
    unsigned long long bad = 0ull, example = 0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h0 = mix64(seed + 3ull * i), h1 = mix64(seed + 3ull * i + 1ull), h2 = mix64(seed + 3ull * i + 2ull);
        const int bx = 1 + (int)(h2 % 53ull), ba = 1 + (int)((h2 >> 8) % 53ull);
        const uint64_t xi = h0 & ((1ull << bx) - 1ull);
        const uint64_t ai = (h1 & ((1ull << ba) - 1ull)) | (1ull << (ba - 1));
        const double x = ((h2 >> 16) & 1ull) ? -(double)xi * 0x1p-16 : (double)xi * 0x1p-16;       // -0 included
        const double area = (double)ai * 0x1p-16;
        const double want = x / area, got = div_area(x, area, 1.0 / area);
        if (__double_as_longlong(want) != __double_as_longlong(got)) { ++bad; example = i; }
    }
    if (bad) { atomicAdd(out, bad); out[1] = example; }
}

// perspective-correct interpolation of two per-vertex values from the first two barycentrics (as doubles, before their
// conversion to float)
__device__ __forceinline__ void interp2_from(const SubTri& s, double q0, double q1, const float a[3], const float b[3],
                                             float& oa, float& ob) {
    const float c0 = (float)q0, c1 = (float)q1, c2 = 1.0f - c0 - c1;
    const float r0 = c0 * s.iw[0], r1 = c1 * s.iw[1], r2 = c2 * s.iw[2];
    const float rs = __fdiv_rn(1.0f, r0 + r1 + r2);
    oa = (r0 * a[0] + r1 * a[1] + r2 * a[2]) * rs;
    ob = (r0 * b[0] + r1 * b[1] + r2 * b[2]) * rs;
}

// The sub-triangle's perspective-correct interpolation of two per-vertex values at the centre of pixel (qx, qy), with
// no coverage test: how texture() obtains its implicit derivatives -- the fragment's quad neighbours (qx ^ 1, qy) and
// (qx, qy ^ 1), helper invocations when they fall outside the triangle (oracle raster_triangle uv_at).
__device__ __forceinline__ void interp2_at(const SubTri& s, double rcp, int qx, int qy, const float a[3],
                                           const float b[3], float& oa, float& ob) {
    const double cx = (double)qx + 0.5, cy = (double)qy + 0.5;
    double f[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (k + 1) % 3, j = (k + 2) % 3;
        const double dx = (s.sx[j] - s.sx[i]) * s.sgn, dy = (s.sy[j] - s.sy[i]) * s.sgn;
        f[k] = dx * (cy - s.sy[i]) - dy * (cx - s.sx[i]);
    }
    interp2_from(s, div_area(f[0], s.area, rcp), div_area(f[1], s.area, rcp), a, b, oa, ob);
}

struct RasterParams {
    const float* pos;            // [ntri][9] model space
    int32_t ntri;
    float model_scale;
    float vp[16];                // column-major view-projection applied to world positions
    int32_t W, H;
    int32_t ys0, ys1;            // scissor: pixel rows [ys0, ys1)
    unsigned long long* vis;     // [H][W] (depth bits << 32) | (tri * 2 + sub); ~0 = empty
    int32_t* wave_list;          // (tri * 2 + sub) of medium sub-triangles (one wave each)
    uint32_t* wave_count;
    int32_t* group_list;         // (tri * 2 + sub) of small-medium sub-triangles (one 16-lane group each)
    uint32_t* group_count;
    uint32_t* vis32;             // depth-only pass: the shadow-map words (depth24 bits + epoch base), else null
    uint32_t vis32_ebase;
    uint2* items;                // tile work items of huge sub-triangles: (tri * 2 + sub, tile_y << 16 | tile_x)
    uint32_t* item_count;
    uint32_t item_capacity;
    uint32_t* next_counts;       // [3] the counters the NEXT pass will use: zeroed by this pass (no memset launch)
    SubTriRec* recs;             // [2 * ntri] set-up records of the listed sub-triangles: group entries from the front, wave entries from the back
    // alpha test of the main draw (null material: depth-only pass, S/Shadow.fs has no alpha test)
    // tri_alpha (optional): per triangle alpha_mode | (diffuse texture + 1) << 2, precomputed once per mesh / texture set
    // (k_tri_alpha) -- one load that travels with the triangle's positions instead of the chain material -> texture
    // index -> descriptor flags (three dependent round trips in a kernel that is nothing but dependent round trips)
    const int32_t* tri_alpha;
    const int32_t* material;
    const float* albedo;
    VctTextures tex;
};

__device__ __forceinline__ void load_clip_tri(const RasterParams& p, int t, RVert in[3]) {
    const VctTri9 q = *reinterpret_cast<const VctTri9*>(p.pos + (size_t)t * 9);       // three wide loads, not nine
#pragma unroll
    for (int k = 0; k < 3; ++k)
        xform4(p.vp, q.v[3 * k] * p.model_scale, q.v[3 * k + 1] * p.model_scale, q.v[3 * k + 2] * p.model_scale, in[k].c);
}

// Work granularity by bounding-box size (the bench scene: half of the visible triangles cover <= 64
// pixels, 91-100 % <= 256, the largest ~2k; a Cornell wall covers the whole frame):
#ifndef VCT_RASTER_SMALL
#define VCT_RASTER_SMALL 16      // <= this many pixels: rasterised inline by the triangle's own thread
#endif
#ifndef VCT_RASTER_GROUP
#define VCT_RASTER_GROUP 256     // <= this many: a 16-lane group (4 sub-triangles per wave), lanes stride the box
#endif
#ifndef VCT_RASTER_WAVE
#define VCT_RASTER_WAVE 4096     // <= this many: one wave, lanes stride the bounding box
#endif
#define VCT_RTILE 16             // larger: cut into 16x16-pixel work items by the wave that met the triangle

// Decides how fragments of triangle t are alpha-tested and, for the per-fragment case, loads the texture
// coordinates of the sub-triangle's three vertices (fan == null: the unclipped triangle; otherwise the vertices
// of the near-clipped polygon's fan sub-triangle, interpolated like every other varying).
// what setup_alpha derives per triangle: alpha_mode | (texture + 1) << 2
__device__ __forceinline__ int alpha_class(const RasterParams& p, int t) {
    const int m = p.material[t];
    const int td = vct_tex_of(p.tex, m, 0);
    if (td < 0) return p.albedo[4 * (size_t)m + 3] < 0.5f ? 1 : 0;
    if (!(p.tex.desc[td].flags & 1u)) return 0;           // every texel opaque: alpha = 1 wherever it is sampled
    return 2 | ((td + 1) << 2);
}
__global__ void __launch_bounds__(256)
k_tri_alpha(const RasterParams p, int32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < p.ntri) out[t] = alpha_class(p, t);
}

// `pre`: the triangle's word of tri_alpha when the caller has already loaded it (>= 0), else -1
__device__ __forceinline__ void setup_alpha(const RasterParams& p, int t, const FanTri* fan, SubTri& s, int pre = -1) {
    if (!p.material) return;
    const int cls = pre >= 0 ? pre : (p.tri_alpha ? p.tri_alpha[t] : alpha_class(p, t));
    if ((cls & 3) != 2) { s.alpha_mode = cls & 3; return; }
    s.alpha_mode = 2;
    s.tex = (cls >> 2) - 1;
    const float* uv = p.tex.uv + (size_t)t * 6;
    if (!fan) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s.tu[k] = uv[2 * k]; s.tv[k] = uv[2 * k + 1]; }
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = fan->v[k].a, b = fan->v[k].b;
        const float ua = uv[2 * a], va = uv[2 * a + 1];
        s.tu[k] = b < 0 ? ua : ua + (uv[2 * b] - ua) * fan->v[k].t;
        s.tv[k] = b < 0 ? va : va + (uv[2 * b + 1] - va) * fan->v[k].t;
    }
}

template <bool FAST>
__device__ __forceinline__ void plot(const RasterParams& p, const SubTri& s, const FastEdges& fe, int px, int py,
                                     unsigned long long id) {
    float b0, b1, b2, z;
    double e0 = 0.0, e1 = 0.0;         // FAST: the fragment's first two edge functions, for the quad neighbours below
    if (s.alpha_mode == 1) return;
    if (!cover<FAST>(s, fe, px, py, b0, b1, b2, z, e0, e1)) return;
    const size_t pix = (size_t)py * p.W + px;
    if (p.vis32) {        // depth-only pass (shadow map): the nearest depth is all that is kept, as the 24-bit depth
        // the map will show (quantisation is monotonic: the minimum of the quantised depths is the quantised minimum)
        atomicMin(&p.vis32[pix], vct_depth24_bits(z) + p.vis32_ebase);
        return;
    }
    const unsigned long long word = ((unsigned long long)__float_as_uint(z) << 32) | id;
    // Early depth test in front of the alpha test: the visibility word of a pixel only ever decreases during a pass
    // (atomicMin), so a fragment whose word is not below what the pixel shows NOW can never win later -- dropping it is
    // exact, whatever the order the fragments arrive in.  It keeps alpha-tested foliage affordable: most fragments of
    // a tree crown lie behind others and never reach the mip-mapped texture fetch (Bistro-class street, G-buffer
    // pass at 1080p: 20.8 -> 10.4 ms).  The load is served by L2, where the atomics execute (a CU's L1 is never
    // refreshed by other CUs' atomics); a stale value would only be conservative.  Opaque fragments and the shadow
    // pass go straight to the atomic: there the extra round trip costs more than it saves (atrium: +15 %).
    if (s.alpha_mode == 2 &&
        word >= __hip_atomic_load(&p.vis[pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (s.alpha_mode == 2) {
        const float q0 = b0 * s.iw[0], q1 = b1 * s.iw[1], q2 = b2 * s.iw[2];
        const float qs = __fdiv_rn(1.0f, q0 + q1 + q2);
        const float u = (q0 * s.tu[0] + q1 * s.tu[1] + q2 * s.tu[2]) * qs;
        const float v = (q0 * s.tv[0] + q1 * s.tv[1] + q2 * s.tv[2]) * qs;
        float alpha;
        if (p.tex.mips) {      // texture(DiffuseTexture, tex) with the quad differences of tex
            float ux, vx, uy, vy;
            if (FAST) {     // one pixel to the side: e_k changes by exactly a_k (b_k), the same numbers interp2_at forms
                const double sx = (px & 1) ? -1.0 : 1.0, sy = (py & 1) ? -1.0 : 1.0;
                interp2_from(s, div_area(fma(sx, fe.a[0], e0), s.area, fe.rcp),
                             div_area(fma(sx, fe.a[1], e1), s.area, fe.rcp), s.tu, s.tv, ux, vx);
                interp2_from(s, div_area(fma(sy, fe.b[0], e0), s.area, fe.rcp),
                             div_area(fma(sy, fe.b[1], e1), s.area, fe.rcp), s.tu, s.tv, uy, vy);
            } else {
                interp2_at(s, fe.rcp, px ^ 1, py, s.tu, s.tv, ux, vx);
                interp2_at(s, fe.rcp, px, py ^ 1, s.tu, s.tv, uy, vy);
            }
            alpha = vct_tex_sample_lod(p.tex, s.tex, u, v, ux - u, vx - v, uy - u, vy - v).w;
        } else {
            alpha = vct_tex_sample(p.tex, s.tex, u, v).w;
        }
        if (alpha < 0.5f) return;                                                // trace.fs:171 discard
    }
    atomicMin(&p.vis[pix], word);
}

// tile range of a sub-triangle's bounding box
struct TileBox { int tx0, ty0, tw, th; };
__device__ __forceinline__ TileBox tile_box(const SubTri& s) {
    TileBox t;
    t.tx0 = s.x0 / VCT_RTILE; t.ty0 = s.y0 / VCT_RTILE;
    t.tw = s.x1 / VCT_RTILE - t.tx0 + 1; t.th = s.y1 / VCT_RTILE - t.ty0 + 1;
    return t;
}

// forward: the sub-triangle of a list entry, rebuilt from the mesh
__device__ __forceinline__ bool rebuild_subtri(const RasterParams& p, int id, SubTri& s);

// A huge sub-triangle met by one lane of k_raster_vis is cut into 16x16-pixel work items by the whole wave:
// one atomic reserves the item range, the 64 lanes write the items (a full-screen Cornell wall at 1080p is
// 8160 items = 128 stores per lane).  Items that do not fit the buffer are rasterised right here.
__device__ __forceinline__ void emit_big_wave(const RasterParams& p, int id, int lane) {
    SubTri s;
    if (!rebuild_subtri(p, id, s)) return;              // wave-uniform
    const TileBox tb = tile_box(s);
    const int n = tb.tw * tb.th;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(p.item_count, (uint32_t)n);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    for (int i = lane; i < n; i += 64) {
        const int tx = tb.tx0 + i % tb.tw, ty = tb.ty0 + i / tb.tw;
        const uint32_t slot = base + (uint32_t)i;
        if (slot < p.item_capacity) {
            p.items[slot] = make_uint2((uint32_t)id, ((uint32_t)ty << 16) | (uint32_t)tx);
            continue;
        }
        const int x0 = max(s.x0, tx * VCT_RTILE), x1 = min(s.x1, tx * VCT_RTILE + VCT_RTILE - 1);
        const int y0 = max(s.y0, ty * VCT_RTILE), y1 = min(s.y1, ty * VCT_RTILE + VCT_RTILE - 1);
        const FastEdges fe = slow_edges(s);
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) plot<false>(p, s, fe, px, py, (unsigned long long)(uint32_t)id);
    }
}

// Wave-aggregated list append, called by all 64 lanes: the slot of this lane's entry (meaningful where `pred`).
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred, int lane) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
    if (m == 0ull) return 0u;
    const int leader = (int)__ffsll((long long)m) - 1;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// one thread per triangle: clip, cull, classify; tiny sub-triangles are rasterised inline, huge ones are cut into
// tile work items by the wave, the rest go to the group / wave lists of k_raster_mid
#ifndef VCT_VIS_MIN_BLOCKS
#define VCT_VIS_MIN_BLOCKS 3
#endif
__global__ void __launch_bounds__(256, VCT_VIS_MIN_BLOCKS)
k_raster_vis(const RasterParams p) {
    __shared__ uint32_t lds_cnt[4][2];
    __shared__ unsigned long long lds_base;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (t == 0) { p.next_counts[0] = 0u; p.next_counts[1] = 0u; p.next_counts[2] = 0u; }
    const bool valid = t < p.ntri;          // no early return: the wave cooperates on huge triangles below
    RVert in[3];
    ClipPoly poly;
    poly.n = 0;
    bool whole = false;
    int acls = -1;
    if (valid) {
        if (p.material && p.tri_alpha) acls = p.tri_alpha[t];      // requested with the positions
        load_clip_tri(p, t, in);
        whole = unclipped(in);
        poly.n = 3;
        if (!whole) clip_near(in, poly);
    }
    if (whole || !valid) {
        poly.p0.v.c[0] = poly.p0.v.c[1] = poly.p0.v.c[2] = poly.p0.v.c[3] = 0.0f;
        poly.p0.a = 0; poly.p0.b = -1; poly.p0.t = 0.0f;
        poly.p1 = poly.p0; poly.p2 = poly.p0; poly.p3 = poly.p0;
    }
    unsigned long long big1 = 0ull, big2 = 0ull;
#pragma unroll
    for (int f = 1; f <= 2; ++f) {          // a near-clipped triangle is at most a quad: two sub-triangles
        bool big = false, to_group = false, to_wave = false;
        const int id = t * 2 + (f - 1);
        SubTriRec rec;
        rec.alpha_tex = 0;
        if (valid && f + 1 < poly.n) {
            SubTri s;
            const FanTri fan = fan_tri(poly, f);
            if (whole) setup_subtri(&in[0], &in[1], &in[2], p.W, p.H, p.ys0, p.ys1, s);
            else setup_subtri(&fan.v[0].v, &fan.v[1].v, &fan.v[2].v, p.W, p.H, p.ys0, p.ys1, s);
            if (s.ok) setup_alpha(p, t, whole ? nullptr : &fan, s, acls);
            if (s.ok && s.alpha_mode != 1) {
                const long long box = (long long)(s.x1 - s.x0 + 1) * (s.y1 - s.y0 + 1);
                if (box <= VCT_RASTER_SMALL) {
                    const FastEdges fe = slow_edges(s);
                    for (int py = s.y0; py <= s.y1; ++py)
                        for (int px = s.x0; px <= s.x1; ++px) plot<false>(p, s, fe, px, py, (unsigned long long)(uint32_t)id);
                } else if (box <= VCT_RASTER_GROUP) {
                    to_group = true;
                    pack_subtri(s, rec);
                } else if (box <= VCT_RASTER_WAVE) {
                    to_wave = true;
                    pack_subtri(s, rec);
                } else {
                    big = true;
                }
            }
        }
        // List appends.  Returning atomics on ONE address are executed by the L2 one after the other (~10 ns each): a
        // per-lane atomicAdd on the two counters was 1.2 ms per pass (round 2), one per wave and list still 4,000
        // (atrium) to 90,000 (the street's 2.8 M triangles) in a row -- 40 us and 0.5 ms, most of this kernel's time
        // (round 3, found on the second bounce's list).  Now the workgroup's four waves pool their counts in LDS and
        // ONE 64-bit atomic reserves both lists for all 256 triangles (wave count low, group count high: neither can
        // carry, the lists hold at most 2 * ntri entries).  The second sub-triangle exists only for near-clipped
        // triangles: rare, it keeps the per-wave append.
        uint32_t gslot, wslot;
        if (f == 1) {
            const unsigned long long mg = __builtin_amdgcn_ballot_w64(to_group), mw = __builtin_amdgcn_ballot_w64(to_wave);
            const int wv = (int)(threadIdx.x >> 6);
            if (lane == 0) { lds_cnt[wv][0] = (uint32_t)__popcll(mg); lds_cnt[wv][1] = (uint32_t)__popcll(mw); }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t tg = 0u, tw = 0u;
                for (int w = 0; w < 4; ++w) { tg += lds_cnt[w][0]; tw += lds_cnt[w][1]; }
                lds_base = (tg | tw) ? atomicAdd(reinterpret_cast<unsigned long long*>(p.wave_count),
                                                 (unsigned long long)tw | ((unsigned long long)tg << 32)) : 0ull;
            }
            __syncthreads();
            const unsigned long long base = lds_base;
            uint32_t gb = (uint32_t)(base >> 32), wb = (uint32_t)base;
            for (int w = 0; w < wv; ++w) { gb += lds_cnt[w][0]; wb += lds_cnt[w][1]; }
            gslot = gb + __builtin_amdgcn_mbcnt_hi((uint32_t)(mg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mg, 0u));
            wslot = wb + __builtin_amdgcn_mbcnt_hi((uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, 0u));
        } else {
            gslot = wave_append(p.group_count, to_group, lane);
            wslot = wave_append(p.wave_count, to_wave, lane);
        }
        if (to_group) { p.group_list[gslot] = (int32_t)id; p.recs[gslot] = rec; }
        if (to_wave) { p.wave_list[wslot] = (int32_t)id; p.recs[(size_t)2 * p.ntri - 1 - wslot] = rec; }
        if (f == 1) big1 = __builtin_amdgcn_ballot_w64(big); else big2 = __builtin_amdgcn_ballot_w64(big);
    }
    // huge sub-triangles of this wave's 64 triangles, one after the other, all lanes helping (outside the loop
    // above so that its triangle set-up is dead here: fewer live registers)
    const int t0 = t - lane;
    // (bits of the second mask are moved behind the first: one loop, one copy of the emit code)
    unsigned long long pending = big1;
    int k = 0;
    while (true) {
        if (pending == 0ull) {
            if (k == 1) break;
            k = 1;
            pending = big2;
            continue;
        }
        const int src = (int)__ffsll((long long)pending) - 1;
        pending &= pending - 1ull;
        emit_big_wave(p, (t0 + src) * 2 + k, lane);
    }
}

__device__ __forceinline__ bool rebuild_subtri(const RasterParams& p, int id, SubTri& s) {
    const int t = id >> 1, f = (id & 1) + 1;
    RVert in[3];
    load_clip_tri(p, t, in);
    if (unclipped(in)) {
        setup_subtri(&in[0], &in[1], &in[2], p.W, p.H, p.ys0, p.ys1, s);
        if (s.ok) setup_alpha(p, t, nullptr, s);
        return s.ok;
    }
    ClipPoly poly;
    clip_near(in, poly);
    const FanTri fan = fan_tri(poly, f);
    setup_subtri(&fan.v[0].v, &fan.v[1].v, &fan.v[2].v, p.W, p.H, p.ys0, p.ys1, s);
    if (s.ok) setup_alpha(p, t, &fan, s);
    return s.ok;
}

// The three list consumers in ONE launch (they are independent of each other, and a dependent dispatch costs
// ~5 us of drain + launch latency on this GPU even when its list is empty): blocks [0, gblocks) serve the group
// list, [gblocks, gblocks + wblocks) the wave list, the rest the tile work items.
//   group: one 16-lane group per small-medium sub-triangle (half of the visible sub-triangles of the bench scene
//          cover <= 64 pixels: a whole wave per triangle would leave most lanes idle after the first iteration)
//   wave:  one wave per medium sub-triangle, 64 bounding-box pixels per iteration
//   tile:  one workgroup per 16x16-pixel work item of a huge sub-triangle, one pixel per thread
#ifndef VCT_MID_MIN_BLOCKS
#define VCT_MID_MIN_BLOCKS 4
#endif
__global__ void __launch_bounds__(256, VCT_MID_MIN_BLOCKS)
k_raster_mid(const RasterParams p, const int gblocks, const int wblocks) {
    int b = blockIdx.x;
    if (b < gblocks) {
        const uint32_t n = *p.group_count;
        const int l16 = threadIdx.x & 15;
        const uint32_t ngroups = ((uint32_t)gblocks * blockDim.x) >> 4;
        for (uint32_t g = ((uint32_t)b * blockDim.x + threadIdx.x) >> 4; g < n; g += ngroups) {
            const int id = p.group_list[g];
            SubTri s;
            if (!unpack_subtri(p.recs[g], s) && !rebuild_subtri(p, id, s)) continue;
            const int bw = s.x1 - s.x0 + 1;
            const int box = bw * (s.y1 - s.y0 + 1);
            FastEdges fe;
            make_fast(s, fe);
            // the box is walked in strides of 16 without a division per pixel: (x, y) advance by (16 % bw, 16 / bw)
            const int qs = 16 / bw, rs = 16 - qs * bw;
            int x = l16 % bw, y = l16 / bw;
            if (fe.ok) {
                for (int i = l16; i < box; i += 16) {
                    plot<true>(p, s, fe, s.x0 + x, s.y0 + y, (unsigned long long)(uint32_t)id);
                    x += rs; y += qs;
                    if (x >= bw) { x -= bw; ++y; }
                }
            } else {
                for (int i = l16; i < box; i += 16)
                    plot<false>(p, s, fe, s.x0 + i % bw, s.y0 + i / bw, (unsigned long long)(uint32_t)id);
            }
        }
        return;
    }
    b -= gblocks;
    if (b < wblocks) {
        const uint32_t n = *p.wave_count;
        const int lane = threadIdx.x & 63;
        const uint32_t nwaves = ((uint32_t)wblocks * blockDim.x) >> 6;
        for (uint32_t w = ((uint32_t)b * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
            const int id = p.wave_list[w];
            SubTri s;
            if (!unpack_subtri(p.recs[(size_t)2 * p.ntri - 1 - w], s) && !rebuild_subtri(p, id, s)) continue;
            const int bw = s.x1 - s.x0 + 1;
            const int box = bw * (s.y1 - s.y0 + 1);
            FastEdges fe;
            make_fast(s, fe);
            const int qs = 64 / bw, rs = 64 - qs * bw;
            int x = lane % bw, y = lane / bw;
            if (fe.ok) {
                for (int i = lane; i < box; i += 64) {
                    plot<true>(p, s, fe, s.x0 + x, s.y0 + y, (unsigned long long)(uint32_t)id);
                    x += rs; y += qs;
                    if (x >= bw) { x -= bw; ++y; }
                }
            } else {
                for (int i = lane; i < box; i += 64)
                    plot<false>(p, s, fe, s.x0 + i % bw, s.y0 + i / bw, (unsigned long long)(uint32_t)id);
            }
        }
        return;
    }
    b -= wblocks;
    const int tblocks = (int)gridDim.x - gblocks - wblocks;
    const uint32_t n = min(*p.item_count, p.item_capacity);
    for (uint32_t it = (uint32_t)b; it < n; it += (uint32_t)tblocks) {
        const uint2 e = p.items[it];
        SubTri s;
        if (!rebuild_subtri(p, (int)e.x, s)) continue;
        const int px = (int)(e.y & 0xffffu) * VCT_RTILE + (int)(threadIdx.x % VCT_RTILE);
        const int py = (int)(e.y >> 16) * VCT_RTILE + (int)(threadIdx.x / VCT_RTILE);
        if (px < s.x0 || px > s.x1 || py < s.y0 || py > s.y1) continue;
        FastEdges fe;
        make_fast(s, fe);
        if (fe.ok) plot<true>(p, s, fe, px, py, (unsigned long long)e.x);
        else plot<false>(p, s, fe, px, py, (unsigned long long)e.x);
    }
}


// ====================================================================================================================
// Tile-binned visibility (round 4; SURVEY.md 8 f1 / f2, R/Voxel_Cone_Tracing.h:146-190,192-211).
//
// The direct form above resolves visibility with one L2 atomicMin per covered pixel and pays, per listed sub-triangle
// and 16-lane group, for a set-up that every lane repeats.  On the Bistro-class street (BASELINE configs[4]: 1.27 M
// visible sub-triangles whose bounding boxes add up to 305 M pixels at 4K, 80 % of them alpha-tested foliage) that was
// 2.4 of the 3.45 ms G-buffer pass.  The binned form is four dependent launches:
//
//   k_bin_setup   one thread per triangle: clip, cull, set up (exactly as k_raster_vis), write ONE 160-byte record per
//                 visible sub-triangle -- plane equations of the three edge functions, reciprocal area, depths, 1/w,
//                 alpha-test state -- and count it into every 16x16-pixel bin its bounding box overlaps (fire-and-
//                 forget atomics on counters a cache line apart).  Sub-triangles over more than VCT_BIN_INLINE bins
//                 are counted by the whole wave, those over more than VCT_BIN_HUGE (a Cornell wall) go to a short
//                 list that every bin scans.
//   k_bin_alloc   one thread per bin: its range of the entry array (wave-aggregated cursor), its work items -- slices of
//                 at most VCT_BIN_SLICE entries -- and the bin counter back to zero.
//   k_bin_fill    one wave per 64 records, lane = (record, bin) pair: an 8-byte entry (sort key | quadrants touched |
//                 record) into each bin a record reaches.
//   k_bin_raster  one workgroup per work item; wave w owns the 8x8-pixel quadrant w of the bin, LANE = PIXEL.  The
//                 entries ARE the sort keys: opaque first, then alpha-tested front to back by a lower bound of the
//                 sub-triangle's depth (bitonic in LDS, only when the slice has alpha-tested entries).  Records are
//                 gathered into LDS 64 at a time by the whole workgroup (one round trip per chunk instead of one per
//                 entry) and read back as broadcasts: the coverage test of 64 pixels is nine fp64 operations on
//                 wave-uniform operands instead of a per-lane set-up.  A pixel's visibility word lives in ITS lane's
//                 registers: no atomic, and the early depth test in front of the alpha fetch costs nothing.  An entry
//                 whose depth bound lies behind everything the quadrant already shows is dropped before its record is
//                 touched.  texture()'s implicit derivatives are what they are in hardware: the quad neighbours' own
//                 interpolations, exchanged by DPP (lanes are laid out so that a 2x2 quad is four consecutive lanes)
//                 -- the same numbers as the oracle's helper invocations, because a lane evaluates the identical
//                 expression at its own centre.  One plain 8-byte store per visible pixel at the end (atomicMin only
//                 where several slices share a bin).
//
// Every arithmetic step that decides a result (snapping, edge functions, barycentrics, depth, alpha) is the code of the
// direct form, so the two are bit-identical; VCT_RASTER_PATH=direct keeps the old path selectable for A/B runs.
#define VCT_BIN 16
#define VCT_BIN_SHIFT 4
#define VCT_BIN_SLICE 512           // entries per work item of k_bin_raster (LDS: 4 KiB of sort keys)
#ifndef VCT_BIN_FILL_AGGREGATE
#define VCT_BIN_FILL_AGGREGATE 1      // k_bin_fill: one returning atomic per distinct bin of a wave's 64 pairs
#endif
#ifndef VCT_BIN_BUCKET_ORDER
#define VCT_BIN_BUCKET_ORDER 1        // k_bin_raster: slice entries bucketed by depth instead of sorted
#endif
#ifndef VCT_BIN_ORDER_MIN
#define VCT_BIN_ORDER_MIN 16          // slices without alpha-tested entries are ordered too from this many entries on (front to back:
                                      // the hierarchical depth test rejects more; atrium 95 -> 62 us, depth-only street 300 -> 212)
#endif
#ifndef VCT_BIN_BUCKETS
#define VCT_BIN_BUCKETS 32            // per class (opaque / alpha-tested); a multiple of 32
#endif
#ifndef VCT_BIN_COUNT_AGGREGATE
#define VCT_BIN_COUNT_AGGREGATE 1     // k_bin_setup: the same for the counting atomics
#endif
#ifndef VCT_BIN_COUNT_AGG_ROUNDS
#define VCT_BIN_COUNT_AGG_ROUNDS 8
#endif
#ifndef VCT_BIN_FILL_AGG_ROUNDS
#define VCT_BIN_FILL_AGG_ROUNDS 16
#endif
#ifndef VCT_BIN_CHUNK
#define VCT_BIN_CHUNK 64            // records staged in LDS at a time (10 KiB); <= 64: one header per lane.  16 / 32 / 96 / 128 measured: all slower
#endif
#define VCT_BIN_INLINE 16           // bins a thread counts / fills by itself
#define VCT_BIN_HUGE 2048           // above: the record goes to the huge list instead of one entry per bin
// VCT_BIN_CSTRIDE (vct_internal.h): words between two bins' counters -- device-scope atomics on one line queue up

struct BinRec {
    double ea[3], eb[3], ec[3];     // fast: e_k(cx, cy) = ea*cx + eb*cy + ec (top-left bias folded in); general: sx[3], sy[3], -
    double area, rcp;
    float sz[3], iw[3];
    float tu[3], tv[3];
    uint32_t id;                    // triangle * 2 + sub-triangle: the tie-breaker of the visibility word
    uint32_t binmask;               // (at most VCT_BIN_INLINE bins) bit j: the triangle -- not only its box -- reaches bin j of the box's bins, row by row
    // the next four are one aligned 16-byte word: all k_bin_fill reads of a record
    uint32_t flags;                 // 0-1 alpha_mode, 2-26 texture + 1, 27 "has per-bin entries", 28-30 "edge k does not own its boundary", 31 fast form
    uint32_t xx, yy;                // x0 | x1 << 16, y0 | y1 << 16 (clipped to the frame and the scissor)
    uint32_t zmin_bits;             // float bits of a lower bound of every depth the sub-triangle can produce (>= 0)
};
static_assert(sizeof(BinRec) == 160, "BinRec is staged as ten 16-byte words");
static_assert(offsetof(BinRec, flags) == 144, "flags / xx / yy / zmin_bits are read as one dwordx4");
#define VCT_BINREC_BINNED 0x08000000u

struct BinParams {
    RasterParams r;
    BinRec* recs;
    uint32_t rec_cap;
    uint2* entries;                 // [entry_cap + 1] (record | quadrants << 28, alpha-tested << 31 | depth bound), bin after bin; the last is a spare
    uint32_t entry_cap;
    // A bin's counter / cursor has 128 bytes to itself: atomics of several XCDs on one cache line make the line migrate
    // between their L2s (~11 ns each); with the counters of neighbouring bins -- the bins a tree crown covers -- in one
    // line, k_bin_setup took 0.52 ms instead of 0.27 on the Bistro-class street.
    uint32_t* bin_count;            // [nbins * VCT_BIN_CSTRIDE] zero between passes
    uint32_t* bin_cursor;           // [nbins * VCT_BIN_CSTRIDE]
    uint4* items;                   // (bin, first entry, entries, flags: 1 = the bin has several slices, 2 = first slice)
    uint32_t item_cap;
    uint32_t* huge;                 // records every bin scans
    uint32_t huge_cap;
    // [0] entries promised, [1] records (one 64-bit word), [2] items, [3] entry cursor (one 64-bit word), [4] huge
    // records, [5] "something was rasterised in place" (capacity overflow): k_bin_raster must merge with atomicMin
    uint32_t* ctr;
    uint32_t* next_ctr;             // the other counter set: zeroed by this pass for the next
    int32_t bins_x, bins_y;
};

__device__ __forceinline__ void pack_binrec(const SubTri& s, const FastEdges& fe, int id, bool binned, uint32_t binmask, BinRec& r) {
    uint32_t nb = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (fe.ok) { r.ea[k] = fe.a[k]; r.eb[k] = fe.b[k]; r.ec[k] = fe.c[k]; }
        else { r.ea[k] = s.sx[k]; r.eb[k] = s.sy[k]; r.ec[k] = 0.0; }
        if (fe.bias[k] != 0.0) nb |= 1u << (28 + k);
        r.sz[k] = s.sz[k]; r.iw[k] = s.iw[k];
        r.tu[k] = s.alpha_mode == 2 ? s.tu[k] : 0.0f; r.tv[k] = s.alpha_mode == 2 ? s.tv[k] : 0.0f;
    }
    r.area = s.area;
    r.rcp = fe.rcp;
    r.id = (uint32_t)id;
    r.binmask = binmask;
    r.flags = (uint32_t)s.alpha_mode | ((uint32_t)(s.tex + 1) << 2) | nb | (fe.ok ? 0x80000000u : 0u) | (binned ? VCT_BINREC_BINNED : 0u);
    r.xx = (uint32_t)s.x0 | ((uint32_t)s.x1 << 16);
    r.yy = (uint32_t)s.y0 | ((uint32_t)s.y1 << 16);
    // z = b0*sz0 + b1*sz1 + b2*sz2 with b0, b1 in [0, 1] and b2 = 1 - b0 - b1 >= -2^-23: within 1e-6 of a convex
    // combination for |sz| <= 2; the bound is only used to DROP entries that cannot win, so it errs low
    const float lo = fminf(fminf(s.sz[0], s.sz[1]), s.sz[2]);
    const float lb = (lo > -2.0f && lo < 2.0f) ? lo - 4e-6f : 0.0f;
    r.zmin_bits = lb > 0.0f ? __float_as_uint(lb) : 0u;
}

// Which of the (at most VCT_BIN_INLINE) bins of its bounding box does the sub-triangle itself reach?  A bin is dropped
// when one edge function is negative on every pixel centre of the bin's part of the box -- its largest value over that
// rectangle is taken at the corner its gradient points to.  Exact on the plane equations (make_fast), so no bin with a
// covered pixel is ever dropped; the general form keeps every bin.  A quarter of the (sub-triangle, bin) pairs of the
// Bistro-class street go: fewer counter atomics here and in k_bin_fill, fewer entries for k_bin_raster to walk.
__device__ __forceinline__ uint32_t bin_reach_mask(const SubTri& s, const FastEdges& fe, int bx0, int by0, int bw, int nb) {
    if (!fe.ok || nb <= 1) return nb >= 32 ? 0xffffffffu : (1u << nb) - 1u;
    uint32_t mask = 0u;
    int bx = 0, by = 0;
    for (int j = 0; j < nb; ++j) {
        const int xa = max(s.x0, (bx0 + bx) << VCT_BIN_SHIFT), xb = min(s.x1, ((bx0 + bx) << VCT_BIN_SHIFT) + VCT_BIN - 1);
        const int ya = max(s.y0, (by0 + by) << VCT_BIN_SHIFT), yb = min(s.y1, ((by0 + by) << VCT_BIN_SHIFT) + VCT_BIN - 1);
        bool reach = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double cx = (double)(fe.a[k] >= 0.0 ? xb : xa) + 0.5, cy = (double)(fe.b[k] >= 0.0 ? yb : ya) + 0.5;
            if (fma(fe.a[k], cx, fma(fe.b[k], cy, fe.c[k])) < 0.0) reach = false;
        }
        if (reach) mask |= 1u << j;
        if (++bx == bw) { bx = 0; ++by; }
    }
    return mask;
}

#ifndef VCT_BINSETUP_MIN_BLOCKS
#define VCT_BINSETUP_MIN_BLOCKS 3
#endif
__global__ void __launch_bounds__(256, VCT_BINSETUP_MIN_BLOCKS)
k_bin_setup(const BinParams p) {
    __shared__ uint32_t lds_cnt[4][2];
    __shared__ unsigned long long lds_base;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) p.next_ctr[k] = 0u;
    }
    const bool valid = t < p.r.ntri;
    RVert in[3];
    ClipPoly poly;
    poly.n = 0;
    bool whole = false;
    int acls = -1;
    if (valid) {
        if (p.r.material && p.r.tri_alpha) acls = p.r.tri_alpha[t];      // requested with the positions
        load_clip_tri(p.r, t, in);
        whole = unclipped(in);
        poly.n = 3;
        if (!whole) clip_near(in, poly);
    }
    if (whole || !valid) {
        poly.p0.v.c[0] = poly.p0.v.c[1] = poly.p0.v.c[2] = poly.p0.v.c[3] = 0.0f;
        poly.p0.a = 0; poly.p0.b = -1; poly.p0.t = 0.0f;
        poly.p1 = poly.p0; poly.p2 = poly.p0; poly.p3 = poly.p0;
    }
    // sub-triangles whose bins the wave counts together (more than VCT_BIN_INLINE): first bin, bins per row | bins
    unsigned long long coop[2] = {0ull, 0ull};
    uint32_t coop_b0[2] = {0u, 0u}, coop_wn[2] = {0u, 0u};
    uint32_t in_place = 0u;       // sub-triangles that found neither a record, entry room nor a slot of the huge list
#pragma unroll
    for (int f = 1; f <= 2; ++f) {
        const int id = t * 2 + (f - 1);
        bool have = false;
        SubTri s;
        s.ok = false;
        if (valid && f + 1 < poly.n) {
            const FanTri fan = fan_tri(poly, f);
            if (whole) setup_subtri(&in[0], &in[1], &in[2], p.r.W, p.r.H, p.r.ys0, p.r.ys1, s);
            else setup_subtri(&fan.v[0].v, &fan.v[1].v, &fan.v[2].v, p.r.W, p.r.H, p.r.ys0, p.r.ys1, s);
            if (s.ok) setup_alpha(p.r, t, whole ? nullptr : &fan, s, acls);
            have = s.ok && s.alpha_mode != 1;
        }
        int bx0 = 0, by0 = 0, bw = 0, nb = 0;
        if (have) {
            bx0 = s.x0 >> VCT_BIN_SHIFT; by0 = s.y0 >> VCT_BIN_SHIFT;
            bw = (s.x1 >> VCT_BIN_SHIFT) - bx0 + 1;
            nb = bw * ((s.y1 >> VCT_BIN_SHIFT) - by0 + 1);
        }
        FastEdges fe;
        fe.ok = false;
        uint32_t binmask = 0xffffffffu;
        if (have) {
            make_fast(s, fe);
            if (nb <= VCT_BIN_INLINE) {
                binmask = bin_reach_mask(s, fe, bx0, by0, bw, nb);
                if (binmask == 0u) have = false;          // the box reaches pixel centres, the triangle none of them
            }
        }
        // entries this sub-triangle asks for
        const uint32_t np = !have ? 0u : (nb <= VCT_BIN_INLINE ? (uint32_t)__popc(binmask) : (nb <= VCT_BIN_HUGE ? (uint32_t)nb : 0u));
        // Reservation of the record and of room in the entry array.  Returning atomics on one address execute one
        // after the other (~11 ns each): one 64-bit atomic per WORKGROUP (entries low, records high) for the first
        // sub-triangle; the second exists for near-clipped triangles only and keeps a per-wave reservation.
        const unsigned long long mh = __builtin_amdgcn_ballot_w64(have);
        uint32_t incl = np;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const uint32_t wave_ent = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t rslot, eslot;
        if (f == 1) {
            if (lane == 0) { lds_cnt[wv][0] = wave_ent; lds_cnt[wv][1] = (uint32_t)__popcll(mh); }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t te = 0u, tr = 0u;
                for (int w = 0; w < 4; ++w) { te += lds_cnt[w][0]; tr += lds_cnt[w][1]; }
                lds_base = (te | tr) ? atomicAdd(reinterpret_cast<unsigned long long*>(p.ctr),
                                                 (unsigned long long)te | ((unsigned long long)tr << 32)) : 0ull;
            }
            __syncthreads();
            const unsigned long long base = lds_base;
            uint32_t eb = (uint32_t)base, rb = (uint32_t)(base >> 32);
            for (int w = 0; w < wv; ++w) { eb += lds_cnt[w][0]; rb += lds_cnt[w][1]; }
            eslot = eb + incl - np;
            rslot = rb + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
        } else {
            unsigned long long base = 0ull;
            if (mh != 0ull) {       // wave-uniform
                if (lane == 0) base = atomicAdd(reinterpret_cast<unsigned long long*>(p.ctr),
                                                (unsigned long long)wave_ent | ((unsigned long long)__popcll(mh) << 32));
                base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            }
            eslot = (uint32_t)base + incl - np;
            rslot = (uint32_t)(base >> 32) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
        }
        bool wave_bins = false;
        if (have) {
            if (rslot >= p.rec_cap) {
                in_place |= 1u << (f - 1);
            } else {
                // the promise of entry room holds when the whole promised range lies inside the array: then every
                // entry k_bin_fill writes for this record has a place (the bins' ranges are packed by k_bin_alloc)
                const bool room = np != 0u && eslot + np <= p.entry_cap && eslot + np >= eslot;
                bool listed = false;
                if (!room) {                                             // huge, or the entry array is full
                    const uint32_t h = atomicAdd(&p.ctr[4], 1u);
                    if (h < p.huge_cap) { p.huge[h] = rslot; listed = true; }
                }
                pack_binrec(s, fe, id, room, binmask, p.recs[rslot]);
                if (room) {
                    if (nb <= VCT_BIN_INLINE) {
                        int bx = 0, by = 0;
                        for (int j = 0; j < nb; ++j) {
                            const uint32_t bin = (uint32_t)((by0 + by) * p.bins_x + bx0 + bx);
#if VCT_BIN_COUNT_AGGREGATE
                            // the lanes still in this loop that count the same bin send ONE atomic (see k_bin_fill)
                            uint32_t want = ((binmask >> j) & 1u) ? bin : 0xffffffffu, add = 0u;
                            unsigned long long left = __builtin_amdgcn_ballot_w64(want != 0xffffffffu);
                            int singles = 0;
                            for (int round = 0; round < VCT_BIN_COUNT_AGG_ROUNDS && left != 0ull && singles < 2; ++round) {
                                const int l = (int)__ffsll((long long)left) - 1;
                                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)want, l);
                                const unsigned long long m = __builtin_amdgcn_ballot_w64(want == b);
                                if (lane == l) add = (uint32_t)__popcll(m);
                                singles = (m & (m - 1ull)) == 0ull ? singles + 1 : 0;
                                left &= ~m;
                            }
                            if ((left >> lane) & 1ull) add = 1u;
                            if (add != 0u) atomicAdd(&p.bin_count[(size_t)bin * VCT_BIN_CSTRIDE], add);
#else
                            if ((binmask >> j) & 1u) atomicAdd(&p.bin_count[(size_t)bin * VCT_BIN_CSTRIDE], 1u);        // result unused: no round trip
#endif
                            if (++bx == bw) { bx = 0; ++by; }
                        }
                    } else {
                        wave_bins = true;
                        coop_b0[f - 1] = (uint32_t)bx0 | ((uint32_t)by0 << 16);
                        coop_wn[f - 1] = (uint32_t)bw | ((uint32_t)nb << 16);
                    }
                } else if (!listed) {
                    in_place |= 1u << (f - 1);
                }
            }
        }
        coop[f - 1] = __builtin_amdgcn_ballot_w64(wave_bins);
    }
    // sub-triangles over many bins, one after the other, 64 bins per step
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned long long pending = coop[k];
        while (pending != 0ull) {
            const int src = (int)__ffsll((long long)pending) - 1;
            pending &= pending - 1ull;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)coop_b0[k], src);
            const uint32_t wn = (uint32_t)__builtin_amdgcn_readlane((int)coop_wn[k], src);
            const int bx0 = (int)(b0 & 0xffffu), by0 = (int)(b0 >> 16), bw = (int)(wn & 0xffffu), nb = (int)(wn >> 16);
            for (int j = lane; j < nb; j += 64)
                atomicAdd(&p.bin_count[(size_t)((by0 + j / bw) * p.bins_x + bx0 + j % bw) * VCT_BIN_CSTRIDE], 1u);
        }
    }
    // Last resort (every capacity exhausted): the owner rasterises in place with global atomics -- correct, slow -- and
    // raises the flag that makes k_bin_raster merge its words with atomicMin instead of storing them.
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        if (!((in_place >> k) & 1u)) continue;
        SubTri s;
        if (!rebuild_subtri(p.r, t * 2 + k, s)) continue;
        p.ctr[5] = 1u;
        const FastEdges fe = slow_edges(s);
        for (int py = s.y0; py <= s.y1; ++py)
            for (int px = s.x0; px <= s.x1; ++px) plot<false>(p.r, s, fe, px, py, (unsigned long long)(uint32_t)(t * 2 + k));
    }
}

// one thread per bin: entry range, work items, counter reset
__global__ void __launch_bounds__(256)
k_bin_alloc(const BinParams p) {
    const int nbins = p.bins_x * p.bins_y;
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t nhuge = min(p.ctr[4], p.huge_cap);
    uint32_t n = 0u;
    bool rows = false;
    if (bin < nbins) {
        n = p.bin_count[(size_t)bin * VCT_BIN_CSTRIDE];
        p.bin_count[(size_t)bin * VCT_BIN_CSTRIDE] = 0u;
        const int y = (bin / p.bins_x) << VCT_BIN_SHIFT;
        rows = y < p.r.ys1 && y + VCT_BIN > p.r.ys0;
    }
    // a bin without entries still gets one item while the huge list is not empty (a Cornell wall covers every bin)
    const uint32_t ni = n ? (n + VCT_BIN_SLICE - 1u) / VCT_BIN_SLICE : ((nhuge && rows) ? 1u : 0u);
    uint32_t in_n = n, in_i = ni;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(in_n, off), b = __shfl_up(in_i, off);
        if (lane >= off) { in_n += a; in_i += b; }
    }
    const uint32_t tot_n = (uint32_t)__builtin_amdgcn_readlane((int)in_n, 63), tot_i = (uint32_t)__builtin_amdgcn_readlane((int)in_i, 63);
    unsigned long long base = 0ull;
    if (tot_i != 0u) {
        if (lane == 0) base = atomicAdd(reinterpret_cast<unsigned long long*>(p.ctr + 2),
                                        (unsigned long long)tot_i | ((unsigned long long)tot_n << 32));
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    }
    const uint32_t ebase = (uint32_t)(base >> 32) + in_n - n, ibase = (uint32_t)base + in_i - ni;
    if (bin < nbins) p.bin_cursor[(size_t)bin * VCT_BIN_CSTRIDE] = ebase;
    for (uint32_t sl = 0; sl < ni; ++sl) {
        const uint32_t at = ibase + sl;
        const uint32_t cnt = n ? min((uint32_t)VCT_BIN_SLICE, n - sl * VCT_BIN_SLICE) : 0u;
        if (at < p.item_cap) p.items[at] = make_uint4((uint32_t)bin, ebase + sl * VCT_BIN_SLICE, cnt, (ni > 1u ? 1u : 0u) | (sl == 0u ? 2u : 0u));
    }
}

// the entry of record `rec` (header q = flags, xx, yy, zmin_bits) in bin (bx, by): sort key high, quadrants | record low
__device__ __forceinline__ uint2 bin_entry(uint32_t rec, const uint4& q, int bx, int by, bool depth_only) {
    const int x0 = (int)(q.y & 0xffffu), x1 = (int)(q.y >> 16), y0 = (int)(q.z & 0xffffu), y1 = (int)(q.z >> 16);
    const int ox = bx << VCT_BIN_SHIFT, oy = by << VCT_BIN_SHIFT;
    const bool l = x0 <= ox + 7, r = x1 >= ox + 8, u = y0 <= oy + 7, d = y1 >= oy + 8;      // halves of the bin the box reaches
    const uint32_t qm = (l && u ? 1u : 0u) | (r && u ? 2u : 0u) | (l && d ? 4u : 0u) | (r && d ? 8u : 0u);
    const bool alpha = !depth_only && (q.x & 3u) == 2u;
    return make_uint2((qm << 28) | rec, (alpha ? 0x80000000u : 0u) | q.w);
}

// (Measured and dropped: counters per class of workgroups (blockIdx % 8, one XCD each) with the slots of k_bin_setup's
// returning atomics stored per record, so that this kernel is a pure permutation -- it fell to 0.10 ms, but k_bin_setup
// rose from 0.27 to 0.44 ms waiting for its atomics' results: 4.1 M returning atomics cost ~0.18 ms wherever they are.)
// one wave per 64 records: its lanes load the headers, then share the records' (record, bin) pairs evenly -- pair k of
// the wave belongs to the record whose prefix range holds k (binary search over the 64 prefix sums in LDS).  One
// returning atomic + one 8-byte store per pair and lane, every lane busy whatever the records' bin counts are
// (a thread looping over its own record's bins waited for each store in turn: 200 us on the street).
__global__ void __launch_bounds__(256)
k_bin_fill(const BinParams p) {
    __shared__ uint32_t s_pre[4][64];
    __shared__ uint4 s_hdr[4][64];
    __shared__ uint32_t s_msk[4][64];
    const uint32_t nrec = min(p.ctr[1], p.rec_cap);
    const bool depth_only = p.r.vis32 != nullptr;
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const uint32_t nround = (nrec + 63u) & ~63u;          // whole waves
    for (uint32_t rec = blockIdx.x * blockDim.x + threadIdx.x; rec < nround; rec += gridDim.x * blockDim.x) {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        uint32_t msk = 0u;
        if (rec < nrec) { q = *reinterpret_cast<const uint4*>(&p.recs[rec].flags); msk = p.recs[rec].binmask; }
        const bool binned = (q.x & VCT_BINREC_BINNED) != 0u;
        const int bx0 = (int)(q.y & 0xffffu) >> VCT_BIN_SHIFT, by0 = (int)(q.z & 0xffffu) >> VCT_BIN_SHIFT;
        const int bw = ((int)(q.y >> 16) >> VCT_BIN_SHIFT) - bx0 + 1;
        const uint32_t nbox = (uint32_t)(bw * (((int)(q.z >> 16) >> VCT_BIN_SHIFT) - by0 + 1));
        // a record of at most VCT_BIN_INLINE bins has an entry in the bins of its mask, a larger one in every bin of its box
        const uint32_t nb = !binned ? 0u : (nbox <= VCT_BIN_INLINE ? (uint32_t)__popc(msk) : nbox);
        if (nbox > VCT_BIN_INLINE) msk = 0u;           // 0: "every bin"
        uint32_t incl = nb;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        s_pre[wv][lane] = incl;                        // inclusive: pair k belongs to the first lane with incl > k
        s_hdr[wv][lane] = q;
        s_msk[wv][lane] = msk;
        // (LDS operations of one wave execute in order: no barrier between these writes and the reads below)
        const uint32_t rec0 = rec - (uint32_t)lane;
        // four pairs per lane in flight: their atomics are issued together, and the stores are unconditional (a pair that
        // does not exist writes the spare entry behind the array) -- a store under a branch waits for the one before it
        for (uint32_t k0 = 0u; k0 < total; k0 += 256u) {
            uint32_t at[4];
            uint2 ent[4];
#if VCT_BIN_FILL_AGGREGATE
            uint32_t bin_of[4];
#endif
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + 64u * (uint32_t)u + (uint32_t)lane;
                at[u] = p.entry_cap;
                ent[u] = make_uint2(0u, 0u);
#if VCT_BIN_FILL_AGGREGATE
                bin_of[u] = 0xffffffffu;
#endif
                if (k < total) {
                    int lo = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (s_pre[wv][lo + step - 1] <= k) lo += step;
                    const uint4 h = s_hdr[wv][lo];
                    uint32_t j = k - (lo ? s_pre[wv][lo - 1] : 0u);
                    uint32_t hm = s_msk[wv][lo];
                    if (hm != 0u) {                   // the j-th set bit of the mask is the bin
                        for (uint32_t i = 0; i < j; ++i) hm &= hm - 1u;
                        j = (uint32_t)__ffs((int)hm) - 1u;
                    }
                    const int hx0 = (int)(h.y & 0xffffu) >> VCT_BIN_SHIFT, hy0 = (int)(h.z & 0xffffu) >> VCT_BIN_SHIFT;
                    const int hw = ((int)(h.y >> 16) >> VCT_BIN_SHIFT) - hx0 + 1;
                    const int by = hy0 + (int)j / hw, bx = hx0 + (int)j % hw;
                    ent[u] = bin_entry(rec0 + (uint32_t)lo, h, bx, by, depth_only);
#if VCT_BIN_FILL_AGGREGATE
                    bin_of[u] = (uint32_t)(by * p.bins_x + bx);
#else
                    at[u] = atomicAdd(&p.bin_cursor[(size_t)(by * p.bins_x + bx) * VCT_BIN_CSTRIDE], 1u);
#endif
                }
            }
#if VCT_BIN_FILL_AGGREGATE
            // One returning atomic per DISTINCT bin of the wave's 64 pairs instead of one per pair (neighbouring triangles
            // land in the same bins; atomics on one address execute one after the other in the L2): the lanes of a bin
            // elect a leader and take consecutive slots behind what it reserved.  The leaders' atomics of the four rounds
            // are all issued before any result is used.
            int leader[4];
            uint32_t rank[4], cnt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                leader[u] = lane; rank[u] = 0u; cnt[u] = 0u;
                unsigned long long left = __builtin_amdgcn_ballot_w64(bin_of[u] != 0xffffffffu);
                // (at most VCT_BIN_FILL_AGG_ROUNDS distinct bins are grouped, and the election stops after two groups of one
                // in a row; pairs left over take their own atomic: where every pair of the wave has a bin of its own --
                // large triangles at 4K -- the election would cost more than the atomics it saves)
                int singles = 0;
                for (int round = 0; round < VCT_BIN_FILL_AGG_ROUNDS && left != 0ull && singles < 2; ++round) {
                    const int l = (int)__ffsll((long long)left) - 1;
                    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bin_of[u], l);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(bin_of[u] == b);
                    if (bin_of[u] == b) {
                        leader[u] = l;
                        rank[u] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    }
                    if (lane == l) cnt[u] = (uint32_t)__popcll(m);
                    singles = (m & (m - 1ull)) == 0ull ? singles + 1 : 0;     // two groups of one in a row: no crowding here
                    left &= ~m;
                }
                if ((left >> lane) & 1ull) cnt[u] = 1u;       // not grouped: leader of itself
            }
            uint32_t got[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                got[u] = 0u;
                if (cnt[u] != 0u) got[u] = atomicAdd(&p.bin_cursor[(size_t)bin_of[u] * VCT_BIN_CSTRIDE], cnt[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t base = (uint32_t)__shfl((int)got[u], leader[u]);
                if (bin_of[u] != 0xffffffffu) at[u] = base + rank[u];
            }
#endif
#pragma unroll
            for (int u = 0; u < 4; ++u) p.entries[at[u] < p.entry_cap ? at[u] : p.entry_cap] = ent[u];
        }
    }
}

// edge functions of the sub-triangle at this lane's pixel centre: inside?, and the first two (unbiased) for the
// barycentrics -- the arithmetic of cover<FAST>, split so that lanes outside the triangle keep their values too
// (they are the helper invocations of the alpha test's texture fetch)
template <bool FAST>
__device__ __forceinline__ bool edges_at(const SubTri& s, const FastEdges& f, int px, int py, double& e0, double& e1) {
    const double cx = (double)px + 0.5, cy = (double)py + 0.5;
    double e[3];
    bool in = true;
    if (FAST) {
#pragma unroll
        for (int k = 0; k < 3; ++k) e[k] = fma(f.a[k], cx, fma(f.b[k], cy, f.c[k]));
        in = !(e[0] < 0.0 || e[1] < 0.0 || e[2] < 0.0);
        e0 = e[0] + f.bias[0];
        e1 = e[1] + f.bias[1];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int a = (k + 1) % 3, b = (k + 2) % 3;
            const double dx = (s.sx[b] - s.sx[a]) * s.sgn, dy = (s.sy[b] - s.sy[a]) * s.sgn;
            e[k] = dx * (cy - s.sy[a]) - dy * (cx - s.sx[a]);
            const bool top_left = (dy > 0.0) || (dy == 0.0 && dx < 0.0);
            if (e[k] < 0.0 || (e[k] == 0.0 && !top_left)) in = false;
        }
        e0 = e[0];
        e1 = e[1];
    }
    return in;
}

__device__ __forceinline__ float dpp_quad_x(float v) {      // the value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_quad_y(float v) {      // the value of lane ^ 2 (quad_perm [2,3,0,1])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
}
// maximum over the wave, result uniform (quad permutes, half-row / row mirrors, then the four rows through readlane)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));     // row_half_mirror
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));     // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

#ifndef VCT_BINRASTER_MIN_BLOCKS
#define VCT_BINRASTER_MIN_BLOCKS 6   // (waves per SIMD, __launch_bounds__' second argument: the 26 KB of LDS allow six workgroups per CU,
                                     // so the register allocator is held to 80 -- left alone it drifts between 72 and 91)
#endif
#ifndef VCT_BINRASTER_GRID
#define VCT_BINRASTER_GRID 16384u    // workgroups of k_bin_raster: each strides the work items
#endif
#if defined(VCT_BIN_STATS) && VCT_BIN_STATS
#define BIN_STAT(k, v) do { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); if ((int)(threadIdx.x & 63) == (int)__ffsll((long long)m_) - 1) atomicAdd(&p.huge[p.huge_cap + 16 + (k)], (uint32_t)(v)); } while (0)
#else
#define BIN_STAT(k, v) do { } while (0)
#endif

// A pixel's visibility word lives in its lane's registers; what goes through LDS is the "mail" of the alpha-test queue:
// a queued fragment is fetched by whichever lane the flush assigns it to, and a survivor is lowered into the slot of the
// pixel it belongs to (ds_min_u64), which that pixel's lane reads back (ds_read_b64) before its next depth test.  (A
// `volatile` access through a plain pointer compiles to a FLAT load with system scope and a wait for every outstanding
// memory operation -- on the critical path of every step; with the address space spelled out these are LDS operations.)
__device__ __forceinline__ unsigned long long lds_peek(const unsigned long long* w) {
    return *(const volatile __attribute__((address_space(3))) unsigned long long*)w;      // w points into __shared__ memory
}
__device__ __forceinline__ void lds_lower(unsigned long long* w, unsigned long long v) {
    (void)__hip_atomic_fetch_min(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// alpha-test queue of ONE wave: fragments that passed coverage and the early depth test, waiting for a FULL wave of
// texture fetches (a fetch issued per entry ran with one lane in twelve busy on the street's foliage)
struct BinAlphaQueue {
    unsigned long long word[64];
    float uv[64][6];                                      // u, v, du/dx, dv/dx, du/dy, dv/dy
    uint32_t pix_tex[64];                                 // pixel slot in the bin's words | texture << 16
};

// fetches the alpha of the queued fragments -- one per lane -- and merges the survivors into the bin's words
template <class Vis>
__device__ __forceinline__ void bin_flush_alpha(const BinParams& p, BinAlphaQueue& Q, Vis& vis, int lane, int qn) {
    BIN_STAT(7, 1);
    if (lane < qn) {
        const unsigned long long word = Q.word[lane];
        const uint32_t pt = Q.pix_tex[lane];
        unsigned long long* slot = &vis[pt & 0xffffu];
        if (word < lds_peek(slot)) {       // still in front of what the pixel shows now
            const float u = Q.uv[lane][0], v = Q.uv[lane][1];
            float alpha;
            if (p.r.tex.mips) alpha = vct_tex_sample_lod(p.r.tex, (int)(pt >> 16), u, v, Q.uv[lane][2], Q.uv[lane][3], Q.uv[lane][4], Q.uv[lane][5]).w;
            else alpha = vct_tex_sample(p.r.tex, (int)(pt >> 16), u, v).w;
#if defined(VCT_BIN_STATS) && VCT_BIN_STATS
            { const int n_f = (int)__popcll(__builtin_amdgcn_ballot_w64(true)); BIN_STAT(8, n_f); }
#endif
            if (!(alpha < 0.5f)) lds_lower(slot, word);                                       // trace.fs:171 discard
        }
    }
}

template <bool DEPTH_ONLY>
__global__ void __launch_bounds__(256, VCT_BINRASTER_MIN_BLOCKS)
k_bin_raster(const BinParams p) {
    __shared__ unsigned long long s_kv[VCT_BIN_SLICE];
    __shared__ uint4 s_rec[VCT_BIN_CHUNK * 10];
    __shared__ unsigned long long s_mail[4][64];          // per wave: what its alpha-queue flushes won, by pixel
    __shared__ uint32_t s_bk[2 * VCT_BIN_BUCKETS], s_mm[4];                // bucket counters / cursors and depth ranges of the slice order
    __shared__ BinAlphaQueue s_q[DEPTH_ONLY ? 1 : 4];
    const uint32_t nitems = min(p.ctr[2], p.item_cap);
    const bool any_in_place = p.ctr[5] != 0u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    BinAlphaQueue& Q = s_q[DEPTH_ONLY ? 0 : wave];
    unsigned long long* mail = s_mail[wave];
    // LANE = PIXEL of the wave's 8x8 quadrant of the bin; a 2x2 quad is four consecutive lanes
    const int lx = (lane & 1) | ((lane >> 1) & 6), ly = ((lane >> 1) & 1) | ((lane >> 3) & 6);
    const uint4* rec16 = reinterpret_cast<const uint4*>(p.recs);
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
#if defined(VCT_BIN_REVERSE) && VCT_BIN_REVERSE
        const uint4 it = p.items[nitems - 1u - item];       // order-sensitivity probe (profiles/experiments/README.md)
#else
        const uint4 it = p.items[item];
#endif
        const int bin = (int)it.x, n = (int)it.z;
        const uint32_t first = it.y;
        const int by = bin / p.bins_x, bx = bin - by * p.bins_x;
        // ---- the slice's entries ARE the sort keys; sorted when the slice holds alpha-tested ones ----
        int m = 1;
        while (m < n) m <<= 1;
        bool alpha_here = false;
        for (int i = (int)threadIdx.x; i < m && n > 0; i += 256) {
            unsigned long long kv = ~0ull;
            if (i < n) {
                const uint2 e = p.entries[first + (uint32_t)i];
                kv = ((unsigned long long)e.y << 32) | e.x;
                alpha_here = alpha_here || (e.y >> 31);
            }
            s_kv[i] = kv;
        }
        mail[lane] = ~0ull;
        const bool any_alpha = __syncthreads_or(alpha_here ? 1 : 0) != 0;
        if ((any_alpha || n >= VCT_BIN_ORDER_MIN) && n > 1) {     // opaque first, then alpha-tested front to back
#if VCT_BIN_BUCKET_ORDER
            // The order only has to be GOOD (what comes first hides what comes later; any order gives the same words):
            // instead of a bitonic network (28 barrier rounds for 128 keys, 45 for 512: a tenth of this kernel's time) the
            // keys are dealt into VCT_BIN_BUCKETS + VCT_BIN_BUCKETS buckets of their class' depth range -- five barriers.
            unsigned long long* tmp = reinterpret_cast<unsigned long long*>(s_rec);      // (free until the first chunk is staged)
            if (threadIdx.x < 2 * VCT_BIN_BUCKETS) s_bk[threadIdx.x] = 0u;
            if (threadIdx.x < 4) s_mm[threadIdx.x] = (threadIdx.x & 1) ? 0u : 0xffffffffu;
            __syncthreads();
            for (int i = (int)threadIdx.x; i < n; i += 256) {
                const uint32_t hi = (uint32_t)(s_kv[i] >> 32), c = hi >> 31, d = hi & 0x7fffffffu;
                atomicMin(&s_mm[2 * c], d);
                atomicMax(&s_mm[2 * c + 1], d);
            }
            __syncthreads();
            static_assert(VCT_BIN_SLICE <= 512, "the bucket order keeps two keys per thread (256 threads)");
            uint32_t bk[2] = {0u, 0u};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = (int)threadIdx.x + 256 * r;
                if (i < n) {
                    const uint32_t hi = (uint32_t)(s_kv[i] >> 32), c = hi >> 31, d = hi & 0x7fffffffu;
                    const uint32_t lo = s_mm[2 * c], span = s_mm[2 * c + 1] - lo;
                    const uint32_t q = min((uint32_t)VCT_BIN_BUCKETS - 1u, (uint32_t)((float)(d - lo) * __fdividef((float)VCT_BIN_BUCKETS, (float)span + 1.0f)));
                    bk[r] = c * (uint32_t)VCT_BIN_BUCKETS + q;
                    atomicAdd(&s_bk[bk[r]], 1u);
                }
            }
            __syncthreads();
            if (wave == 0) {            // exclusive scan of the counters: they become the buckets' cursors
                uint32_t carry = 0u;
#pragma unroll
                for (int h = 0; h < 2 * VCT_BIN_BUCKETS; h += 64) {
                    const uint32_t v = s_bk[h + lane];
                    uint32_t incl = v;
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t t = __shfl_up(incl, off);
                        if (lane >= off) incl += t;
                    }
                    s_bk[h + lane] = carry + incl - v;
                    carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = (int)threadIdx.x + 256 * r;
                if (i < n) tmp[atomicAdd(&s_bk[bk[r]], 1u)] = s_kv[i];
            }
            __syncthreads();
            for (int i = (int)threadIdx.x; i < n; i += 256) s_kv[i] = tmp[i];
            // (the chunk loop starts with a barrier)
#else
#if defined(VCT_PROBE_SORT_TWICE)
            for (int rep = 0; rep < 2; ++rep)       // timing probe: what the sort costs (sorted input, same network)
#endif
            for (int k = 2; k <= m; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = (int)threadIdx.x; i < m; i += 256) {
                        const int l = i ^ j;
                        if (l > i) {
                            const unsigned long long a = s_kv[i], b = s_kv[l];
                            if ((a > b) == ((i & k) == 0)) { s_kv[i] = b; s_kv[l] = a; }
                        }
                    }
                    __syncthreads();
                }
#endif
        }
        const int qx0 = (bx << VCT_BIN_SHIFT) + (wave & 1) * 8, qy0 = (by << VCT_BIN_SHIFT) + (wave >> 1) * 8;
        const int px = qx0 + lx, py = qy0 + ly;
        const double cx = (double)px + 0.5, cy = (double)py + 0.5;
        const bool inscr = px < p.r.W && py >= p.r.ys0 && py < p.r.ys1;
        unsigned long long best = ~0ull;        // this pixel's visibility word (DEPTH_ONLY: the depth's float bits in the low word)
        uint32_t zmax = 0xffffffffu;            // bits of the farthest depth the quadrant shows (all ones while a pixel is open)
        const uint32_t nh = (it.w & 2u) ? min(p.ctr[4], p.huge_cap) : 0u;
        const int total = n + (int)nh;
        int qn = 0;                             // fragments in this wave's alpha queue
        for (int c0 = 0; c0 < total; c0 += VCT_BIN_CHUNK) {
            const int cn = min(VCT_BIN_CHUNK, total - c0);
            // ---- the chunk's records into LDS: ten 16-byte words each, all 256 threads, one round trip ----
            __syncthreads();        // the previous chunk has been consumed
            for (int w = (int)threadIdx.x; w < cn * 10; w += 256) {
                const int j = w / 10, part = w - j * 10;
                const int i = c0 + j;
                const uint32_t e = i < n ? ((uint32_t)s_kv[i] & 0x0fffffffu) : p.huge[i - n];
                s_rec[w] = rec16[(size_t)e * 10 + part];
            }
            __syncthreads();
            // Lane j looks at entry j of the chunk: its header, and whether the entry concerns this wave at all (its box
            // reaches the quadrant; it does not lie behind everything the quadrant shows).  The wave then walks the set
            // bits -- an entry that does not concern it costs nothing, the others one v_readlane per header word.
            uint4 hd_l = make_uint4(0u, 0u, 0u, 0xffffffffu);
            bool mine = false;
            if (lane < cn) {
                hd_l = s_rec[lane * 10 + 9];                               // flags, xx, yy, zmin_bits
                const int x0 = (int)(hd_l.y & 0xffffu), x1 = (int)(hd_l.y >> 16), y0 = (int)(hd_l.z & 0xffffu), y1 = (int)(hd_l.z >> 16);
                mine = x1 >= qx0 && x0 <= qx0 + 7 && y1 >= qy0 && y0 <= qy0 + 7 && hd_l.w <= zmax;
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(mine);
            while (todo != 0ull) {
                const int j = (int)__ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const uint32_t zmin_bits = (uint32_t)__builtin_amdgcn_readlane((int)hd_l.w, j);
                if (zmin_bits > zmax) continue;                            // the quadrant closed meanwhile
                const uint32_t xx = (uint32_t)__builtin_amdgcn_readlane((int)hd_l.y, j), yy = (uint32_t)__builtin_amdgcn_readlane((int)hd_l.z, j);
                const int x0 = (int)(xx & 0xffffu), x1 = (int)(xx >> 16), y0 = (int)(yy & 0xffffu), y1 = (int)(yy >> 16);
                const uint4* R = s_rec + j * 10;
                // Hierarchical depth test, in registers: a pixel that already shows something nearer than the
                // sub-triangle's nearest point cannot be won; where that holds for every pixel the box reaches (the
                // inside of a tree crown after its front layers) the record is not even unpacked.
                const bool open = px >= x0 && px <= x1 && py >= y0 && py <= y1 && (uint32_t)(DEPTH_ONLY ? best : best >> 32) >= zmin_bits;
                if (__builtin_amdgcn_ballot_w64(open) == 0ull) continue;
                const uint32_t flags = (uint32_t)__builtin_amdgcn_readlane((int)hd_l.x, j);
                const double* D = reinterpret_cast<const double*>(R);      // ea[3], eb[3], ec[3], area, rcp
                const float* F = reinterpret_cast<const float*>(R) + 22;   // sz[3], iw[3], tu[3], tv[3], id
                bool in;
                double e0, e1;
                if (flags >> 31) {       // plane equations (every coordinate below 2^16)
                    const double f0 = fma(D[0], cx, fma(D[3], cy, D[6])), f1 = fma(D[1], cx, fma(D[4], cy, D[7]));
                    const double f2 = fma(D[2], cx, fma(D[5], cy, D[8]));
                    in = !(f0 < 0.0 || f1 < 0.0 || f2 < 0.0);
                    e0 = f0 + (((flags >> 28) & 1u) ? 0x1p-16 : 0.0);
                    e1 = f1 + (((flags >> 29) & 1u) ? 0x1p-16 : 0.0);
                } else {                 // the general form on the snapped coordinates (sx in ea, sy in eb)
                    SubTri s;
                    FastEdges fe;
                    s.sgn = 1.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { s.sx[k] = D[k]; s.sy[k] = D[3 + k]; }
                    in = edges_at<false>(s, fe, px, py, e0, e1);
                }
                const bool cov = in && open;
                if (__builtin_amdgcn_ballot_w64(cov) == 0ull) continue;
                const double area = D[9], rcp = D[10];
                const float b0 = (float)div_area(e0, area, rcp), b1 = (float)div_area(e1, area, rcp);
                const float b2 = 1.0f - b0 - b1;
                float z = b0 * F[0] + b1 * F[1] + b2 * F[2];
                z = z + 0.0f;
                bool pass = cov && z >= 0.0f && z < 1.0f;
                if (DEPTH_ONLY) {
                    const unsigned long long word = (unsigned long long)__float_as_uint(z);
                    pass = pass && word < best;
                    if (pass) best = word;
                } else {
                    const unsigned long long word = ((unsigned long long)__float_as_uint(z) << 32) | __float_as_uint(F[12]);
                    pass = pass && word < best;        // the exact early depth test, for free
                    if ((flags & 3u) == 2u) {
                        const unsigned long long mp = __builtin_amdgcn_ballot_w64(pass);
                        if (mp == 0ull) continue;
                        // every lane interpolates the texture coordinate at its own centre (helper invocations included)
                        const float w0 = b0 * F[3], w1 = b1 * F[4], w2 = b2 * F[5];
                        const float ws = __fdiv_rn(1.0f, w0 + w1 + w2);
                        const float u = (w0 * F[6] + w1 * F[7] + w2 * F[8]) * ws;
                        const float v = (w0 * F[9] + w1 * F[10] + w2 * F[11]) * ws;
                        const float ux = dpp_quad_x(u), vx = dpp_quad_x(v), uy = dpp_quad_y(u), vy = dpp_quad_y(v);
                        const int cnt = (int)__popcll(mp);
                        if (qn + cnt > 64) {
                            bin_flush_alpha(p, Q, mail, lane, qn);
                            qn = 0;
                            best = min(best, lds_peek(&mail[lane]));
                        }
                        if (pass) {
                            const int at = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mp, 0u));
                            Q.word[at] = word;
                            Q.uv[at][0] = u; Q.uv[at][1] = v;
                            Q.uv[at][2] = ux - u; Q.uv[at][3] = vx - v; Q.uv[at][4] = uy - u; Q.uv[at][5] = vy - v;
                            Q.pix_tex[at] = (uint32_t)lane | ((((flags >> 2) & 0x1ffffffu) - 1u) << 16);
                        }
                        qn += cnt;
                        continue;
                    }
                    if (pass) best = word;
                }
                if (__builtin_amdgcn_ballot_w64(pass) != 0ull)
                    zmax = wave_max_u32(inscr ? (uint32_t)(DEPTH_ONLY ? best : best >> 32) : 0u);
            }
            // a fragment waiting in the queue does not yet hide what lies behind it: resolve the fuller queues between chunks
            if (!DEPTH_ONLY && qn >= 32) {
                bin_flush_alpha(p, Q, mail, lane, qn);
                qn = 0;
                best = min(best, lds_peek(&mail[lane]));
                zmax = wave_max_u32(inscr ? (uint32_t)(best >> 32) : 0u);
            }
        }
        if (!DEPTH_ONLY && qn > 0) {
            bin_flush_alpha(p, Q, mail, lane, qn);
            qn = 0;
            best = min(best, lds_peek(&mail[lane]));
        }
        // ---- one word per visible pixel ----
        if (inscr && best != ~0ull) {
            const bool merge = (it.w & 1u) || any_in_place;
            const size_t pix = (size_t)py * p.r.W + px;
            if (DEPTH_ONLY) {
                const uint32_t word = vct_depth24_bits(__uint_as_float((uint32_t)best)) + p.r.vis32_ebase;
                if (merge) atomicMin(&p.r.vis32[pix], word); else p.r.vis32[pix] = word;
            } else {
                if (merge) atomicMin(&p.r.vis[pix], best); else p.r.vis[pix] = best;
            }
        }
        __syncthreads();        // s_kv is reused by the next item
    }
}

// float depths in [0, 1] <-> shadow-map words of epoch base `ebase`
__global__ void __launch_bounds__(256)
k_shadow_encode(const float* __restrict__ depth, uint32_t* __restrict__ words, size_t n, uint32_t ebase) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = depth[i];
        const float c = !(d > 0.0f) ? 0.0f : (d > 1.0f ? 1.0f : d);        // [GL] depth textures are clamped to [0, 1]
        words[i] = __float_as_uint(c) + ebase;
    }
}
__global__ void __launch_bounds__(256)
k_shadow_decode(const uint32_t* __restrict__ words, float* __restrict__ depth, size_t n, uint32_t ebase) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        depth[i] = vct_shadow_depth(words[i], ebase);
}

struct ShadeParams {
    RasterParams r;
    const float* nrm;            // [ntri][9]
    const float* tan;
    const float* bit;
    const int32_t* material;     // [ntri]
    const float* albedo;         // [nmat][4]
    const float* specular;       // [nmat][3]
    const uint32_t* shadow;      // [S*S] shadow-map words or null
    uint32_t shadow_ebase;
    int32_t shadow_size;
    const uint2* shadow_tiles;   // per-tile depth bounds of the map (vct_launch_shadow_minmax) or null
    float light_vp[16];
    float* tiled;                // [tile][23][64]
    int32_t tiles_x;
    int32_t tile0, tile1;        // tiles [tile0, tile1) are shaded (whole tile rows)
};

// [GL] bilinear clamp-to-edge fetch with the operation order of host/vct_host.cpp shadow_fetch
__device__ __forceinline__ float shadow_fetch(const uint32_t* __restrict__ words, uint32_t eb, int S, float u, float v) {
    const float x = u * (float)S - 0.5f, y = v * (float)S - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const float top = (float)(S - 1);
    auto cl = [&](float f) -> int { return f < 0.0f ? 0 : (f > top ? S - 1 : (int)f); };
    const int i0 = cl(fx), i1 = cl(fx + 1.0f), j0 = cl(fy), j1 = cl(fy + 1.0f);
    const float d00 = vct_shadow_depth(words[(size_t)j0 * S + i0], eb), d10 = vct_shadow_depth(words[(size_t)j0 * S + i1], eb);
    const float d01 = vct_shadow_depth(words[(size_t)j1 * S + i0], eb), d11 = vct_shadow_depth(words[(size_t)j1 * S + i1], eb);
    return (1.0f - a) * (1.0f - b) * d00 + a * (1.0f - b) * d10 + (1.0f - a) * b * d01 + a * b * d11;
}

// one thread per pixel, 64 threads = one 8x8 tile of the tiled G-buffer
#ifndef VCT_SHADE_MIN_BLOCKS
#define VCT_SHADE_MIN_BLOCKS 4       // waves per SIMD (round 2, dword PCF loads, G-buffer pass in us: 5: 204, 7: 201, 8: 212); the wide row loads of round 3 need register tuples: 7 spills
#endif
// TEX: the scene has material textures.  Two instantiations because the mip-mapped fetches (two levels x four texels,
// their derivatives from two more perspective-correct interpolations) do not fit the 72-VGPR budget of 7 waves per
// SIMD: compiled into one kernel they spilled 76 registers to scratch (560 B per lane) and the pass of the textured
// atrium took 0.73 ms instead of 0.27.  Flat scenes keep the lean kernel; the textured one gets 96 VGPRs (5 waves).
#ifndef VCT_SHADE_TEX_MIN_BLOCKS
#define VCT_SHADE_TEX_MIN_BLOCKS 4
#endif
#ifndef VCT_SHADE_BLOCK
#define VCT_SHADE_BLOCK 256        // threads per workgroup = 64 x tiles per workgroup (the kernel has no workgroup-level state)
#endif
template <bool TEX>
__global__ void __launch_bounds__(256, TEX ? VCT_SHADE_TEX_MIN_BLOCKS : VCT_SHADE_MIN_BLOCKS)
k_gbuffer_shade(const ShadeParams p) {
    const int W = p.r.W, H = p.r.H;
    const int tile = p.tile0 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    if (tile >= p.tile1 || ty * VCT_TILE >= H) return;
    const int px = tx * VCT_TILE + (lane & 7), py = ty * VCT_TILE + (lane >> 3);
    float* out = p.tiled + (size_t)tile * (VCT_GB_NPLANES * VCT_TILE_PIX) + lane;
    float g[VCT_GB_NPLANES];
#pragma unroll
    for (int k = 0; k < VCT_GB_NPLANES; ++k) g[k] = 0.0f;
    unsigned long long v = ~0ull;
    if (px < W && py < H) {
        v = p.r.vis[(size_t)py * W + px];
        p.r.vis[(size_t)py * W + px] = ~0ull;       // consumed: leave the word "empty" for the next pass (no clear launch)
    }
    if (v != ~0ull) {
        const int id = (int)(uint32_t)v;
        const int t = id >> 1, f = (id & 1) + 1;
        RVert in[3];
        load_clip_tri(p.r, t, in);
#if VCT_SHADE_PREFETCH
        // Everything that depends on the triangle index only is requested NOW, ahead of the fp64 set-up: the four
        // per-vertex records (the position once more: the clip-space copy above is consumed by the set-up) and the
        // material index.  The kernel is a chain of dependent round trips (visibility word -> records -> material ->
        // colours -> shadow window); these used to be five of its eight.
        VctTri9 recs4[4];
#pragma unroll
        for (int arr = 0; arr < 4; ++arr) {
            const float* src = arr == 0 ? p.r.pos : (arr == 1 ? p.nrm : (arr == 2 ? p.tan : p.bit));
            recs4[arr] = *reinterpret_cast<const VctTri9*>(src + (size_t)t * 9);
        }
        const int m_early = p.material[t];
        // ... and the material's colours as soon as the index is there (it arrives with the records), so that they too
        // travel during the set-up
        const float alb_early[4] = {p.albedo[4 * (size_t)m_early], p.albedo[4 * (size_t)m_early + 1],
                                    p.albedo[4 * (size_t)m_early + 2], p.albedo[4 * (size_t)m_early + 3]};
        const float sp_early[3] = {p.specular[3 * (size_t)m_early], p.specular[3 * (size_t)m_early + 1],
                                   p.specular[3 * (size_t)m_early + 2]};
        // ... and, in a scene with textures, the material's three texture indices and the triangle's texture coordinates
        // (round 4: they used to be asked for after the set-up -- material -> indices -> descriptor -> texels was four
        // dependent round trips behind it, now two)
        int tex_early[3] = {-1, -1, -1};
        float uv_early[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (TEX) {
#pragma unroll
            for (int k = 0; k < 3; ++k) tex_early[k] = vct_tex_of(p.r.tex, m_early, k);
#pragma unroll
            for (int k = 0; k < 6; ++k) uv_early[k] = p.r.tex.uv[(size_t)t * 6 + k];
        }
#endif
        const bool whole = unclipped(in);
        FanTri fan;
        SubTri s;
        if (whole) {
            setup_subtri(&in[0], &in[1], &in[2], W, H, 0, H, s);
#pragma unroll
            for (int k = 0; k < 3; ++k) { fan.v[k].v = in[k]; fan.v[k].a = k; fan.v[k].b = -1; fan.v[k].t = 0.0f; }
        } else {
            ClipPoly poly;
            clip_near(in, poly);
            fan = fan_tri(poly, f);
            setup_subtri(&fan.v[0].v, &fan.v[1].v, &fan.v[2].v, W, H, 0, H, s);
        }
        float b0, b1, b2, z;
        double e0, e1;
        const FastEdges fe = slow_edges(s);
        cover<false>(s, fe, px, py, b0, b1, b2, z, e0, e1);
        // perspective-correct interpolation of the 12 varyings (trace.vs:27,31-33)
        const float q0 = b0 * s.iw[0], q1 = b1 * s.iw[1], q2 = b2 * s.iw[2];
        const float qs = __fdiv_rn(1.0f, q0 + q1 + q2);
        // the four per-vertex arrays one after the other, each as one 36-byte record (three wide loads): nine values
        // live at a time.  A vertex of a near-clipped polygon is picked by selects (a run-time register index would
        // put the record into scratch).
#pragma unroll
        for (int arr = 0; arr < 4; ++arr) {
#if VCT_SHADE_PREFETCH
            const VctTri9 rec = recs4[arr];
#else
            const float* src = arr == 0 ? p.r.pos : (arr == 1 ? p.nrm : (arr == 2 ? p.tan : p.bit));
            const VctTri9 rec = *reinterpret_cast<const VctTri9*>(src + (size_t)t * 9);
#endif
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) {
                const float a0 = rec.v[comp] * p.r.model_scale, a1 = rec.v[3 + comp] * p.r.model_scale,
                            a2 = rec.v[6 + comp] * p.r.model_scale;
                float var[3];
                if (whole) {
                    var[0] = a0; var[1] = a1; var[2] = a2;
                } else {
                    auto pick = [&](int vert) -> float { return vert == 0 ? a0 : (vert == 1 ? a1 : a2); };
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float va = pick(fan.v[k].a);
                        var[k] = fan.v[k].b < 0 ? va : va + (pick(fan.v[k].b) - va) * fan.v[k].t;
                    }
                }
                g[3 * arr + comp] = (q0 * var[0] + q1 * var[1] + q2 * var[2]) * qs;
            }
        }
#if VCT_SHADE_PREFETCH
        const int td = tex_early[0], tsp = tex_early[1], th = tex_early[2];
#else
        const int m = p.material[t];
        const int td = TEX ? vct_tex_of(p.r.tex, m, 0) : -1, tsp = TEX ? vct_tex_of(p.r.tex, m, 1) : -1,
                  th = TEX ? vct_tex_of(p.r.tex, m, 2) : -1;
#endif
        float tcu = 0.0f, tcv = 0.0f;                                             // tex (trace.vs:36)
        float dq[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                   // its quad differences (mip-mapped textures)
        if (TEX && (td >= 0 || tsp >= 0 || th >= 0)) {
            const float* uv = p.r.tex.uv + (size_t)t * 6;
            float vu[3], vv[3];
            if (whole) {
#pragma unroll
#if VCT_SHADE_PREFETCH
                for (int k = 0; k < 3; ++k) { vu[k] = uv_early[2 * k]; vv[k] = uv_early[2 * k + 1]; }
#else
                for (int k = 0; k < 3; ++k) { vu[k] = uv[2 * k]; vv[k] = uv[2 * k + 1]; }
#endif
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int a = fan.v[k].a, b = fan.v[k].b;
                    const float ua = uv[2 * a], va = uv[2 * a + 1];
                    vu[k] = b < 0 ? ua : ua + (uv[2 * b] - ua) * fan.v[k].t;
                    vv[k] = b < 0 ? va : va + (uv[2 * b + 1] - va) * fan.v[k].t;
                }
            }
            tcu = (q0 * vu[0] + q1 * vu[1] + q2 * vu[2]) * qs;
            tcv = (q0 * vv[0] + q1 * vv[1] + q2 * vv[2]) * qs;
            if (p.r.tex.mips) {
                float ux, vx, uy, vy;
                interp2_at(s, fe.rcp, px ^ 1, py, vu, vv, ux, vx);
                interp2_at(s, fe.rcp, px, py ^ 1, vu, vv, uy, vy);
                dq[0] = ux - tcu; dq[1] = vx - tcv; dq[2] = uy - tcu; dq[3] = vy - tcv;
            }
        }
        // texture(sampler, tex [+ constant]): the implicit derivatives are those of tex
        // (always_inline: an out-of-line lambda takes the address of the kernel arguments, and the whole 430-byte
        // argument block is then copied to scratch at kernel entry)
        auto fetch = [&](int ti, float uu, float vv2) __attribute__((always_inline)) {
            return vct_tex_sample_lod(p.r.tex, ti, uu, vv2, dq[0], dq[1], dq[2], dq[3]);
        };
        // TBN = inverse(transpose(mat3(T,B,N))): columns (BxN, NxT, TxB) / det                    trace.fs:175
        const float Tx = g[6], Ty = g[7], Tz = g[8], Bx = g[9], By = g[10], Bz = g[11];
        const float Nx = g[3], Ny = g[4], Nz = g[5];
        const float c2x = Ty * Bz - Tz * By, c2y = Tz * Bx - Tx * Bz, c2z = Tx * By - Ty * Bx;   // T x B
        const float bnx = By * Nz - Bz * Ny, bny = Bz * Nx - Bx * Nz, bnz = Bx * Ny - By * Nx;   // B x N
        const float det = Tx * bnx + Ty * bny + Tz * bnz;
        const float inv = __fdiv_rn(1.0f, det);
        float ux, uy, uz;
        if (!TEX || th < 0) {
            // CalcBumpNormal with a flat height map: normalize(TBN * (0,0,1))
            ux = c2x * inv; uy = c2y * inv; uz = c2z * inv;
        } else {
            // CalcBumpNormal (trace.fs:110-128): three HeightTexture taps one texel apart
            const VctTexDesc hd = p.r.tex.desc[th];
            const float ox = __fdiv_rn(1.0f, (float)hd.w), oy = __fdiv_rn(1.0f, (float)hd.h);      // :112
            const float cur = fetch(th, tcu, tcv).x;                                                 // :114
            const float dx = fetch(th, tcu + ox, tcv).x - cur;                                       // :115
            const float dy = fetch(th, tcu, tcv + oy).x - cur;                                       // :116
            // t1 = normalize(1,0,dx), t2 = normalize(0,1,dy)   (host normalize(): a * (1/l))
            const float l1 = __builtin_sqrtf(1.0f * 1.0f + 0.0f * 0.0f + dx * dx), i1 = __fdiv_rn(1.0f, l1);
            const float l2 = __builtin_sqrtf(0.0f * 0.0f + 1.0f * 1.0f + dy * dy), i2 = __fdiv_rn(1.0f, l2);
            const float t1x = 1.0f * i1, t1y = 0.0f * i1, t1z = dx * i1;
            const float t2x = 0.0f * i2, t2y = 1.0f * i2, t2z = dy * i2;
            float bx = t1y * t2z - t1z * t2y, by = t1z * t2x - t1x * t2z, bz = t1x * t2y - t1y * t2x;   // cross(t1, t2)
            const float lb = __builtin_sqrtf(bx * bx + by * by + bz * bz);
            const float ib = lb > 0.0f ? __fdiv_rn(1.0f, lb) : 0.0f;
            bx = lb > 0.0f ? bx * ib : 0.0f; by = lb > 0.0f ? by * ib : 0.0f; bz = lb > 0.0f ? bz * ib : 0.0f;
            const float c1x = Ny * Tz - Nz * Ty, c1y = Nz * Tx - Nx * Tz, c1z = Nx * Ty - Ny * Tx;     // N x T
            const float k0x = bnx * inv, k0y = bny * inv, k0z = bnz * inv;
            const float k1x = c1x * inv, k1y = c1y * inv, k1z = c1z * inv;
            const float k2x = c2x * inv, k2y = c2y * inv, k2z = c2z * inv;
            ux = k0x * bx + k1x * by + k2x * bz;
            uy = k0y * bx + k1y * by + k2y * bz;
            uz = k0z * bx + k1z * by + k2z * bz;
        }
        const float len = __builtin_sqrtf(ux * ux + uy * uy + uz * uz);
        const float il = len > 0.0f ? __fdiv_rn(1.0f, len) : 0.0f;      // host normalize(): a * (1/l)
        g[12] = len > 0.0f ? ux * il : 0.0f;
        g[13] = len > 0.0f ? uy * il : 0.0f;
        g[14] = len > 0.0f ? uz * il : 0.0f;
#if VCT_SHADE_PREFETCH
        const float* alb = alb_early;
#else
        const float* alb = p.albedo + 4 * (size_t)m;
#endif
        if (TEX && td >= 0) {
            const float4 c = fetch(td, tcu, tcv);                                 // trace.fs:167
            g[15] = c.x; g[16] = c.y; g[17] = c.z; g[18] = c.w;
        } else {
            g[15] = alb[0]; g[16] = alb[1]; g[17] = alb[2]; g[18] = alb[3];
        }
#if VCT_SHADE_PREFETCH
        float sp[3] = {sp_early[0], sp_early[1], sp_early[2]};
#else
        float sp[3] = {p.specular[3 * (size_t)m], p.specular[3 * (size_t)m + 1], p.specular[3 * (size_t)m + 2]};
#endif
        if (TEX && tsp >= 0) {
            const float4 c = fetch(tsp, tcu, tcv);                                // trace.fs:209
            sp[0] = c.x; sp[1] = c.y; sp[2] = c.z;
        }
        const bool has_gb = __builtin_sqrtf(sp[1] * sp[1] + sp[2] * sp[2]) > 0.0f;
        g[19] = sp[0];
        g[20] = has_gb ? sp[1] : sp[0];                                           // trace.fs:210
        g[21] = has_gb ? sp[2] : sp[0];
        float shadow = 25.0f * 0.111f;
        if (p.shadow) {
            float d[4];
            xform4(p.light_vp, g[0], g[1], g[2], d);                              // trace.vs:28
            const float cx = d[0] * 0.5f + 0.5f, cy = d[1] * 0.5f + 0.5f, cz = d[2] * 0.5f + 0.5f;
            const float cur = __fdiv_rn(cz, d[3]) - 0.002f;
            const float step = __fdiv_rn(1.0f, (float)p.shadow_size);
            // The 25 taps sit one texel apart: when every tap's upper texel index equals the next
            // tap's lower one on both axes (checked per lane) the shared 6x6 window is loaded once
            // and each tap is evaluated with its own exact fractions; otherwise tap by tap.
            const int S = p.shadow_size;
            const float fS = (float)S, top = (float)(S - 1);
            int xi0[5], xi1[5], yj0[5], yj1[5];
            float xa[5], yb[5];
            bool regular = true;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float ux = (cx + step * (float)(k - 2)) * fS - 0.5f, uy = (cy + step * (float)(k - 2)) * fS - 0.5f;
                const float fx = floorf(ux), fy = floorf(uy);
                xa[k] = ux - fx; yb[k] = uy - fy;
                const float fx1 = fx + 1.0f, fy1 = fy + 1.0f;
                xi0[k] = fx < 0.0f ? 0 : (fx > top ? S - 1 : (int)fx);
                xi1[k] = fx1 < 0.0f ? 0 : (fx1 > top ? S - 1 : (int)fx1);
                yj0[k] = fy < 0.0f ? 0 : (fy > top ? S - 1 : (int)fy);
                yj1[k] = fy1 < 0.0f ? 0 : (fy1 > top ? S - 1 : (int)fy1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) regular = regular && xi1[k] == xi0[k + 1] && yj1[k] == yj0[k + 1];
            float cnt = 0.0f;
            if (regular) {
                int col[6], row[6];
#pragma unroll
                for (int k = 0; k < 5; ++k) { col[k] = xi0[k]; row[k] = yj0[k]; }
                col[5] = xi1[4]; row[5] = yj1[4];
                const bool wide = col[5] - col[0] == 5;     // six consecutive texels per row: one dwordx4 + one dwordx2 (VctWords6)
                auto tap_row = [&](int y, const float r0[6], const float r1[6]) __attribute__((always_inline)) {
#pragma unroll
                    for (int x = 0; x < 5; ++x) {
                        const float a = xa[x], b = yb[y];
                        const float tap = (1.0f - a) * (1.0f - b) * r0[x] + a * (1.0f - b) * r0[x + 1] +
                                          (1.0f - a) * b * r1[x] + a * b * r1[x + 1];
                        if (cur <= tap) cnt += 1.0f;
                    }
                };
                int tile_verdict = -1;
                if (wide && p.shadow_tiles) tile_verdict = vct_pcf_tile_verdict(p.shadow_tiles, S, (uint32_t)col[0], (uint32_t)row[0], cur);
                if (tile_verdict >= 0) {
                    cnt = (float)tile_verdict;      // the tiles the window lies in bound it: no window fetch
                } else if (wide) {
                    // all six rows in flight at once (round 3: the rows used to be fetched one after the other, two
                    // live at a time to save registers -- six dependent round trips per pixel were the price, 36 of
                    // the pass's 188 us); the taps are counted, so their order is free
                    VctWords6 w[6];
#pragma unroll
                    for (int y = 0; y < 6; ++y) w[y] = *reinterpret_cast<const VctWords6*>(p.shadow + (size_t)row[y] * S + col[0]);
                    uint32_t wmin = 0xffffffffu, wmax = 0u;
#pragma unroll
                    for (int y = 0; y < 6; ++y)
#pragma unroll
                        for (int i = 0; i < 6; ++i) { wmin = min(wmin, w[y].v[i]); wmax = max(wmax, w[y].v[i]); }
                    const int verdict = vct_pcf_window_verdict(wmin, wmax, p.shadow_ebase, cur);      // vct_internal.h "PCF short cut"
                    if (verdict >= 0) {
                        cnt = (float)verdict;
                    } else {
                        float r0[6], r1[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) r0[i] = vct_shadow_depth(w[0].v[i], p.shadow_ebase);
#pragma unroll
                        for (int y = 0; y < 5; ++y) {
#pragma unroll
                            for (int i = 0; i < 6; ++i) r1[i] = vct_shadow_depth(w[y + 1].v[i], p.shadow_ebase);
                            tap_row(y, r0, r1);
#pragma unroll
                            for (int i = 0; i < 6; ++i) r0[i] = r1[i];
                        }
                    }
                } else {       // a window at a clamped border: texel by texel, two rows live at a time
                    float r0[6], r1[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) r0[i] = vct_shadow_depth(p.shadow[(size_t)row[0] * S + col[i]], p.shadow_ebase);
#pragma unroll
                    for (int y = 0; y < 5; ++y) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) r1[i] = vct_shadow_depth(p.shadow[(size_t)row[y + 1] * S + col[i]], p.shadow_ebase);
                        tap_row(y, r0, r1);
#pragma unroll
                        for (int i = 0; i < 6; ++i) r0[i] = r1[i];
                    }
                }
            } else {
                for (int x = -2; x <= 2; ++x)
                    for (int y = -2; y <= 2; ++y) {
                        const float ox = step * (float)x, oy = step * (float)y;       // trace.fs:147
                        if (cur <= shadow_fetch(p.shadow, p.shadow_ebase, p.shadow_size, cx + ox, cy + oy)) cnt += 1.0f;
                    }
            }
            shadow = cnt * 0.111f;                                                // trace.fs:158
        }
        g[22] = shadow;
    }
#pragma unroll
    for (int k = 0; k < VCT_GB_NPLANES; ++k) out[k * VCT_TILE_PIX] = g[k];
}

// glGenerateMipmap (R/Model.h:168) for one level: rounded mean of the 2x2 parent texels, indices clamped to the parent
__global__ void __launch_bounds__(256)
k_tex_mip(const uint32_t* __restrict__ parent, int pw, int ph, uint32_t* __restrict__ level, int w, int h) {
    const size_t n = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)w), x = (int)(i - (size_t)y * w);
        const int x0 = min(2 * x, pw - 1), x1 = min(2 * x + 1, pw - 1), y0 = min(2 * y, ph - 1), y1 = min(2 * y + 1, ph - 1);
        const uint32_t a = parent[(size_t)y0 * pw + x0], b = parent[(size_t)y0 * pw + x1];
        const uint32_t c = parent[(size_t)y1 * pw + x0], d = parent[(size_t)y1 * pw + x1];
        uint32_t o = 0u;
#pragma unroll
        for (int sh = 0; sh < 32; sh += 8)
            o |= ((((a >> sh) & 0xffu) + ((b >> sh) & 0xffu) + ((c >> sh) & 0xffu) + ((d >> sh) & 0xffu) + 2u) >> 2) << sh;
        level[i] = o;
    }
}

// tiled [tile][23][64] -> linear planes [23][h*w] (downloads / tests)
__global__ void k_untile_gbuffer(const float* __restrict__ tiled, float* __restrict__ planes, int w, int h,
                                 int tiles_x) {
    const size_t npix = (size_t)w * h;
    const size_t total = npix * VCT_GB_NPLANES;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int plane = (int)(i / npix);
        const size_t pix = i - (size_t)plane * npix;
        const int y = (int)(pix / w), x = (int)(pix - (size_t)y * w);
        const size_t tile = (size_t)(y / VCT_TILE) * tiles_x + x / VCT_TILE;
        const int lane = (y % VCT_TILE) * VCT_TILE + (x % VCT_TILE);
        planes[i] = tiled[(tile * VCT_GB_NPLANES + plane) * VCT_TILE_PIX + lane];
    }
}

RasterParams make_raster(const VctRasterArgs& a, const float vp[16], int W, int H, int ys0, int ys1) {
    RasterParams r;
    r.ys0 = ys0 < 0 ? 0 : ys0;
    r.ys1 = ys1 > H ? H : ys1;
    r.pos = a.pos;
    r.ntri = a.ntri;
    r.model_scale = a.model_scale;
    for (int i = 0; i < 16; ++i) r.vp[i] = vp[i];
    r.W = W;
    r.H = H;
    r.vis = a.vis;
    r.wave_list = a.wave_list;
    r.wave_count = a.wave_count;
    r.group_list = a.group_list;
    r.group_count = a.group_count;
    r.vis32 = nullptr;
    r.vis32_ebase = 0u;
    r.items = a.items;
    r.item_count = a.item_count;
    r.item_capacity = a.item_capacity;
    r.next_counts = a.next_counts;
    r.recs = (SubTriRec*)a.recs;
    r.material = nullptr;
    r.tri_alpha = nullptr;
    r.albedo = nullptr;
    memset(&r.tex, 0, sizeof(r.tex));
    return r;
}

// Visibility of one pass.  The caller guarantees (vct_capi.hip raster_args) that the visibility words of the
// scissor rows are all-ones and that this pass's three counters are zero: both are re-established by the pass
// itself (the consumer of a word writes ~0 back, k_raster_vis zeroes the next pass's counters), so a pass is two
// dependent launches and no memset.
hipError_t run_visibility(const RasterParams& r, hipStream_t s) {
    if (r.ys1 <= r.ys0 || r.ntri <= 0)        // nothing to draw: keep the "next counters are zero" invariant
        return hipMemsetAsync(r.next_counts, 0, 3 * sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_raster_vis, dim3((r.ntri + 255) / 256), dim3(256), 0, s, r);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int gblocks = 4096, wblocks = 2048, tblocks = 2048;
    hipLaunchKernelGGL(k_raster_mid, dim3(gblocks + wblocks + tblocks), dim3(256), 0, s, r, gblocks, wblocks);
    return hipGetLastError();
}

// The tile-binned form of run_visibility: four dependent launches (set-up + pairs, bin ranges + work items, scatter,
// raster).  Same caller guarantees; the counters of the next pass are zeroed by k_bin_setup, the bins' counters by
// k_bin_alloc.
hipError_t run_visibility_binned(const VctRasterArgs& a, const RasterParams& r, hipStream_t s) {
    BinParams p;
    p.r = r;
    p.recs = (BinRec*)a.bin_recs; p.rec_cap = a.bin_rec_cap;
    p.entries = a.bin_entries; p.entry_cap = a.bin_entry_cap;
    p.bin_count = a.bin_count; p.bin_cursor = a.bin_cursor;
    p.items = a.bin_items; p.item_cap = a.bin_item_cap;
    p.huge = a.bin_huge; p.huge_cap = a.bin_huge_cap;
    p.ctr = a.bin_ctr; p.next_ctr = a.bin_next_ctr;
    p.bins_x = (r.W + VCT_BIN - 1) / VCT_BIN;
    p.bins_y = (r.H + VCT_BIN - 1) / VCT_BIN;
    if (r.ys1 <= r.ys0 || r.ntri <= 0) return hipMemsetAsync(p.next_ctr, 0, 8 * sizeof(uint32_t), s);
    const int nbins = p.bins_x * p.bins_y;
    hipLaunchKernelGGL(k_bin_setup, dim3((r.ntri + 255) / 256), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bin_alloc, dim3((nbins + 255) / 256), dim3(256), 0, s, p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const uint32_t fill_blocks = p.rec_cap / 256u + 1u < 4096u ? p.rec_cap / 256u + 1u : 4096u;
    hipLaunchKernelGGL(k_bin_fill, dim3(fill_blocks), dim3(256), 0, s, p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const uint32_t raster_blocks = p.item_cap < VCT_BINRASTER_GRID ? p.item_cap : VCT_BINRASTER_GRID;
    if (r.vis32) hipLaunchKernelGGL(k_bin_raster<true>, dim3(raster_blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_bin_raster<false>, dim3(raster_blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t vct_launch_shadow_raster(const VctRasterArgs& a, const float light_vp[16], int S, hipStream_t s) {
    RasterParams r = make_raster(a, light_vp, S, S, 0, S);
    r.vis32 = a.vis32;                                  // depth only, straight into the shadow-map words
    r.vis32_ebase = a.vis32_ebase;
    return a.binned ? run_visibility_binned(a, r, s) : run_visibility(r, s);
}

namespace {
// A workgroup per 64 x 64-texel block of the map: the block + the 5-texel reach to the right / top is decoded into LDS
// (69 x 69), then min / max separably: per row over the 13 texels each of the block's 8 tile columns reaches, then per
// tile over its 13 rows.  Texels beyond the map do not exist (a regular window never touches them): neutral values.
#define VCT_SMM_BLOCK 64
#define VCT_SMM_SPAN (VCT_SMM_BLOCK + VCT_SHADOW_TILE_REACH)
__global__ void __launch_bounds__(256)
k_shadow_minmax(const uint32_t* __restrict__ words, uint32_t ebase, int S, int nb, uint2* __restrict__ tiles) {
    __shared__ uint32_t tex[VCT_SMM_SPAN][VCT_SMM_SPAN + 1];
    __shared__ uint32_t rlo[VCT_SMM_SPAN][8], rhi[VCT_SMM_SPAN][8];
    const int bx = blockIdx.x * VCT_SMM_BLOCK, by = blockIdx.y * VCT_SMM_BLOCK;
    for (int i = threadIdx.x; i < VCT_SMM_SPAN * VCT_SMM_SPAN; i += 256) {
        const int r = i / VCT_SMM_SPAN, c = i - r * VCT_SMM_SPAN;
        const int x = bx + c, y = by + r;
        tex[r][c] = (x < S && y < S) ? min(words[(size_t)y * S + x] - ebase, VCT_SHADOW_ONE) : 0xffffffffu;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < VCT_SMM_SPAN * 8; i += 256) {
        const int r = i >> 3, t = i & 7;
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int k = 0; k < 8 + VCT_SHADOW_TILE_REACH; ++k) {
            const uint32_t v = tex[r][t * 8 + k];
            lo = min(lo, v);
            hi = v == 0xffffffffu ? hi : max(hi, v);          // (outside the map: neutral for both)
        }
        rlo[r][t] = lo; rhi[r][t] = hi;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ty = threadIdx.x >> 3, tx = threadIdx.x & 7;
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int k = 0; k < 8 + VCT_SHADOW_TILE_REACH; ++k) { lo = min(lo, rlo[ty * 8 + k][tx]); hi = max(hi, rhi[ty * 8 + k][tx]); }
        const int gx = (bx >> 3) + tx, gy = (by >> 3) + ty;
        if (gx < nb && gy < nb) tiles[(size_t)gy * nb + gx] = make_uint2(lo, hi);
    }
}
}  // namespace

hipError_t vct_launch_shadow_minmax(const uint32_t* words, uint32_t ebase, int S, uint2* tiles, hipStream_t s) {
    const int nb = (S + 7) >> 3, g = (S + VCT_SMM_BLOCK - 1) / VCT_SMM_BLOCK;
    hipLaunchKernelGGL(k_shadow_minmax, dim3(g, g), dim3(256), 0, s, words, ebase, S, nb, tiles);
    return hipGetLastError();
}

hipError_t vct_launch_shadow_encode(const float* depth, uint32_t* words, size_t n, uint32_t ebase, hipStream_t s) {
    hipLaunchKernelGGL(k_shadow_encode, dim3(256 * 8), dim3(256), 0, s, depth, words, n, ebase);
    return hipGetLastError();
}

hipError_t vct_launch_shadow_decode(const uint32_t* words, float* depth, size_t n, uint32_t ebase, hipStream_t s) {
    hipLaunchKernelGGL(k_shadow_decode, dim3(256 * 8), dim3(256), 0, s, words, depth, n, ebase);
    return hipGetLastError();
}

static ShadeParams make_shade(const VctRasterArgs& a, const float view_proj[16], int W, int H, int row0, int row1) {
    ShadeParams p;
    p.r = make_raster(a, view_proj, W, H, row0 * VCT_TILE, row1 * VCT_TILE);
    p.r.material = a.material;        // main draw: alpha test before the depth write (trace.fs:169-172)
    p.r.tri_alpha = a.tri_alpha;
    p.r.albedo = a.albedo;
    p.r.tex = a.tex;
    return p;
}

hipError_t vct_launch_gbuffer_visibility(const VctRasterArgs& a, const float view_proj[16], int W, int H, int row0, int row1,
                                         hipStream_t s) {
    const ShadeParams p = make_shade(a, view_proj, W, H, row0, row1);
    return a.binned ? run_visibility_binned(a, p.r, s) : run_visibility(p.r, s);
}

hipError_t vct_launch_gbuffer_shade(const VctRasterArgs& a, const float view_proj[16], int W, int H, int row0, int row1,
                                    const uint32_t* shadow, uint32_t shadow_ebase, int shadow_size, const uint2* shadow_tiles,
                                    const float light_vp[16], float* tiled, hipStream_t s) {
    ShadeParams p = make_shade(a, view_proj, W, H, row0, row1);
    p.shadow_tiles = shadow ? shadow_tiles : nullptr;
    p.nrm = a.nrm; p.tan = a.tan; p.bit = a.bit;
    p.material = a.material; p.albedo = a.albedo; p.specular = a.specular;
    p.shadow = shadow; p.shadow_ebase = shadow_ebase; p.shadow_size = shadow_size;
    for (int i = 0; i < 16; ++i) p.light_vp[i] = light_vp[i];
    p.tiled = tiled;
    p.tiles_x = (W + VCT_TILE - 1) / VCT_TILE;
    p.tile0 = row0 * p.tiles_x;
    p.tile1 = row1 * p.tiles_x;
    const int tiles = p.tile1 - p.tile0;
    if (tiles <= 0) return hipSuccess;
    const int per_block = VCT_SHADE_BLOCK / 64;       // tiles (waves) per workgroup
    if (a.tex.texels) hipLaunchKernelGGL(k_gbuffer_shade<true>, dim3((tiles + per_block - 1) / per_block), dim3(VCT_SHADE_BLOCK), 0, s, p);
    else hipLaunchKernelGGL(k_gbuffer_shade<false>, dim3((tiles + per_block - 1) / per_block), dim3(VCT_SHADE_BLOCK), 0, s, p);
    return hipGetLastError();
}

hipError_t vct_launch_tri_alpha(const VctRasterArgs& a, int32_t* out, hipStream_t s) {
    if (a.ntri <= 0) return hipSuccess;
    RasterParams r;
    memset(&r, 0, sizeof(r));
    r.ntri = a.ntri;
    r.material = a.material;
    r.albedo = a.albedo;
    r.tex = a.tex;
    hipLaunchKernelGGL(k_tri_alpha, dim3((a.ntri + 255) / 256), dim3(256), 0, s, r, out);
    return hipGetLastError();
}

hipError_t vct_launch_tex_mip(const uint32_t* parent, int pw, int ph, uint32_t* level, int w, int h, hipStream_t s) {
    const size_t n = (size_t)w * h;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_tex_mip, dim3(blocks ? blocks : 1u), dim3(256), 0, s, parent, pw, ph, level, w, h);
    return hipGetLastError();
}

hipError_t vct_launch_untile_gbuffer(const float* tiled, float* planes_linear, int w, int h, hipStream_t s) {
    const int tx = (w + VCT_TILE - 1) / VCT_TILE;
    hipLaunchKernelGGL(k_untile_gbuffer, dim3(256 * 8), dim3(256), 0, s, tiled, planes_linear, w, h, tx);
    return hipGetLastError();
}

hipError_t vct_launch_area_divide_selftest(uint64_t seed, uint64_t count, unsigned long long* out, hipStream_t s) {
    hipLaunchKernelGGL(k_area_divide_selftest, dim3(256 * 16), dim3(256), 0, s, seed, count, out);
    return hipGetLastError();
}
