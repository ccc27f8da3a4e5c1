// vct_oracle.cpp -- scalar CPU restatement of the voxel-cone-tracing GI path.
// TEST INFRASTRUCTURE ONLY; pinned against the reference's own GLSL (tests/test_ref_gl.py; see vct_oracle.h for the full header).
// Build: g++ -O2 -ffp-contract=off -mfma (oracle/Makefile).  Every fused multiply-add is an
// explicit fmaf(); everything else is one IEEE fp32 operation per C operator.
#include "vct_oracle.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// GLSL normalize(): v / length(v).
inline V3 normalize(V3 a) {
    float l = sqrtf(dot(a, a));
    return {a.x / l, a.y / l, a.z / l};
}
// GLSL reflect(I, N) = I - 2*dot(N,I)*N.
inline V3 reflect(V3 I, V3 N) {
    float d = 2.0f * dot(N, I);
    return {I.x - d * N.x, I.y - d * N.y, I.z - d * N.z};
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// [GL] A.1: unorm8 -> float = c / 255.
inline float unorm8(uint8_t c) { return (float)c / 255.0f; }

// [GL] float -> unorm8: round to nearest of f*255, clamped.
inline uint8_t to_unorm8(float f) {
    float s = f * 255.0f + 0.5f;
    if (!(s > 0.0f)) return 0;
    if (s >= 255.0f) return 255;
    return (uint8_t)(int)s;
}

// [GL] A.2 tri(level): trilinear with texel centres at (i+0.5)/N, GL_REPEAT (or clamp-to-edge).
// `chain` points at texel 0 of a buffer in which level `level` starts at (level_offset - skip)
void tri_sample(const vcto_params* p, const uint8_t* chain, int level, float ux, float uy, float uz,
                float out[4], size_t skip = 0) {
    const int V = p->V;
    const int N = V >> level;
    const uint8_t* base = chain + 4 * (vcto_level_offset_texels(V, level) - skip);
    const float fN = (float)N;
    const float u = ux * fN - 0.5f, v = uy * fN - 0.5f, w = uz * fN - 0.5f;
    const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    const float a = u - fu, b = v - fv, c = w - fw;
    int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
    int i1 = i0 + 1, j1 = j0 + 1, k1 = k0 + 1;
    if (p->wrap_repeat) {
        const int m = N - 1;
        i0 &= m; i1 &= m; j0 &= m; j1 &= m; k0 &= m; k1 &= m;
    } else {
        auto cl = [N](int i) { return i < 0 ? 0 : (i > N - 1 ? N - 1 : i); };
        i0 = cl(i0); i1 = cl(i1); j0 = cl(j0); j1 = cl(j1); k0 = cl(k0); k1 = cl(k1);
    }
    const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
    // weights in the order of the GL spec's sum: i fastest, then j, then k.
    const float wgt[8] = {(a0 * b0) * c0, (a * b0) * c0, (a0 * b) * c0, (a * b) * c0,
                          (a0 * b0) * c,  (a * b0) * c,  (a0 * b) * c,  (a * b) * c};
    const int xi[2] = {i0, i1}, yj[2] = {j0, j1}, zk[2] = {k0, k1};
    float acc[4] = {0, 0, 0, 0};
    for (int t = 0; t < 8; ++t) {
        const int x = xi[t & 1], y = yj[(t >> 1) & 1], z = zk[t >> 2];
        const uint8_t* tx = base + 4 * (((size_t)z * N + y) * N + x);
        for (int ch = 0; ch < 4; ++ch) {
            const float tv = unorm8(tx[ch]);
            acc[ch] = (t == 0) ? wgt[0] * tv : fmaf(wgt[t], tv, acc[ch]);
        }
    }
    for (int ch = 0; ch < 4; ++ch) out[ch] = acc[ch];
}

// [GL] A.2 textureLod, LINEAR_MIPMAP_LINEAR min / LINEAR mag, no bias.
// Directional sample of a level >= 1 (vct_oracle.h, anisotropic mip volumes)
void tri_sample_aniso(const vcto_params* p, const uint8_t* aniso, int level, float ux, float uy, float uz,
                      V3 dir, float out[4]) {
    const size_t V3n = (size_t)p->V * p->V * p->V;
    const size_t stride = 4 * (vcto_chain_texels(p->V) - V3n);
    const float w[3] = {dir.x * dir.x, dir.y * dir.y, dir.z * dir.z};
    const float comp[3] = {dir.x, dir.y, dir.z};
    float t[3][4];
    for (int a = 0; a < 3; ++a) {
        const int d = 2 * a + (comp[a] >= 0.0f ? 0 : 1);
        tri_sample(p, aniso + (size_t)d * stride, level, ux, uy, uz, t[a], V3n);
    }
    for (int ch = 0; ch < 4; ++ch) {
        float r = w[0] * t[0][ch];
        r = fmaf(w[1], t[1][ch], r);
        r = fmaf(w[2], t[2][ch], r);
        out[ch] = r;
    }
}

void texture_lod(const vcto_params* p, const uint8_t* chain, float ux, float uy, float uz, float lod,
                 float out[4], const uint8_t* aniso = nullptr, V3 dir = V3{0, 0, 0}) {
    const int maxl = ilog2(p->V);
    float lam = lod;
    if (!(lam > 0.0f)) {  // magnification (and NaN): level 0 only
        tri_sample(p, chain, 0, ux, uy, uz, out);
        return;
    }
    if (lam > (float)maxl) lam = (float)maxl;
    const float fl = floorf(lam);
    const int d1 = (int)fl;
    const int d2 = d1 + 1 > maxl ? maxl : d1 + 1;
    const float f = lam - fl;
    float t1[4], t2[4];
    if (aniso && d1 >= 1) tri_sample_aniso(p, aniso, d1, ux, uy, uz, dir, t1);
    else tri_sample(p, chain, d1, ux, uy, uz, t1);
    if (aniso && d2 >= 1) tri_sample_aniso(p, aniso, d2, ux, uy, uz, dir, t2);
    else tri_sample(p, chain, d2, ux, uy, uz, t2);
    const float g = 1.0f - f;
    for (int ch = 0; ch < 4; ++ch) out[ch] = fmaf(f, t2[ch], g * t1[ch]);
}

// trace.fs:59-66
void sample_voxels(const vcto_params* p, const uint8_t* chain, V3 pos, float lod, float out[4],
                   const uint8_t* aniso = nullptr, V3 dir = V3{0, 0, 0}) {
    const float half = p->G * 0.5f;
    float ux = pos.x / half, uy = pos.y / half, uz = pos.z / half;
    ux = ux * 0.5f + 0.5f;
    uy = uy * 0.5f + 0.5f;
    uz = uz * 0.5f + 0.5f;
    texture_lod(p, chain, ux, uy, uz, lod, out, aniso, dir);
}

// trace.fs:82-107
int cone_trace(const vcto_params* p, const uint8_t* chain, V3 P, V3 Nw, V3 dir, float tan_half,
               float out[4], const uint8_t* aniso = nullptr) {
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    float alpha = 0.0f, occlusion = 0.0f;
    const float vs = p->G / (float)p->V;                 // trace.fs:90
    float dist = vs;                                     // trace.fs:91
    const V3 start = add(P, mul(Nw, vs));                // trace.fs:92
    int steps = 0;
    while (dist < p->max_distance && alpha < p->max_alpha) {   // trace.fs:94
        const float diameter = fmaxf(vs, 2.0f * tan_half * dist);   // :96
        const float lod = log2f(diameter / vs);                     // :97
        float vc[4];
        sample_voxels(p, chain, add(start, mul(dir, dist)), lod, vc, aniso, dir);   // :98
        const float oma = 1.0f - alpha;
        cr = fmaf(oma, vc[0], cr);                                  // :100
        cg = fmaf(oma, vc[1], cg);
        cb = fmaf(oma, vc[2], cb);
        occlusion = occlusion + (oma * vc[3]) / (1.0f + 0.03f * diameter);   // :101
        alpha = fmaf(oma, vc[3], alpha);                            // :102
        dist = dist + diameter;                                     // :103
        ++steps;
    }
    out[0] = cr; out[1] = cg; out[2] = cb; out[3] = occlusion;      // :106
    return steps;
}

const float kConeDirs[18] = {0.0f, 0.0f, 1.0f,
                             0.0f, 0.866025f, 0.5f,
                             0.823639f, 0.267617f, 0.5f,
                             0.509037f, -0.700629f, 0.5f,
                             -0.509037f, -0.700629f, 0.5f,
                             -0.823639f, 0.267617f, 0.5f};                 // trace.fs:49-57
const float kConeWeights[6] = {0.25f, 0.15f, 0.15f, 0.15f, 0.15f, 0.15f};  // trace.fs:48

}  // namespace

extern "C" {

void vcto_default_params(vcto_params* p) {
    memset(p, 0, sizeof(*p));
    p->V = 128;
    p->G = 150.0f;
    p->camera_pos[0] = 0.0f; p->camera_pos[1] = 4.0f; p->camera_pos[2] = 0.0f;   // VCT.h:8
    p->light_dir[0] = 0.0f; p->light_dir[1] = 1.0f; p->light_dir[2] = 0.25f;     // VCT.h:14
    p->ambient_factor = 0.1f;
    p->shininess = 20.0f;
    p->max_distance = 75.0f;
    p->max_alpha = 0.95f;
    p->tan_diffuse = 0.577f;
    p->tan_specular = 0.07f;
    p->wrap_repeat = 1;
}

int vcto_num_levels(int V) { return ilog2(V) + 1; }

size_t vcto_level_offset_texels(int V, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) {
        size_t n = (size_t)(V >> l);
        off += n * n * n;
    }
    return off;
}

size_t vcto_chain_texels(int V) { return vcto_level_offset_texels(V, vcto_num_levels(V)); }

void vcto_build_mips(uint8_t* chain, int V) {
    const int nl = vcto_num_levels(V);
    for (int l = 1; l < nl; ++l) {
        const int Ns = V >> (l - 1), Nd = V >> l;
        const uint8_t* src = chain + 4 * vcto_level_offset_texels(V, l - 1);
        uint8_t* dst = chain + 4 * vcto_level_offset_texels(V, l);
        for (int z = 0; z < Nd; ++z)
            for (int y = 0; y < Nd; ++y)
                for (int x = 0; x < Nd; ++x)
                    for (int ch = 0; ch < 4; ++ch) {
                        unsigned sum = 0;
                        for (int t = 0; t < 8; ++t) {
                            const size_t sx = 2 * x + (t & 1), sy = 2 * y + ((t >> 1) & 1),
                                         sz = 2 * z + (t >> 2);
                            sum += src[4 * ((sz * Ns + sy) * Ns + sx) + ch];
                        }
                        // mean of 8 unorm8 values, rounded to the nearest unorm8 (half up).
                        dst[4 * (((size_t)z * Nd + y) * Nd + x) + ch] = (uint8_t)((sum + 4u) >> 3);
                    }
    }
}

void vcto_build_mips_aniso(const uint8_t* level0, int V, uint8_t* aniso) {
    const int nl = vcto_num_levels(V);
    const size_t V3n = (size_t)V * V * V;
    const size_t stride = 4 * (vcto_chain_texels(V) - V3n);
    for (int d = 0; d < 6; ++d) {
        const int axis = d >> 1;
        const bool toward_plus = (d & 1) == 0;
        uint8_t* chain_d = aniso + (size_t)d * stride;
        for (int l = 1; l < nl; ++l) {
            const int Ns = V >> (l - 1), Nd = V >> l;
            const uint8_t* src = l == 1 ? level0 : chain_d + 4 * (vcto_level_offset_texels(V, l - 1) - V3n);
            uint8_t* dst = chain_d + 4 * (vcto_level_offset_texels(V, l) - V3n);
            const int oa = axis == 0 ? 1 : 0, ob = axis == 2 ? 1 : 2;     // the two other axes, lower first
            for (int z = 0; z < Nd; ++z)
                for (int y = 0; y < Nd; ++y)
                    for (int x = 0; x < Nd; ++x) {
                        float acc[4] = {0, 0, 0, 0};
                        for (int pr = 0; pr < 4; ++pr) {
                            int c[3] = {0, 0, 0};
                            c[oa] = pr & 1;
                            c[ob] = pr >> 1;
                            float F[4], B[4];
                            for (int side = 0; side < 2; ++side) {
                                c[axis] = side;
                                const size_t sx = 2 * (size_t)x + c[0], sy = 2 * (size_t)y + c[1], sz = 2 * (size_t)z + c[2];
                                const uint8_t* t = src + 4 * ((sz * Ns + sy) * Ns + sx);
                                float* o = (side == 0) == toward_plus ? F : B;     // lower coordinate first when travelling +
                                for (int ch = 0; ch < 4; ++ch) o[ch] = unorm8(t[ch]);
                            }
                            const float oma = 1.0f - F[3];
                            for (int ch = 0; ch < 4; ++ch) {
                                const float comp = fmaf(oma, B[ch], F[ch]);
                                acc[ch] = pr == 0 ? comp : acc[ch] + comp;
                            }
                        }
                        uint8_t* o = dst + 4 * (((size_t)z * Nd + y) * Nd + x);
                        for (int ch = 0; ch < 4; ++ch) o[ch] = to_unorm8(acc[ch] * 0.25f);
                    }
        }
    }
}

void vcto_sample(const vcto_params* p, const uint8_t* chain, const float pos[3], float lod,
                 float out[4]) {
    sample_voxels(p, chain, v3(pos), lod, out);
}

void vcto_texture_lod(const vcto_params* p, const uint8_t* chain, const float uvw[3], float lod, float out[4]) {
    texture_lod(p, chain, uvw[0], uvw[1], uvw[2], lod, out);
}

int vcto_cone(const vcto_params* p, const uint8_t* chain, const float P[3], const float Nw[3],
              const float dir[3], float tan_half, float out[4]) {
    return cone_trace(p, chain, v3(P), v3(Nw), v3(dir), tan_half, out);
}

void vcto_cone_constants(float dirs[18], float weights[6]) {
    memcpy(dirs, kConeDirs, sizeof(kConeDirs));
    memcpy(weights, kConeWeights, sizeof(kConeWeights));
}

int vcto_max_steps(const vcto_params* p, float tan_half, float* last_lod) {
    const float vs = p->G / (float)p->V;
    float dist = vs, lod = 0.0f;
    int steps = 0;
    while (dist < p->max_distance) {
        const float diameter = fmaxf(vs, 2.0f * tan_half * dist);
        lod = log2f(diameter / vs);
        dist = dist + diameter;
        ++steps;
    }
    if (last_lod) *last_lod = lod;
    return steps;
}

}  // extern "C" (reopened below)

namespace {
int shade_pixel(const vcto_params* p, const uint8_t* chain, const uint8_t* aniso, const float gb[23],
                float out[4], uint8_t steps[7], float cones[28]) {
    const float* alb = gb + VCTO_GB_ALBEDO;
    uint8_t st[7] = {0, 0, 0, 0, 0, 0, 0};
    float cn[28];
    memset(cn, 0, sizeof(cn));
    if (alb[3] < 0.5f) {                                       // trace.fs:171 discard
        const float c = p->ambient_factor < 0.5f ? 0.5f : 1.0f;   // VCT.h:156-159
        out[0] = c; out[1] = c; out[2] = c; out[3] = 1.0f;
        if (steps) memcpy(steps, st, 7);
        if (cones) memcpy(cones, cn, sizeof(cn));
        return 0;
    }
    const V3 P = v3(gb + VCTO_GB_P), Nw = v3(gb + VCTO_GB_NW);
    const V3 T = v3(gb + VCTO_GB_TW), B = v3(gb + VCTO_GB_BW);
    const V3 N = v3(gb + VCTO_GB_BUMPN);
    const float shadow = gb[VCTO_GB_SHADOW];

    // trace.fs:175  TBN = inverse(transpose(mat3(T,B,N))): rows of the transposed matrix are
    // T,B,N; its inverse has columns (BxN, NxT, TxB)/det.
    const V3 c0 = cross(B, Nw), c1 = cross(Nw, T), c2 = cross(T, B);
    const float inv_det = 1.0f / dot(T, c0);
    const V3 k0 = mul(c0, inv_det), k1 = mul(c1, inv_det), k2 = mul(c2, inv_det);

    const V3 L = normalize(v3(p->light_dir));                              // :179
    const V3 E = normalize(sub(v3(p->camera_pos), P));                     // trace.vs:34, :181
    const float cos_theta = fmaxf(dot(N, L), 0.0f);                        // :188
    const float direct_diffuse = shadow * cos_theta;                       // :192

    float ind[4] = {0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) {                                          // :196-199
        const float* d = kConeDirs + 3 * i;
        V3 dir = {k0.x * d[0] + k1.x * d[1] + k2.x * d[2],
                  k0.y * d[0] + k1.y * d[1] + k2.y * d[2],
                  k0.z * d[0] + k1.z * d[1] + k2.z * d[2]};
        dir = normalize(dir);
        float c[4];
        st[i] = (uint8_t)cone_trace(p, chain, P, Nw, dir, p->tan_diffuse, c, aniso);
        for (int ch = 0; ch < 4; ++ch) {
            cn[4 * i + ch] = c[ch];
            ind[ch] = fmaf(kConeWeights[i], c[ch], ind[ch]);
        }
    }
    const float occlusion = 1.0f - ind[3];                                 // :201
    const float dr = (direct_diffuse + occlusion * ind[0]) * alb[0];       // :205
    const float dg = (direct_diffuse + occlusion * ind[1]) * alb[1];
    const float db = (direct_diffuse + occlusion * ind[2]) * alb[2];

    const float* sc = gb + VCTO_GB_SPEC;
    const V3 R = normalize(reflect(mul(L, -1.0f), N));                     // :212
    const float spec = powf(fmaxf(dot(E, R), 0.0f), p->shininess);         // :213
    const float direct_spec = spec * shadow;                               // :214
    const V3 Rd = normalize(reflect(mul(E, -1.0f), N));                    // :217
    float s[4];
    st[6] = (uint8_t)cone_trace(p, chain, P, Nw, Rd, p->tan_specular, s, aniso);  // :218
    for (int ch = 0; ch < 4; ++ch) cn[24 + ch] = s[ch];
    const float spec_occ = 1.0f - s[3];                                    // :221
    const float sr = (s[0] + spec_occ * direct_spec) * sc[0];              // :223
    const float sg = (s[1] + spec_occ * direct_spec) * sc[1];
    const float sb = (s[2] + spec_occ * direct_spec) * sc[2];

    const float ar = p->ambient_factor * alb[0] * occlusion;               // :225
    const float ag = p->ambient_factor * alb[1] * occlusion;
    const float ab = p->ambient_factor * alb[2] * occlusion;

    out[0] = ar + dr + sr;                                                 // :227
    out[1] = ag + dg + sg;
    out[2] = ab + db + sb;
    out[3] = alb[3];
    if (steps) memcpy(steps, st, 7);
    if (cones) memcpy(cones, cn, sizeof(cn));
    return 1;
}
}  // namespace

extern "C" {

int vcto_shade_pixel(const vcto_params* p, const uint8_t* chain, const float gb[23], float out[4],
                     uint8_t steps[7], float cones[28]) {
    return shade_pixel(p, chain, nullptr, gb, out, steps, cones);
}

uint16_t vcto_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return (uint16_t)sign;                // <= 2^-25 rounds to zero
    const int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }          // half subnormal
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    const uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    uint32_t h = base + q;
    if (rem > halfway || (rem == halfway && (q & 1u))) ++h;
    return (uint16_t)(sign | h);
}

float vcto_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float f;
    if (e == 0) f = ldexpf((float)m, -24);
    else if (e == 31) f = m ? NAN : INFINITY;
    else f = ldexpf((float)(m | 0x400u), (int)e - 25);
    uint32_t x;
    memcpy(&x, &f, 4);
    x |= sign;
    memcpy(&f, &x, 4);
    return f;
}

uint64_t vcto_trace(const vcto_params* p, const uint8_t* chain, const float* planes, size_t npix,
                    float* out32f, uint16_t* out16f, uint8_t* steps, float* cones, int nthreads) {
    return vcto_trace_aniso(p, chain, nullptr, planes, npix, out32f, out16f, steps, cones, nthreads);
}

uint64_t vcto_trace_aniso(const vcto_params* p, const uint8_t* chain, const uint8_t* aniso,
                          const float* planes, size_t npix, float* out32f, uint16_t* out16f,
                          uint8_t* steps, float* cones, int nthreads) {
    auto work = [&](size_t lo, size_t hi, uint64_t* total) {
        uint64_t t = 0;
        for (size_t i = lo; i < hi; ++i) {
            float gb[23], o[4], cn[28];
            uint8_t st[7];
            for (int k = 0; k < 23; ++k) gb[k] = planes[(size_t)k * npix + i];
            shade_pixel(p, chain, aniso, gb, o, st, cn);
            for (int k = 0; k < 7; ++k) t += st[k];
            if (out32f) memcpy(out32f + 4 * i, o, 16);
            if (out16f) for (int k = 0; k < 4; ++k) out16f[4 * i + k] = vcto_f32_to_f16(o[k]);
            if (steps) memcpy(steps + 7 * i, st, 7);
            if (cones) memcpy(cones + 28 * i, cn, sizeof(cn));
        }
        *total = t;
    };
    if (nthreads <= 1) {
        uint64_t t = 0;
        work(0, npix, &t);
        return t;
    }
    // static partition in 64-pixel blocks, round-robin so every thread sees the whole frame
    std::vector<uint64_t> totals((size_t)nthreads, 0);
    std::vector<std::thread> th;
    const size_t blk = 64, nblk = (npix + blk - 1) / blk;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            uint64_t acc = 0;
            for (size_t b = (size_t)t; b < nblk; b += (size_t)nthreads) {
                uint64_t part = 0;
                work(b * blk, std::min(npix, (b + 1) * blk), &part);
                acc += part;
            }
            totals[(size_t)t] = acc;
        });
    for (auto& x : th) x.join();
    uint64_t sum = 0;
    for (auto v : totals) sum += v;
    return sum;
}

// ---- PCF -----------------------------------------------------------------------------

float vcto_shadow_tex(const float* depth, int S, float u, float v) {
    // [GL] bilinear, clamp-to-edge (VCT.h:93-96)
    const float fS = (float)S;
    const float x = u * fS - 0.5f, y = v * fS - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    auto cl = [S](float f) {
        if (!(f > 0.0f)) return 0;
        if (f >= (float)(S - 1)) return S - 1;
        return (int)f;
    };
    const int i0 = cl(fx), i1 = cl(fx + 1.0f), j0 = cl(fy), j1 = cl(fy + 1.0f);
    const float d00 = depth[(size_t)j0 * S + i0], d10 = depth[(size_t)j0 * S + i1];
    const float d01 = depth[(size_t)j1 * S + i0], d11 = depth[(size_t)j1 * S + i1];
    const float a0 = 1.0f - a, b0 = 1.0f - b;
    float acc = (a0 * b0) * d00;
    acc = fmaf(a * b0, d10, acc);
    acc = fmaf(a0 * b, d01, acc);
    acc = fmaf(a * b, d11, acc);
    return acc;
}

void vcto_pcf25_batch(const float* depth, int S, const float* coords, size_t n, float bias, int32_t* counts) {
    for (size_t i = 0; i < n; ++i) counts[i] = vcto_pcf25(depth, S, coords + 3 * i, bias);
}

int vcto_pcf25(const float* depth, int S, const float coord[3], float bias) {
    const float cur = coord[2];   // coord.z / coord.w with w = 1 (orthographic light)
    const float inv = 1.0f / (float)S;
    int count = 0;
    for (int x = -2; x <= 2; ++x)
        for (int y = -2; y <= 2; ++y) {
            const float ox = inv * (float)x, oy = inv * (float)y;
            const float closest = vcto_shadow_tex(depth, S, coord[0] + ox, coord[1] + oy);
            if (cur - bias <= closest) ++count;
        }
    return count;
}

// ---- voxelization --------------------------------------------------------------------

static void mat_mul(const float a[16], const float b[16], float o[16]) {   // column-major o = a*b
    float t[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
            t[c * 4 + r] = s;
        }
    memcpy(o, t, sizeof(t));
}

static void glm_ortho(float l, float r, float b, float t, float n, float f, float m[16]) {
    memset(m, 0, 64);
    m[0] = 2.0f / (r - l);
    m[5] = 2.0f / (t - b);
    m[10] = -2.0f / (f - n);
    m[12] = -(r + l) / (r - l);
    m[13] = -(t + b) / (t - b);
    m[14] = -(f + n) / (f - n);
    m[15] = 1.0f;
}

static void glm_lookat(V3 eye, V3 center, V3 up, float m[16]) {
    const V3 f = normalize(sub(center, eye));
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    memset(m, 0, 64);
    m[0] = s.x; m[4] = s.y; m[8] = s.z;
    m[1] = u.x; m[5] = u.y; m[9] = u.z;
    m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
    m[12] = -dot(s, eye); m[13] = -dot(u, eye); m[14] = dot(f, eye);
    m[15] = 1.0f;
}

void vcto_voxel_proj(float G, int axis, float m[16]) {
    float o[16], v[16];
    glm_ortho(-G * 0.5f, G * 0.5f, -G * 0.5f, G * 0.5f, G * 0.5f, G * 1.5f, o);   // VCT.h:130
    if (axis == 1) glm_lookat({G, 0, 0}, {0, 0, 0}, {0, 1, 0}, v);                // VCT.h:132
    else if (axis == 2) glm_lookat({0, G, 0}, {0, 0, 0}, {0, 0, -1}, v);          // VCT.h:133
    else glm_lookat({0, 0, G}, {0, 0, 0}, {0, 1, 0}, v);                          // VCT.h:134
    mat_mul(o, v, m);
}

int vcto_dominant_axis(const float a[3], const float b[3], const float c[3]) {
    const V3 e1 = sub(v3(a), v3(b)), e2 = sub(v3(c), v3(a));   // vox.gs:24-25
    V3 n = normalize(cross(e1, e2));                           // :27
    const float nx = fabsf(n.x), ny = fabsf(n.y), nz = fabsf(n.z);
    if (nx >= ny && nx >= nz) return 1;                        // :34-39
    if (ny >= nx && ny >= nz) return 2;
    return 3;
}

void vcto_frag_to_voxel(int V, int axis, float fx, float fy, float fz, int32_t out[3]) {
    const int cx = (int)fx, cy = (int)fy, cz = (int)((float)V * fz);   // vox.fs:58
    if (axis == 1) { out[0] = V - 1 - cz; out[2] = V - 1 - cx; out[1] = cy; }        // :70-75
    else if (axis == 2) { out[2] = V - 1 - cy; out[1] = V - 1 - cz; out[0] = cx; }   // :76-81
    else { out[0] = cx; out[1] = cy; out[2] = V - 1 - cz; }                          // :82-86
}

namespace {

struct TriSetup {
    V3 w[3];    // world-space vertices  (ModelMatrix * position, vox.vs:21)
    V3 dc[3];   // DepthCoord.xyz*0.5+0.5 per vertex (vox.vs:18-19)
    float uv[3][2];   // TexCoord per vertex (vox.vs:17)
    int tex;    // diffuse texture of the triangle's material or -1
    int axis;
};

inline V3 xform_point(const float m[16], V3 p) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}

TriSetup setup_tri(const vcto_scene* s, int t) {
    TriSetup r;
    for (int k = 0; k < 3; ++k) {
        const float* q = s->pos + (size_t)t * 9 + 3 * k;
        r.w[k] = {q[0] * s->model_scale, q[1] * s->model_scale, q[2] * s->model_scale};
        const V3 d = xform_point(s->light_vp, r.w[k]);
        r.dc[k] = {d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f};
        r.uv[k][0] = s->uv ? s->uv[(size_t)t * 6 + 2 * k] : 0.0f;
        r.uv[k][1] = s->uv ? s->uv[(size_t)t * 6 + 2 * k + 1] : 0.0f;
    }
    r.tex = -1;
    if (s->uv && s->mat_tex && s->textures) {
        const int ti = s->mat_tex[3 * (size_t)s->material[t]];
        if (ti >= 0 && ti < s->ntex) r.tex = ti;
    }
    const float a[3] = {r.w[0].x, r.w[0].y, r.w[0].z}, b[3] = {r.w[1].x, r.w[1].y, r.w[1].z},
                c[3] = {r.w[2].x, r.w[2].y, r.w[2].z};
    r.axis = vcto_dominant_axis(a, b, c);
    return r;
}

// vox.fs:56,88 value: unorm8(texture(DiffuseTexture, uv).rgb * PCF/25), a = 1 (flat material colour when the
// material has no diffuse texture).  b0..b2: the fragment's barycentrics; alb_out (optional): the albedo used.
// duv (mip-mapped textures): (ds_dx, dt_dx, ds_dy, dt_dy) of the voxelization raster, vct_oracle.h vcto_scene
inline void frag_value(const vcto_scene* s, int t, const TriSetup& ts, float b0, float b1, float b2, V3 dc,
                       uint8_t out[3], float* alb_out = nullptr, const float* duv = nullptr) {
    float texel[4];
    const float* alb = s->albedo + 4 * (size_t)s->material[t];
    if (ts.tex >= 0) {
        const float u = b0 * ts.uv[0][0] + b1 * ts.uv[1][0] + b2 * ts.uv[2][0];
        const float v = b0 * ts.uv[0][1] + b1 * ts.uv[1][1] + b2 * ts.uv[2][1];
        if (duv) vcto_tex_sample_lod(&s->textures[ts.tex], u, v, duv[0], duv[1], duv[2], duv[3], texel);
        else vcto_tex_sample(&s->textures[ts.tex], u, v, texel);
        alb = texel;
    }
    if (alb_out) { alb_out[0] = alb[0]; alb_out[1] = alb[1]; alb_out[2] = alb[2]; }
    float sh = 1.0f;
    if (s->shadow_depth) {
        const float c[3] = {dc.x, dc.y, dc.z};
        sh = (float)vcto_pcf25(s->shadow_depth, s->shadow_size, c, 0.002f) / 25.0f;   // vox.fs:46
    }
    out[0] = to_unorm8(alb[0] * sh);
    out[1] = to_unorm8(alb[1] * sh);
    out[2] = to_unorm8(alb[2] * sh);
}

}  // namespace

void vcto_voxelize_reference(const vcto_params* p, const vcto_scene* s, uint8_t* l0) {
    const int V = p->V;
    float proj[4][16];
    for (int a = 1; a <= 3; ++a) vcto_voxel_proj(p->G, a, proj[a]);
    const float fV = (float)V;
    for (int t = 0; t < s->ntri; ++t) {
        const TriSetup ts = setup_tri(s, t);
        // window coordinates: xy in [0,V], z in [0,1]  [GL viewport V x V, depth range 0..1]
        float wx[3], wy[3], wz[3];
        for (int k = 0; k < 3; ++k) {
            const V3 ndc = xform_point(proj[ts.axis], ts.w[k]);   // vox.gs:47 (w = 1)
            wx[k] = (ndc.x * 0.5f + 0.5f) * fV;
            wy[k] = (ndc.y * 0.5f + 0.5f) * fV;
            wz[k] = ndc.z * 0.5f + 0.5f;
        }
        // signed doubled area; orient edges so that "inside" is >= 0 for either winding (cull off)
        const float area = (wx[1] - wx[0]) * (wy[2] - wy[0]) - (wx[2] - wx[0]) * (wy[1] - wy[0]);
        if (area == 0.0f || area != area) continue;
        const float sgn = area > 0.0f ? 1.0f : -1.0f;
        int x0 = (int)floorf(std::min({wx[0], wx[1], wx[2]}));
        int x1 = (int)floorf(std::max({wx[0], wx[1], wx[2]}));
        int y0 = (int)floorf(std::min({wy[0], wy[1], wy[2]}));
        int y1 = (int)floorf(std::max({wy[0], wy[1], wy[2]}));
        x0 = std::max(x0, 0); y0 = std::max(y0, 0);
        x1 = std::min(x1, V - 1); y1 = std::min(y1, V - 1);
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) {
                const float cx = (float)px + 0.5f, cy = (float)py + 0.5f;
                float e[3];
                bool inside = true;
                for (int k = 0; k < 3; ++k) {
                    const int a = (k + 1) % 3, b = (k + 2) % 3;   // edge opposite vertex k
                    const float dx = (wx[b] - wx[a]) * sgn, dy = (wy[b] - wy[a]) * sgn;
                    e[k] = dx * (cy - wy[a]) - dy * (cx - wx[a]);
                    // [GL] top-left rule: a sample exactly on an edge belongs to the triangle only
                    // for left edges (dy < 0 in a y-up window... ) or top edges.
                    const bool top_left = (dy > 0.0f) || (dy == 0.0f && dx < 0.0f);
                    if (e[k] < 0.0f || (e[k] == 0.0f && !top_left)) { inside = false; break; }
                }
                if (!inside) continue;
                const float aa = area * sgn;
                const float l0b = e[0] / aa, l1b = e[1] / aa, l2b = 1.0f - l0b - l1b;
                const float fz = l0b * wz[0] + l1b * wz[1] + l2b * wz[2];
                const V3 dc = {l0b * ts.dc[0].x + l1b * ts.dc[1].x + l2b * ts.dc[2].x,
                               l0b * ts.dc[0].y + l1b * ts.dc[1].y + l2b * ts.dc[2].y,
                               l0b * ts.dc[0].z + l1b * ts.dc[1].z + l2b * ts.dc[2].z};
                int32_t vp[3];
                vcto_frag_to_voxel(V, ts.axis, cx, cy, fz, vp);
                if (vp[0] < 0 || vp[1] < 0 || vp[2] < 0 || vp[0] >= V || vp[1] >= V || vp[2] >= V)
                    continue;   // [GL] out-of-bounds imageStore is discarded
                uint8_t rgb[3];
                float duv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (ts.tex >= 0 && s->textures[ts.tex].mips) {
                    // texture() derivatives: the neighbouring pixel centres of the fragment's 2x2 quad, on this triangle
                    auto uv_at = [&](int qx, int qy, float o[2]) {
                        const float nx = (float)qx + 0.5f, ny = (float)qy + 0.5f;
                        float f[2];
                        for (int k = 0; k < 2; ++k) {
                            const int a = (k + 1) % 3, b = (k + 2) % 3;
                            const float dx = (wx[b] - wx[a]) * sgn, dy = (wy[b] - wy[a]) * sgn;
                            f[k] = dx * (ny - wy[a]) - dy * (nx - wx[a]);
                        }
                        const float c0 = f[0] / aa, c1 = f[1] / aa, c2 = 1.0f - c0 - c1;
                        o[0] = c0 * ts.uv[0][0] + c1 * ts.uv[1][0] + c2 * ts.uv[2][0];
                        o[1] = c0 * ts.uv[0][1] + c1 * ts.uv[1][1] + c2 * ts.uv[2][1];
                    };
                    float me[2], nx[2], ny[2];
                    uv_at(px, py, me);
                    uv_at(px ^ 1, py, nx);
                    uv_at(px, py ^ 1, ny);
                    duv[0] = nx[0] - me[0]; duv[1] = nx[1] - me[1]; duv[2] = ny[0] - me[0]; duv[3] = ny[1] - me[1];
                }
                frag_value(s, t, ts, l0b, l1b, l2b, dc, rgb, nullptr, duv);
                uint8_t* d = l0 + 4 * (((size_t)vp[2] * V + vp[1]) * V + vp[0]);
                d[0] = rgb[0]; d[1] = rgb[1]; d[2] = rgb[2]; d[3] = 255;   // last writer wins
            }
    }
}

namespace {

// Schwarz & Seidel 2010 conservative triangle / unit-box overlap, voxel-space coordinates.
struct ConsSetup {
    V3 n;
    float d1, d2;
    float ne[3][3][2];   // [plane xy,yz,zx][edge][2]
    float de[3][3];
};

inline float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

ConsSetup cons_setup(const V3 g[3]) {
    ConsSetup c;
    const V3 e[3] = {sub(g[1], g[0]), sub(g[2], g[1]), sub(g[0], g[2])};
    c.n = cross(e[0], e[1]);
    const V3 cp = {c.n.x > 0.0f ? 1.0f : 0.0f, c.n.y > 0.0f ? 1.0f : 0.0f, c.n.z > 0.0f ? 1.0f : 0.0f};
    c.d1 = dot(c.n, sub(cp, g[0]));
    c.d2 = dot(c.n, sub(sub(V3{1.0f, 1.0f, 1.0f}, cp), g[0]));
    // planes: 0 = xy (normal z), 1 = yz (normal x), 2 = zx (normal y)
    const int ax[3][2] = {{0, 1}, {1, 2}, {2, 0}};
    const float nsel[3] = {c.n.z, c.n.x, c.n.y};
    for (int pl = 0; pl < 3; ++pl) {
        const float sg = nsel[pl] >= 0.0f ? 1.0f : -1.0f;
        for (int i = 0; i < 3; ++i) {
            const float ea = comp(e[i], ax[pl][0]), eb = comp(e[i], ax[pl][1]);
            const float na = -eb * sg, nb = ea * sg;
            c.ne[pl][i][0] = na;
            c.ne[pl][i][1] = nb;
            const float va = comp(g[i], ax[pl][0]), vb = comp(g[i], ax[pl][1]);
            c.de[pl][i] = -(na * va + nb * vb) + fmaxf(0.0f, na) + fmaxf(0.0f, nb);
        }
    }
    return c;
}

inline bool cons_overlap(const ConsSetup& c, int i, int j, int k) {
    const V3 pp = {(float)i, (float)j, (float)k};
    const float np = dot(c.n, pp);
    if ((np + c.d1) * (np + c.d2) > 0.0f) return false;
    const int ax[3][2] = {{0, 1}, {1, 2}, {2, 0}};
    for (int pl = 0; pl < 3; ++pl) {
        const float pa = comp(pp, ax[pl][0]), pb = comp(pp, ax[pl][1]);
        for (int e = 0; e < 3; ++e)
            if (c.ne[pl][e][0] * pa + c.ne[pl][e][1] * pb + c.de[pl][e] < 0.0f) return false;
    }
    return true;
}

}  // namespace

void vcto_voxelize_conservative(const vcto_params* p, const vcto_scene* s, uint8_t* l0,
                                uint32_t* acc_out) {
    vcto_voxelize_conservative_attr(p, s, l0, acc_out, nullptr, nullptr);
}

namespace {
// one conservative fragment: voxel (linear index), vox.fs:88 value, and the triangle's attributes
struct ConsFrag {
    uint64_t vox;
    uint8_t rgb[3];
    uint8_t attr[6];     // albedo rgb (unorm8), biased face normal xyz
};
}  // namespace

// Fragments are collected per triangle range (one std::thread each -- the per-fragment arithmetic is the
// scalar restatement; integer sums make the result independent of the order), sorted by voxel and
// reduced, so memory scales with the surface (fragments), not with V^3: a 1024^3 grid needs no 16 GiB
// accumulator on the checker side either.
namespace {
void voxelize_conservative_z(const vcto_params* p, const vcto_scene* s, int z0, int z1, uint8_t* l0,
                             uint32_t* acc_out, uint8_t* attr_albedo, uint8_t* attr_normal);
}
void vcto_voxelize_conservative_attr(const vcto_params* p, const vcto_scene* s, uint8_t* l0,
                                     uint32_t* acc_out, uint8_t* attr_albedo, uint8_t* attr_normal) {
    voxelize_conservative_z(p, s, 0, p->V, l0, acc_out, attr_albedo, attr_normal);
}
/* z-slices [z0, z1) of the same voxelization: l0_slab is [(z1 - z0)][V][V][4].  A voxel's value depends only on the
 * triangles that overlap it, so a slab of the full result costs a slab of memory (checks of 1024^3 on small hosts). */
void vcto_voxelize_conservative_zslab(const vcto_params* p, const vcto_scene* s, int32_t z0, int32_t z1, uint8_t* l0_slab) {
    voxelize_conservative_z(p, s, z0, z1, l0_slab, nullptr, nullptr, nullptr);
}
namespace {
void voxelize_conservative_z(const vcto_params* p, const vcto_scene* s, int z0, int z1, uint8_t* l0,
                             uint32_t* acc_out, uint8_t* attr_albedo, uint8_t* attr_normal) {
    const int V = p->V;
    const float fV = (float)V;
    const size_t nvox = (size_t)V * V * (size_t)(z1 - z0);
    const bool want_attr = attr_albedo || attr_normal;
    auto work = [&](int t0, int t1, std::vector<ConsFrag>* out) {
      for (int t = t0; t < t1; ++t) {
        const TriSetup ts = setup_tri(s, t);
        ConsFrag f;
        memset(&f, 0, sizeof(f));
        if (want_attr) {
            const V3 fn = normalize(cross(sub(ts.w[1], ts.w[0]), sub(ts.w[2], ts.w[0])));
            const float fc[3] = {fn.x, fn.y, fn.z};
            for (int c = 0; c < 3; ++c) f.attr[3 + c] = (uint8_t)((int)floorf(fc[c] * 127.0f + 0.5f) + 128);
        }
        V3 g[3];
        for (int k = 0; k < 3; ++k)
            g[k] = {(ts.w[k].x / p->G + 0.5f) * fV, (ts.w[k].y / p->G + 0.5f) * fV,
                    (ts.w[k].z / p->G + 0.5f) * fV};
        const ConsSetup cs = cons_setup(g);
        if (cs.n.x == 0.0f && cs.n.y == 0.0f && cs.n.z == 0.0f) continue;   // degenerate
        if (cs.n.x != cs.n.x || cs.n.y != cs.n.y || cs.n.z != cs.n.z) continue;
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            const float mn = fminf(fminf(comp(g[0], a), comp(g[1], a)), comp(g[2], a));
            const float mx = fmaxf(fmaxf(comp(g[0], a), comp(g[1], a)), comp(g[2], a));
            lo[a] = std::max((int)floorf(mn), 0);
            hi[a] = std::min((int)floorf(mx), V - 1);
        }
        lo[2] = std::max(lo[2], z0);          // only the slices asked for
        hi[2] = std::min(hi[2], z1 - 1);
        // 2-D barycentric set-up in the plane orthogonal to the dominant axis (voxel space)
        const int ua = ts.axis == 1 ? 1 : 0, ub = ts.axis == 3 ? 1 : 2;   // X:(y,z) Y:(x,z) Z:(x,y)
        const float ax0 = comp(g[0], ua), ay0 = comp(g[0], ub);
        const float ax1 = comp(g[1], ua), ay1 = comp(g[1], ub);
        const float ax2 = comp(g[2], ua), ay2 = comp(g[2], ub);
        const float area = (ax1 - ax0) * (ay2 - ay0) - (ax2 - ax0) * (ay1 - ay0);
        for (int k = lo[2]; k <= hi[2]; ++k)
            for (int j = lo[1]; j <= hi[1]; ++j)
                for (int i = lo[0]; i <= hi[0]; ++i) {
                    if (!cons_overlap(cs, i, j, k)) continue;
                    const V3 ctr = {(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
                    const float cx = comp(ctr, ua), cy = comp(ctr, ub);
                    float b0 = ((ax1 - cx) * (ay2 - cy) - (ax2 - cx) * (ay1 - cy)) / area;
                    float b1 = ((ax2 - cx) * (ay0 - cy) - (ax0 - cx) * (ay2 - cy)) / area;
                    b0 = fminf(fmaxf(b0, 0.0f), 1.0f);
                    b1 = fminf(fmaxf(b1, 0.0f), 1.0f);
                    const float sum = b0 + b1;
                    if (sum > 1.0f) { b0 = b0 / sum; b1 = b1 / sum; }
                    const float b2 = fmaxf(1.0f - b0 - b1, 0.0f);
                    const V3 dc = {b0 * ts.dc[0].x + b1 * ts.dc[1].x + b2 * ts.dc[2].x,
                                   b0 * ts.dc[0].y + b1 * ts.dc[1].y + b2 * ts.dc[2].y,
                                   b0 * ts.dc[0].z + b1 * ts.dc[1].z + b2 * ts.dc[2].z};
                    float alb[3];
                    float duv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (ts.tex >= 0 && s->textures[ts.tex].mips) {
                        // no reference code for this mode: the UNCLAMPED barycentrics one voxel further along each
                        // in-plane axis give the texture coordinate's differences (vct_oracle.h vcto_scene)
                        auto uv_at = [&](float qx, float qy, float o[2]) {
                            const float c0 = ((ax1 - qx) * (ay2 - qy) - (ax2 - qx) * (ay1 - qy)) / area;
                            const float c1 = ((ax2 - qx) * (ay0 - qy) - (ax0 - qx) * (ay2 - qy)) / area;
                            const float c2 = 1.0f - c0 - c1;
                            o[0] = c0 * ts.uv[0][0] + c1 * ts.uv[1][0] + c2 * ts.uv[2][0];
                            o[1] = c0 * ts.uv[0][1] + c1 * ts.uv[1][1] + c2 * ts.uv[2][1];
                        };
                        float me[2], nx[2], ny[2];
                        uv_at(cx, cy, me);
                        uv_at(cx + 1.0f, cy, nx);
                        uv_at(cx, cy + 1.0f, ny);
                        duv[0] = nx[0] - me[0]; duv[1] = nx[1] - me[1]; duv[2] = ny[0] - me[0]; duv[3] = ny[1] - me[1];
                    }
                    frag_value(s, t, ts, b0, b1, b2, dc, f.rgb, alb, duv);
                    if (want_attr)
                        for (int c = 0; c < 3; ++c) f.attr[c] = to_unorm8(alb[c]);      // the fragment's albedo
                    f.vox = ((uint64_t)(k - z0) * V + j) * V + i;
                    out->push_back(f);
                }
      }
    };
    int nthreads = s->ntri >= 20000 ? (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    std::vector<std::vector<ConsFrag>> parts((size_t)nthreads);
    if (nthreads == 1) {
        work(0, s->ntri, &parts[0]);
    } else {
        // dynamic chunks: large triangles (floors, walls) cluster in the triangle list
        std::atomic<int> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t]() {
                for (;;) {
                    const int t0 = next.fetch_add(64);
                    if (t0 >= s->ntri) break;
                    work(t0, std::min(s->ntri, t0 + 64), &parts[(size_t)t]);
                }
            });
        for (auto& x : th) x.join();
    }
    std::vector<ConsFrag> frags;
    {
        size_t n = 0;
        for (auto& v : parts) n += v.size();
        frags.reserve(n);
        for (auto& v : parts) { frags.insert(frags.end(), v.begin(), v.end()); std::vector<ConsFrag>().swap(v); }
    }
    std::sort(frags.begin(), frags.end(), [](const ConsFrag& a, const ConsFrag& b) { return a.vox < b.vox; });

    memset(l0, 0, nvox * 4);
    if (acc_out) memset(acc_out, 0, nvox * 16);
    if (attr_albedo) memset(attr_albedo, 0, nvox * 4);
    if (attr_normal) memset(attr_normal, 0, nvox * 4);
    for (size_t b = 0; b < frags.size();) {
        size_t e = b;
        uint32_t a[3] = {0, 0, 0}, q[6] = {0, 0, 0, 0, 0, 0}, c = 0;
        for (; e < frags.size() && frags[e].vox == frags[b].vox; ++e) {
            for (int k = 0; k < 3; ++k) a[k] += frags[e].rgb[k];
            for (int k = 0; k < 6; ++k) q[k] += frags[e].attr[k];
            ++c;
        }
        const size_t v = (size_t)frags[b].vox;
        const uint32_t h = c >> 1;                        // rounded integer mean
        uint8_t* d = l0 + 4 * v;
        d[0] = (uint8_t)((a[0] + h) / c);
        d[1] = (uint8_t)((a[1] + h) / c);
        d[2] = (uint8_t)((a[2] + h) / c);
        d[3] = 255;
        if (acc_out) { uint32_t* o = acc_out + 4 * v; o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = c; }
        for (int k = 0; k < 3; ++k) {
            if (attr_albedo) attr_albedo[4 * v + k] = (uint8_t)((q[k] + h) / c);
            if (attr_normal) attr_normal[4 * v + k] = (uint8_t)((q[3 + k] + h) / c);
        }
        if (attr_albedo) attr_albedo[4 * v + 3] = 255;
        if (attr_normal) attr_normal[4 * v + 3] = 255;
        b = e;
    }
}
}  // namespace

uint64_t vcto_bounce(const vcto_params* p, const uint8_t* chain0, const uint8_t* attr_albedo,
                     const uint8_t* attr_normal, uint8_t* out_l0, int nthreads) {
    const int V = p->V;
    const size_t nvox = (size_t)V * V * V;
    const float fV = (float)V;
    memcpy(out_l0, chain0, nvox * 4);
    auto work = [&](size_t lo, size_t hi, uint64_t* total) {
        uint64_t steps = 0;
        for (size_t v = lo; v < hi; ++v) {
            const uint8_t* src = chain0 + 4 * v;
            if (src[3] == 0) continue;
            const uint8_t* nq = attr_normal + 4 * v;
            const V3 nraw = {(float)((int)nq[0] - 128), (float)((int)nq[1] - 128), (float)((int)nq[2] - 128)};
            if (nraw.x == 0.0f && nraw.y == 0.0f && nraw.z == 0.0f) continue;
            const int i = (int)(v % (size_t)V), j = (int)((v / (size_t)V) % (size_t)V), k = (int)(v / ((size_t)V * V));
            const V3 P = {(((float)i + 0.5f) / fV - 0.5f) * p->G, (((float)j + 0.5f) / fV - 0.5f) * p->G,
                          (((float)k + 0.5f) / fV - 0.5f) * p->G};
            const V3 n = normalize(nraw);
            const V3 helper = fabsf(n.y) < 0.9f ? V3{0.0f, 1.0f, 0.0f} : V3{1.0f, 0.0f, 0.0f};
            const V3 t = normalize(cross(helper, n));
            const V3 b = cross(n, t);
            float ind[4] = {0, 0, 0, 0};
            for (int c = 0; c < 6; ++c) {
                const float* d = kConeDirs + 3 * c;
                V3 dir = {t.x * d[0] + b.x * d[1] + n.x * d[2], t.y * d[0] + b.y * d[1] + n.y * d[2],
                          t.z * d[0] + b.z * d[1] + n.z * d[2]};
                dir = normalize(dir);
                float cone[4];
                steps += (uint64_t)cone_trace(p, chain0, P, n, dir, p->tan_diffuse, cone);
                for (int ch = 0; ch < 4; ++ch) ind[ch] = fmaf(kConeWeights[c], cone[ch], ind[ch]);
            }
            const float occlusion = 1.0f - ind[3];
            const uint8_t* alb = attr_albedo + 4 * v;
            uint8_t* dst = out_l0 + 4 * v;
            for (int ch = 0; ch < 3; ++ch)
                dst[ch] = to_unorm8(unorm8(src[ch]) + unorm8(alb[ch]) * (occlusion * ind[ch]));
        }
        *total = steps;
    };
    if (nthreads <= 1) {
        uint64_t t = 0;
        work(0, nvox, &t);
        return t;
    }
    std::vector<uint64_t> totals((size_t)nthreads, 0);
    std::vector<std::thread> th;
    const size_t per = (nvox + (size_t)nthreads - 1) / (size_t)nthreads;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() { work(std::min(nvox, per * t), std::min(nvox, per * (t + 1)), &totals[(size_t)t]); });
    for (auto& x : th) x.join();
    uint64_t sum = 0;
    for (auto v : totals) sum += v;
    return sum;
}

}  // extern "C"
