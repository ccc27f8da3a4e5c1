// vct_oracle_raster.cpp -- CPU restatement of the two raster input stages and of the
// material texture fetches around them.  TEST INFRASTRUCTURE ONLY (see vct_oracle.h): these are the checkers of
// csrc/vct_raster.hip; nothing in the product links them.
//
// Reference code restated: R/Voxel_Cone_Tracing.h:192-211 + S/Shadow.vs/.fs (depth pass), :161-189 +
// S/VoxelConeTracing.vs:23-37 (vertex stage), S/VoxelConeTracing.fs:110-128 (CalcBumpNormal), :132-163 (PCF),
// :167-172 (matColor, alpha test), :209-210 (specColor), R/main.cpp:55-58 (depth LESS, cull back), and the
// OpenGL 4.3 rasterisation rules those draws invoke.
#include "vct_oracle.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 normalize(V3 a) {
    const float l = sqrtf(dot(a, a));
    return l > 0.0f ? a * (1.0f / l) : V3{0, 0, 0};
}
inline void xform(const float* m, V3 p, float out[4]) {
    for (int r = 0; r < 4; ++r) out[r] = m[r] * p.x + m[4 + r] * p.y + m[8 + r] * p.z + m[12 + r];
}

constexpr int kMaxVar = 14;
struct RVert { float c[4]; float var[kMaxVar]; };

inline RVert lerp_vert(const RVert& a, const RVert& b, float t, int nvar) {
    RVert r;
    for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + (b.c[i] - a.c[i]) * t;
    for (int i = 0; i < nvar; ++i) r.var[i] = a.var[i] + (b.var[i] - a.var[i]) * t;
    return r;
}

// One clip-space triangle.  frag(x, y, z, var, duv) is called for every fragment that passes the depth test and
// returns false to discard it (trace.fs:171): a discarded fragment leaves the depth buffer untouched.  duv[4]
// (nvar == 14 only) = quad differences of the texture coordinate var[12..13]: (ds_dx, dt_dx, ds_dy, dt_dy) with
// d/dx = f(x ^ 1, y) - f(x, y), d/dy = f(x, y ^ 1) - f(x, y), the neighbour evaluated on this triangle's own
// interpolation (a helper invocation when it lies outside) -- vct_oracle.h "Mip-mapped sampling".
template <class Frag>
void raster_triangle(const RVert in[3], int nvar, bool cull_back, int W, int H, float* zbuf, Frag frag) {
    RVert poly[4];
    int np = 0;
    for (int i = 0; i < 3; ++i) {             // clip against z >= -w
        const RVert& a = in[i];
        const RVert& b = in[(i + 1) % 3];
        const float da = a.c[2] + a.c[3], db = b.c[2] + b.c[3];
        if (da >= 0.0f) poly[np++] = a;
        if ((da >= 0.0f) != (db >= 0.0f)) poly[np++] = lerp_vert(a, b, da / (da - db), nvar);
    }
    if (np < 3) return;
    for (int t = 1; t + 1 < np; ++t) {
        const RVert* v[3] = {&poly[0], &poly[t], &poly[t + 1]};
        double sx[3], sy[3], ux[3], uy[3];
        float sz[3], iw[3];
        bool bad = false;
        for (int k = 0; k < 3; ++k) {
            if (!(v[k]->c[3] > 1e-20f)) { bad = true; break; }
            iw[k] = 1.0f / v[k]->c[3];
            ux[k] = (double)((v[k]->c[0] * iw[k] * 0.5f + 0.5f) * (float)W);
            uy[k] = (double)((v[k]->c[1] * iw[k] * 0.5f + 0.5f) * (float)H);
            sx[k] = floor(ux[k] * 256.0 + 0.5) / 256.0;
            sy[k] = floor(uy[k] * 256.0 + 0.5) / 256.0;
            sz[k] = v[k]->c[2] * iw[k] * 0.5f + 0.5f;
        }
        if (bad) continue;
        double area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
        if (area == 0.0 || area != area) continue;
        if (area < 0.0 && cull_back) continue;
        const double sgn = area > 0.0 ? 1.0 : -1.0;
        area *= sgn;
        // Interpolation positions: the snapped ones coverage is decided on (mode 0), or -- vcto_set_gl_choices(1), what
        // Mesa llvmpipe does -- the unsnapped window coordinates (its plane equations are set up in float from them).
        const bool unsnapped = (vcto_get_gl_choices() & 4) != 0;
        const double* ix = unsnapped ? ux : sx;
        const double* iy = unsnapped ? uy : sy;
        const double iarea = unsnapped ? ((ux[1] - ux[0]) * (uy[2] - uy[0]) - (ux[2] - ux[0]) * (uy[1] - uy[0])) * sgn : area;
        if (!(iarea > 0.0)) continue;
        // barycentrics of a pixel centre on the interpolation positions (identical to the coverage edge functions in mode 0)
        auto bary = [&](double cx, double cy, float& b0, float& b1) {
            double f[2];
            for (int k = 0; k < 2; ++k) {
                const int a = (k + 1) % 3, b = (k + 2) % 3;
                const double dx = (ix[b] - ix[a]) * sgn, dy = (iy[b] - iy[a]) * sgn;
                f[k] = dx * (cy - iy[a]) - dy * (cx - ix[a]);
            }
            b0 = (float)(f[0] / iarea);
            b1 = (float)(f[1] / iarea);
        };
        const int x0 = std::max(0, (int)floor(std::min({sx[0], sx[1], sx[2]})));
        const int x1 = std::min(W - 1, (int)floor(std::max({sx[0], sx[1], sx[2]})));
        const int y0 = std::max(0, (int)floor(std::min({sy[0], sy[1], sy[2]})));
        const int y1 = std::min(H - 1, (int)floor(std::max({sy[0], sy[1], sy[2]})));
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) {
                const double cx = (double)px + 0.5, cy = (double)py + 0.5;
                double e[3];
                bool inside = true;
                for (int k = 0; k < 3; ++k) {
                    const int a = (k + 1) % 3, b = (k + 2) % 3;
                    const double dx = (sx[b] - sx[a]) * sgn, dy = (sy[b] - sy[a]) * sgn;
                    e[k] = dx * (cy - sy[a]) - dy * (cx - sx[a]);
                    const bool top_left = (dy > 0.0) || (dy == 0.0 && dx < 0.0);
                    if (e[k] < 0.0 || (e[k] == 0.0 && !top_left)) { inside = false; break; }
                }
                if (!inside) continue;
                float b0, b1;
                if (unsnapped) bary(cx, cy, b0, b1);
                else { b0 = (float)(e[0] / area); b1 = (float)(e[1] / area); }
                const float b2 = 1.0f - b0 - b1;
                const float z = b0 * sz[0] + b1 * sz[1] + b2 * sz[2];
                if (!(z >= 0.0f && z <= 1.0f)) continue;               // far-plane clip
                float& zb = zbuf[(size_t)py * W + px];
                if (!(z < zb)) continue;                                 // GL_LESS
                const float q0 = b0 * iw[0], q1 = b1 * iw[1], q2 = b2 * iw[2];
                const float qs = 1.0f / (q0 + q1 + q2);
                float var[kMaxVar];
                for (int i = 0; i < nvar; ++i)
                    var[i] = (q0 * v[0]->var[i] + q1 * v[1]->var[i] + q2 * v[2]->var[i]) * qs;
                float duv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (nvar == kMaxVar) {
                    auto uv_at = [&](int qx, int qy, float out[2]) {       // the same interpolation at another pixel centre
                        float c0, c1;
                        bary((double)qx + 0.5, (double)qy + 0.5, c0, c1);
                        const float c2 = 1.0f - c0 - c1;
                        const float r0 = c0 * iw[0], r1 = c1 * iw[1], r2 = c2 * iw[2];
                        const float rs = 1.0f / (r0 + r1 + r2);
                        for (int i = 0; i < 2; ++i)
                            out[i] = (r0 * v[0]->var[12 + i] + r1 * v[1]->var[12 + i] + r2 * v[2]->var[12 + i]) * rs;
                    };
                    float nx[2], ny[2];
                    if (vcto_get_gl_choices() & 2) {                    // one pair per quad, at its (even, even) pixel
                        float tl[2];
                        uv_at(px & ~1, py & ~1, tl);
                        uv_at(px | 1, py & ~1, nx);
                        uv_at(px & ~1, py | 1, ny);
                        duv[0] = nx[0] - tl[0]; duv[1] = nx[1] - tl[1];
                        duv[2] = ny[0] - tl[0]; duv[3] = ny[1] - tl[1];
                    } else {
                        uv_at(px ^ 1, py, nx);
                        uv_at(px, py ^ 1, ny);
                        duv[0] = nx[0] - var[12]; duv[1] = nx[1] - var[13];
                        duv[2] = ny[0] - var[12]; duv[3] = ny[1] - var[13];
                    }
                }
                if (frag(px, py, z, var, duv)) zb = z;
            }
    }
}

float shadow_fetch(const float* depth, int S, float u, float v) {   // bilinear, clamp-to-edge
    const float x = u * (float)S - 0.5f, y = v * (float)S - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    auto cl = [S](float f) { return f < 0.0f ? 0 : (f > (float)(S - 1) ? S - 1 : (int)f); };
    const int i0 = cl(fx), i1 = cl(fx + 1.0f), j0 = cl(fy), j1 = cl(fy + 1.0f);
    const float d00 = depth[(size_t)j0 * S + i0], d10 = depth[(size_t)j0 * S + i1];
    const float d01 = depth[(size_t)j1 * S + i0], d11 = depth[(size_t)j1 * S + i1];
    return (1 - a) * (1 - b) * d00 + a * (1 - b) * d10 + (1 - a) * b * d01 + a * b * d11;
}

inline int tex_of(const vcto_mesh* m, int mat, int slot) {
    if (!m->mat_tex || !m->textures) return -1;
    const int t = m->mat_tex[3 * (size_t)mat + slot];
    return t >= 0 && t < m->ntex ? t : -1;
}

}  // namespace

extern "C" {

int vcto_tex_num_levels(int width, int height) {
    int n = 1;
    for (int m = width > height ? width : height; m > 1; m >>= 1) ++n;
    return n;
}

size_t vcto_tex_level_offset(int width, int height, int level) {
    size_t off = 0;
    for (int k = 0; k < level; ++k) {
        const int w = std::max(1, width >> k), h = std::max(1, height >> k);
        off += (size_t)w * h;
    }
    return off;
}

void vcto_tex_build_mips(const uint8_t* rgba, int width, int height, uint8_t* chain) {
    memcpy(chain, rgba, (size_t)width * height * 4);
    const int nlev = vcto_tex_num_levels(width, height);
    for (int k = 1; k < nlev; ++k) {
        const int pw = std::max(1, width >> (k - 1)), ph = std::max(1, height >> (k - 1));
        const int w = std::max(1, width >> k), h = std::max(1, height >> k);
        const uint8_t* src = chain + 4 * vcto_tex_level_offset(width, height, k - 1);
        uint8_t* dst = chain + 4 * vcto_tex_level_offset(width, height, k);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int x0 = std::min(2 * x, pw - 1), x1 = std::min(2 * x + 1, pw - 1);
                const int y0 = std::min(2 * y, ph - 1), y1 = std::min(2 * y + 1, ph - 1);
                for (int c = 0; c < 4; ++c) {
                    const uint32_t sum = (uint32_t)src[4 * ((size_t)y0 * pw + x0) + c] + src[4 * ((size_t)y0 * pw + x1) + c] +
                                         src[4 * ((size_t)y1 * pw + x0) + c] + src[4 * ((size_t)y1 * pw + x1) + c];
                    dst[4 * ((size_t)y * w + x) + c] = (uint8_t)((sum + 2u) >> 2);
                }
            }
    }
}

// ---- implementation-defined GL choices, switchable for the reference-GLSL cross-check (tests/test_ref_gl.py) -----
// GL leaves (a) the precision of the level-of-detail lambda and (b) where inside the 2x2 quad the implicit
// derivatives are taken to the implementation.  Mode 0 (the build's definition, what csrc/vct_raster.hip implements):
// lambda from vcto_log2_det, per-pixel differences towards the quad neighbour.  Mode 1 = what Mesa llvmpipe does (the
// GL implementation oracle/_ref runs the reference's shaders on): lambda = 0.5 * ((exponent - 1) + mantissa) of rho^2
// (Mesa's lp_build_fast_log2: exact at powers of two, piecewise linear between); bit 1 (b) one derivative pair per quad taken
// at its (even x, even y) pixel: d/dx = f(x | 1, y & ~1) - f(x & ~1, y & ~1), d/dy = f(x & ~1, y | 1) - f(x & ~1, y & ~1);
// bit 2 (c) varyings and depth interpolated on the UNSNAPPED window positions (coverage stays on the 1/256 grid).
static int g_gl_choices = 0;
void vcto_set_gl_choices(int mode) { g_gl_choices = mode; }
int vcto_get_gl_choices(void) { return g_gl_choices; }
static float fast_log2_mesa(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    const int e = (int)((b >> 23) & 0xffu) - 127 - 1;
    const uint32_t mb = (b & 0x7fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mb, 4);
    return (float)e + m;
}

// log2 without a math library (the GPU runs the same sequence, csrc/vct_internal.h vct_log2_det): exponent +
// 2/ln2 * atanh((f - 1) / (f + 1)) of the mantissa f, centred on [sqrt(1/2), sqrt(2)); series to s^9
float vcto_log2_det(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    int e = (int)(b >> 23) - 127;
    const uint32_t mb = (b & 0x7fffffu) | 0x3f800000u;
    float f;
    memcpy(&f, &mb, 4);
    if (f > 1.41421354f) { f = f * 0.5f; e += 1; }
    const float s = (f - 1.0f) / (f + 1.0f);
    const float s2 = s * s;
    float p = 0.111111112f;
    p = fmaf(p, s2, 0.142857149f);
    p = fmaf(p, s2, 0.2f);
    p = fmaf(p, s2, 0.333333343f);
    p = fmaf(p, s2, 1.0f);
    return fmaf(s * p, 2.88539004f, (float)e);
}

namespace {
// [GL] bilinear, GL_REPEAT, on one level (W x H texels at `texels`)
void tex_bilinear(const uint8_t* texels, int W, int H, float u, float v, float out[4]) {
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    auto wrap = [](int i, int n) { const int r = i % n; return r < 0 ? r + n : r; };     // GL_REPEAT
    const int i0 = wrap((int)fx, W), i1 = wrap((int)fx + 1, W), j0 = wrap((int)fy, H), j1 = wrap((int)fy + 1, H);
    const uint8_t* p00 = texels + 4 * ((size_t)j0 * W + i0);
    const uint8_t* p10 = texels + 4 * ((size_t)j0 * W + i1);
    const uint8_t* p01 = texels + 4 * ((size_t)j1 * W + i0);
    const uint8_t* p11 = texels + 4 * ((size_t)j1 * W + i1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    for (int c = 0; c < 4; ++c)
        out[c] = w00 * ((float)p00[c] / 255.0f) + w10 * ((float)p10[c] / 255.0f) + w01 * ((float)p01[c] / 255.0f) +
                 w11 * ((float)p11[c] / 255.0f);
}
}  // namespace

void vcto_tex_sample(const vcto_texture* t, float u, float v, float out[4]) {
    tex_bilinear(t->rgba, t->width, t->height, u, v, out);
}

void vcto_tex_sample_lod(const vcto_texture* t, float u, float v, float ds_dx, float dt_dx, float ds_dy, float dt_dy,
                         float out[4]) {
    const int W = t->width, H = t->height;
    if (!t->mips || t->nlev <= 1) { tex_bilinear(t->rgba, W, H, u, v, out); return; }
    const float du_dx = ds_dx * (float)W, dv_dx = dt_dx * (float)H;
    const float du_dy = ds_dy * (float)W, dv_dy = dt_dy * (float)H;
    const float ax = du_dx * du_dx + dv_dx * dv_dx, ay = du_dy * du_dy + dv_dy * dv_dy;
    const float m = ax > ay ? ax : ay;
    auto level = [&](int k, float o[4]) {
        tex_bilinear(t->mips + 4 * vcto_tex_level_offset(W, H, k), std::max(1, W >> k), std::max(1, H >> k), u, v, o);
    };
    if (!(m > 1.0f)) { level(0, out); return; }                // magnification (and NaN)
    const float lam = (g_gl_choices & 1) ? fast_log2_mesa(m) * 0.5f : 0.5f * vcto_log2_det(m);
    const int q = t->nlev - 1;
    if (lam >= (float)q) { level(q, out); return; }
    const int d = (int)lam;
    const float f = lam - (float)d, g = 1.0f - f;
    float t1[4], t2[4];
    level(d, t1);
    level(d + 1, t2);
    for (int c = 0; c < 4; ++c) out[c] = fmaf(f, t2[c], g * t1[c]);
}

void vcto_render_shadow_map(const vcto_mesh* s, const float light_vp[16], int32_t S, float* depth) {
    const size_t n = (size_t)S * S;
    for (size_t i = 0; i < n; ++i) depth[i] = 1.0f;                      // glClear depth
    for (int t = 0; t < s->ntri; ++t) {
        RVert v[3];
        for (int k = 0; k < 3; ++k) {
            const float* p = &s->pos[(size_t)t * 9 + 3 * k];
            xform(light_vp, V3{p[0] * s->model_scale, p[1] * s->model_scale, p[2] * s->model_scale}, v[k].c);
        }
        raster_triangle(v, 0, true, S, S, depth, [](int, int, float, const float*, const float*) { return true; });
    }
    const float q = 16777215.0f;                                         // DEPTH_COMPONENT24
    for (size_t i = 0; i < n; ++i) depth[i] = (float)(floor((double)depth[i] * q + 0.5) / q);
}

void vcto_render_gbuffer(const vcto_mesh* s, const float view_proj[16], int32_t W, int32_t H,
                         const float* shadow_depth, int32_t shadow_size, const float light_vp[16],
                         float* planes) {
    const size_t npix = (size_t)W * H;
    memset(planes, 0, npix * 23 * sizeof(float));
    std::vector<float> zbuf(npix, 1.0f);
    std::vector<int32_t> mat(npix, -1);
    std::vector<float> uvbuf(npix * 2, 0.0f);
    std::vector<float> duvbuf(npix * 4, 0.0f);      // quad differences of the winning fragment's texture coordinate
    const float ms = s->model_scale;
    for (int t = 0; t < s->ntri; ++t) {
        RVert v[3];
        for (int k = 0; k < 3; ++k) {
            const size_t o9 = (size_t)t * 9 + 3 * k;
            const V3 w = {s->pos[o9] * ms, s->pos[o9 + 1] * ms, s->pos[o9 + 2] * ms};   // trace.vs:27
            xform(view_proj, w, v[k].c);                                                  // trace.vs:25
            float* o = v[k].var;
            o[0] = w.x; o[1] = w.y; o[2] = w.z;
            for (int i = 0; i < 3; ++i) {
                o[3 + i] = s->nrm[o9 + i] * ms;    // trace.vs:31 (w = 0)
                o[6 + i] = s->tan[o9 + i] * ms;    // trace.vs:32
                o[9 + i] = s->bit[o9 + i] * ms;    // trace.vs:33
            }
            o[12] = s->uv ? s->uv[(size_t)t * 6 + 2 * k] : 0.0f;                           // trace.vs:36
            o[13] = s->uv ? s->uv[(size_t)t * 6 + 2 * k + 1] : 0.0f;
        }
        const int32_t m = s->material[t];
        const int td = tex_of(s, m, 0);
        raster_triangle(v, 14, true, W, H, zbuf.data(), [&](int x, int y, float, const float* var, const float* duv) {
            float alpha = s->albedo[4 * (size_t)m + 3];
            if (td >= 0) {
                float c[4];
                vcto_tex_sample_lod(&s->textures[td], var[12], var[13], duv[0], duv[1], duv[2], duv[3], c);   // trace.fs:167
                alpha = c[3];
            }
            if (alpha < 0.5f) return false;                                                  // trace.fs:171 discard
            const size_t i = (size_t)y * W + x;
            for (int k = 0; k < 12; ++k) planes[(size_t)k * npix + i] = var[k];
            uvbuf[2 * i] = var[12]; uvbuf[2 * i + 1] = var[13];
            for (int k = 0; k < 4; ++k) duvbuf[4 * i + k] = duv[k];
            mat[i] = m;
            return true;
        });
    }
    // per-pixel material + bump normal + shadow term (the non-cone part of trace.fs)
    for (size_t i = 0; i < npix; ++i) {
        if (mat[i] < 0) continue;
        auto G = [&](int k) -> float& { return planes[(size_t)k * npix + i]; };
        const V3 P = {G(0), G(1), G(2)}, N = {G(3), G(4), G(5)}, T = {G(6), G(7), G(8)}, B = {G(9), G(10), G(11)};
        const int m = mat[i];
        const float u = uvbuf[2 * i], v = uvbuf[2 * i + 1];
        const float* dq = &duvbuf[4 * i];
        // texture(sampler, tex [+ constant]) with the implicit derivatives of `tex` (a constant offset has none)
        auto fetch = [&](const vcto_texture* tx, float uu, float vv, float o[4]) {
            vcto_tex_sample_lod(tx, uu, vv, dq[0], dq[1], dq[2], dq[3], o);
        };
        const int td = tex_of(s, m, 0), tsp = tex_of(s, m, 1), th = tex_of(s, m, 2);
        const V3 c2 = cross(T, B);
        const float det = dot(T, cross(B, N));
        if (th < 0) {
            // CalcBumpNormal with a flat height map: normalize(TBN * (0,0,1)), TBN = inverse(transpose(M))
            const V3 bn = normalize(c2 * (1.0f / det));                               // trace.fs:127,175
            G(12) = bn.x; G(13) = bn.y; G(14) = bn.z;
        } else {
            const vcto_texture* ht = &s->textures[th];
            const float ox = 1.0f / (float)ht->width, oy = 1.0f / (float)ht->height;   // trace.fs:112
            float c0[4], c1[4], c3[4];
            fetch(ht, u, v, c0);                                                        // :114
            fetch(ht, u + ox, v, c1);                                                   // :115
            fetch(ht, u, v + oy, c3);                                                   // :116
            const float dx = c1[0] - c0[0], dy = c3[0] - c0[0];
            const V3 t1 = normalize(V3{1.0f, 0.0f, dx}), t2 = normalize(V3{0.0f, 1.0f, dy});   // :120-121
            const V3 bump = normalize(cross(t1, t2));                                   // :125
            // TBN = inverse(transpose(mat3(T,B,N))): columns (BxN, NxT, TxB) / det    trace.fs:175
            const float inv = 1.0f / det;
            const V3 k0 = cross(B, N) * inv, k1 = cross(N, T) * inv, k2 = c2 * inv;
            const V3 r = {k0.x * bump.x + k1.x * bump.y + k2.x * bump.z, k0.y * bump.x + k1.y * bump.y + k2.y * bump.z,
                          k0.z * bump.x + k1.z * bump.y + k2.z * bump.z};
            const V3 bn = normalize(r);                                                 // :127
            G(12) = bn.x; G(13) = bn.y; G(14) = bn.z;
        }
        if (td >= 0) {
            float c[4];
            fetch(&s->textures[td], u, v, c);                                           // trace.fs:167
            for (int k = 0; k < 4; ++k) G(15 + k) = c[k];
        } else {
            for (int k = 0; k < 4; ++k) G(15 + k) = s->albedo[4 * (size_t)m + k];
        }
        float sp[3] = {s->specular[3 * (size_t)m], s->specular[3 * (size_t)m + 1], s->specular[3 * (size_t)m + 2]};
        if (tsp >= 0) {
            float c[4];
            fetch(&s->textures[tsp], u, v, c);                                          // trace.fs:209
            sp[0] = c[0]; sp[1] = c[1]; sp[2] = c[2];
        }
        const bool has_gb = sqrtf(sp[1] * sp[1] + sp[2] * sp[2]) > 0.0f;
        G(19) = sp[0];
        G(20) = has_gb ? sp[1] : sp[0];                                                 // trace.fs:210
        G(21) = has_gb ? sp[2] : sp[0];
        float shadow = 25.0f * 0.111f;
        if (shadow_depth) {
            float d[4];
            xform(light_vp, P, d);                                                     // trace.vs:28
            const float cx = d[0] * 0.5f + 0.5f, cy = d[1] * 0.5f + 0.5f, cz = d[2] * 0.5f + 0.5f;   // :29
            float cnt = 0.0f;
            for (int x = -2; x <= 2; ++x)
                for (int y = -2; y <= 2; ++y) {
                    const float ox = 1.0f / (float)shadow_size * (float)x;            // trace.fs:147
                    const float oy = 1.0f / (float)shadow_size * (float)y;
                    if (cz / d[3] - 0.002f <= shadow_fetch(shadow_depth, shadow_size, cx + ox, cy + oy))
                        cnt += 1.0f;                                                   // trace.fs:151-152
                }
            shadow = cnt * 0.111f;                                                     // trace.fs:158
        }
        G(22) = shadow;
    }
}

}  // extern "C"
