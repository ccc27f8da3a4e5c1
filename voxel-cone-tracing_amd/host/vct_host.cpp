// vct_host.cpp -- procedural scenes, shadow-map raster and G-buffer raster (see vct_host.h).
#include "vct_host.h"

#include <math.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 normalize(V3 a) {
    const float l = sqrtf(dot(a, a));
    return l > 0.0f ? a * (1.0f / l) : V3{0, 0, 0};
}

// column-major 4x4, glm conventions (right-handed, NDC z in [-1,1])
struct M4 { float m[16]; };
M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.m[k * 4 + rr] * b.m[c * 4 + k];
            r.m[c * 4 + rr] = s;
        }
    return r;
}
M4 ortho(float l, float r, float b, float t, float n, float f) {
    M4 o;
    memset(o.m, 0, sizeof(o.m));
    o.m[0] = 2.0f / (r - l);
    o.m[5] = 2.0f / (t - b);
    o.m[10] = -2.0f / (f - n);
    o.m[12] = -(r + l) / (r - l);
    o.m[13] = -(t + b) / (t - b);
    o.m[14] = -(f + n) / (f - n);
    o.m[15] = 1.0f;
    return o;
}
M4 perspective(float fovy, float aspect, float n, float f) {
    M4 o;
    memset(o.m, 0, sizeof(o.m));
    const float t = tanf(fovy * 0.5f);
    o.m[0] = 1.0f / (aspect * t);
    o.m[5] = 1.0f / t;
    o.m[10] = -(f + n) / (f - n);
    o.m[11] = -1.0f;
    o.m[14] = -(2.0f * f * n) / (f - n);
    return o;
}
M4 look_at(V3 eye, V3 center, V3 up) {
    const V3 f = normalize(center - eye);
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    M4 o;
    memset(o.m, 0, sizeof(o.m));
    o.m[0] = s.x; o.m[4] = s.y; o.m[8] = s.z;
    o.m[1] = u.x; o.m[5] = u.y; o.m[9] = u.z;
    o.m[2] = -f.x; o.m[6] = -f.y; o.m[10] = -f.z;
    o.m[12] = -dot(s, eye); o.m[13] = -dot(u, eye); o.m[14] = dot(f, eye);
    o.m[15] = 1.0f;
    return o;
}
inline void xform(const float* m, V3 p, float out[4]) {
    for (int r = 0; r < 4; ++r) out[r] = m[r] * p.x + m[4 + r] * p.y + m[8 + r] * p.z + m[12 + r];
}

struct Material { float albedo[4]; float spec[3]; };

}  // namespace

struct vcth_scene {
    std::vector<float> pos, nrm, tan, bit;   // per triangle-vertex, 3 floats each
    std::vector<int32_t> mat;                // per triangle
    std::vector<Material> materials;
};

namespace {

struct Builder {
    vcth_scene* s;
    void tri(V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc, int m) {
        const V3 p[3] = {a, b, c}, n[3] = {na, nb, nc};
        for (int k = 0; k < 3; ++k) {
            const V3 nn = normalize(n[k]);
            const V3 hint = fabsf(nn.y) < 0.9f ? V3{0, 1, 0} : V3{1, 0, 0};
            const V3 t = normalize(cross(hint, nn));
            const V3 bt = cross(nn, t);   // T x B = N
            s->pos.insert(s->pos.end(), {p[k].x, p[k].y, p[k].z});
            s->nrm.insert(s->nrm.end(), {nn.x, nn.y, nn.z});
            s->tan.insert(s->tan.end(), {t.x, t.y, t.z});
            s->bit.insert(s->bit.end(), {bt.x, bt.y, bt.z});
        }
        s->mat.push_back(m);
    }
    // o + u*du + v*dv, front face normal = normalize(du x dv); nu x nv cells; optional displacement
    template <class F>
    void patch(V3 o, V3 du, V3 dv, int nu, int nv, int m, F disp) {
        nu = std::max(nu, 1); nv = std::max(nv, 1);
        const V3 n0 = normalize(cross(du, dv));
        auto P = [&](int i, int j) {
            const float u = (float)i / nu, v = (float)j / nv;
            return o + du * u + dv * v + n0 * disp(u, v);
        };
        auto Nn = [&](int i, int j) {
            const float e = 1e-3f;
            const float u = (float)i / nu, v = (float)j / nv;
            const V3 a = du + n0 * ((disp(u + e, v) - disp(u - e, v)) / (2 * e));
            const V3 b = dv + n0 * ((disp(u, v + e) - disp(u, v - e)) / (2 * e));
            return normalize(cross(a, b));
        };
        for (int j = 0; j < nv; ++j)
            for (int i = 0; i < nu; ++i) {
                const V3 a = P(i, j), b = P(i + 1, j), c = P(i + 1, j + 1), d = P(i, j + 1);
                tri(a, b, c, Nn(i, j), Nn(i + 1, j), Nn(i + 1, j + 1), m);
                tri(a, c, d, Nn(i, j), Nn(i + 1, j + 1), Nn(i, j + 1), m);
            }
    }
    void quad(V3 o, V3 du, V3 dv, int nu, int nv, int m) {
        patch(o, du, dv, nu, nv, m, [](float, float) { return 0.0f; });
    }
    // axis-aligned box, outward faces; cells ~ `cell` world units
    void box(V3 lo, V3 hi, float cell, int m) {
        const V3 d = hi - lo;
        auto n = [&](float len) { return std::max(1, (int)lroundf(len / cell)); };
        const int nx = n(d.x), ny = n(d.y), nz = n(d.z);
        quad({lo.x, lo.y, hi.z}, {d.x, 0, 0}, {0, d.y, 0}, nx, ny, m);      // +z
        quad({hi.x, lo.y, lo.z}, {-d.x, 0, 0}, {0, d.y, 0}, nx, ny, m);     // -z
        quad({hi.x, lo.y, hi.z}, {0, 0, -d.z}, {0, d.y, 0}, nz, ny, m);     // +x
        quad({lo.x, lo.y, lo.z}, {0, 0, d.z}, {0, d.y, 0}, nz, ny, m);      // -x
        quad({lo.x, hi.y, hi.z}, {d.x, 0, 0}, {0, 0, -d.z}, nx, nz, m);     // +y
        quad({lo.x, lo.y, lo.z}, {d.x, 0, 0}, {0, 0, d.z}, nx, nz, m);      // -y
    }
    void cylinder(V3 base, float radius, float height, int seg, int rings, int m) {
        const float two_pi = 6.28318530718f;
        for (int j = 0; j < rings; ++j)
            for (int i = 0; i < seg; ++i) {
                const float a0 = two_pi * i / seg, a1 = two_pi * (i + 1) / seg;
                const float y0 = height * j / rings, y1 = height * (j + 1) / rings;
                const V3 n0 = {cosf(a0), 0, sinf(a0)}, n1 = {cosf(a1), 0, sinf(a1)};
                const V3 p00 = base + n0 * radius + V3{0, y0, 0}, p10 = base + n1 * radius + V3{0, y0, 0};
                const V3 p01 = base + n0 * radius + V3{0, y1, 0}, p11 = base + n1 * radius + V3{0, y1, 0};
                tri(p00, p01, p11, n0, n0, n1, m);     // CCW seen from outside
                tri(p00, p11, p10, n0, n1, n1, m);
            }
    }
    void sphere(V3 c, float r, int seg, int rings, int m) {
        const float pi = 3.14159265359f;
        auto P = [&](int i, int j) {
            const float th = pi * j / rings, ph = 2 * pi * i / seg;
            return V3{sinf(th) * cosf(ph), cosf(th), sinf(th) * sinf(ph)};
        };
        for (int j = 0; j < rings; ++j)
            for (int i = 0; i < seg; ++i) {
                const V3 a = P(i, j), b = P(i + 1, j), cc = P(i + 1, j + 1), d = P(i, j + 1);
                if (j > 0) tri(c + a * r, c + b * r, c + cc * r, a, b, cc, m);
                if (j < rings - 1) tri(c + a * r, c + cc * r, c + d * r, a, cc, d, m);
            }
    }
};

int add_material(vcth_scene* s, float r, float g, float b, float sr, float sg, float sb) {
    Material m = {{r, g, b, 1.0f}, {sr, sg, sb}};
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}

// All builders work in WORLD units and scale to model space (world / 0.05) at the end.
void finish(vcth_scene* s) {
    for (auto& v : s->pos) v *= 20.0f;
}

void build_cornell(vcth_scene* s) {
    Builder b{s};
    const int white = add_material(s, 0.73f, 0.73f, 0.73f, 0.2f, 0.2f, 0.2f);
    const int red = add_material(s, 0.65f, 0.05f, 0.05f, 0.1f, 0.1f, 0.1f);
    const int green = add_material(s, 0.12f, 0.45f, 0.15f, 0.1f, 0.1f, 0.1f);
    const int block = add_material(s, 0.6f, 0.6f, 0.75f, 0.6f, 0.6f, 0.6f);
    const float L = 60.0f;   // half extent; box spans [-60,60] x [-60,60] x [-60,60] (inside +-67)
    b.quad({-L, -L, L}, {2 * L, 0, 0}, {0, 0, -2 * L}, 1, 1, white);        // floor (+y)
    b.quad({-L, -L, -L}, {2 * L, 0, 0}, {0, 2 * L, 0}, 1, 1, white);        // back wall (+z)
    b.quad({-L, -L, L}, {0, 0, -2 * L}, {0, 2 * L, 0}, 1, 1, red);          // left wall (+x)
    b.quad({L, -L, -L}, {0, 0, 2 * L}, {0, 2 * L, 0}, 1, 1, green);         // right wall (-x)
    // ceiling with a central opening so that lightDirection (0,1,0.25) reaches the floor
    const float h = 30.0f;
    b.quad({-L, L, -L}, {2 * L, 0, 0}, {0, 0, L - h}, 1, 1, white);         // faces -y
    b.quad({-L, L, h}, {2 * L, 0, 0}, {0, 0, L - h}, 1, 1, white);
    b.quad({-L, L, -h}, {L - h, 0, 0}, {0, 0, 2 * h}, 1, 1, white);
    b.quad({h, L, -h}, {L - h, 0, 0}, {0, 0, 2 * h}, 1, 1, white);
    b.box({-35, -L, -30}, {-5, -L + 70, 0}, 1000.0f, block);                // tall block
    b.box({8, -L, 5}, {40, -L + 35, 37}, 1000.0f, block);                   // short block
    finish(s);
}

void build_atrium(vcth_scene* s, float detail, uint32_t seed) {
    Builder b{s};
    uint32_t rng = seed * 2654435761u + 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 16777216.0f; };
    const int stone = add_material(s, 0.62f, 0.58f, 0.50f, 0.15f, 0.15f, 0.15f);
    const int floorm = add_material(s, 0.45f, 0.42f, 0.38f, 0.35f, 0.35f, 0.35f);
    const int brick = add_material(s, 0.55f, 0.33f, 0.25f, 0.1f, 0.1f, 0.1f);
    const int redc = add_material(s, 0.75f, 0.08f, 0.06f, 0.05f, 0.05f, 0.05f);
    const int greenc = add_material(s, 0.10f, 0.55f, 0.14f, 0.05f, 0.05f, 0.05f);
    const int bluec = add_material(s, 0.10f, 0.18f, 0.70f, 0.05f, 0.05f, 0.05f);
    const int bronze = add_material(s, 0.55f, 0.40f, 0.20f, 0.8f, 0.6f, 0.3f);
    const int curtains[3] = {redc, greenc, bluec};

    // hall: x in [-64,64] (long axis), z in [-28,28], floor y = -20, roof line y = 34
    const float X = 64.0f, Z = 28.0f, Y0 = -20.0f, Y1 = 34.0f, aisle = 13.0f, gallery = 6.0f;
    const float cell = 0.79f / std::max(detail, 0.05f);   // tessellation cell in world units (detail 1 ~ 262k tris)
    auto n = [&](float len) { return std::max(1, (int)lroundf(len / cell)); };

    b.quad({-X, Y0, Z}, {2 * X, 0, 0}, {0, 0, -2 * Z}, n(2 * X), n(2 * Z), floorm);            // floor
    b.quad({-X, Y0, -Z}, {2 * X, 0, 0}, {0, Y1 - Y0, 0}, n(2 * X), n(Y1 - Y0), brick);          // -z wall
    b.quad({X, Y0, Z}, {-2 * X, 0, 0}, {0, Y1 - Y0, 0}, n(2 * X), n(Y1 - Y0), brick);           // +z wall
    b.quad({-X, Y0, Z}, {0, 0, -2 * Z}, {0, Y1 - Y0, 0}, n(2 * Z), n(Y1 - Y0), stone);          // -x end
    b.quad({X, Y0, -Z}, {0, 0, 2 * Z}, {0, Y1 - Y0, 0}, n(2 * Z), n(Y1 - Y0), stone);           // +x end
    // roofs over the two side aisles (the nave z in [-aisle,aisle] is open to the sky)
    b.quad({-X, Y1, -Z}, {2 * X, 0, 0}, {0, 0, Z - aisle}, n(2 * X), n(Z - aisle), stone);
    b.quad({-X, Y1, aisle}, {2 * X, 0, 0}, {0, 0, Z - aisle}, n(2 * X), n(Z - aisle), stone);
    // gallery slabs
    b.box({-X, gallery - 1.5f, -Z}, {X, gallery, -aisle}, cell, stone);
    b.box({-X, gallery - 1.5f, aisle}, {X, gallery, Z}, cell, stone);

    // colonnades: two storeys of columns along both sides of the nave, lintels on top
    const int ncol = 11;
    const int seg = std::max(8, (int)lroundf(20 * sqrtf(detail)));
    for (int side = -1; side <= 1; side += 2)
        for (int i = 0; i < ncol; ++i) {
            const float x = -X + 8.0f + (2 * X - 16.0f) * i / (ncol - 1);
            const float z = side * aisle;
            const float r = 1.6f;
            b.box({x - 2.2f, Y0, z - 2.2f}, {x + 2.2f, Y0 + 1.5f, z + 2.2f}, cell, stone);       // plinth
            b.cylinder({x, Y0 + 1.5f, z}, r, gallery - 1.5f - (Y0 + 1.5f), seg, n(gallery - Y0), stone);
            b.cylinder({x, gallery, z}, r * 0.8f, Y1 - 3.0f - gallery, seg, n(Y1 - gallery), stone);
            if (i + 1 < ncol) {
                const float x2 = -X + 8.0f + (2 * X - 16.0f) * (i + 1) / (ncol - 1);
                b.box({x, Y1 - 3.0f, z - 1.5f}, {x2, Y1, z + 1.5f}, cell, stone);                // lintel
                // hanging cloth between upper columns
                if ((i % 2) == 0) {
                    const int cm = curtains[(i / 2 + (side > 0)) % 3];
                    const float ph = rnd() * 6.28f;
                    const V3 o = {x + 1.5f, gallery + 1.0f, z + side * 0.2f};
                    const V3 du = {x2 - x - 3.0f, 0, 0};
                    const V3 dv = {0, Y1 - 6.0f - gallery, 0};
                    auto wave = [ph](float u, float v) { return 0.6f * sinf(u * 18.0f + ph) * (0.3f + v); };
                    // cloth is two-sided: one patch per facing (back faces are culled in the draws)
                    auto wave_back = [ph](float u, float v) { return -0.6f * sinf((1.0f - u) * 18.0f + ph) * (0.3f + v); };
                    b.patch(o, du, dv, n(du.x) * 2, n(dv.y) * 2, cm, wave);
                    b.patch(o + du, du * -1.0f, dv, n(du.x) * 2, n(dv.y) * 2, cm, wave_back);
                }
            }
        }
    // objects in the nave
    b.sphere({-20.0f, Y0 + 6.0f, 0.0f}, 6.0f, seg * 3, seg * 2, bronze);
    b.box({10, Y0, -6}, {22, Y0 + 9, 6}, cell, stone);
    b.sphere({16.0f, Y0 + 13.0f, 0.0f}, 4.0f, seg * 2, seg, bronze);
    for (int i = 0; i < 6; ++i) {
        const float x = -50.0f + 20.0f * i + 4.0f * rnd(), z = (rnd() - 0.5f) * 14.0f;
        const float hgt = 2.0f + 5.0f * rnd();
        b.box({x, Y0, z}, {x + 3.0f + 3.0f * rnd(), Y0 + hgt, z + 3.0f + 2.0f * rnd()}, cell,
              (i & 1) ? brick : stone);
    }
    finish(s);
}

// ---- rasteriser ------------------------------------------------------------------------

constexpr int kMaxVar = 12;
struct RVert { float c[4]; float var[kMaxVar]; };

inline RVert lerp_vert(const RVert& a, const RVert& b, float t, int nvar) {
    RVert r;
    for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + (b.c[i] - a.c[i]) * t;
    for (int i = 0; i < nvar; ++i) r.var[i] = a.var[i] + (b.var[i] - a.var[i]) * t;
    return r;
}

// Rasterises one clip-space triangle (GL rules: near-plane clip, pixel-centre sampling, top-left
// fill rule, optional back-face cull with CCW front faces, depth test LESS).  frag(x, y, z, var)
// is called for every fragment that passes the depth test.
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
        // window coordinates snapped to 1/256 pixel (GL sub-pixel precision) and edge functions in
        // double: with snapped inputs every product is exact, so shared edges are watertight.
        double sx[3], sy[3];
        float sz[3], iw[3];
        bool bad = false;
        for (int k = 0; k < 3; ++k) {
            if (!(v[k]->c[3] > 1e-20f)) { bad = true; break; }
            iw[k] = 1.0f / v[k]->c[3];
            sx[k] = floor((double)((v[k]->c[0] * iw[k] * 0.5f + 0.5f) * (float)W) * 256.0 + 0.5) / 256.0;
            sy[k] = floor((double)((v[k]->c[1] * iw[k] * 0.5f + 0.5f) * (float)H) * 256.0 + 0.5) / 256.0;
            sz[k] = v[k]->c[2] * iw[k] * 0.5f + 0.5f;
        }
        if (bad) continue;
        double area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
        if (area == 0.0 || area != area) continue;
        if (area < 0.0 && cull_back) continue;
        const double sgn = area > 0.0 ? 1.0 : -1.0;
        area *= sgn;
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
                const float b0 = (float)(e[0] / area), b1 = (float)(e[1] / area), b2 = 1.0f - b0 - b1;
                const float z = b0 * sz[0] + b1 * sz[1] + b2 * sz[2];
                if (!(z >= 0.0f && z <= 1.0f)) continue;               // far-plane clip
                float& zb = zbuf[(size_t)py * W + px];
                if (!(z < zb)) continue;                                 // GL_LESS
                zb = z;
                const float q0 = b0 * iw[0], q1 = b1 * iw[1], q2 = b2 * iw[2];
                const float qs = 1.0f / (q0 + q1 + q2);
                float var[kMaxVar];
                for (int i = 0; i < nvar; ++i)
                    var[i] = (q0 * v[0]->var[i] + q1 * v[1]->var[i] + q2 * v[2]->var[i]) * qs;
                frag(px, py, z, var);
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

}  // namespace

extern "C" {

void vcth_default_camera(vcth_camera* cam) {
    cam->position[0] = 0.0f; cam->position[1] = 4.0f; cam->position[2] = 0.0f;   // VCT.h:8
    cam->yaw = -90.0f;      // Camera.h:21
    cam->pitch = 0.0f;      // Camera.h:22
    cam->zoom = 45.0f;      // Camera.h:47
    cam->z_near = 0.1f;     // VCT.h:162
    cam->z_far = 1000.0f;
}

vcth_scene* vcth_scene_create(int kind, float detail, uint32_t seed) {
    vcth_scene* s = new vcth_scene();
    if (kind == 0) build_cornell(s);
    else if (kind == 1) build_atrium(s, detail, seed);
    else { delete s; return nullptr; }
    return s;
}

vcth_scene* vcth_scene_load_obj(const char* path, char* error) {
    auto failmsg = [&](const std::string& m) -> vcth_scene* {
        if (error) snprintf(error, 256, "%s", m.c_str());
        return nullptr;
    };
    if (!path) return failmsg("null path");
    FILE* fp = fopen(path, "r");
    if (!fp) return failmsg(std::string("cannot open ") + path);
    const std::string dir = std::string(path).find_last_of('/') == std::string::npos
                                ? std::string() : std::string(path).substr(0, std::string(path).find_last_of('/') + 1);
    std::vector<V3> vp, vn;
    std::vector<float> vt;                       // u, v pairs
    struct Corner { int p, t, n; };
    std::vector<Corner> corners;                 // 3 per triangle
    std::vector<int32_t> tri_mat;
    std::vector<Material> mats;
    std::map<std::string, int> mat_index;
    auto material = [&](const std::string& name) {
        auto it = mat_index.find(name);
        if (it != mat_index.end()) return it->second;
        Material m = {{0.7f, 0.7f, 0.7f, 1.0f}, {0.2f, 0.2f, 0.2f}};
        mats.push_back(m);
        return mat_index[name] = (int)mats.size() - 1;
    };
    auto load_mtl = [&](const std::string& file) {
        FILE* mf = fopen((dir + file).c_str(), "r");
        if (!mf) return;                          // like assimp: a missing .mtl leaves default materials
        char line[1024];
        int cur = -1;
        while (fgets(line, sizeof(line), mf)) {
            char name[512];
            float a, b, c;
            if (sscanf(line, " newmtl %511s", name) == 1) cur = material(name);
            else if (cur >= 0 && sscanf(line, " Kd %f %f %f", &a, &b, &c) == 3) {
                mats[(size_t)cur].albedo[0] = a; mats[(size_t)cur].albedo[1] = b; mats[(size_t)cur].albedo[2] = c;
            } else if (cur >= 0 && sscanf(line, " Ks %f %f %f", &a, &b, &c) == 3) {
                mats[(size_t)cur].spec[0] = a; mats[(size_t)cur].spec[1] = b; mats[(size_t)cur].spec[2] = c;
            } else if (cur >= 0 && sscanf(line, " d %f", &a) == 1) mats[(size_t)cur].albedo[3] = a;
        }
        fclose(mf);
    };
    int cur_mat = -1;
    char line[4096];
    while (fgets(line, sizeof(line), fp)) {
        float a, b, c;
        char name[512];
        if (line[0] == 'v' && line[1] == ' ' && sscanf(line + 2, "%f %f %f", &a, &b, &c) == 3) vp.push_back({a, b, c});
        else if (line[0] == 'v' && line[1] == 'n' && sscanf(line + 3, "%f %f %f", &a, &b, &c) == 3) vn.push_back({a, b, c});
        else if (line[0] == 'v' && line[1] == 't' && sscanf(line + 3, "%f %f", &a, &b) >= 1) { vt.push_back(a); vt.push_back(b); }
        else if (sscanf(line, " usemtl %511s", name) == 1) cur_mat = material(name);
        else if (sscanf(line, " mtllib %511s", name) == 1) load_mtl(name);
        else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
            std::vector<Corner> poly;
            const char* q = line + 2;
            while (*q) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == '\0' || *q == '\n' || *q == '\r') break;
                Corner cn = {0, 0, 0};
                char* end;
                cn.p = (int)strtol(q, &end, 10);
                if (end == q) break;
                q = end;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') { cn.t = (int)strtol(q, &end, 10); q = end; }
                    if (*q == '/') { ++q; cn.n = (int)strtol(q, &end, 10); q = end; }
                }
                auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
                cn.p = fix(cn.p, vp.size()); cn.t = fix(cn.t, vt.size() / 2); cn.n = fix(cn.n, vn.size());
                if (cn.p < 0 || cn.p >= (int)vp.size()) { fclose(fp); return failmsg("face references a missing vertex"); }
                if (cn.t >= (int)(vt.size() / 2)) cn.t = -1;
                if (cn.n >= (int)vn.size()) cn.n = -1;
                poly.push_back(cn);
            }
            if (cur_mat < 0) cur_mat = material("(default)");
            for (size_t k = 1; k + 1 < poly.size(); ++k) {          // aiProcess_Triangulate: fan
                corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]);
                tri_mat.push_back(cur_mat);
            }
        }
    }
    fclose(fp);
    if (corners.empty()) return failmsg("no faces in the file");
    // aiProcess_GenSmoothNormals for corners without a normal: area-weighted per position
    std::vector<V3> smooth(vp.size(), V3{0, 0, 0});
    for (size_t t = 0; t < tri_mat.size(); ++t) {
        const V3 a = vp[(size_t)corners[3 * t].p], b = vp[(size_t)corners[3 * t + 1].p], c = vp[(size_t)corners[3 * t + 2].p];
        const V3 fn = cross(b - a, c - a);
        for (int k = 0; k < 3; ++k) smooth[(size_t)corners[3 * t + k].p] = smooth[(size_t)corners[3 * t + k].p] + fn;
    }
    vcth_scene* s = new vcth_scene();
    s->materials = mats;
    s->mat = tri_mat;
    for (size_t t = 0; t < tri_mat.size(); ++t) {
        const Corner* c = &corners[3 * t];
        const V3 p0 = vp[(size_t)c[0].p], p1 = vp[(size_t)c[1].p], p2 = vp[(size_t)c[2].p];
        // aiProcess_CalcTangentSpace: per-triangle tangent from the UV gradient when UVs exist
        V3 tri_tan = {0, 0, 0};
        if (c[0].t >= 0 && c[1].t >= 0 && c[2].t >= 0) {
            const float du1 = vt[2 * (size_t)c[1].t] - vt[2 * (size_t)c[0].t], dv1 = vt[2 * (size_t)c[1].t + 1] - vt[2 * (size_t)c[0].t + 1];
            const float du2 = vt[2 * (size_t)c[2].t] - vt[2 * (size_t)c[0].t], dv2 = vt[2 * (size_t)c[2].t + 1] - vt[2 * (size_t)c[0].t + 1];
            const float det = du1 * dv2 - du2 * dv1;
            if (fabsf(det) > 1e-20f) tri_tan = ((p1 - p0) * dv2 - (p2 - p0) * dv1) * (1.0f / det);
        }
        for (int k = 0; k < 3; ++k) {
            const V3 pos = vp[(size_t)c[k].p];
            V3 n = c[k].n >= 0 ? vn[(size_t)c[k].n] : smooth[(size_t)c[k].p];
            n = normalize(n);
            if (dot(n, n) == 0.0f) n = normalize(cross(p1 - p0, p2 - p0));
            V3 tg = tri_tan - n * dot(n, tri_tan);                   // Gram-Schmidt against the normal
            if (dot(tg, tg) < 1e-20f) {
                const V3 hint = fabsf(n.y) < 0.9f ? V3{0, 1, 0} : V3{1, 0, 0};
                tg = cross(hint, n);
            }
            tg = normalize(tg);
            const V3 bt = cross(n, tg);
            s->pos.insert(s->pos.end(), {pos.x, pos.y, pos.z});
            s->nrm.insert(s->nrm.end(), {n.x, n.y, n.z});
            s->tan.insert(s->tan.end(), {tg.x, tg.y, tg.z});
            s->bit.insert(s->bit.end(), {bt.x, bt.y, bt.z});
        }
    }
    return s;
}

void vcth_scene_destroy(vcth_scene* s) { delete s; }
int32_t vcth_scene_num_triangles(const vcth_scene* s) { return s ? (int32_t)s->mat.size() : 0; }
int32_t vcth_scene_num_materials(const vcth_scene* s) { return s ? (int32_t)s->materials.size() : 0; }

void vcth_scene_get(const vcth_scene* s, float* pos, int32_t* material, float* albedo, float* specular) {
    if (pos) memcpy(pos, s->pos.data(), s->pos.size() * sizeof(float));
    if (material) memcpy(material, s->mat.data(), s->mat.size() * sizeof(int32_t));
    for (size_t i = 0; i < s->materials.size(); ++i) {
        if (albedo) memcpy(albedo + 4 * i, s->materials[i].albedo, 16);
        if (specular) memcpy(specular + 3 * i, s->materials[i].spec, 12);
    }
}

void vcth_scene_get_frames(const vcth_scene* s, float* normal, float* tangent, float* bitangent) {
    if (normal) memcpy(normal, s->nrm.data(), s->nrm.size() * sizeof(float));
    if (tangent) memcpy(tangent, s->tan.data(), s->tan.size() * sizeof(float));
    if (bitangent) memcpy(bitangent, s->bit.data(), s->bit.size() * sizeof(float));
}

static M4 camera_vp(const vcth_camera* cam, int32_t W, int32_t H) {
    const float deg = 3.14159265358979f / 180.0f;
    const V3 pos = {cam->position[0], cam->position[1], cam->position[2]};
    const V3 front = normalize(V3{cosf(cam->yaw * deg) * cosf(cam->pitch * deg), sinf(cam->pitch * deg),
                                  sinf(cam->yaw * deg) * cosf(cam->pitch * deg)});   // Camera.h:136-143
    const V3 right = normalize(cross(front, V3{0, 1, 0}));
    const V3 up = normalize(cross(right, front));
    const M4 view = look_at(pos, pos + front, up);                                   // Camera.h:77
    const M4 proj = perspective(cam->zoom * deg, (float)W / (float)H, cam->z_near, cam->z_far);   // VCT.h:162
    return mul(proj, view);
}

void vcth_camera_view_proj(const vcth_camera* cam, int32_t width, int32_t height, float out_vp[16]) {
    const M4 vp = camera_vp(cam, width, height);
    memcpy(out_vp, vp.m, 64);
}

void vcth_light_view_proj(const float L[3], float out_vp[16]) {
    const M4 v = look_at({L[0], L[1], L[2]}, {0, 0, 0}, {0, 1, 0});     // VCT.h:84
    const M4 p = ortho(-120, 120, -120, 120, -100, 100);                // VCT.h:85
    const M4 vp = mul(p, v);                                            // VCT.h:86
    memcpy(out_vp, vp.m, 64);
}

void vcth_render_shadow_map(const vcth_scene* s, float model_scale, const float light_vp[16],
                            int32_t S, float* depth) {
    const size_t n = (size_t)S * S;
    for (size_t i = 0; i < n; ++i) depth[i] = 1.0f;                      // glClear depth
    const size_t ntri = s->mat.size();
    for (size_t t = 0; t < ntri; ++t) {
        RVert v[3];
        for (int k = 0; k < 3; ++k) {
            const float* p = &s->pos[t * 9 + 3 * k];
            xform(light_vp, V3{p[0] * model_scale, p[1] * model_scale, p[2] * model_scale}, v[k].c);
        }
        raster_triangle(v, 0, true, S, S, depth, [](int, int, float, const float*) {});
    }
    const float q = 16777215.0f;                                         // DEPTH_COMPONENT24
    for (size_t i = 0; i < n; ++i) depth[i] = (float)(floor((double)depth[i] * q + 0.5) / q);
}

void vcth_render_gbuffer(const vcth_scene* s, float model_scale, const vcth_camera* cam, int32_t W,
                         int32_t H, const float* shadow_depth, int32_t shadow_size,
                         const float light_vp[16], float* planes) {
    const size_t npix = (size_t)W * H;
    memset(planes, 0, npix * 23 * sizeof(float));
    std::vector<float> zbuf(npix, 1.0f);
    std::vector<int32_t> mat(npix, -1);
    const M4 vp = camera_vp(cam, W, H);
    const size_t ntri = s->mat.size();
    for (size_t t = 0; t < ntri; ++t) {
        RVert v[3];
        for (int k = 0; k < 3; ++k) {
            const float* p = &s->pos[t * 9 + 3 * k];
            const V3 w = {p[0] * model_scale, p[1] * model_scale, p[2] * model_scale};   // trace.vs:27
            xform(vp.m, w, v[k].c);                                                      // trace.vs:25
            float* o = v[k].var;
            o[0] = w.x; o[1] = w.y; o[2] = w.z;
            for (int i = 0; i < 3; ++i) {
                o[3 + i] = s->nrm[t * 9 + 3 * k + i] * model_scale;    // trace.vs:31 (w = 0)
                o[6 + i] = s->tan[t * 9 + 3 * k + i] * model_scale;    // trace.vs:32
                o[9 + i] = s->bit[t * 9 + 3 * k + i] * model_scale;    // trace.vs:33
            }
        }
        const int32_t m = s->mat[t];
        raster_triangle(v, 12, true, W, H, zbuf.data(), [&](int x, int y, float, const float* var) {
            const size_t i = (size_t)y * W + x;
            for (int k = 0; k < 12; ++k) planes[(size_t)k * npix + i] = var[k];
            mat[i] = m;
        });
    }
    // per-pixel material + bump normal + shadow term (the non-cone part of trace.fs)
    for (size_t i = 0; i < npix; ++i) {
        if (mat[i] < 0) continue;
        auto G = [&](int k) -> float& { return planes[(size_t)k * npix + i]; };
        const V3 P = {G(0), G(1), G(2)}, N = {G(3), G(4), G(5)}, T = {G(6), G(7), G(8)}, B = {G(9), G(10), G(11)};
        // CalcBumpNormal with a flat height map: normalize(TBN * (0,0,1)), TBN = inverse(transpose(M))
        const V3 c2 = cross(T, B);
        const float det = dot(T, cross(B, N));
        const V3 bn = normalize(c2 * (1.0f / det));                                   // trace.fs:127,175
        G(12) = bn.x; G(13) = bn.y; G(14) = bn.z;
        const Material& mm = s->materials[(size_t)mat[i]];
        for (int k = 0; k < 4; ++k) G(15 + k) = mm.albedo[k];                          // trace.fs:167
        const bool has_gb = sqrtf(mm.spec[1] * mm.spec[1] + mm.spec[2] * mm.spec[2]) > 0.0f;
        G(19) = mm.spec[0];
        G(20) = has_gb ? mm.spec[1] : mm.spec[0];                                      // trace.fs:210
        G(21) = has_gb ? mm.spec[2] : mm.spec[0];
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
