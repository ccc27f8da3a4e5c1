// vct_host.cpp -- scenes (procedural + OBJ/MTL reader with textures), camera and light matrices (see vct_host.h).
#include "vct_host.h"
#include "vct_image.h"

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

struct Material { float albedo[4]; float spec[3]; int tex[3] = {-1, -1, -1}; };   // tex: diffuse, specular, height
struct Texture { int w = 0, h = 0; std::vector<uint8_t> rgba; };

}  // namespace

struct vcth_scene {
    std::vector<float> pos, nrm, tan, bit;   // per triangle-vertex, 3 floats each
    std::vector<float> uv;                   // per triangle-vertex, 2 floats
    std::vector<int32_t> mat;                // per triangle
    std::vector<Material> materials;
    std::vector<Texture> textures;
};

namespace {

struct UV { float u, v; };
constexpr float kTexTile = 8.0f;     // world units per texture repeat of the procedural scenes

struct Builder {
    vcth_scene* s;
    void tri(V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc, int m, UV ta = {0, 0}, UV tb = {0, 0}, UV tc = {0, 0}) {
        const V3 p[3] = {a, b, c}, n[3] = {na, nb, nc};
        const UV tx[3] = {ta, tb, tc};
        for (int k = 0; k < 3; ++k) {
            s->uv.insert(s->uv.end(), {tx[k].u, tx[k].v});
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
        const float lu = sqrtf(dot(du, du)) / kTexTile, lv = sqrtf(dot(dv, dv)) / kTexTile;
        auto T = [&](int i, int j) { return UV{lu * (float)i / nu, lv * (float)j / nv}; };
        for (int j = 0; j < nv; ++j)
            for (int i = 0; i < nu; ++i) {
                const V3 a = P(i, j), b = P(i + 1, j), c = P(i + 1, j + 1), d = P(i, j + 1);
                tri(a, b, c, Nn(i, j), Nn(i + 1, j), Nn(i + 1, j + 1), m, T(i, j), T(i + 1, j), T(i + 1, j + 1));
                tri(a, c, d, Nn(i, j), Nn(i + 1, j + 1), Nn(i, j + 1), m, T(i, j), T(i + 1, j + 1), T(i, j + 1));
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
                const float u0 = a0 * radius / kTexTile, u1 = a1 * radius / kTexTile;
                const float v0 = y0 / kTexTile, v1 = y1 / kTexTile;
                tri(p00, p01, p11, n0, n0, n1, m, {u0, v0}, {u0, v1}, {u1, v1});     // CCW seen from outside
                tri(p00, p11, p10, n0, n1, n1, m, {u0, v0}, {u1, v1}, {u1, v0});
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
                auto T = [&](int ii, int jj) { return UV{4.0f * ii / seg, 2.0f * jj / rings}; };
                if (j > 0) tri(c + a * r, c + b * r, c + cc * r, a, b, cc, m, T(i, j), T(i + 1, j), T(i + 1, j + 1));
                if (j < rings - 1) tri(c + a * r, c + cc * r, c + d * r, a, cc, d, m, T(i, j), T(i + 1, j + 1), T(i, j + 1));
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

// ---- procedural textures (the reference loads image files with stb_image, R/Model.h:141-226; there are no
// assets and no network here, so the textured scene generates its maps: same data path from there on) ----
inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
inline float hash01(int x, int y, uint32_t seed) {
    return (float)(hash32((uint32_t)x * 374761393u + (uint32_t)y * 668265263u + seed * 2246822519u) >> 8) / 16777216.0f;
}
// tileable value noise in [0,1]: `cells` lattice cells across the texture
float value_noise(float u, float v, int cells, uint32_t seed) {
    const float x = u * cells, y = v * cells;
    const int i = (int)floorf(x), j = (int)floorf(y);
    float a = x - i, b = y - j;
    a = a * a * (3 - 2 * a); b = b * b * (3 - 2 * b);
    auto L = [&](int ii, int jj) { return hash01(((ii % cells) + cells) % cells, ((jj % cells) + cells) % cells, seed); };
    return (L(i, j) * (1 - a) + L(i + 1, j) * a) * (1 - b) + (L(i, j + 1) * (1 - a) + L(i + 1, j + 1) * a) * b;
}
template <class F>
int add_texture(vcth_scene* s, int w, int h, F texel) {     // texel(u, v, x, y, out rgba 0..1)
    Texture t;
    t.w = w; t.h = h;
    t.rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float c[4] = {0, 0, 0, 1};
            texel(((float)x + 0.5f) / w, ((float)y + 0.5f) / h, x, y, c);
            for (int k = 0; k < 4; ++k) {
                const float f = c[k] < 0.0f ? 0.0f : (c[k] > 1.0f ? 1.0f : c[k]);
                t.rgba[((size_t)y * w + x) * 4 + k] = (uint8_t)(f * 255.0f + 0.5f);
            }
        }
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}

// material textures of the textured atrium (kind 2): checker floor with a red-only specular map (the .rrra
// rule of trace.fs:210), brick walls, stone with a noisy height map, bronze with a coloured specular map,
// cloth with alpha cut-outs (the alpha test of trace.fs:169-172)
void texture_atrium(vcth_scene* s, int stone, int floorm, int brick, const int curtains[3], int bronze, uint32_t seed) {
    const int checker = add_texture(s, 256, 256, [&](float u, float v, int x, int y, float* c) {
        const bool k = ((x / 32) + (y / 32)) & 1;
        const float n = 0.08f * (value_noise(u, v, 32, seed + 1) - 0.5f);
        const float base = k ? 0.62f : 0.30f;
        c[0] = base + n; c[1] = base * 0.95f + n; c[2] = base * 0.85f + n;
    });
    const int floor_spec = add_texture(s, 64, 64, [&](float u, float v, int, int, float* c) {
        c[0] = 0.25f + 0.3f * value_noise(u, v, 8, seed + 2); c[1] = 0.0f; c[2] = 0.0f;     // r only -> .rrra
    });
    const int bricks = add_texture(s, 256, 128, [&](float u, float v, int x, int y, float* c) {
        const int row = y / 16, xo = x + ((row & 1) ? 16 : 0);
        const bool mortar = (y % 16) < 2 || (xo % 32) < 2;
        const float n = 0.12f * (value_noise(u, v, 64, seed + 3) - 0.5f);
        const float tone = 0.85f + 0.3f * (hash01(xo / 32, row, seed + 4) - 0.5f);
        if (mortar) { c[0] = 0.55f + n; c[1] = 0.53f + n; c[2] = 0.50f + n; }
        else { c[0] = 0.55f * tone + n; c[1] = 0.30f * tone + n; c[2] = 0.22f * tone + n; }
    });
    const int brick_h = add_texture(s, 256, 128, [&](float u, float v, int x, int y, float* c) {
        const int row = y / 16, xo = x + ((row & 1) ? 16 : 0);
        const bool mortar = (y % 16) < 2 || (xo % 32) < 2;
        const float hgt = mortar ? 0.15f : 0.7f + 0.2f * value_noise(u, v, 64, seed + 5);
        c[0] = c[1] = c[2] = hgt;
    });
    const int stone_d = add_texture(s, 256, 256, [&](float u, float v, int, int, float* c) {
        const float n = 0.55f * value_noise(u, v, 16, seed + 6) + 0.45f * value_noise(u, v, 64, seed + 7);
        c[0] = 0.50f + 0.25f * n; c[1] = 0.47f + 0.23f * n; c[2] = 0.40f + 0.20f * n;
    });
    const int stone_h = add_texture(s, 256, 256, [&](float u, float v, int, int, float* c) {
        const float n = 0.6f * value_noise(u, v, 16, seed + 8) + 0.4f * value_noise(u, v, 96, seed + 9);
        c[0] = c[1] = c[2] = n;
    });
    const int bronze_s = add_texture(s, 64, 64, [&](float u, float v, int, int, float* c) {
        const float n = 0.7f + 0.3f * value_noise(u, v, 8, seed + 10);
        c[0] = 0.8f * n; c[1] = 0.6f * n; c[2] = 0.3f * n;
    });
    int cloth[3];
    const float tint[3][3] = {{0.75f, 0.08f, 0.06f}, {0.10f, 0.55f, 0.14f}, {0.10f, 0.18f, 0.70f}};
    for (int k = 0; k < 3; ++k)
        cloth[k] = add_texture(s, 128, 128, [&](float u, float v, int x, int y, float* c) {
            const float n = 0.85f + 0.3f * (value_noise(u, v, 32, seed + 11 + k) - 0.5f);
            const int dx = (x % 32) - 16, dy = (y % 32) - 16;
            c[0] = tint[k][0] * n; c[1] = tint[k][1] * n; c[2] = tint[k][2] * n;
            c[3] = dx * dx + dy * dy < 36 ? 0.0f : 1.0f;                 // lace holes: alpha test
        });
    auto set = [&](int m, int d, int sp, int h) { s->materials[(size_t)m].tex[0] = d; s->materials[(size_t)m].tex[1] = sp; s->materials[(size_t)m].tex[2] = h; };
    set(floorm, checker, floor_spec, stone_h);
    set(brick, bricks, -1, brick_h);
    set(stone, stone_d, -1, stone_h);
    set(bronze, -1, bronze_s, -1);
    for (int k = 0; k < 3; ++k) set(curtains[k], cloth[k], -1, -1);
}

void build_atrium(vcth_scene* s, float detail, uint32_t seed, bool textured) {
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
    if (textured) texture_atrium(s, stone, floorm, brick, curtains, bronze, seed);
    finish(s);
}

// ---- kind 3: Bistro-exterior-class street scene (BASELINE.json configs[4] names Amazon Bistro exterior; no asset can
// be fetched here, so its CHARACTER is generated: ~2.8 M triangles at detail 1, about 40 % of them alpha-tested
// foliage cards with incoherent normals, facades with recessed windows, balconies and awnings, street clutter --
// tables, chairs, lamps, planters, strings of lights -- and every surface textured, with noisy height maps).
// A street runs along x between two rows of buildings; the camera stands on it.
void texture_bistro(vcth_scene* s, const int m[], uint32_t seed) {
    enum { COBBLE, PLASTER, BRICKM, WOOD, METAL, LEAF, AWNING, GLASS, ROOF, SIDEWALK, LAMP, NMAT };
    const int cobble_d = add_texture(s, 512, 512, [&](float u, float v, int x, int y, float* c) {
        const int cx = x / 32, cy = y / 32, ox = (cy & 1) ? 16 : 0;
        const int lx = (x + ox) % 32, ly = y % 32;
        const bool gap = lx < 3 || ly < 3;
        const float tone = 0.75f + 0.5f * (hash01((x + ox) / 32, cy, seed + 31) - 0.5f);
        const float n = 0.18f * (value_noise(u, v, 128, seed + 32) - 0.5f);
        const float g = gap ? 0.18f : 0.42f * tone;
        c[0] = g + n; c[1] = g * 0.97f + n; c[2] = g * 0.92f + n;
        (void)cx;
    });
    const int cobble_h = add_texture(s, 512, 512, [&](float u, float v, int x, int y, float* c) {
        const int cy = y / 32, ox = (cy & 1) ? 16 : 0;
        const int lx = (x + ox) % 32, ly = y % 32;
        const float ex = fminf((float)lx, 31.0f - lx), ey = fminf((float)ly, 31.0f - ly);
        const float dome = fminf(1.0f, fminf(ex, ey) / 6.0f);
        c[0] = c[1] = c[2] = 0.15f + 0.6f * dome + 0.25f * value_noise(u, v, 256, seed + 33);     // noisy: bump normals decohere
    });
    const int cobble_s = add_texture(s, 128, 128, [&](float u, float v, int, int, float* c) {
        c[0] = 0.15f + 0.35f * value_noise(u, v, 32, seed + 34); c[1] = 0.0f; c[2] = 0.0f;        // wet patches, .rrra
    });
    const int plaster_d = add_texture(s, 512, 512, [&](float u, float v, int, int, float* c) {
        const float n = 0.5f * value_noise(u, v, 8, seed + 35) + 0.3f * value_noise(u, v, 64, seed + 36) +
                        0.2f * value_noise(u, v, 256, seed + 37);
        c[0] = 0.66f + 0.22f * n; c[1] = 0.60f + 0.20f * n; c[2] = 0.48f + 0.18f * n;
    });
    const int plaster_h = add_texture(s, 512, 512, [&](float u, float v, int, int, float* c) {
        c[0] = c[1] = c[2] = 0.4f * value_noise(u, v, 32, seed + 38) + 0.6f * value_noise(u, v, 256, seed + 39);
    });
    const int brick_d = add_texture(s, 256, 128, [&](float u, float v, int x, int y, float* c) {
        const int row = y / 16, xo = x + ((row & 1) ? 16 : 0);
        const bool mortar = (y % 16) < 2 || (xo % 32) < 2;
        const float n = 0.12f * (value_noise(u, v, 64, seed + 40) - 0.5f);
        const float tone = 0.85f + 0.3f * (hash01(xo / 32, row, seed + 41) - 0.5f);
        if (mortar) { c[0] = 0.50f + n; c[1] = 0.48f + n; c[2] = 0.45f + n; }
        else { c[0] = 0.50f * tone + n; c[1] = 0.26f * tone + n; c[2] = 0.20f * tone + n; }
    });
    const int brick_h = add_texture(s, 256, 128, [&](float u, float v, int x, int y, float* c) {
        const int row = y / 16, xo = x + ((row & 1) ? 16 : 0);
        const bool mortar = (y % 16) < 2 || (xo % 32) < 2;
        c[0] = c[1] = c[2] = mortar ? 0.1f : 0.65f + 0.3f * value_noise(u, v, 128, seed + 42);
    });
    const int wood_d = add_texture(s, 256, 256, [&](float u, float v, int, int, float* c) {
        const float g = 0.5f + 0.5f * sinf(40.0f * v + 6.0f * value_noise(u, v, 16, seed + 43));
        c[0] = 0.36f + 0.16f * g; c[1] = 0.22f + 0.10f * g; c[2] = 0.11f + 0.06f * g;
    });
    const int metal_s = add_texture(s, 64, 64, [&](float u, float v, int, int, float* c) {
        const float n = 0.6f + 0.4f * value_noise(u, v, 16, seed + 44);
        c[0] = 0.75f * n; c[1] = 0.75f * n; c[2] = 0.8f * n;
    });
    // leaf clusters: several leaves per card, everything else cut out (alpha test, trace.fs:169-172)
    const int leaf_d = add_texture(s, 256, 256, [&](float u, float v, int, int, float* c) {
        float a = 0.0f, shade = 0.0f;
        for (int k = 0; k < 9; ++k) {
            const float cx = 0.15f + 0.7f * hash01(k, 1, seed + 45), cy = 0.15f + 0.7f * hash01(k, 2, seed + 45);
            const float ang = 6.2831853f * hash01(k, 3, seed + 45);
            const float dx = u - cx, dy = v - cy;
            const float lx = dx * cosf(ang) + dy * sinf(ang), ly = -dx * sinf(ang) + dy * cosf(ang);
            const float e = (lx * lx) / (0.17f * 0.17f) + (ly * ly) / (0.07f * 0.07f);      // an ellipse per leaf
            if (e < 1.0f) { a = 1.0f; shade = 0.6f + 0.4f * hash01(k, 4, seed + 45) - 0.25f * fabsf(ly) / 0.07f; }
        }
        const float n = 0.15f * (value_noise(u, v, 64, seed + 46) - 0.5f);
        c[0] = 0.16f * shade + n; c[1] = 0.46f * shade + n; c[2] = 0.10f * shade + n; c[3] = a;
    });
    const int awning_d = add_texture(s, 128, 128, [&](float u, float v, int x, int, float* c) {
        const bool stripe = (x / 16) & 1;
        const float n = 0.9f + 0.2f * (value_noise(u, v, 32, seed + 47) - 0.5f);
        if (stripe) { c[0] = 0.80f * n; c[1] = 0.78f * n; c[2] = 0.70f * n; }
        else { c[0] = 0.70f * n; c[1] = 0.10f * n; c[2] = 0.08f * n; }
    });
    const int glass_s = add_texture(s, 32, 32, [&](float u, float v, int, int, float* c) {
        const float n = 0.85f + 0.15f * value_noise(u, v, 8, seed + 48);
        c[0] = 0.9f * n; c[1] = 0.9f * n; c[2] = 0.95f * n;
    });
    const int roof_d = add_texture(s, 256, 256, [&](float u, float v, int x, int y, float* c) {
        const int row = y / 16, xo = x + ((row & 1) ? 8 : 0);
        const float tone = 0.8f + 0.4f * (hash01(xo / 16, row, seed + 49) - 0.5f);
        const bool edge = (y % 16) < 2;
        const float n = 0.1f * (value_noise(u, v, 64, seed + 50) - 0.5f);
        const float g = edge ? 0.12f : 0.30f * tone;
        c[0] = g * 1.1f + n; c[1] = g * 0.8f + n; c[2] = g * 0.75f + n;
    });
    auto set = [&](int mi, int d, int sp, int h) { s->materials[(size_t)mi].tex[0] = d; s->materials[(size_t)mi].tex[1] = sp; s->materials[(size_t)mi].tex[2] = h; };
    set(m[COBBLE], cobble_d, cobble_s, cobble_h);
    set(m[SIDEWALK], plaster_d, -1, cobble_h);
    set(m[PLASTER], plaster_d, -1, plaster_h);
    set(m[BRICKM], brick_d, -1, brick_h);
    set(m[WOOD], wood_d, -1, plaster_h);
    set(m[METAL], -1, metal_s, -1);
    set(m[LAMP], -1, metal_s, -1);
    set(m[LEAF], leaf_d, -1, -1);
    set(m[AWNING], awning_d, -1, -1);
    set(m[GLASS], -1, glass_s, -1);
    set(m[ROOF], roof_d, -1, plaster_h);
    (void)NMAT;
}

void build_bistro(vcth_scene* s, float detail, uint32_t seed) {
    enum { COBBLE, PLASTER, BRICKM, WOOD, METAL, LEAF, AWNING, GLASS, ROOF, SIDEWALK, LAMP, NMAT };
    Builder b{s};
    uint32_t rng = seed * 2654435761u + 977u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 16777216.0f; };
    int m[NMAT];
    m[COBBLE] = add_material(s, 0.40f, 0.39f, 0.37f, 0.25f, 0.25f, 0.25f);
    m[PLASTER] = add_material(s, 0.74f, 0.68f, 0.55f, 0.08f, 0.08f, 0.08f);
    m[BRICKM] = add_material(s, 0.50f, 0.28f, 0.22f, 0.08f, 0.08f, 0.08f);
    m[WOOD] = add_material(s, 0.40f, 0.26f, 0.14f, 0.12f, 0.12f, 0.12f);
    m[METAL] = add_material(s, 0.30f, 0.30f, 0.32f, 0.75f, 0.75f, 0.8f);
    m[LEAF] = add_material(s, 0.16f, 0.46f, 0.10f, 0.04f, 0.04f, 0.04f);
    m[AWNING] = add_material(s, 0.70f, 0.12f, 0.10f, 0.05f, 0.05f, 0.05f);
    m[GLASS] = add_material(s, 0.06f, 0.07f, 0.09f, 0.9f, 0.9f, 0.95f);
    m[ROOF] = add_material(s, 0.33f, 0.22f, 0.20f, 0.1f, 0.1f, 0.1f);
    m[SIDEWALK] = add_material(s, 0.58f, 0.56f, 0.52f, 0.12f, 0.12f, 0.12f);
    m[LAMP] = add_material(s, 0.95f, 0.9f, 0.7f, 0.9f, 0.85f, 0.6f);

    const float d = std::max(detail, 0.03f);
    const float cell = 0.235f / d;                // tessellation cell of the large surfaces (world units)
    auto n = [&](float len) { return std::max(1, (int)lroundf(len / cell)); };
    const int seg = std::max(6, (int)lroundf(14 * sqrtf(d)));
    const float X = 68.0f, Y0 = -26.0f, ZS = 10.0f, ZW = 17.0f;      // street half width, facade plane

    // ground: cobbled street, raised sidewalks
    b.quad({-X, Y0, ZS}, {2 * X, 0, 0}, {0, 0, -2 * ZS}, n(2 * X), n(2 * ZS), m[COBBLE]);
    for (int side = -1; side <= 1; side += 2) {
        const float z0 = side < 0 ? -ZW : ZS, z1 = side < 0 ? -ZS : ZW;
        b.quad({-X, Y0 + 0.4f, z1}, {2 * X, 0, 0}, {0, 0, z0 - z1}, n(2 * X), n(ZW - ZS), m[SIDEWALK]);
        const float zc = side < 0 ? -ZS : ZS;                               // kerb, facing the street
        if (side < 0) b.quad({-X, Y0, zc}, {2 * X, 0, 0}, {0, 0.4f, 0}, n(2 * X), 1, m[SIDEWALK]);
        else b.quad({X, Y0, zc}, {-2 * X, 0, 0}, {0, 0.4f, 0}, n(2 * X), 1, m[SIDEWALK]);
    }
    // street ends: walls closing the canyon
    b.quad({-X, Y0, ZW}, {0, 0, -2 * ZW}, {0, 44.0f, 0}, n(2 * ZW), n(44.0f), m[BRICKM]);
    b.quad({X, Y0, -ZW}, {0, 0, 2 * ZW}, {0, 44.0f, 0}, n(2 * ZW), n(44.0f), m[BRICKM]);

    // two rows of buildings
    for (int side = -1; side <= 1; side += 2) {
        float x = -X;
        int bi = 0;
        while (x < X - 1.0f) {
            const float wdt = std::min(14.0f + 9.0f * rnd(), X - x);
            const float hgt = 26.0f + 22.0f * rnd();
            const int wall = (bi % 3 == 1) ? m[BRICKM] : m[PLASTER];
            const float zf = side * ZW;                       // facade plane; the building extends away from the street
            const float zb = side * (ZW + 9.0f);
            // facade (faces the street), roof, the two side walls above the neighbours
            if (side < 0) b.quad({x, Y0, zf}, {wdt, 0, 0}, {0, hgt, 0}, n(wdt), n(hgt), wall);
            else b.quad({x + wdt, Y0, zf}, {-wdt, 0, 0}, {0, hgt, 0}, n(wdt), n(hgt), wall);
            if (side < 0) b.quad({x, Y0 + hgt, zf}, {wdt, 0, 0}, {0, 0, zb - zf}, n(wdt), n(9.0f), m[ROOF]);
            else b.quad({x, Y0 + hgt, zb}, {wdt, 0, 0}, {0, 0, zf - zb}, n(wdt), n(9.0f), m[ROOF]);
            b.quad({x, Y0, zb}, {0, 0, zf - zb}, {0, hgt, 0}, n(9.0f), n(hgt), wall);                        // -x side
            b.quad({x + wdt, Y0, zf}, {0, 0, zb - zf}, {0, hgt, 0}, n(9.0f), n(hgt), wall);                  // +x side
            // windows: glass pane slightly in front of the wall, wooden frame bars, shutters; balconies on some floors
            const int floors = std::max(2, (int)(hgt / 7.5f));
            const int bays = std::max(2, (int)(wdt / 4.5f));
            for (int f = 0; f < floors; ++f)
                for (int k = 0; k < bays; ++k) {
                    const float wx = x + wdt * (k + 0.5f) / bays, wy = Y0 + 2.0f + 7.5f * f + (f == 0 ? 0.5f : 0.0f);
                    const float ww = 2.2f, wh = f == 0 ? 4.6f : 3.6f;
                    const float zo = zf - side * 0.12f;       // towards the street
                    const int gseg = std::max(1, n(ww) / 2);
                    if (side < 0) b.quad({wx - ww / 2, wy, zo}, {ww, 0, 0}, {0, wh, 0}, gseg, gseg, m[GLASS]);
                    else b.quad({wx + ww / 2, wy, zo}, {-ww, 0, 0}, {0, wh, 0}, gseg, gseg, m[GLASS]);
                    const float zfr = zf - side * 0.3f;
                    const float z0 = std::min(zf, zfr), z1 = std::max(zf, zfr);
                    b.box({wx - ww / 2 - 0.25f, wy - 0.25f, z0}, {wx + ww / 2 + 0.25f, wy, z1}, cell, m[WOOD]);          // sill
                    b.box({wx - ww / 2 - 0.25f, wy + wh, z0}, {wx + ww / 2 + 0.25f, wy + wh + 0.25f, z1}, cell, m[WOOD]);   // head
                    b.box({wx - ww / 2 - 0.25f, wy, z0}, {wx - ww / 2, wy + wh, z1}, cell, m[WOOD]);
                    b.box({wx + ww / 2, wy, z0}, {wx + ww / 2 + 0.25f, wy + wh, z1}, cell, m[WOOD]);
                    b.box({wx - 0.06f, wy, z0}, {wx + 0.06f, wy + wh, z1}, cell, m[WOOD]);                                // mullion
                    if (f > 0 && ((k + f + bi) % 2) == 0) {                                                               // balcony
                        const float zbal = zf - side * 1.6f;
                        const float q0 = std::min(zf, zbal), q1 = std::max(zf, zbal);
                        b.box({wx - 1.9f, wy - 0.6f, q0}, {wx + 1.9f, wy - 0.35f, q1}, cell, m[PLASTER]);
                        for (int r = 0; r <= 12; ++r) {                                                                   // railing bars
                            const float rx = wx - 1.85f + 3.7f * r / 12.0f;
                            b.box({rx - 0.04f, wy - 0.35f, zbal - 0.04f}, {rx + 0.04f, wy + 1.0f, zbal + 0.04f}, 10.0f, m[METAL]);
                        }
                        b.box({wx - 1.9f, wy + 1.0f, zbal - 0.06f}, {wx + 1.9f, wy + 1.1f, zbal + 0.06f}, cell, m[METAL]);
                        // flower box with a small bush of leaf cards
                        b.box({wx - 1.5f, wy - 0.3f, zbal - 0.35f * 1.0f}, {wx + 1.5f, wy + 0.1f, zbal + 0.35f}, cell, m[WOOD]);
                        const int cards = std::max(4, (int)lroundf(60 * d));
                        for (int c = 0; c < cards; ++c) {
                            const V3 ctr = {wx - 1.4f + 2.8f * rnd(), wy + 0.2f + 0.7f * rnd(), zbal + (rnd() - 0.5f) * 0.8f};
                            const V3 ax = normalize({rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f});
                            const V3 up = normalize(cross(ax, {0.3f, 1.0f, 0.2f}));
                            const float h2 = 0.15f;
                            const V3 p0 = ctr - ax * h2 - up * h2, du = ax * (2 * h2), dv = up * (2 * h2);
                            const V3 nn = normalize(cross(du, dv));
                            b.tri(p0, p0 + du, p0 + du + dv, nn, nn, nn, m[LEAF], {0, 0}, {1, 0}, {1, 1});
                            b.tri(p0, p0 + du + dv, p0 + dv, nn, nn, nn, m[LEAF], {0, 0}, {1, 1}, {0, 1});
                            b.tri(p0, p0 + du + dv, p0 + du, nn * -1.0f, nn * -1.0f, nn * -1.0f, m[LEAF], {0, 0}, {1, 1}, {1, 0});
                            b.tri(p0, p0 + dv, p0 + du + dv, nn * -1.0f, nn * -1.0f, nn * -1.0f, m[LEAF], {0, 0}, {0, 1}, {1, 1});
                        }
                    }
                }
            // awning over the ground floor of every other building: a sloped striped cloth, both faces
            if ((bi % 2) == 0) {
                const float ay = Y0 + 7.4f, ar = 3.2f;
                const V3 o = side < 0 ? V3{x + 1.0f, ay, zf} : V3{x + wdt - 1.0f, ay, zf};
                const V3 du = side < 0 ? V3{wdt - 2.0f, 0, 0} : V3{-(wdt - 2.0f), 0, 0};
                const V3 dv = {0, -1.3f, -side * ar};
                auto sag = [](float u, float v) { return -0.12f * sinf(u * 25.0f) * v; };
                auto sag_b = [](float u, float v) { return 0.12f * sinf((1.0f - u) * 25.0f) * v; };
                b.patch(o, du, dv, n(wdt) * 2, n(ar) * 2, m[AWNING], sag);
                b.patch(o + du, du * -1.0f, dv, n(wdt) * 2, n(ar) * 2, m[AWNING], sag_b);
            }
            x += wdt;
            ++bi;
        }
    }

    // bistro terraces on both sidewalks: tables with chairs
    auto table = [&](float tx, float tz) {
        b.cylinder({tx, Y0 + 0.4f, tz}, 0.08f, 1.45f, seg, 2, m[METAL]);
        b.cylinder({tx, Y0 + 1.85f, tz}, 0.9f, 0.08f, seg * 2, 1, m[WOOD]);
        b.quad({tx - 0.64f, Y0 + 1.93f, tz + 0.64f}, {1.28f, 0, 0}, {0, 0, -1.28f}, 3, 3, m[WOOD]);      // top disc stand-in
        for (int c = 0; c < 4; ++c) {
            const float a = 1.5708f * c + 0.4f * rnd(), cx = tx + 1.5f * cosf(a), cz = tz + 1.5f * sinf(a);
            b.box({cx - 0.4f, Y0 + 1.2f, cz - 0.4f}, {cx + 0.4f, Y0 + 1.3f, cz + 0.4f}, 10.0f, m[WOOD]);        // seat
            b.box({cx - 0.4f, Y0 + 1.3f, cz + 0.32f}, {cx + 0.4f, Y0 + 2.3f, cz + 0.4f}, 10.0f, m[WOOD]);       // back
            for (int l = 0; l < 4; ++l) {
                const float lx = cx + ((l & 1) ? 0.34f : -0.34f), lz = cz + ((l & 2) ? 0.34f : -0.34f);
                b.box({lx - 0.04f, Y0 + 0.4f, lz - 0.04f}, {lx + 0.04f, Y0 + 1.2f, lz + 0.04f}, 10.0f, m[METAL]);
            }
        }
    };
    const int ntables = std::max(2, (int)lroundf(44 * sqrtf(d)));
    for (int t = 0; t < ntables; ++t) {
        const int side = (t & 1) ? 1 : -1;
        const float tx = -60.0f + 120.0f * (t / 2 + 0.5f) / ((ntables + 1) / 2) + (rnd() - 0.5f) * 2.0f;
        table(tx, side * (ZS + 3.3f + (rnd() - 0.5f)));
    }
    // street lamps, bollards, crates
    for (int i = 0; i < 14; ++i) {
        const int side = (i & 1) ? 1 : -1;
        const float lx = -62.0f + 124.0f * i / 13.0f, lz = side * (ZS + 0.8f);
        b.cylinder({lx, Y0 + 0.4f, lz}, 0.14f, 9.0f, seg, n(9.0f), m[METAL]);
        b.box({lx - 0.08f, Y0 + 9.3f, std::min(lz, lz - side * 1.6f)}, {lx + 0.08f, Y0 + 9.45f, std::max(lz, lz - side * 1.6f)}, cell, m[METAL]);
        b.sphere({lx, Y0 + 9.0f, lz - side * 1.6f}, 0.45f, seg * 2, seg, m[LAMP]);
    }
    for (int i = 0; i < 40; ++i) {
        const int side = (i & 1) ? 1 : -1;
        const float bx = -66.0f + 132.0f * i / 39.0f;
        b.cylinder({bx, Y0 + 0.4f, side * (ZS + 0.35f)}, 0.16f, 1.1f, seg, 2, m[METAL]);
    }
    for (int i = 0; i < 30; ++i) {
        const int side = rnd() < 0.5f ? -1 : 1;
        const float cx = -64.0f + 128.0f * rnd(), cz = side * (ZS + 4.5f + 2.0f * rnd()), sz = 0.6f + 0.7f * rnd();
        b.box({cx, Y0 + 0.4f, cz}, {cx + sz, Y0 + 0.4f + sz, cz + sz}, cell, m[WOOD]);
    }
    // strings of lights across the street
    for (int sidx = 0; sidx < 10; ++sidx) {
        const float sx = -58.0f + 116.0f * sidx / 9.0f;
        const int bulbs = 14;
        for (int k = 0; k <= bulbs; ++k) {
            const float t = (float)k / bulbs, z = -ZW + 0.5f + (2 * ZW - 1.0f) * t;
            const float y = Y0 + 15.0f - 2.5f * sinf(3.14159265f * t);
            b.sphere({sx + 0.6f * sinf(7.0f * t), y, z}, 0.16f, std::max(6, seg), std::max(4, seg / 2), m[LAMP]);
        }
    }
    // trees along both sidewalks: trunk, a few branches, and the foliage -- alpha-tested leaf cards (two faces each)
    // scattered through an ellipsoidal crown with random orientations: the incoherent normals and the in-volume
    // occupancy a foliage-heavy exterior has
    const int ntrees = 14;
    const int cards_per_tree = std::max(30, (int)lroundf(21000 * d * d));
    for (int t = 0; t < ntrees; ++t) {
        const int side = (t & 1) ? 1 : -1;
        const float tx = -60.0f + 120.0f * t / (ntrees - 1) + (rnd() - 0.5f) * 3.0f, tz = side * (ZS + 1.8f);
        const float th = 11.0f + 4.0f * rnd();
        b.cylinder({tx, Y0 + 0.4f, tz}, 0.55f, th, seg * 2, n(th), m[WOOD]);
        const V3 crown = {tx, Y0 + 0.4f + th + 3.0f, tz - side * 1.0f};
        const float rx = 6.5f + 1.5f * rnd(), ry = 5.0f + 1.5f * rnd(), rz = 5.5f + 1.0f * rnd();
        for (int br = 0; br < 5; ++br) {
            const float a = 6.2831853f * rnd();
            b.cylinder({tx + 0.8f * cosf(a), Y0 + 0.4f + th * (0.75f + 0.05f * br), tz + 0.8f * sinf(a)}, 0.16f, 4.0f, seg, 2, m[WOOD]);
        }
        for (int c = 0; c < cards_per_tree; ++c) {
            // rejection-free: a point in the unit ball, biased towards the shell like a real crown
            V3 p;
            do { p = {2 * rnd() - 1, 2 * rnd() - 1, 2 * rnd() - 1}; } while (dot(p, p) > 1.0f);
            const float rr = sqrtf(dot(p, p));
            const float k = rr > 1e-3f ? powf(rr, 0.35f) / rr : 0.0f;
            const V3 ctr = {crown.x + p.x * k * rx, crown.y + p.y * k * ry, crown.z + p.z * k * rz};
            const V3 ax = normalize({rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f});
            V3 up = cross(ax, {rnd() - 0.5f, rnd() + 0.2f, rnd() - 0.5f});
            if (dot(up, up) < 1e-6f) up = cross(ax, {0, 1, 0});
            up = normalize(up);
            // leaf clusters of 0.35 - 0.55 world units: ~22 cards per unit^3 of crown give an optical depth of about one
            // per unit of path through the crown -- one sees INTO a crown, as with real foliage (cards of 1.3 units made
            // every crown an opaque blob with a depth complexity in the thousands next to the camera)
            const float h2 = 0.17f + 0.10f * rnd();
            const V3 p0 = ctr - ax * h2 - up * h2, du = ax * (2 * h2), dv = up * (2 * h2);
            const V3 nn = normalize(cross(du, dv));
            b.tri(p0, p0 + du, p0 + du + dv, nn, nn, nn, m[LEAF], {0, 0}, {1, 0}, {1, 1});
            b.tri(p0, p0 + du + dv, p0 + dv, nn, nn, nn, m[LEAF], {0, 0}, {1, 1}, {0, 1});
            b.tri(p0, p0 + du + dv, p0 + du, nn * -1.0f, nn * -1.0f, nn * -1.0f, m[LEAF], {0, 0}, {1, 1}, {1, 0});
            b.tri(p0, p0 + dv, p0 + du + dv, nn * -1.0f, nn * -1.0f, nn * -1.0f, m[LEAF], {0, 0}, {0, 1}, {1, 1});
        }
    }
    texture_bistro(s, m, seed);
    finish(s);
}

// Image files of an MTL's map_Kd / map_Ks / map_bump (the reference decodes them with stb_image,
// R/Model.h:141-226): PNG, baseline JPEG, BMP, TGA (raw / RLE, colour / grey), PPM / PGM -- host/vct_image.h, decoders
// written from the format specifications.  Rows are stored bottom-up (row 0 at v = 0), which is what the reference's
// aiProcess_FlipUVs + top-down stb rows amount to.
bool load_image(const std::string& path, Texture& t) {
    vct_image::Image im;
    if (!vct_image::load(path, im)) return false;
    t.w = im.w; t.h = im.h;
    t.rgba.swap(im.rgba);
    return true;
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
    else if (kind == 1) build_atrium(s, detail, seed, false);
    else if (kind == 2) build_atrium(s, detail, seed, true);
    else if (kind == 3) build_bistro(s, detail, seed);
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
    std::vector<Texture> texs;
    std::map<std::string, int> tex_index;        // de-duplicated by file name, like R/Model.h:198-208
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
            else if (cur >= 0) {
                // map_Kd -> DiffuseTexture, map_Ks -> SpecularTexture, map_bump / bump -> HeightTexture
                // (R/Model.h:126-136; options such as "-bm 1.0" are skipped: the file name is the last token)
                int slot = -1;
                char key[64];
                if (sscanf(line, " %63s", key) == 1) {
                    if (!strcmp(key, "map_Kd")) slot = 0;
                    else if (!strcmp(key, "map_Ks")) slot = 1;
                    else if (!strcmp(key, "map_bump") || !strcmp(key, "map_Bump") || !strcmp(key, "bump")) slot = 2;
                }
                if (slot >= 0) {
                    std::string rest(line);
                    while (!rest.empty() && (rest.back() == '\n' || rest.back() == '\r' || rest.back() == ' ')) rest.pop_back();
                    const size_t sp = rest.find_last_of(" \t");
                    const std::string fname = sp == std::string::npos ? rest : rest.substr(sp + 1);
                    auto it = tex_index.find(fname);
                    int ti = it != tex_index.end() ? it->second : -2;
                    if (ti == -2) {
                        Texture t;
                        ti = load_image(dir + fname, t) ? (int)texs.size() : -1;     // unreadable / unsupported: flat colour
                        if (ti >= 0) texs.push_back(t);
                        tex_index[fname] = ti;
                    }
                    mats[(size_t)cur].tex[slot] = ti;
                }
            }
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
    s->textures = texs;
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
            if (c[k].t >= 0) s->uv.insert(s->uv.end(), {vt[2 * (size_t)c[k].t], vt[2 * (size_t)c[k].t + 1]});
            else s->uv.insert(s->uv.end(), {0.0f, 0.0f});
        }
    }
    return s;
}

// ---- on-disk scene cache: everything a loaded / generated scene holds, little-endian,
// one file.  Parsing a large OBJ + decoding its maps is the slow part of start-up; the cache is one read.
extern "C++" {
namespace {
const char kCacheMagic[8] = {'V', 'C', 'T', 'S', 'C', 'N', '0', '2'};
template <class T> bool put(FILE* f, const std::vector<T>& v) {
    const uint64_t n = v.size();
    return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T> bool get(FILE* f, std::vector<T>& v, uint64_t max_n) {
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1 || n > max_n) return false;
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace
}  // extern "C++"

int vcth_scene_save(const vcth_scene* s, const char* path) {
    if (!s || !path) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    bool ok = fwrite(kCacheMagic, 8, 1, f) == 1 && put(f, s->pos) && put(f, s->nrm) && put(f, s->tan) && put(f, s->bit) &&
              put(f, s->uv) && put(f, s->mat);
    std::vector<float> mf;
    std::vector<int32_t> mt;
    for (const Material& m : s->materials) {
        mf.insert(mf.end(), m.albedo, m.albedo + 4);
        mf.insert(mf.end(), m.spec, m.spec + 3);
        mt.insert(mt.end(), m.tex, m.tex + 3);
    }
    ok = ok && put(f, mf) && put(f, mt);
    const uint64_t ntex = s->textures.size();
    ok = ok && fwrite(&ntex, 8, 1, f) == 1;
    for (const Texture& t : s->textures) {
        const int32_t wh[2] = {t.w, t.h};
        ok = ok && fwrite(wh, 4, 2, f) == 2 && put(f, t.rgba);
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}

vcth_scene* vcth_scene_load_cache(const char* path, char* error) {
    auto failmsg = [&](const char* m) -> vcth_scene* { if (error) snprintf(error, 256, "%s", m); return nullptr; };
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) return failmsg("cannot open the cache file");
    char magic[8];
    vcth_scene* s = new vcth_scene();
    std::vector<float> mf;
    std::vector<int32_t> mt;
    const uint64_t big = 1ull << 33;
    bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, kCacheMagic, 8) == 0 && get(f, s->pos, big) && get(f, s->nrm, big) &&
              get(f, s->tan, big) && get(f, s->bit, big) && get(f, s->uv, big) && get(f, s->mat, big) && get(f, mf, big) &&
              get(f, mt, big);
    uint64_t ntex = 0;
    ok = ok && fread(&ntex, 8, 1, f) == 1 && ntex < (1u << 20);
    for (uint64_t i = 0; ok && i < ntex; ++i) {
        Texture t;
        int32_t wh[2];
        ok = fread(wh, 4, 2, f) == 2 && get(f, t.rgba, big) && wh[0] > 0 && wh[1] > 0 &&
             t.rgba.size() == (size_t)wh[0] * wh[1] * 4;
        t.w = wh[0]; t.h = wh[1];
        if (ok) s->textures.push_back(t);
    }
    fclose(f);
    const size_t ntri = s->mat.size(), nmat = mt.size() / 3;
    ok = ok && s->pos.size() == ntri * 9 && s->nrm.size() == ntri * 9 && s->tan.size() == ntri * 9 &&
         s->bit.size() == ntri * 9 && s->uv.size() == ntri * 6 && mf.size() == nmat * 7 && ntri > 0 && nmat > 0;
    for (size_t t = 0; ok && t < ntri; ++t) ok = s->mat[t] >= 0 && (size_t)s->mat[t] < nmat;
    for (size_t i = 0; ok && i < mt.size(); ++i) ok = mt[i] >= -1 && mt[i] < (int32_t)s->textures.size();
    if (!ok) { delete s; return failmsg("not a scene cache of this version, or truncated / inconsistent"); }
    for (size_t m = 0; m < nmat; ++m) {
        Material mm;
        memcpy(mm.albedo, &mf[m * 7], 16);
        memcpy(mm.spec, &mf[m * 7 + 4], 12);
        memcpy(mm.tex, &mt[m * 3], 12);
        s->materials.push_back(mm);
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

void vcth_scene_get_uvs(const vcth_scene* s, float* uv) {
    if (uv) memcpy(uv, s->uv.data(), s->uv.size() * sizeof(float));
}

int32_t vcth_scene_num_textures(const vcth_scene* s) { return s ? (int32_t)s->textures.size() : 0; }

void vcth_scene_texture_info(const vcth_scene* s, int32_t i, int32_t* w, int32_t* h) {
    const bool ok = s && i >= 0 && i < (int32_t)s->textures.size();
    if (w) *w = ok ? s->textures[(size_t)i].w : 0;
    if (h) *h = ok ? s->textures[(size_t)i].h : 0;
}

int32_t vcth_image_load(const char* path, int32_t* width, int32_t* height, uint8_t* rgba, size_t capacity) {
    if (!path) return -1;
    vct_image::Image im;
    if (!vct_image::load(path, im)) return -1;
    if (width) *width = im.w;
    if (height) *height = im.h;
    if (rgba) {
        // the file may have changed since the caller asked for its size: never write more than the caller has room for
        if (im.rgba.size() > capacity) return -2;
        memcpy(rgba, im.rgba.data(), im.rgba.size());
    }
    return 0;
}

void vcth_scene_get_texture(const vcth_scene* s, int32_t i, uint8_t* rgba) {
    if (!s || !rgba || i < 0 || i >= (int32_t)s->textures.size()) return;
    memcpy(rgba, s->textures[(size_t)i].rgba.data(), s->textures[(size_t)i].rgba.size());
}

void vcth_scene_get_material_textures(const vcth_scene* s, int32_t* mat_tex) {
    for (size_t i = 0; i < s->materials.size(); ++i)
        for (int k = 0; k < 3; ++k) mat_tex[3 * i + k] = s->materials[i].tex[k];
}

}  // extern "C"
