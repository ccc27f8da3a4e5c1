// glsl_shim.h -- a minimal GLSL-as-C++ stand-in, TEST INFRASTRUCTURE ONLY (tests/test_shader_crosscheck.py).
//
// Just enough of GLSL 4.30's vector / matrix types and built-ins for the text of the reference's cone-trace
// fragment shader (S/VoxelConeTracing.fs) and of its voxelization shaders (S/Voxelization.gs / .fs) to compile as C++ after a handful of mechanical rewrites (qualifiers
// dropped, array constructors -> braces, multi-component swizzles -> calls).  The shader text itself is read from
// /root/reference at test time and never stored.  Arithmetic: fp32, one rounding per operation (-ffp-contract=off),
// built-ins as the GLSL specification defines them (normalize(x) = x / length(x), reflect(I, N) = I - 2 dot(N, I) N,
// inverse() by cofactors) -- deliberately NOT the oracle's operation order: this is an independent evaluation.
// The three samplers are stand-ins: sampler3D -> the oracle's textureLod ([GL] rules restated in oracle/vct_oracle.cpp),
// sampler2D -> per-pixel constants the harness sets (the G-buffer contract of SURVEY.md 8 a5 takes the material fetches,
// the bump normal and the shadow term as inputs).
#ifndef VCT_GLSL_SHIM_H_
#define VCT_GLSL_SHIM_H_

#include <math.h>

namespace glsl {       // its own max / pow / log2 hide the C library's inside the shader text

struct vec2 {
    union { float x, r; };
    union { float y, g; };
    vec2() : x(0), y(0) {}
    explicit vec2(double s) : x((float)s), y((float)s) {}
    vec2(double a, double b) : x((float)a), y((float)b) {}
};
struct vec3 {
    union { float x, r; };
    union { float y, g; };
    union { float z, b; };
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(double s) : x((float)s), y((float)s), z((float)s) {}
    vec3(double a, double b_, double c) : x((float)a), y((float)b_), z((float)c) {}
    vec3 rgb() const { return *this; }
};
struct vec4 {
    union { float x, r; };
    union { float y, g; };
    union { float z, b; };
    union { float w, a; };
    vec4() : x(0), y(0), z(0), w(0) {}
    explicit vec4(double s) : x((float)s), y((float)s), z((float)s), w((float)s) {}
    vec4(double a_, double b_, double c, double d) : x((float)a_), y((float)b_), z((float)c), w((float)d) {}
    vec4(const vec3& v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    vec3 rgb() const { return vec3(x, y, z); }
    vec2 gb() const { return vec2(y, z); }
    vec2 xy() const { return vec2(x, y); }
    vec3 xyz() const { return vec3(x, y, z); }
    vec4 rrra() const { return vec4(x, x, x, w); }
};
struct ivec3 {                          // GLSL int(float): truncation toward zero
    int x, y, z;
    ivec3() : x(0), y(0), z(0) {}
    ivec3(double a, double b, double c) : x((int)a), y((int)b), z((int)c) {}
};

inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec4 operator*(float s, vec4 a) { return vec4(s * a.x, s * a.y, s * a.z, s * a.w); }
inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4& operator+=(vec4& a, vec4 b) { a = a + b; return a; }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline float length(vec2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
inline vec3 normalize(vec3 a) { return a / length(a); }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }
inline float max(float a, float b) { return a < b ? b : a; }     // GLSL: y if x < y else x
inline float abs(float a) { return fabsf(a); }
inline float pow(float a, float b) { return powf(a, b); }
inline float log2(float a) { return log2f(a); }

struct mat3 {
    vec3 c[3];                                              // columns
    mat3() {}
    mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; }
};
inline vec3 operator*(const mat3& m, vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
inline mat3 transpose(const mat3& m) {
    return mat3(vec3(m.c[0].x, m.c[1].x, m.c[2].x), vec3(m.c[0].y, m.c[1].y, m.c[2].y), vec3(m.c[0].z, m.c[1].z, m.c[2].z));
}
inline mat3 inverse(const mat3& m) {                        // adjugate / determinant
    const float a00 = m.c[0].x, a01 = m.c[0].y, a02 = m.c[0].z;     // a[col][row]
    const float a10 = m.c[1].x, a11 = m.c[1].y, a12 = m.c[1].z;
    const float a20 = m.c[2].x, a21 = m.c[2].y, a22 = m.c[2].z;
    const float b01 = a22 * a11 - a12 * a21, b11 = -a22 * a10 + a12 * a20, b21 = a21 * a10 - a11 * a20;
    const float det = a00 * b01 + a01 * b11 + a02 * b21;
    mat3 r;
    r.c[0] = vec3(b01, -a22 * a01 + a02 * a21, a12 * a01 - a02 * a11) / det;
    r.c[1] = vec3(b11, a22 * a00 - a02 * a20, -a12 * a00 + a02 * a10) / det;
    r.c[2] = vec3(b21, -a21 * a00 + a01 * a20, a11 * a00 - a01 * a10) / det;
    return r;
}

struct mat4 {                           // column-major, m * v = sum of columns scaled (GLSL 5.10)
    vec4 c[4];
    mat4() {}
};
inline vec4 operator*(const mat4& m, vec4 v) {
    return vec4(m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z + m.c[3].x * v.w,
                m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z + m.c[3].y * v.w,
                m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z + m.c[3].z * v.w,
                m.c[0].w * v.x + m.c[1].w * v.y + m.c[2].w * v.z + m.c[3].w * v.w);
}
// geometry-shader plumbing (S/Voxelization.gs): the three input vertices, the emitted ones
struct GlInVertex { vec4 gl_Position; };
struct GlInArray {
    GlInVertex v[3];
    GlInVertex& operator[](int i) { return v[i]; }
    int length() const { return 3; }
};
struct image3D { int unused; };

// ---- samplers ------------------------------------------------------------------------------------------------
struct sampler2D { int which; };       // 0 diffuse, 1 specular, 2 mask, 3 height, 4 shadow map
struct sampler3D { int unused; };
struct ShimState {                     // set by the harness per pixel
    vec4 diffuse, specular;
    float height;
    int shadow_pass;                   // the first `shadow_pass` of the 25 PCF taps see an unoccluded depth
    int shadow_calls;
    int lod_calls;                     // textureLod invocations (= executed cone steps)
    int cone_steps[8];                 // per Voxel_Cone_Tracing call (the test instruments the function's entry)
    int ncones;
    bool discarded;
    const void* params;                // vcto_params
    const unsigned char* chain;
    const float* shadow_depth;         // when set: sampler 4 is this S x S depth map ([GL] bilinear, clamp to edge:
    int shadow_S;                      //   the oracle's vcto_shadow_tex), not the counting stand-in
    ivec3 stored_pos;                  // imageStore (S/Voxelization.fs:88)
    vec4 stored_value;
    int stores;
};
extern ShimState g_shim;
}  // namespace glsl
extern "C" void vcto_texture_lod(const void* p, const unsigned char* chain, const float uvw[3], float lod, float out[4]);
extern "C" float vcto_shadow_tex(const float* depth, int S, float u, float v);
namespace glsl {

inline vec4 texture(const sampler2D& s, vec2 uv) {
    if (s.which == 4 && g_shim.shadow_depth) return vec4(vcto_shadow_tex(g_shim.shadow_depth, g_shim.shadow_S, uv.x, uv.y), 0, 0, 1);
    switch (s.which) {
    case 0: return g_shim.diffuse;
    case 1: return g_shim.specular;
    case 3: return vec4(g_shim.height, 0, 0, 1);
    case 4: return vec4(g_shim.shadow_calls++ < g_shim.shadow_pass ? 1.0f : 0.0f, 0, 0, 1);
    default: return vec4(0, 0, 0, 1);
    }
}
inline void imageStore(const image3D&, ivec3 p, vec4 v) { g_shim.stored_pos = p; g_shim.stored_value = v; ++g_shim.stores; }
inline void shim_cone_begin() { ++g_shim.ncones; }
inline vec4 textureLod(const sampler3D&, vec3 uvw, float lod) {
    if (g_shim.ncones >= 1 && g_shim.ncones <= 8) ++g_shim.cone_steps[g_shim.ncones - 1];
    ++g_shim.lod_calls;
    const float c[3] = {uvw.x, uvw.y, uvw.z};
    float o[4];
    vcto_texture_lod(g_shim.params, g_shim.chain, c, lod, o);
    return vec4(o[0], o[1], o[2], o[3]);
}

}  // namespace glsl
#endif
