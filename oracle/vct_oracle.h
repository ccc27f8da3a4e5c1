/*
 * vct_oracle.h -- scalar CPU restatement of the voxel-cone-tracing GI path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline.
 *
 * PARITY PINNED against the reference's own shaders: the reference (AlerianEmperor/
 * Voxel-Cone-Tracing) holds no tests, golden vectors or fixtures, and its C++ host needs
 * GLFW/GLEW/glm/assimp (unbuildable here) -- but the hot path is its seven GLSL files, and
 * those run in the build container: oracle/ref_gl.c (oracle/_ref/libvct_refgl.so) hands
 * them, read unmodified from /root/reference at run time, to Mesa llvmpipe (OpenGL 4.5 core).
 * tests/golden/make_ref_golden.py turns their outputs into the committed fixtures
 * tests/golden/ref_*.npz; tests/test_ref_gl.py holds this oracle to them:
 *   cone trace + composite   <= 1e-5 rel-L2 (measured 1.3e-7 .. 3.5e-7), identical discards
 *   shadow map               identical coverage; depth equal up to GL choice (c) below
 *   voxelization             identical occupancy; values equal under Mesa's choices, +-1 under ours
 *   glGenerateMipmap 2-D/3-D equal except exact .5 ties, which Mesa rounds either way
 *   Render on a textured scene  9.4e-7 rel-L2 under Mesa's choices, 1.2e-3 under ours
 * "choices" = what GL leaves to the implementation and Mesa decides differently: (a) log2
 * precision of the texture level of detail, (b) where in the 2x2 quad implicit derivatives
 * are taken, (c) interpolation on snapped or unsnapped window positions, (d) tie rounding in
 * glGenerateMipmap; vcto_set_gl_choices() switches the oracle to Mesa's so that each is
 * shown to be the ONLY difference.  Besides: the known-answer tests derived from the shader
 * text (tests/test_oracle_kat.py) and an independent numpy restatement (tests/np_restatement.py).
 *
 * Reference files restated (R = Voxel_Cone_Tracing_Final, S = R/Shader):
 *   S/VoxelConeTracing.fs:43-66   constants, SampleVoxels
 *   S/VoxelConeTracing.fs:82-107  Voxel_Cone_Tracing (cone march)
 *   S/VoxelConeTracing.fs:165-228 gather + composite
 *   S/VoxelConeTracing.vs:23-37   varyings (the "G-buffer" fields)
 *   R/Voxel_Cone_Tracing.h:110-126,248  volume format, sampler state, mip build
 *   R/Voxel_Cone_Tracing.h:128-134      voxelization projections
 *   S/Voxelization.vs/.gs/.fs           voxelize + light inject
 *   S/VoxelConeTracing.fs:132-163, S/Voxelization.fs:18-52  PCF
 * Rules marked [GL] restate the OpenGL 4.3 core specification (the behaviour of
 * the driver the reference calls into), see SURVEY.md Appendix A.
 *
 * All arithmetic is fp32, compiled with -ffp-contract=off; every fused
 * multiply-add is an explicit fmaf() so that the HIP kernels can match it
 * operation for operation.
 */
#ifndef VCT_ORACLE_H_
#define VCT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* G-buffer plane indices (fp32 SoA, 23 planes = 92 B / pixel). */
enum {
    VCTO_GB_P = 0,        /* Position_world            trace.vs:27 */
    VCTO_GB_NW = 3,       /* Normal_world (raw)        trace.vs:31 */
    VCTO_GB_TW = 6,       /* Tangent_world (raw)       trace.vs:32 */
    VCTO_GB_BW = 9,       /* BiTangent_world (raw)     trace.vs:33 */
    VCTO_GB_BUMPN = 12,   /* bump normal N (unit)      trace.fs:177 */
    VCTO_GB_ALBEDO = 15,  /* matColor rgba             trace.fs:167 */
    VCTO_GB_SPEC = 19,    /* specColor rgb (resolved)  trace.fs:209-210 */
    VCTO_GB_SHADOW = 22,  /* shadow_value              trace.fs:186 */
    VCTO_GB_PLANES = 23
};

typedef struct vcto_params {
    int32_t V;              /* VoxelDimensions            VCT.h:16  */
    float G;                /* VoxelGridWorldSize         VCT.h:17  */
    float camera_pos[3];    /* CameraPosition             VCT.h:167 */
    float light_dir[3];     /* LightDirection             VCT.h:14,168 */
    float ambient_factor;   /* AmbientFactor              VCT.h:53  */
    float shininess;        /* Shininess = 20             Mesh.h:86 */
    float max_distance;     /* MAX_DISTANCE = 75          trace.fs:43 */
    float max_alpha;        /* MAX_ALPHA = 0.95           trace.fs:44 */
    float tan_diffuse;      /* 0.577                      trace.fs:198 */
    float tan_specular;     /* 0.07                       trace.fs:218 */
    int32_t wrap_repeat;    /* 1 = GL_REPEAT (reference default, VCT.h:110-113), 0 = clamp-to-edge */
} vcto_params;

void vcto_default_params(vcto_params* p);

/* ---- volume: linear RGBA8 mip chain ------------------------------------------------
 * level k has N=V>>k texels per side, texel (x,y,z) at ((z*N+y)*N+x)*4 bytes from the
 * level's offset; levels are concatenated 0..log2(V).  [GL] A.1 */
int vcto_num_levels(int V);
size_t vcto_level_offset_texels(int V, int level);
size_t vcto_chain_texels(int V);
/* [GL] A.8 -- VCT.h:248 glGenerateMipmap: 2x2x2 box, per channel, requantised per level. */
void vcto_build_mips(uint8_t* chain, int V);

/* ---- sampler + cone ---------------------------------------------------------------- */
/* trace.fs:59-66 + [GL] A.2 textureLod with LINEAR_MIPMAP_LINEAR / LINEAR. */
void vcto_sample(const vcto_params* p, const uint8_t* chain, const float pos[3], float lod,
                 float out[4]);
/* [GL] A.2 textureLod(VoxelTexture, uvw, lod) alone -- the sampler behind vcto_sample, for tests that evaluate the
 * reference's shader text through their own GLSL stand-in (tests/test_shader_crosscheck.py). */
void vcto_texture_lod(const vcto_params* p, const uint8_t* chain, const float uvw[3], float lod, float out[4]);
/* trace.fs:82-107.  Returns the executed step count. */
int vcto_cone(const vcto_params* p, const uint8_t* chain, const float P[3], const float Nw[3],
              const float dir[3], float tan_half, float out[4]);
/* trace.fs:46-57 */
void vcto_cone_constants(float dirs[18], float weights[6]);
/* trace.fs:90-104 on an empty volume: steps a cone of this aperture executes, and the
 * lod of its last sample. */
int vcto_max_steps(const vcto_params* p, float tan_half, float* last_lod);

/* ---- per-pixel gather + composite  (trace.fs:165-228) -------------------------------
 * gb: 23 floats of one pixel.  out: rgba.  steps[7] / cones[28] optional (6 diffuse + 1
 * specular; cones = the vec4 each Voxel_Cone_Tracing call returned).  Returns 0 if the
 * fragment was discarded (albedo.a < 0.5; out = clear colour VCT.h:156-159). */
int vcto_shade_pixel(const vcto_params* p, const uint8_t* chain, const float gb[23],
                     float out[4], uint8_t steps[7], float cones[28]);

/* Whole-frame trace.  planes: [23][npix] fp32 SoA (linear pixel order).  out32f [npix][4],
 * out16f [npix][4] (IEEE half, round-to-nearest-even), steps [npix][7], cones [npix][7][4];
 * any output may be NULL.  nthreads<=1: scalar single thread; else static partition over
 * std::thread.  Returns total executed cone steps. */
uint64_t vcto_trace(const vcto_params* p, const uint8_t* chain, const float* planes,
                    size_t npix, float* out32f, uint16_t* out16f, uint8_t* steps, float* cones,
                    int nthreads);

uint16_t vcto_f32_to_f16(float f);
float vcto_f16_to_f32(uint16_t h);

/* ---- shadow map PCF  (vox.fs:18-52, trace.fs:132-163; [GL] A.9) ---------------------
 * depth: S*S fp32 in [0,1] (already quantised to 24-bit fixed by the producer), row-major,
 * bilinear, clamp-to-edge.  coord = xyz*0.5+0.5 shadow coordinate.  Returns the number of
 * the 25 taps that pass (caller scales by 1/25 (inject) or 0.111 (trace)). */
int vcto_pcf25(const float* depth, int S, const float coord[3], float bias);
/* the same for n coordinates ([n][3]) -> counts[n] (fixture generation at frame sizes) */
void vcto_pcf25_batch(const float* depth, int S, const float* coords, size_t n, float bias, int32_t* counts);
float vcto_shadow_tex(const float* depth, int S, float u, float v);

/* ---- voxelization ------------------------------------------------------------------- */
/* VCT.h:128-134: ProjX/Y/Z = ortho(-G/2,G/2,-G/2,G/2,G/2,3G/2) * lookAt(...), column-major
 * mat4 like glm.  axis 1,2,3 = X,Y,Z (vox.gs:34-39). */
void vcto_voxel_proj(float G, int axis, float m[16]);
/* vox.gs:24-39: dominant axis (1,2,3) of a world-space triangle. */
int vcto_dominant_axis(const float v0[3], const float v1[3], const float v2[3]);
/* vox.fs:58-86: fragment (window x, y, depth z in [0,1]) -> voxel index. */
void vcto_frag_to_voxel(int V, int axis, float fx, float fy, float fz, int32_t out[3]);

/* Triangle soup in MODEL space: pos [ntri][3][3] fp32, material id per triangle, material
 * albedo table [nmat][4] (flat per-material albedo -- SURVEY.md A.7 allows this in place of
 * the texture fetch at vox.fs:56).  model_scale = 0.05 (VCT.h:240).  light_vp: column-major
 * DepthViewProjectionMatrix (VCT.h:84-86) applied to WORLD positions (the reference applies
 * DepthVP*Model to model positions: the same point).  shadow may be NULL (PCF = 1). */
/* Material textures (R/Model.h:126-136,141-226 loads them, R/Mesh.h:91-108 binds them).  RGBA8, row 0 at
 * v = 0, texel = byte/255, wrap GL_REPEAT (R/Model.h:170-171).
 *
 * Level-0 sampling (vcto_tex_sample; textures without a mip chain): BILINEAR on level 0,
 *   x = u*W - 0.5, i0 = floor(x), a = x - i0, indices wrapped mod W (same for v), per channel
 *   ((1-a)*(1-b))*t00 + (a*(1-b))*t10 + ((1-a)*b)*t01 + (a*b)*t11 in that order, fp32.
 *
 * Mip-mapped sampling (vcto_tex_sample_lod; textures carrying `mips`): what the reference's sampler state asks for --
 * glGenerateMipmap + MIN = LINEAR_MIPMAP_LINEAR (R/Model.h:168,172), MAG = LINEAR (the LINEAR_MIPMAP_LINEAR
 * at :173 is not a valid mag filter: the call is an error and the default stays) -- with the implicit
 * derivatives of texture() in a fragment shader (S/VoxelConeTracing.fs:114-116,167,209, S/Voxelization.fs:56).
 * [GL 4.3 8.14] restated with the choices an implementation is free to make written down:
 *   chain     level k is max(1, W >> k) x max(1, H >> k), down to 1 x 1; a texel is the rounded mean
 *             (a + b + c + d + 2) >> 2 per channel of parent texels (2x, 2y) .. (2x+1, 2y+1), indices clamped to the
 *             parent's size (box filter; odd sizes drop the last row / column like the common drivers).
 *   scale     derivatives are differences inside the fragment's 2x2 pixel quad, evaluated on the fragment's own
 *             triangle (helper invocations extrapolate): d/dx = f(x ^ 1, y) - f(x, y), d/dy = f(x, y ^ 1) - f(x, y)
 *             (the sign does not matter below); in texels of level 0: du = ds * W, dv = dt * H.
 *             m = max(du_dx^2 + dv_dx^2, du_dy^2 + dv_dy^2)  (rho^2),  lambda = 0.5 * log2(m).
 *   log2      the library-free evaluation vcto_log2_det below (same instruction sequence on the GPU): exponent +
 *             atanh series of the mantissa, |error| < 1e-6 -- GL leaves the precision of lambda to the implementation.
 *   filter    m <= 1 (lambda <= 0, also NaN): magnification, BILINEAR on level 0.  lambda >= q = nlev - 1: level q.
 *             Otherwise d = floor(lambda), f = lambda - d, per channel fma(f, tau(d + 1), (1 - f) * tau(d)).
 * The level-0 form stays available (config.texture_mipmaps = 0 on the GPU side, textures without `mips` here): the
 * round-1/2 fixtures were made with it. */
typedef struct vcto_texture {
    const uint8_t* rgba;      /* level 0: [height*width*4] */
    int32_t width, height;
    const uint8_t* mips;      /* NULL: level-0 sampling; else the chain of vcto_tex_build_mips (level 0 first) */
    int32_t nlev;             /* levels in `mips` */
} vcto_texture;
int vcto_tex_num_levels(int width, int height);
size_t vcto_tex_level_offset(int width, int height, int level);      /* in texels; level == nlev: the chain's size */
void vcto_tex_build_mips(const uint8_t* rgba, int width, int height, uint8_t* chain);
float vcto_log2_det(float x);                                        /* x > 0, normal */
/* The two choices above that GL leaves to the implementation (log2 precision, where in the quad the derivatives are
 * taken) plus whether varyings are interpolated on the snapped or the unsnapped window positions: 0 = as stated above
 * (default; what the HIP path implements); bits select Mesa llvmpipe's choice instead -- 1: piecewise-linear log2,
 * 2: one derivative pair per quad at its even/even pixel, 4: unsnapped interpolation positions (7 = all).  Used only
 * to show that the default differs from the reference's GLSL as run by oracle/_ref in nothing else
 * (tests/test_ref_gl.py).  Process-global; tests reset it to 0. */
void vcto_set_gl_choices(int mode);
int vcto_get_gl_choices(void);
void vcto_tex_sample(const vcto_texture* t, float u, float v, float out[4]);
/* ds_dx .. dt_dy: quad differences of the NORMALISED coordinates (s, t) as defined above */
void vcto_tex_sample_lod(const vcto_texture* t, float u, float v, float ds_dx, float dt_dx, float ds_dy, float dt_dy,
                         float out[4]);

typedef struct vcto_scene {
    const float* pos;         /* [ntri*9] */
    const int32_t* material;  /* [ntri] */
    const float* albedo;      /* [nmat*4] */
    int32_t ntri, nmat;
    float model_scale;
    const float* shadow_depth; /* [S*S] or NULL */
    int32_t shadow_size;
    float light_vp[16];
    /* optional (NULL / 0: flat per-material colours): texture coordinates and diffuse textures.  A
     * fragment of a material with a diffuse texture takes albedo = texture(DiffuseTexture, uv) (vox.fs:56),
     * uv interpolated with the barycentrics the fragment's shadow coordinate is interpolated with.  Mip-mapped
     * textures take their derivatives from the voxelization raster: reference mode, the neighbouring pixel centres
     * (x ^ 1, y) and (x, y ^ 1) of the V x V window evaluated on the fragment's triangle; conservative mode (no
     * reference code), the UNCLAMPED barycentrics one voxel further along each in-plane axis, (cx + 1, cy) and
     * (cx, cy + 1). */
    const float* uv;           /* [ntri*6] */
    const int32_t* mat_tex;    /* [nmat*3]: diffuse, specular, height texture index or -1 */
    const vcto_texture* textures;
    int32_t ntex;
} vcto_scene;

/* Reference mode (A.7): dominant-axis VxV raster at pixel centres, top-left rule, value =
 * unorm8(albedo.rgb * PCF25/25), a=255, last writer in triangle order wins (vox.fs:88).
 * l0: V^3*4 bytes, linear, zero-initialised by the caller. */
void vcto_voxelize_reference(const vcto_params* p, const vcto_scene* s, uint8_t* l0);
/* North-star mode: conservative triangle/voxel-box overlap; every overlapped voxel receives
 * one fragment whose value is evaluated at the voxel centre projected along the dominant
 * axis onto the triangle (barycentrics clamped into the triangle); integer sum + count per
 * voxel, resolved to the rounded mean.  acc (optional, may be NULL): [V^3][4] uint32
 * (sum r, sum g, sum b, count). */
void vcto_voxelize_conservative(const vcto_params* p, const vcto_scene* s, uint8_t* l0,
                                uint32_t* acc);
/* z-slices [z0, z1) of vcto_voxelize_conservative's level 0: l0_slab [(z1 - z0)][V][V][4] (a slab of memory for a slab
 * of the result: the 1024^3 check on hosts that cannot hold the whole volume twice). */
void vcto_voxelize_conservative_zslab(const vcto_params* p, const vcto_scene* s, int32_t z0, int32_t z1, uint8_t* l0_slab);

/* ---- raster input stages -- the checkers of csrc/vct_raster.hip ---------------
 * GL rules restated once: near-plane clip (z >= -w), window coordinates snapped to 1/256 pixel, edge
 * functions in double (exact on snapped inputs: shared edges are watertight), pixel-centre sampling,
 * top-left fill rule, CCW front faces with back faces culled (R/main.cpp:55-58), depth test LESS in
 * triangle order, perspective-correct varyings. */
typedef struct vcto_mesh {
    const float* pos;          /* [ntri*9] model space                      R/Mesh.h:12-19 */
    const float* nrm;          /* [ntri*9] per-vertex normal   (G-buffer only; may be NULL for the shadow pass) */
    const float* tan;          /* [ntri*9] tangent */
    const float* bit;          /* [ntri*9] bitangent */
    const float* uv;           /* [ntri*6] texture coordinates or NULL */
    const int32_t* material;   /* [ntri] */
    const float* albedo;       /* [nmat*4] flat colour of materials without a diffuse texture */
    const float* specular;     /* [nmat*3] same for the specular colour */
    const int32_t* mat_tex;    /* [nmat*3] diffuse / specular / height texture index or -1; NULL = none */
    const vcto_texture* textures;
    int32_t ntri, nmat, ntex;
    float model_scale;         /* ModelMatrix = scale(0.05)                 VCT.h:183,204 */
} vcto_mesh;
/* DrawDepthTexture (VCT.h:192-211, S/Shadow.vs/.fs): depth [S*S] in [0,1], cleared to 1, quantised to
 * DEPTH_COMPONENT24.  No alpha test (Shadow.fs has none). */
void vcto_render_shadow_map(const vcto_mesh* m, const float light_vp[16], int32_t S, float* depth);
/* The vertex + fixed-function part of Render (VCT.h:161-189, S/VoxelConeTracing.vs) and the non-cone
 * fragment work of S/VoxelConeTracing.fs: matColor fetch + alpha test (:167-172; a discarded fragment
 * writes neither colour nor depth), CalcBumpNormal from three HeightTexture taps (:110-128), specColor
 * with the .rrra rule (:209-210), PCF x 0.111 (:132-163).  planes: [23][W*H] (include/vct.h VCT_GB_*);
 * pixels without a fragment keep albedo.a = 0.  Row 0 = bottom row of the GL window. */
void vcto_render_gbuffer(const vcto_mesh* m, const float view_proj[16], int32_t W, int32_t H,
                         const float* shadow_depth, int32_t shadow_size, const float light_vp[16],
                         float* planes);

/* ---- second bounce (north-star; BASELINE.json config 3) ---------------------------------------
 * The reference has NO code for this: its README claims "2 bounces" (README.md:16) but the
 * orchestrator injects direct light once and gathers once (VCT.h:138-139, vox.fs:88).  The
 * definition below is the build's own ("every occupied voxel gathers 6
 * diffuse cones from the bounce-0 volume along a stored voxel normal and adds albedo*irradiance,
 * re-mip, then screen trace"), restated here so the HIP kernel has something to be checked against.
 *
 * Per-voxel attributes come from the conservative voxelizer: every fragment also contributes its
 * material albedo (unorm8, before the shadow factor) and its triangle's front-face unit normal
 * n = normalize(cross(v1-v0, v2-v0)) (CCW = front, R/main.cpp:57-58), quantised to
 * floor(n*127 + 0.5) + 128 in [1,255]; the voxel keeps the rounded integer means.
 *   attr_albedo [V^3][4] uint8 (rgb, a = 255 where occupied), attr_normal [V^3][4] uint8 (biased
 *   xyz, w = 255 where occupied), linear voxel order like l0.  Either may be NULL. */
void vcto_voxelize_conservative_attr(const vcto_params* p, const vcto_scene* s, uint8_t* l0,
                                     uint32_t* acc, uint8_t* attr_albedo, uint8_t* attr_normal);
/* One bounce: for every voxel of level 0 with a != 0 and a non-zero stored normal,
 *   P = ((idx + 0.5)/V - 0.5) * G,  n = normalize(stored normal),
 *   frame t = normalize(cross(|n.y| < 0.9 ? (0,1,0) : (1,0,0), n)), b = cross(n, t),
 *   cone_i = Voxel_Cone_Tracing(from P with Normal_world = n, normalize(t*d.x + b*d.y + n*d.z), tan_diffuse)
 *   gathered like trace.fs:194-201: ind = sum w_i * cone_i, occlusion = 1 - ind.a,
 *   out.rgb = unorm8(l0.rgb/255 + albedo.rgb/255 * (occlusion * ind.rgb)),  out.a = l0.a.
 * Other voxels are copied.  chain0: full bounce-0 mip chain (linear); out_l0: V^3*4 bytes.
 * Returns the number of cone steps executed. */
uint64_t vcto_bounce(const vcto_params* p, const uint8_t* chain0, const uint8_t* attr_albedo,
                     const uint8_t* attr_normal, uint8_t* out_l0, int nthreads);

/* ---- anisotropic (directional) mip volumes -- north-star option, no reference code -----------
 * BASELINE.json north_star names "the anisotropic mip-filter downsample" as one of the kernels; the
 * reference has a single isotropic chain (glGenerateMipmap, VCT.h:248), so this is an OPTION and the
 * definition is the build's own (directional pre-integration as in Crassin et al. 2011):
 *   six chains, direction index = 2*axis + (0: travelling towards +axis, 1: towards -axis), levels
 *   1..log2 V only (level 0 is the shared isotropic level 0).  A parent texel of direction d is built
 *   from its 8 children of the same direction (of level 0 for level 1): the 2x2x2 block is four
 *   pairs along the axis; per pair  comp = F + (1 - F.a) * B  (front F = the child met first when
 *   travelling in direction d, per channel, fp32 on unorm8-decoded values, comp = fmaf(1-F.a, B, F));
 *   parent = unorm8(((c0 + c1) + c2 + c3) * 0.25) with the pairs in (lower, higher) order of the two
 *   other coordinates (first varying fastest = the lower axis).
 *   Sampling a level l >= 1 for a unit cone direction dir:  T(l) = dir.x^2 * tri(chain[x, sign]) +
 *   dir.y^2 * tri(chain[y, sign]) + dir.z^2 * tri(chain[z, sign])  (r = wx*tx; r = fma(wy,ty,r);
 *   r = fma(wz,tz,r)), sign = 0 if the component is >= 0; level 0 is the isotropic trilinear
 *   sample.  The two-level blend and the march are unchanged.
 * aniso: [6][chain_texels(V) - V^3][4] bytes (level k of a direction at texel offset
 * level_offset(k) - V^3, linear layout). */
void vcto_build_mips_aniso(const uint8_t* level0, int V, uint8_t* aniso);
/* Like vcto_trace, sampling levels >= 1 from the directional chains. */
uint64_t vcto_trace_aniso(const vcto_params* p, const uint8_t* chain, const uint8_t* aniso,
                          const float* planes, size_t npix, float* out32f, uint16_t* out16f,
                          uint8_t* steps, float* cones, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
