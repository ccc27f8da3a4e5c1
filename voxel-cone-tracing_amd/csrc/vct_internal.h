// vct_internal.h -- shared between the C-ABI translation unit and the kernel files.
#ifndef VCT_INTERNAL_H_
#define VCT_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vct_layout.h"

#if defined(__HIPCC__)
// unorm8 -> float, bit-identical to (float)c / 255.0f for every c in [0,255]: c * RN(1/255) misses for
// 126 of the 256 bytes; the two-term product below never does (tests/test_abi.py).
__device__ __forceinline__ float vct_unorm8_to_float(uint32_t c) {
    const float f = (float)c;
    return fmaf(f, 0x1.010102p-8f, f * -0x1.fdfdfep-33f);
}
// [GL] float -> unorm8, round to nearest
__device__ __forceinline__ uint32_t vct_float_to_unorm8(float f) {
    const float s = f * 255.0f + 0.5f;
    if (!(s > 0.0f)) return 0u;
    if (s >= 255.0f) return 255u;
    return (uint32_t)(int)s;
}
#endif

// ---- material textures (R/Model.h:126-136 loads them, R/Mesh.h:91-108 binds them) ---------------------------
// All textures of a scene live in one packed RGBA8 buffer, each with its mip chain behind level 0 when
// config.texture_mipmaps (glGenerateMipmap, R/Model.h:168).  texture(sampler2D, uv) is restated with the operation
// order of oracle/vct_oracle_raster.cpp: vcto_tex_sample (level 0, BILINEAR, GL_REPEAT) and vcto_tex_sample_lod
// (LINEAR_MIPMAP_LINEAR with the quad differences of the coordinate; oracle/vct_oracle.h "Mip-mapped sampling").
#define VCT_TEX_MAX_LEVELS 15       // 16384 x 16384 down to 1 x 1
struct VctTexDesc {
    uint32_t off;        // first texel of level 0 in the packed buffer
    int32_t w, h;
    uint32_t flags;      // bit 0: some texel has alpha != 255 (fragments need the alpha test of trace.fs:169-172); bit 1: square power of two
    int32_t nlev;        // levels stored (1: level 0 only)
    uint32_t lvl[VCT_TEX_MAX_LEVELS];     // texel offset of level k behind `off`
};
struct VctTextures {
    const uint32_t* texels;    // null: the scene has no textures
    const VctTexDesc* desc;
    const int32_t* mat_tex;    // [nmat][3] diffuse / specular / height texture index or -1
    const float* uv;           // [ntri][3][2]
    int32_t ntex;
    int32_t mips;              // 1: the textures carry mip chains and are sampled with implicit derivatives
};
#if defined(__HIPCC__)
__device__ __forceinline__ int vct_tex_of(const VctTextures& t, int material, int slot) {
    if (!t.texels) return -1;
    const int i = t.mat_tex[3 * (size_t)material + slot];
    return i >= 0 && i < t.ntex ? i : -1;
}
// [GL] bilinear, GL_REPEAT, on one level of W x H texels
__device__ __forceinline__ float4 vct_tex_bilinear(const uint32_t* __restrict__ base, int W, int H, float u, float v) {
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    // GL_REPEAT.  An integer remainder by a run-time divisor is ~25 instructions on this GPU and a bilinear tap used to
    // take four: the upper index follows from the lower one ((i + 1) mod n = wrap(i) + 1, or 0 at the end), and a size
    // that is a power of two (most maps) wraps with a mask -- the same integers either way.
    auto wrap = [](int i, int n) {
        if ((n & (n - 1)) == 0) return i & (n - 1);
        const int r = i % n;
        return r < 0 ? r + n : r;
    };
    const int i0 = wrap((int)fx, W), j0 = wrap((int)fy, H);
    const int i1 = i0 + 1 == W ? 0 : i0 + 1, j1 = j0 + 1 == H ? 0 : j0 + 1;
    const uint32_t p00 = base[(size_t)j0 * W + i0], p10 = base[(size_t)j0 * W + i1];
    const uint32_t p01 = base[(size_t)j1 * W + i0], p11 = base[(size_t)j1 * W + i1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 o;
#define VCT_TEXCH(ch, sh)                                                                                  \
    o.ch = w00 * vct_unorm8_to_float((p00 >> sh) & 0xffu) + w10 * vct_unorm8_to_float((p10 >> sh) & 0xffu) + \
           w01 * vct_unorm8_to_float((p01 >> sh) & 0xffu) + w11 * vct_unorm8_to_float((p11 >> sh) & 0xffu);
    VCT_TEXCH(x, 0) VCT_TEXCH(y, 8) VCT_TEXCH(z, 16) VCT_TEXCH(w, 24)
#undef VCT_TEXCH
    return o;
}
__device__ __forceinline__ float4 vct_tex_sample(const VctTextures& t, int ti, float u, float v) {
    const VctTexDesc* d = t.desc + ti;
    return vct_tex_bilinear(t.texels + d->off, d->w, d->h, u, v);
}
// log2 without the math library, instruction for instruction oracle/vct_oracle_raster.cpp vcto_log2_det
__device__ __forceinline__ float vct_log2_det(float x) {
    const uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 127;
    float f = __uint_as_float((b & 0x7fffffu) | 0x3f800000u);
    if (f > 1.41421354f) { f = f * 0.5f; e += 1; }
    const float s = __fdiv_rn(f - 1.0f, f + 1.0f);
    const float s2 = s * s;
    float p = 0.111111112f;
    p = fmaf(p, s2, 0.142857149f);
    p = fmaf(p, s2, 0.2f);
    p = fmaf(p, s2, 0.333333343f);
    p = fmaf(p, s2, 1.0f);
    return fmaf(s * p, 2.88539004f, (float)e);
}
// texture(sampler, (u, v)) with the quad differences (ds_dx, dt_dx, ds_dy, dt_dy) of the normalised coordinate
__device__ __forceinline__ float4 vct_tex_sample_lod(const VctTextures& t, int ti, float u, float v, float ds_dx,
                                                     float dt_dx, float ds_dy, float dt_dy) {
    const VctTexDesc* d = t.desc + ti;
    const int W = d->w, H = d->h, nlev = d->nlev;
    const uint32_t* base = t.texels + d->off;
    const uint32_t dflags = d->flags;
    // Texel offset of level k behind level 0.  A square power-of-two map stores W^2 (1 + 1/4 + ... ) texels in front of
    // level k = 4/3 (W^2 - W^2 / 4^k): the difference is a multiple of 3, so the division is one multiply by the inverse
    // of 3 modulo 2^32 -- the same integers as the table, without the round trip that depended on k.
    auto level_offset = [&](int k) -> uint32_t {
        if (dflags & 2u) {
            const uint32_t wh = (uint32_t)W * (uint32_t)W;
            return ((wh - (wh >> (2 * k))) << 2) * 0xaaaaaaabu;
        }
        return d->lvl[k];
    };
    if (!t.mips || nlev <= 1) return vct_tex_bilinear(base, W, H, u, v);
    const float du_dx = ds_dx * (float)W, dv_dx = dt_dx * (float)H;
    const float du_dy = ds_dy * (float)W, dv_dy = dt_dy * (float)H;
    const float ax = du_dx * du_dx + dv_dx * dv_dx, ay = du_dy * du_dy + dv_dy * dv_dy;
    const float m = ax > ay ? ax : ay;
    if (!(m > 1.0f)) return vct_tex_bilinear(base, W, H, u, v);                     // magnification (and NaN)
    const float lam = 0.5f * vct_log2_det(m);
    const int q = nlev - 1;
    if (lam >= (float)q) return vct_tex_bilinear(base + level_offset(q), max(1, W >> q), max(1, H >> q), u, v);
    const int k = (int)lam;
    const float f = lam - (float)k, g = 1.0f - f;
    const float4 t1 = vct_tex_bilinear(base + level_offset(k), max(1, W >> k), max(1, H >> k), u, v);
    const float4 t2 = vct_tex_bilinear(base + level_offset(k + 1), max(1, W >> (k + 1)), max(1, H >> (k + 1)), u, v);
    float4 o;
    o.x = fmaf(f, t2.x, g * t1.x); o.y = fmaf(f, t2.y, g * t1.y);
    o.z = fmaf(f, t2.z, g * t1.z); o.w = fmaf(f, t2.w, g * t1.w);
    return o;
}
#endif

// ---- shadow map words (R/Voxel_Cone_Tracing.h:79-105 keeps a DEPTH_COMPONENT24 texture) ----------------------------
// The map is stored as the very words the shadow pass's atomicMin leaves behind: the fp32 bits of the depth, already
// quantised to 24 bits ([GL] fixed-point depth: float(round(z * (2^24 - 1)) / (2^24 - 1)), in [0, 1] so <= 0x3f800000),
// plus an EPOCH in the two top bits: pass n writes with epoch 3 - (n & 3).  Positive floats order like their bit
// patterns and a lower epoch is a smaller word, so a new pass simply overwrites what older passes left (no clear, no
// conversion pass: the buffer is memset once every fourth pass, when the epoch wraps); a reader subtracts the pass's
// epoch base and clamps -- words of older passes and never-written ones (all ones) come out as depth 1.0, the
// cleared depth buffer.  Two VALU per texel instead of a 192 MiB conversion pass per shadow map.
#define VCT_SHADOW_ONE 0x3f800000u
#define VCT_SHADOW_EPOCH(e) ((uint32_t)(e) << 30)
#if defined(__HIPCC__)
__device__ __forceinline__ float vct_shadow_depth(uint32_t word, uint32_t ebase) {
    const uint32_t v = word - ebase;
    return __uint_as_float(v < VCT_SHADOW_ONE ? v : VCT_SHADOW_ONE);
}
// [GL] DEPTH_COMPONENT24: z in [0, 1) -> float(q / (2^24 - 1)), q = floor(z * (2^24 - 1) + 0.5) in double exactly as
// the oracle computes it (the product is exact in double, so the fma is the same single rounding as mul + add);
// float(q) / 16777215.0f (IEEE) equals float(double(q) / 16777215.0) for every q (tests/test_oracle_kat.py).
__device__ __forceinline__ uint32_t vct_depth24_bits(float z) {
    const uint32_t q = (uint32_t)floor(fma((double)z, 16777215.0, 0.5));
    return __float_as_uint(__fdiv_rn((float)q, 16777215.0f));
}
#endif

// PCF short cut.  A tap is a convex combination of four depths of the 6 x 6 window, so it lies between the window's
// smallest and largest depth -- up to rounding: the four weights are products of (a, 1 - a) x (b, 1 - b), each within
// 3 ulp of the exact product, and the four-term sum adds at most 4 more (mul + add or fma chain alike), so every tap of
// the window is within a factor 1 +- 2^-21 of [dmin, dmax].  With margins of 2^-19: when the compared depth is
// <= dmin * (1 - 2^-19) all 25 taps pass, when it is > dmax * (1 + 2^-19) none does -- and only a window the shadow
// boundary crosses pays for the 25 bilinear taps (~330 of the shade kernel's ~1200 instructions per pixel, the larger
// part of the voxelizer's per-fragment work).  Raw words order like their depths (an older epoch or a never-written
// word is larger than every word of the current pass and decodes to 1.0), so the minimum / maximum are taken on the
// words.  Returns 25 / 0, or -1: evaluate the taps.
#if defined(__HIPCC__)
__device__ __forceinline__ int vct_pcf_window_verdict(uint32_t wmin, uint32_t wmax, uint32_t ebase, float cur) {
    const float lo = vct_shadow_depth(wmin, ebase) * 0.999998f, hi = vct_shadow_depth(wmax, ebase) * 1.000002f;
    return cur <= lo ? 25 : (cur > hi ? 0 : -1);
}
// The same verdict from the tile table: (col0, row0) = first texel of a regular (unclamped, six consecutive texels per
// axis) window.  Tile (tx, ty) holds the bounds of texels [8 tx, 8 tx + 13) x [8 ty, 8 ty + 13) -- every window that
// STARTS in the tile lies inside that region -- so one 8-byte load bounds the window from outside.  -1: undecided.
#define VCT_SHADOW_TILE_SHIFT 3
#define VCT_SHADOW_TILE_REACH 5      // a window spans 6 texels: it ends at most 5 texels beyond the tile it starts in
__device__ __forceinline__ int vct_pcf_tile_verdict(const uint2* __restrict__ tiles, int S, uint32_t col0, uint32_t row0, float cur) {
    const uint32_t nb = ((uint32_t)S + 7u) >> VCT_SHADOW_TILE_SHIFT;
    const uint2 t = tiles[(row0 >> VCT_SHADOW_TILE_SHIFT) * nb + (col0 >> VCT_SHADOW_TILE_SHIFT)];
    return cur <= __uint_as_float(t.x) * 0.999998f ? 25 : (cur > __uint_as_float(t.y) * 1.000002f ? 0 : -1);
}
#endif

// Six consecutive shadow-map words (one row of a PCF window) as ONE dwordx4 + ONE dwordx2 load: the row starts at any
// 4-byte boundary (gfx950 under HSA runs with unaligned vector access enabled, and hipcc emits the wide loads for this
// packed type).  The PCF gathers are bound by the number of load instructions the texture-address unit has to spread
// over cache lines (36 dword loads per fragment: 70 of the voxelizer's 100 us at configs[1]), not by bytes.
struct __attribute__((packed, aligned(4))) VctWords6 { uint32_t v[6]; };
// The nine floats of one triangle in a per-vertex attribute array ([ntri][3][3]: 36 B records at 4-byte alignment) as
// dwordx4 + dwordx4 + dword instead of nine dword loads -- same reason.
struct __attribute__((packed, aligned(4))) VctTri9 { float v[9]; };

#define VCT_BIN_CSTRIDE 32          // words between the counters of two bins (vct_raster.hip)
#define VCT_BIN_HUGE_CAP 1024u      // records the tile-binned raster lets every bin scan (vct_raster.hip k_bin_setup)
#define VCT_NO_SLOT 0xffffffffu
#define VCT_TILE 8
#define VCT_TILE_PIX 64
#define VCT_GB_NPLANES 23
#define VCT_MAX_STEPS 1024
// Executed steps.  The screen trace STORES each tile's count into its own slot, tile_steps[tile] (frame-wide tile
// index; the tile's waves add theirs in LDS and the last one to arrive stores the sum): no global atomic, nothing to
// zero between launches, and the host gets the total (sum over the launched rows) as well as the per-tile-row cost
// histogram (vct_last_row_steps: load-aware slabs) from the same words.  (Device-scope atomicAdd into a shared bank
// cost 2.5 % of the trace when neighbouring tiles shared addresses.)  The bounce kernels, whose waves loop over many
// voxels, keep a small bank of atomic counters.
#define VCT_STEP_COUNTERS 256   // bounce kernels only (power of two)

// One entry per march step of a cone aperture.  The step sequence of trace.fs:90-104 (dist,
// diameter, lod) does not depend on the pixel, only on (V, G, tanHalfAngle, MAX_DISTANCE): the
// host evaluates it once with the reference's operation order and the kernel reads it through
// the scalar cache.
struct VctLevelRef {    // one mip level as the sampler needs it (all wave-uniform)
    uint32_t off;      // texel offset of the level in the chain
    uint32_t mask_x;   // x bits of the level's Morton index: 0x09249249 & (N^3 - 1)
    float fN;          // N = V >> level, as float
    int32_t m;         // N - 1
};

// The verified constant division of the march (vct_trace.hip div_const<1>): 1 = two-term product
// fma(x, r_hi, x * r_lo) with r_hi + r_lo = 1/d to ~48 bits (2 VALU), 0 = one FMA correction round of x * RN(1/d)
// (3 VALU, rounds 1-3).  Either is exact only for divisors the device has checked over every fp32 input.
#ifndef VCT_DIV2
#define VCT_DIV2 1
#endif
// second argument of div_const<1> for divisor d with r = RN(1 / d): the low word of the reciprocal, or d itself
static inline float vct_div_aux(float d, float r) {
#if VCT_DIV2
    return (float)(1.0 / (double)d - (double)r);
#else
    (void)r;
    return d;
#endif
}
struct VctStep {       // 64 B: one s_load_dwordx16 per march step
    float dist;        // trace.fs:91,103
    float occ_rcp;     // RN(1 / occ_den): reciprocal for the exact constant division (vct_trace.hip)
    float occ_den;     // 1 + 0.03*diameter            trace.fs:101 -- or, in a table built for the verified two-term
                       // division (VCT_DIV2, vct_capi.hip refresh_steps), the low word r_lo of its reciprocal
    float frac;        // fract(lod) after [GL] clamp  trace.fs:97
    int32_t level;     // floor(lod)
    int32_t level2;    // min(level+1, maxLevel)
    int32_t two_levels;  // lod > 0 (minification) and frac != 0: blend level/level2; else `level` only
    float omf;         // 1 - frac (the level blend's other factor: one VALU per step less than computing it per lane)
    VctLevelRef l1, l2;
};

struct VctTraceParams {
    const uint32_t* chain;              // Morton chain, RGBA8 packed
    uint32_t level_off[VCT_MAX_LEVELS]; // texel offsets
    int32_t V, nlev;
    float G, half_G, vs;
    float half_G_rcp;                   // RN(1 / half_G)
    float half_G_aux;                   // what div_const takes beside it: half_G, or (VCT_DIV2 and fast_div) the low word of 1 / half_G
    int32_t fast_div;                   // 1: every constant divisor admits the FMA-corrected division
    float cam[3];
    float light[3];
    float ambient, shininess, max_alpha;
    int32_t wrap_repeat;
    const uint32_t* spread_lut;         // [1024] spread3(i) << 2: dilated byte offsets of an x coordinate (scalar loads)
    // footprint records of the levels >= 1 (vct_volume.hip k_build_cells), biased so that the record of Morton index c of
    // the level at texel offset `off` sits at cells_biased + (off + c) * 32; null: per-texel gathers everywhere
    const char* cells_biased;
    const VctStep* steps_diffuse;
    const VctStep* steps_specular;
    int32_t n_diffuse, n_specular;
    int32_t width, height, tiles_x, tiles_y;
    int32_t tile_row0, tile_row1;       // slab [row0,row1)
    // Interleaved slabs (vct_multi.hip): of the rows [row0, row1) only every row_stride-th is traced, starting with row0
    // (0 / 1: all of them); pack_rows = 1 writes traced row j to pixel rows 8j .. 8j+7 of `out` (a rank's packed slab of
    // an interleaved frame) instead of the row's own place in the frame.  k_trace_tile_split only.
    int32_t row_stride, pack_rows;
    int32_t ntiles;                     // tiles of this launch (rows traced x tiles_x)
    int32_t spec_prio;                  // 1: the specular waves raise their issue priority (slab launches, vct_trace.hip)
    const float* gbuf;                  // tiled [tile][23][64]
    uint16_t* out;                      // RGBA16F [h][w][4]
    uint8_t* dbg_steps;                 // [npix][7] or null
    float* dbg_cones;                   // [npix][7][4] or null
    unsigned long long* step_counter;   // [VCT_STEP_COUNTERS] bounce kernels: partial sums of executed steps (zero at launch)
    uint32_t* tile_steps;               // [tiles] screen trace: executed steps of each tile (stored, not added)
    unsigned long long* stats;          // [8] wave-level march counters (builds with -DVCT_STATS=1 only)
    // second bounce (k_bounce): per-voxel attributes (pooled like the accumulators: [slot][512]), touched-brick
    // flags, output level 0
    const uint32_t* attr_albedo;
    const uint32_t* attr_normal;
    const uint32_t* brick_slot;
    const uint32_t* brick_prev;
    const uint32_t* bounce_seen;        // bricks the bounce chain showed when its mips were last built
    uint32_t* bounce_out;
    uint32_t nbricks;
    const uint32_t* slot_brick;         // [nslots] the bricks a fragment of the mesh can land in (the voxelizer's slots)
    uint32_t nslots;
    // anisotropic option: six directional chains (levels >= 1), Morton per level, each
    // `aniso_stride` texels; level k of a direction at texel offset level_off[k] - level_off[1]
    const uint32_t* aniso;
    uint32_t aniso_stride;
    uint32_t aniso_alt_slab;            // float4 offset from a level's LDS slab to its second ("-axis") slab
    uint32_t* bounce_list;              // global list of occupied voxels (Morton indices), brick by brick
    uint32_t* bounce_list_count;
    uint32_t bounce_list_cap;
    uint32_t* brick_over;               // bricks whose voxels did not fit the list
    // Live-pixel compaction (config.trace_variant = 4; vct_trace.hip k_compact_tiles): the live pixels (albedo.a >= 0.5)
    // of every 16x16-pixel super-tile packed into whole waves.  vt_pix[v * 64 + lane] = tile << 6 | pixel of the tile
    // (all ones: no pixel), vt_count[0] = virtual tiles.  Null: lane = pixel of the launched tile.
    uint32_t* vt_pix;
    uint32_t* vt_count;
};

#ifndef VCT_VOX_CHUNK
#define VCT_VOX_CHUNK 4096u       // most fragments one work item (workgroup) of the voxelize pass takes (vct_capi.hip build_voxel_slots)
#endif
struct VctVoxParams {
    int32_t V;
    float G, model_scale;
    const float* pos;          // [ntri][9]
    const int32_t* material;   // [ntri]
    const float* albedo;       // [nmat][4]
    int32_t ntri;
    const uint32_t* shadow;    // [S*S] shadow-map words (vct_shadow_depth) or null
    uint32_t shadow_ebase;     // epoch base of the pass that produced them
    int32_t shadow_size;
    const uint2* shadow_tiles; // [ceil(S/8)]^2 decoded (min, max) per dilated 8 x 8 tile (vct_launch_shadow_minmax) or null
    float light_vp[16];
    // Per brick the uploaded mesh can touch (found once per mesh) there is a SLOT: brick_slot maps brick -> slot
    // (VCT_NO_SLOT: no fragment ever lands there).  A 1024^3 grid costs the surface, not 16 GiB.
    // North-star mode: the fragments of the mesh sorted by slot (frag_sorted[slot_first[s] .. slot_first[s + 1]),
    // entry = triangle << 9 | voxel inside the brick), accumulated in LDS by one workgroup per slot and resolved into
    // stage[slot][512] (+ the voxel attributes).  Reference mode: acc[nslots][512][2] in HBM, (triangle + 1) << 32 | rgb
    // by atomicMax.
    unsigned long long* acc;   // [nslots][512][2] (reference mode)
    const uint32_t* brick_slot;   // [V^3 / 512]
    uint32_t* brick_mark;      // mark_only (reference mode, at upload): bricks a fragment lands in
    int32_t mark_only;
    const uint32_t* frag_sorted;  // [nfrag]
    // light-independent per-fragment values, written by k_frag_geom: the fragment's clamped barycentrics (b0, b1) -- once
    // per mesh -- and, in a scene with textures, its albedo (once per change of the textures / texture coordinates)
    const float2* frag_bary;      // [nfrag]
    const float* frag_alb;        // [nfrag][3] or null (no textures: the material's colour is read per fragment)
    const uint32_t* tri_qnrm;     // [ntri][3] quantised front-face normal (voxel attributes) or null
    const uint32_t* slot_first;   // [nslots + 1]
    const uint32_t* slot_brick;   // [nslots]
    uint32_t nslots;
    // Work items of the pass: a slot's fragments in chunks of at most VCT_VOX_CHUNK, one workgroup per chunk, heaviest
    // slots first (built on the host once per mesh).  A slot of one chunk is accumulated and resolved in LDS; the chunks of a
    // larger one add their LDS partial sums to the slot's accumulators in HBM (acc2: [multi][512][2], acc2_attr:
    // [multi][512][3]); a second small kernel resolves those slots (multi_slot[multi]) and re-zeroes the accumulators.
    const uint4* items;        // [nitems] (slot, first fragment, fragments, index among the multi-chunk slots or ~0)
    uint32_t nitems;
    uint32_t chunk;            // fragments per work item
    unsigned long long* acc2;
    unsigned long long* acc2_attr;
    const uint32_t* multi_slot;
    uint32_t nmulti;
    uint32_t* stage;           // [nslots][512] resolved RGBA8 of the pass (Morton order inside the brick)
    uint32_t* stage_albedo;    // [nslots][512] or null (config.voxel_attributes)
    uint32_t* stage_normal;
    uint32_t* brick_flags;     // [V^3 / 512] raised for bricks written by a pass, consumed by the sparse resolve
    float proj[48];            // ProjX, ProjY, ProjZ (VCT.h:128-134), column-major; reference mode only
    int32_t mode;
    VctTextures tex;           // diffuse textures + texture coordinates (vox.fs:56); texels == null: flat colours
};

// inputs of the raster stages (vct_raster.hip); all device pointers
struct VctRasterArgs {
    const float* pos;            // [ntri][9] model space
    const float* nrm;            // [ntri][9] per-vertex normal / tangent / bitangent (G-buffer only)
    const float* tan;
    const float* bit;
    const int32_t* material;     // [ntri]
    const int32_t* tri_alpha;    // [ntri] alpha-test class of the triangle (vct_raster.hip k_tri_alpha) or null
    const float* albedo;         // [nmat][4]
    const float* specular;       // [nmat][3]
    int32_t ntri;
    float model_scale;
    // Visibility words: 64-bit (depth | id) of the main draw, W*H; 32-bit depth-only of the shadow pass, S*S.
    // Invariant between passes: every word is all-ones ("empty") -- the consumer of a word resets it.
    unsigned long long* vis;
    uint32_t* vis32;             // = the shadow-map words of the context
    uint32_t vis32_ebase;        // epoch base this shadow pass writes with
    int32_t* wave_list;          // [2*ntri] medium sub-triangles
    uint32_t* wave_count;
    int32_t* group_list;         // [2*ntri] small-medium sub-triangles (one 16-lane group each)
    uint32_t* group_count;
    uint2* items;                // tile work items of huge sub-triangles
    uint32_t* item_count;
    uint32_t item_capacity;
    uint32_t* next_counts;       // [3] the other counter set (item, wave, group): zeroed by this pass for the next
    void* recs;                  // [2 * ntri] 96-byte set-up records of the listed sub-triangles (vct_raster.hip SubTriRec)
    VctTextures tex;             // material textures + texture coordinates (G-buffer pass)
    // Tile-binned visibility (vct_raster.hip "Tile-binned visibility"; binned == 0: the direct form above).  Scratch of
    // one pass kind: records, (bin, record) pairs, per-bin entry ranges, work items, the list of huge records, and two
    // alternating sets of eight counters (a pass zeroes the other set; bin_count is left zero by k_bin_alloc).
    int32_t binned;
    void* bin_recs;              // [bin_rec_cap] 160-byte BinRec
    uint32_t bin_rec_cap;
    uint2* bin_entries;          // [bin_entry_cap] one per (sub-triangle, 16x16 bin) overlap
    uint32_t bin_entry_cap;
    uint32_t* bin_count;         // [bins * VCT_BIN_CSTRIDE] (a bin's counter has a cache line to itself)
    uint32_t* bin_cursor;        // [bins * VCT_BIN_CSTRIDE]
    uint4* bin_items;            // [bin_item_cap]
    uint32_t bin_item_cap;
    uint32_t* bin_huge;          // [bin_huge_cap]
    uint32_t bin_huge_cap;
    uint32_t* bin_ctr;           // [8] this pass's counters
    uint32_t* bin_next_ctr;      // [8] the next pass's
};

hipError_t vct_launch_shadow_raster(const VctRasterArgs& a, const float light_vp[16], int S, hipStream_t s);
// float depths in [0, 1] <-> shadow-map words (uploads of maps made elsewhere, downloads)
hipError_t vct_launch_shadow_encode(const float* depth, uint32_t* words, size_t n, uint32_t ebase, hipStream_t s);
hipError_t vct_launch_shadow_decode(const uint32_t* words, float* depth, size_t n, uint32_t ebase, hipStream_t s);
// Depth bounds of the shadow map per 8 x 8-texel tile, each over the tile DILATED by the reach of a PCF window (13 x 13
// texels; decoded words: monotonic in depth), rebuilt after a pass / upload: [ceil(S / 8)]^2 (min, max) pairs.  Where the
// compared depth lies outside the bounds of the tile a window starts in, all 25 taps pass or fail
// (vct_pcf_window_verdict's margins hold for any superset of the window) and the window is never fetched.
hipError_t vct_launch_shadow_minmax(const uint32_t* words, uint32_t ebase, int S, uint2* tiles, hipStream_t s);
// tile rows [row0, row1) only (whole frame: 0 .. ceil(H/8)); visibility (k_raster_vis + k_raster_mid) and shading
// (k_gbuffer_shade, the only part that reads the shadow map) are separate calls so that a stream wait can sit between
hipError_t vct_launch_gbuffer_visibility(const VctRasterArgs& a, const float view_proj[16], int W, int H, int row0, int row1,
                                         hipStream_t s);
hipError_t vct_launch_gbuffer_shade(const VctRasterArgs& a, const float view_proj[16], int W, int H, int row0, int row1,
                                    const uint32_t* shadow, uint32_t shadow_ebase, int shadow_size, const uint2* shadow_tiles,
                                    const float light_vp[16], float* tiled, hipStream_t s);
// one level of a texture's mip chain from its parent (pw x ph -> w x h), glGenerateMipmap restated as in the oracle
hipError_t vct_launch_tri_alpha(const VctRasterArgs& a, int32_t* out, hipStream_t s);
hipError_t vct_launch_tex_mip(const uint32_t* parent, int pw, int ph, uint32_t* level, int w, int h, hipStream_t s);
hipError_t vct_launch_untile_gbuffer(const float* tiled, float* planes_linear, int w, int h, hipStream_t s);
hipError_t vct_launch_trace(const VctTraceParams& p, int variant, hipStream_t s);
hipError_t vct_launch_divide_selftest(float d, unsigned long long* mismatches, hipStream_t s);
// typed-buffer texel loads (RGBA8 UNORM -> four floats in the texture path) against the exact decode, n texels
hipError_t vct_launch_texel_buffer_selftest(const uint32_t* texels, uint32_t n, unsigned long long* out, hipStream_t s);
hipError_t vct_launch_area_divide_selftest(uint64_t seed, uint64_t count, unsigned long long* out, hipStream_t s);
hipError_t vct_launch_linear_to_morton(const uint32_t* lin, uint32_t* mor, int N, hipStream_t s);
hipError_t vct_launch_morton_to_linear(const uint32_t* mor, uint32_t* lin, int N, hipStream_t s);
// bricks_now / bricks_seen (optional): per-8^3-brick occupancy flags of level 0 for the sparse form
hipError_t vct_launch_build_mips(uint32_t* chain, int V, const uint32_t* bricks_now, uint32_t* bricks_seen,
                                 hipStream_t s);
hipError_t vct_launch_build_mips_aniso(const uint32_t* level0, uint32_t* aniso, int V, hipStream_t s);
hipError_t vct_launch_build_cells(const uint32_t* chain, uint4* cells, int V, hipStream_t s);
hipError_t vct_launch_tile_gbuffer(const float* planes_linear, float* tiled, int w, int h,
                                   hipStream_t s);
// fragment list of the conservative voxelizer, built at upload: small triangles (+ the list of the big ones), big
// triangles, then the counting sort by brick slot
hipError_t vct_launch_vox_plan(const VctVoxParams& p, uint32_t* plan, uint2* frags,
                               int32_t* big_list, bool write, hipStream_t s);
hipError_t vct_launch_vox_plan_big(const VctVoxParams& p, const int32_t* big_list, int n_big, uint32_t* plan, uint2* frags,
                                   bool write, hipStream_t s);
hipError_t vct_launch_frag_mark(const uint2* frags, uint32_t n, uint32_t* mark, hipStream_t s);
hipError_t vct_launch_frag_count(const uint2* frags, uint32_t n, const uint32_t* brick_slot, uint32_t* count, hipStream_t s);
hipError_t vct_launch_frag_scatter(const uint2* frags, uint32_t n, const uint32_t* brick_slot, const uint32_t* first,
                                   uint32_t* cursor, uint32_t* sorted, uint32_t* slot_brick, hipStream_t s);
hipError_t vct_launch_voxelize(const VctVoxParams& p, hipStream_t s);
hipError_t vct_launch_frag_geom(const VctVoxParams& p, float2* bary, float* frag_alb, hipStream_t s);
hipError_t vct_launch_tri_nrm(const VctVoxParams& p, uint32_t* tri_nrm, hipStream_t s);
hipError_t vct_launch_resolve(unsigned long long* acc, const uint32_t* brick_slot, uint32_t* level0, uint32_t* flags,
                              uint32_t* prev, int V, bool dense, unsigned long long* acc_attr,
                              uint32_t* attr_albedo, uint32_t* attr_normal, bool reference, const uint32_t* stage,
                              const uint32_t* stage_albedo, const uint32_t* stage_normal, hipStream_t s);
// mark[b] != 0 -> slot[b] = next free slot (order irrelevant), else VCT_NO_SLOT; *count = slots handed out
hipError_t vct_launch_assign_slots(const uint32_t* mark, uint32_t* slot, uint32_t* count, uint32_t nbricks, hipStream_t s);
hipError_t vct_launch_slot_bricks(const uint32_t* slot, uint32_t nbricks, uint32_t* slot_brick, hipStream_t s);
// pooled per-voxel attribute -> dense Morton volume (downloads)
hipError_t vct_launch_unpool(const uint32_t* pooled, const uint32_t* brick_slot, uint32_t* dense, uint32_t nbricks, hipStream_t s);
hipError_t vct_launch_voxelize_reference(const VctVoxParams& p, int32_t* big_list, int32_t* big_count,
                                         hipStream_t s);
hipError_t vct_launch_bounce(const VctTraceParams& p, hipStream_t s);

#endif
