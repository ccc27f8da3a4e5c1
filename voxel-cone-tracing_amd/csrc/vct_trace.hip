// vct_trace.hip -- per-pixel cone trace (6 diffuse + 1 specular) through the Morton brick chain.
//
// Replaces the fragment stage of the reference's main draw: S/VoxelConeTracing.fs:59-66
// (SampleVoxels), :82-107 (Voxel_Cone_Tracing), :165-228 (gather + composite), with the driver's
// textureLod (trilinear x 2 levels, GL_REPEAT, texel centres -- OpenGL 4.3 core)
// written out by hand because there is no texture unit behind a HIP pointer.
//
// Arithmetic contract: the march is operation-for-operation the scalar oracle's (fp32, explicit
// fmaf only where the oracle has one or where the fused form is provably the same bits), so
// per-cone step counts and the raw cone vec4s are bit-identical; compile with -ffp-contract=off.
//
// Kernels in this file: k_trace_tile_split (default: an 8x8 screen tile = 3 waves -- cones 0-2, cones
// 3-5, specular -- with an LDS hand-off and a last-arriver composite), k_trace_tile (one wave per
// tile; A/B variants), k_bounce_list/_march/_bricks (second bounce, same march), k_divide_selftest.
//
// Mapping: lane = pixel of the tile, a wave marches its cones one after the other.  The kernel
// was VALU-issue bound from round 1 (profiles/r01a: 995 M VALU wave-instructions per 1080p frame, half of the
// wave cycles spent waiting to issue, 12 % waiting on memory) to round 5 (543 M), and on gfx950 only fp32
// fma/mul/add and plain logic ops issue in 2 cycles per wave -- conversions, floor, bit-field,
// 3-operand integer ops take 4 (tools/valu_bench.hip).  The dominant cost per march step was
// turning 2 x 8 RGBA8 texels into 64 floats (cvt + exact /255) and the Morton address of each, so
// the sampler is organised to do that work once per wave instead of once per lane -- and since round 6 the
// conversion itself is left to the texture path: a level is read as an RGBA8 UNORM TEXEL BUFFER (typed-buffer
// loads, level_texel_buffer / texel_f32 below), whose UNORM8 -> fp32 conversion is bit for bit (float)c / 255.0f
// on this GPU (tools/unorm_probe.hip, vct_selftest_texel_buffer).  447 M VALU wave-instructions per frame now,
// vector pipes 83 % busy, 38 % of the wave-cycles waiting on memory (profiles/r06f_final.txt).
//
//   cooperative sample (the common case): the 64 trilinear footprints of a tile at one march
//   step almost always fall inside one 4x4x4 texel block of the level (neighbouring pixels trace
//   near-parallel cones; ~85 % of wave-level samples of the 1080p bench frame).  The block is
//   anchored at the centre pixel's footprint; lane l fetches block texel (l&3, (l>>2)&3, l>>4)
//   with ONE load (Morton index = scalar-unit spread of the anchor + a per-lane dilated-integer
//   add; the four channels arrive as floats), parks the float4 in a wave-private 1 KiB LDS slab, and
//   every lane gathers its own 8 texels with ds_read_b128 at constant offsets {0,1,4,5,16,..}.
//   If the whole block is zero (ballot) the sample is exactly 0 and everything else is skipped.
//
//   per-lane sample (incoherent waves: silhouettes, random G-buffers): 8 texel loads per lane as in a
//   plain gather (phases of 2 + 2 + 4), the dilated coordinates from the anchor path's table, the +1
//   neighbours derived by dilated increments.
//
// Both produce the same bits.  Divisions by wave-uniform constants use one FMA correction round
// instead of the 10-instruction IEEE sequence, for divisors the device has verified exhaustively
// (vct_capi.hip: divisor_verified).
#include <hip/hip_fp16.h>

#include "vct_internal.h"

typedef int vct_v4i32 __attribute__((ext_vector_type(4)));
typedef float vct_v4f32 __attribute__((ext_vector_type(4)));
__device__ vct_v4f32 vct_struct_buffer_load_format_v4f32(vct_v4i32 rsrc, int vindex, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.struct.buffer.load.format.v4f32");

namespace {

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// IEEE-correct fp32 divide / sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; the
// __fsqrt_rn intrinsic maps to the 1-ulp native sqrt and must not be used here).
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ F3 normalize3(F3 a) {
    const float l = sqrt_rn(dot3(a, a));
    return {div_rn(a.x, l), div_rn(a.y, l), div_rn(a.z, l)};
}
__device__ __forceinline__ F3 reflect3(F3 I, F3 N) {
    const float d = 2.0f * dot3(N, I);
    return {I.x - d * N.x, I.y - d * N.y, I.z - d * N.z};
}

// x / d for a wave-uniform divisor d: the IEEE division costs v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup.
// Rounds 1-3: q = x*r; e = fma(-d, q, x); q = fma(e, r, q) with r = RN(1/d) -- three instructions, one correction
// round.  Round 4 (VCT_DIV2): t = x * r_lo; q = fma(x, r_hi, t) with r_hi = RN(1/d), r_lo = RN(1/d - r_hi) -- TWO
// instructions: r_hi + r_lo is 1/d to ~48 bits, the fma adds the two products exactly and rounds once, so q is the
// correctly rounded quotient unless x/d lies within ~2^-47 of a rounding boundary.  Neither form is correctly rounded
// for every divisor, so both are used only for divisors the DEVICE has verified: before a step table is used,
// vct_capi.hip runs k_divide_selftest for each of its divisors (half_G and the per-step occlusion denominators) over
// EVERY fp32 x of the domain below and requires the IEEE quotient bit for bit (results cached per divisor; all 334
// divisors of the BASELINE grids and apertures pass, in either form: csrc/vct_divisors.h); a table with a divisor that
// fails runs the IEEE-divide instantiation.  Atrium 0.6177 -> 0.6107 ms, street at 1024^3 / 4K 2.727 -> 2.711 ms.
// Domain: x == +0 and every finite |x| >= 2^-100 whose quotient is a normal number (host-side precondition on
// d: vct_capi.hip divisor_ok; below that the low product x * r_lo -- or the remainder e of the older form -- loses
// bits to underflow; the sign of a zero quotient is not preserved).
// The march stays inside the domain by construction:
//   * coordinates: |x| < 2^-100 or x == -0 gives |q| < 2^-26, and u = fma(q, .5, .5) = 0.5 for any
//     such q, exactly as with the IEEE quotient;
//   * occlusion numerator oma * vc.w: never -0, and either +0 or >= 2^-98 -- a non-zero filter
//     fraction is >= 2^-25 (u = fma(ux, N, -0.5) is exact and a multiple of 2^-25 near 0), so a
//     non-zero trilinear weight is >= 2^-75, a non-zero texel >= 1/255, the level blend factors are
//     0 or >= 2^-10 and oma >= 2^-5 (both checked on the host: vct_capi.hip refresh_steps).
#define VCT_DIV_TINY 0x1p-100f
// MODE 0: the IEEE division, 1: the verified form (two-term product, or one correction round without VCT_DIV2), 2: x * r alone -- NOT exact: only the opt-in "loose" trace
// variant that prices the exactness (config.trace_variant = 3, k_trace_tile_split<.., 2, ..>) uses it
// `d`: the divisor -- or, for MODE 1 under VCT_DIV2, the low word r_lo of its reciprocal (the host puts it where the
// divisor used to be: VctStep::occ_den, VctTraceParams::half_G_aux)
template <int MODE>
__device__ __forceinline__ float div_const(float x, float d, float r) {
    if (MODE == 0) return x / d;
    if (MODE == 2) return x * r;
#if VCT_DIV2
    // x / d = x * (r_hi + r_lo) (1 + e), |e| < 2^-47: the one rounding of the fma is the rounding of the quotient unless
    // x / d lies within 2^-47 of a rounding boundary -- which the device rules out per divisor, over every fp32 x
    return fmaf(x, r, x * d);
#else
    const float q = x * r;
    const float e = fmaf(-d, q, x);
    return fmaf(e, r, q);
#endif
}

// unorm8 -> float, bit-identical to (float)c / 255.0f for every c in [0,255]:
// c * RN(1/255) misses for 126 of the 256 bytes; the two-term product below never does
// (checked exhaustively in tests/test_abi.py::test_unorm8_decode_exact).
__device__ __forceinline__ float unorm8(uint32_t c) { return vct_unorm8_to_float(c); }

// A level of the chain as a TEXEL BUFFER: buffer resource with stride 4 and format 8_8_8_8 UNORM; the structured load takes
// the texel's Morton INDEX (a 4 GiB level -- 1024^3 level 0 -- is addressed in full; the raw byte-offset form fails its
// range check on that level's last texel) and returns the four channels converted by the texture path.  Declared like
// composable_kernel declares its buffer loads, so that the compiler tracks the load's completion itself.
// (vct_v4i32 / vct_v4f32 / vct_struct_buffer_load_format_v4f32: declared in front of this namespace)
__device__ __forceinline__ vct_v4i32 level_texel_buffer(const uint32_t* level_base) {
    const uint64_t a = (uint64_t)level_base;
    vct_v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(((uint32_t)(a >> 32) & 0xffffu) | (4u << 16)));     // base[47:32] | stride 4
    r.z = 0x40000000;                   // records: no level has more than 2^30 texels
    r.w = 0x50fac;                      // DST_SEL x,y,z,w = R,G,B,A | NUM_FORMAT_UNORM << 12 | DATA_FORMAT_8_8_8_8 << 15
    return r;
}
__device__ __forceinline__ float4 texel_f32(vct_v4i32 rsrc, uint32_t index) {
    const vct_v4f32 v = vct_struct_buffer_load_format_v4f32(rsrc, (int)index, 0, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}

// LDS operations of one wave execute in order, so a slab written and then read by the lanes of
// the same wave needs no s_barrier -- only the compiler must not reorder across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// HIP's __ballot()/__any() lower to v_cndmask + v_cmp_ne; the builtin is the compare mask itself.
__device__ __forceinline__ unsigned long long ballot64(bool pred) {
    return __builtin_amdgcn_ballot_w64(pred);
}

// Dilated anchor coordinates come from a table: computing spread3() of three scalars took 41 scalar-unit instructions
// per level sample, and the scalar pipe issues one instruction per 4 cycles per SIMD (tools/valu_bench.hip: s_add /
// s_and / s_mul 4.17 cycles at any occupancy) -- with half as many SALU as VALU instructions it was 68 % busy, and a
// padding experiment showed the march paying 2.1 us per scalar instruction added to a step against 1.4 us per vector
// one.  One s_load_dword per axis replaces 13 instructions.  Entry i = spread3(i) << 2 (byte offset of the x axis).
typedef const __attribute__((address_space(4))) uint32_t* SpreadLut;
// byte offset of x coordinate (a & m) in dilated form; m4 = m << 2.  All 32-bit, so the table entry is one
// s_load_dword with an SGPR offset.
__device__ __forceinline__ uint32_t spread_byte(SpreadLut lut, int a, uint32_t m4) {
    const uint32_t off = ((uint32_t)a << 2) & m4;
    return *(SpreadLut)((const __attribute__((address_space(4))) char*)lut + off);
}

// EXPERIMENT / option (-DVCT_LUT=1 | 2 | 3; round 4): the exact unorm8 -> float decode through a
// 256-entry table in LDS (1 KiB per workgroup, entry i = (float)i / 255.0f computed by the same two-term product) --
// one SDWA byte-select shift + one ds_read_b32 per channel instead of cvt + mul + fma.  Bit 0: the cooperative
// block's four decodes per lane, bit 1: the per-lane gather's 32.  Same bits either way; measurements in
// profiles/experiments/README.md.
#ifndef VCT_LUT
#define VCT_LUT 0
#endif
typedef const __attribute__((address_space(3))) float* UnormLut;
struct LaneBlock {      // this lane's texel inside the cooperative 4x4x4 block
    SpreadLut lut;              // (wave-uniform) the dilated-coordinate table
    const uint32_t* lut_vec;    // the same table through a plain global pointer (per-lane gather: VCT_LANE_SPREAD_LUT)
    const __attribute__((address_space(3))) uint32_t* lut_lds = nullptr;   // VCT_LANE_SPREAD_LUT == 2: a copy in LDS (split kernel)
    int lane;
    uint32_t sbx, sby, sbz;     // texel-INDEX offsets: dilated (l&3), ((l>>2)&3)<<1, (l>>4)<<2
    UnormLut unorm;             // VCT_LUT: the decode table in LDS
};
#if VCT_LUT
#define VCT_LUT_DECL __shared__ float lds_unorm[256];
// (called by every thread of the workgroup before its first barrier / before any sample)
#define VCT_LUT_FILL(lb)                                                                                      \
    for (uint32_t i_ = threadIdx.x; i_ < 256u; i_ += blockDim.x) lds_unorm[i_] = vct_unorm8_to_float(i_);     \
    (lb).unorm = (UnormLut)lds_unorm;
#else
#define VCT_LUT_DECL
#define VCT_LUT_FILL(lb) (lb).unorm = nullptr;
#endif
template <int SHIFT, bool LOOSE = false>
__device__ __forceinline__ float unorm8_of(UnormLut lut, uint32_t t, bool use_lut) {
    if (LOOSE) return (float)((t >> SHIFT) & 0xffu) * 0x1.010102p-8f;     // one multiply: wrong in the last bit for 126 of the 256 bytes
    if (use_lut) return lut[(t >> SHIFT) & 0xffu];
    return vct_unorm8_to_float((t >> SHIFT) & 0xffu);
}

// Instrumented build (-DVCT_STATS=1, tools/trace_stats.py): wave-level counters of the march, kept in
// SGPRs and flushed once per wave.  The production build compiles all of it away.
#ifndef VCT_STATS
#define VCT_STATS 0
#endif
#ifndef VCT_UNROLL2
#define VCT_UNROLL2 1     // A/B: 0.6281 -> 0.6216 ms at 256^3, 2.659 -> 2.623 ms at 512^3 / 4K
#endif
#ifndef VCT_FRACT
#define VCT_FRACT 0       // A/B (round 3, vector pipes binding): 0.6184 / 0.6210 ms without, 0.6194 / 0.6178 ms with: no gain (v_fract and
                          // v_cvt_flr are 4-cycle operations: 8 cycles per axis against 4 + 2 + 4)
#endif
#ifndef VCT_HALF_GATHER
#define VCT_HALF_GATHER 1     // gather + interpolate the lower z plane, then the upper one (half the texel registers live)
#endif
#ifndef VCT_CELLS
#define VCT_CELLS 1           // footprint records (round 5; vct_set_footprint_records): a per-lane sample of a level >= 1 is ONE
                              // 32-byte fetch of the footprint's 8 texels (vct_volume.hip k_build_cells) instead of eight 4-byte
                              // ones from 2-4 cache lines.  Same bits.  Dense random 1024^3 chain + random G-buffer (the one
                              // HBM-bound case): 5.61 -> 2.86 ms; cache-resident scenes: street at 1024^3 / 4K 2.72 -> 2.70 ms,
                              // atrium 0.611 -> 0.613 (profiles/experiments/README.md).  Off unless the context asks for it.
#endif
#ifndef VCT_PAIR_LOAD
#define VCT_PAIR_LOAD 0       // EXPERIMENT (round 5, review item 8): per-lane gather with the x-adjacent texel pair of an even x in one
                              // 8-byte load (Morton order keeps (x, x+1) adjacent for even x); odd lanes fetch x + 1 with a masked
                              // 4-byte load.  Same bits.  Result: profiles/experiments/README.md
#endif
#ifndef VCT_LANE_SPREAD_LUT
#define VCT_LANE_SPREAD_LUT 1  // round 6: the per-lane gather's dilated coordinates by table look-up (profiles/experiments/README.md)
#endif
#ifndef VCT_HW_UNORM
#define VCT_HW_UNORM 1         // round 6: texels are fetched through the texture path's typed-buffer load (RGBA8 UNORM, stride 4),
                               // which returns a texel as four floats -- bit for bit (float)c / 255.0f for every byte
                               // (tools/unorm_probe.hip) -- so the exact decode (cvt + mul + fma per channel) is not issued at all
#endif
#ifndef VCT_ANISO_HW
#define VCT_ANISO_HW 0         // EXPERIMENT: the anisotropic sampler's directional blocks through typed loads too
#endif
#ifndef VCT_LANE_HYBRID
#define VCT_LANE_HYBRID 0
#endif
#ifndef VCT_CELLS_HW
#define VCT_CELLS_HW 0
#endif
#ifndef VCT_LANE_PAIRS
#define VCT_LANE_PAIRS 0
#endif
#ifndef VCT_LANE_224
#define VCT_LANE_224 1         // (A/B: 4 + 4 spills five registers at 72 VGPRs and runs 1-2.5 % slower; pairs only: slower on the street)
#endif
#ifndef VCT_LANE_RECOMPUTE_XY
#define VCT_LANE_RECOMPUTE_XY 0
#endif
#ifndef VCT_QUARTER_GATHER
#define VCT_QUARTER_GATHER 0
#endif
#ifndef VCT_LOAD_PRIO
#define VCT_LOAD_PRIO 1        // (A/B, 4 interleaved rounds: kernel -1.7 %, one-stream step -1.3 %, vct_gi_pass -0.9 % on the atrium; street 4K -0.5 %)
#endif
#ifndef VCT_LOAD_PRIO_MODE
#define VCT_LOAD_PRIO_MODE 1   // (A/B) 1: reset behind the load instructions; 2: cooperative path resets behind its LDS reads; 3: raised from the start of the march step
#endif
#ifndef VCT_CELLS_ARITH
#define VCT_CELLS_ARITH 1      // the footprint-record instantiation keeps the arithmetic (it is memory bound: see sample_level)
#endif
#ifndef VCT_TWO_BLOCKS
#define VCT_TWO_BLOCKS 0      // 1: a second cooperative block before the per-lane gather (profiles/experiments/README.md)
#endif
struct MarchStats {
    uint32_t wave_steps;       // march-loop iterations executed by the wave
    uint32_t lane_steps;       // sum over those iterations of the live lanes (== executed cone steps)
    uint32_t coop_zero;        // level samples served by the cooperative block, block all zero (skipped)
    uint32_t coop_hit;         // level samples served by the cooperative block, gathered through LDS
    uint32_t fallback;         // level samples that took the per-lane gather
    uint32_t fallback_lanes;   // live lanes in those
    uint32_t fallback_fits;    // per-lane samples whose live footprints WOULD fit one 4x4x4 block (anchored at their minimum)
    uint32_t greedy_blocks, greedy_le2, greedy_le3, greedy_le4;   // blocks a greedy multi-anchor cover of them would need
    uint32_t two_blocks;       // level samples served by two cooperative blocks (VCT_TWO_BLOCKS)
    // round 5 (review item 3): would sharing inside smaller lane groups serve the per-lane samples?  Per per-lane sample:
    uint32_t quads_live;       // 2x2-pixel quads with a live lane
    uint32_t quads_fit333;     // ... whose live footprints span <= 1 texel per axis (their union fits a 3x3x3 block)
    uint32_t quads_same;       // ... whose live footprints are one and the same 2x2x2 cell
    uint32_t quadrants_live;   // 4x4-pixel quadrants with a live lane
    uint32_t quadrants_fit444; // ... whose live footprints fit a 4x4x4 block of their own
};

// [GL] tri(level): trilinear, texel centres, REPEAT (or clamp).  `level` is wave-uniform; must be
// called in wave-uniform control flow with at least one lane `act`.  Lanes without `act` help
// fetch the block and return garbage-free zeros / unused values.
template <bool WRAP, bool COOP, bool LOOSE = false, bool CELLS = false, bool PRIO = false>
__device__ __forceinline__ F4 sample_level(const uint32_t* __restrict__ chain, const VctLevelRef lv,
                                           float ux, float uy, float uz, bool act, unsigned long long am,
                                           float4* __restrict__ blk, const LaneBlock& lb, MarchStats& ms,
                                           const char* __restrict__ cells = nullptr) {
    // `am` is ballot64(act), passed in so that compound predicates are ANDed as lane masks on the
    // scalar unit (a ballot of `x && y` costs a v_cndmask + v_cmp_ne pair to materialise the bool).
    // Issue priority raised from here until the sample's loads are out (VCT_LOAD_PRIO, round 6): a wave that is forming
    // addresses gets its instructions ahead of the waves that are folding texels, so its loads leave earlier and more of
    // their latency lies under the other waves' arithmetic.  PRIO is a template parameter: the launches of whole frames
    // take the instantiation with it, slab launches the one without -- there the priority goes to the specular waves, the
    // tail of a short launch (spec_prio; 8-way slabs 0.0881 ms with that against 0.0900 with this).  The same choice behind
    // a wave-uniform flag cost the default kernel 60 B of scratch per lane.
    if (VCT_LOAD_PRIO && PRIO) __builtin_amdgcn_s_setprio(1);
    const int m = lv.m;
    const float fN = lv.fN;
    // ux * fN is exact (power of two), so the fused form is the oracle's (ux*fN) - 0.5f bit for bit
    const float u = fmaf(ux, fN, -0.5f), v = fmaf(uy, fN, -0.5f), w = fmaf(uz, fN, -0.5f);
    float a, b, c;
    int i0, j0, k0;
#if VCT_FRACT
    if (m != 0) {
        // Footprint in two instructions per axis instead of three (v_floor + v_sub + v_cvt): v_fract_f32 and
        // v_cvt_flr_i32_f32.  v_fract returns u - floor(u) except that it never returns 1.0 (it clamps to 1 - 2^-24),
        // while the subtraction rounds up to 1.0 for u in [-2^-25, 0).  For N >= 2 such a u cannot occur here: u < 0
        // means ux < 0.5 / N <= 0.25, and ux = fma(q, 0.5, 0.5) with |q| >= 0.5 there, so ux is a multiple of 2^-25
        // (q's ulp is >= 2^-24 and the sum is exact), u = ux * N - 0.5 is a multiple of 2^-24, and 1 - |u| is
        // representable: the subtraction is exact and equals v_fract.  The one-texel level (N = 1) keeps the
        // subtraction.  Bit-exact in every parity test -- and no faster, twice (round 2, round 3): off by default.
        a = __builtin_amdgcn_fractf(u); b = __builtin_amdgcn_fractf(v); c = __builtin_amdgcn_fractf(w);
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i0) : "v"(u));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(j0) : "v"(v));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k0) : "v"(w));
    } else
#endif
    {
        const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
        a = u - fu; b = v - fv; c = w - fw;
        i0 = (int)fu; j0 = (int)fv; k0 = (int)fw;
    }
    // Texels are addressed by their Morton INDEX inside the level (< 2^30): the dilated-integer arithmetic yields it
    // directly, and the level is a texel buffer whose structured load takes the index (no shift, no 64-bit address add).
    const uint32_t* __restrict__ base = chain + lv.off;
    const uint32_t MX = lv.mask_x, MY = MX << 1, MZ = MX << 2;
    const vct_v4i32 tb = level_texel_buffer(base);

    F4 r = {0.0f, 0.0f, 0.0f, 0.0f};
    bool coop = false;
    int ax = 0, ay = 0, az = 0, dx = 0, dy = 0, dz = 0;
#if VCT_TWO_BLOCKS
    bool two = false, mine = act, mine2 = false;   // mine: this lane's footprint lies in the block being fetched
    int ex = 0, ey = 0, ez = 0;                    // second anchor - first anchor
#endif
    if (COOP) {
        // anchor = footprint of the tile's centre pixel (lane 27) if it is live, else the first live lane
        const int src = ((am >> 27) & 1ull) ? 27 : (int)__ffsll((long long)am) - 1;
        ax = __builtin_amdgcn_readlane(i0, src) - 1;
        ay = __builtin_amdgcn_readlane(j0, src) - 1;
        az = __builtin_amdgcn_readlane(k0, src) - 1;
        dx = i0 - ax; dy = j0 - ay; dz = k0 - az;
        const uint32_t far = max(max((uint32_t)dx, (uint32_t)dy), (uint32_t)dz);
        const unsigned long long out = ballot64(far > 2u) & am;
        coop = out == 0ull;
#if VCT_TWO_BLOCKS
        if (!coop) {
            // Second chance before the per-lane gather: a second block anchored at the first live lane the first one
            // misses.  If the two blocks hold every live footprint, each is fetched cooperatively in turn and a lane
            // gathers from the one that holds its footprint (any block that holds the 8 texels gives the same bits).
            const int s2 = (int)__ffsll((long long)out) - 1;
            ex = __builtin_amdgcn_readlane(dx, s2) - 1;
            ey = __builtin_amdgcn_readlane(dy, s2) - 1;
            ez = __builtin_amdgcn_readlane(dz, s2) - 1;
            const uint32_t far2 = max(max((uint32_t)(dx - ex), (uint32_t)(dy - ey)), (uint32_t)(dz - ez));
            two = (ballot64(far2 > 2u) & out) == 0ull;
            coop = two;
            mine = act && far <= 2u;
            mine2 = act && far > 2u;
        }
#endif
    }
#if VCT_TWO_BLOCKS
    if (COOP && coop)
#pragma unroll 1
    for (int pass = 0;; ++pass) {
#else
    if (COOP && coop) {
#endif
        uint32_t idx;
        if (WRAP) {
            // scalar unit: dilate the anchor; vector unit: one dilated add per axis
            const uint32_t m4 = (uint32_t)m << 2;
            // (the table holds spread3(i) << 2: the scalar unit shifts it into place -- one shift for two of the axes, as before)
            const uint32_t sax = spread_byte(lb.lut, ax, m4) >> 2;
            const uint32_t say = spread_byte(lb.lut, ay, m4) >> 1;
            const uint32_t saz = spread_byte(lb.lut, az, m4);
            idx = (((sax | ~MX) + lb.sbx) & MX) | (((say | ~MY) + lb.sby) & MY) |
                  (((saz | ~MZ) + lb.sbz) & MZ);
        } else {
            const int x = min(max(ax + (lb.lane & 3), 0), m);
            const int y = min(max(ay + ((lb.lane >> 2) & 3), 0), m);
            const int z = min(max(az + (lb.lane >> 4), 0), m);
            idx = vct_morton3((uint32_t)x, (uint32_t)y, (uint32_t)z);
        }
#if VCT_HW_UNORM && !VCT_LUT
        const float4 d = texel_f32(tb, idx);
        if (VCT_LOAD_PRIO && PRIO && VCT_LOAD_PRIO_MODE != 2) __builtin_amdgcn_s_setprio(0);
        // (channels are >= +0: the block is empty iff no channel has a bit set)
        const bool any_texel = ballot64((__float_as_uint(d.x) | __float_as_uint(d.y) | __float_as_uint(d.z) | __float_as_uint(d.w)) != 0u) != 0ull;
#else
        const uint32_t t = base[idx];
        const bool any_texel = ballot64(t != 0u) != 0ull;
#endif
        if (VCT_STATS) { if (any_texel) ++ms.coop_hit; else ++ms.coop_zero; }
        if (VCT_LOAD_PRIO && PRIO && VCT_LOAD_PRIO_MODE == 2 && !any_texel) __builtin_amdgcn_s_setprio(0);
        if (any_texel) {     // all 64 texels zero: every footprint sums to exactly +0
#if !(VCT_HW_UNORM && !VCT_LUT)
            float4 d;
            d.x = unorm8_of<0, LOOSE>(lb.unorm, t, (VCT_LUT & 1) != 0);
            d.y = unorm8_of<8, LOOSE>(lb.unorm, t, (VCT_LUT & 1) != 0);
            d.z = unorm8_of<16, LOOSE>(lb.unorm, t, (VCT_LUT & 1) != 0);
            d.w = unorm8_of<24, LOOSE>(lb.unorm, t, (VCT_LUT & 1) != 0);
#endif
            blk[lb.lane] = d;
            wave_sync();
#if VCT_TWO_BLOCKS
            if (mine) {
#endif
            const int slot = act ? (dz * 4 + dy) * 4 + dx : 0;
            const float4* q = blk + slot;
            const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
            const float ab00 = a0 * b0, ab10 = a * b0, ab01 = a0 * b, ab11 = a * b;
#if VCT_QUARTER_GATHER
            {   // EXPERIMENT (round 6): two texels at a time -- 8 texel registers live instead of 16 (for 64 VGPRs / 8 waves per SIMD)
                const float w0 = ab00 * c0, w1 = ab10 * c0;
                { const float4 t0 = q[0], t1 = q[1];
#define VCT_ACC(ch) r.ch = w0 * t0.ch; r.ch = fmaf(w1, t1.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
                }
                __builtin_amdgcn_sched_barrier(0);
                const float w2 = ab01 * c0, w3 = ab11 * c0;
                { const float4 t2 = q[4], t3 = q[5];
#define VCT_ACC(ch) r.ch = fmaf(w2, t2.ch, r.ch); r.ch = fmaf(w3, t3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
                }
                __builtin_amdgcn_sched_barrier(0);
                const float w4 = ab00 * c, w5 = ab10 * c;
                { const float4 t4 = q[16], t5 = q[17];
#define VCT_ACC(ch) r.ch = fmaf(w4, t4.ch, r.ch); r.ch = fmaf(w5, t5.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
                }
                __builtin_amdgcn_sched_barrier(0);
                const float w6 = ab01 * c, w7 = ab11 * c;
                { const float4 t6 = q[20], t7 = q[21];
                wave_sync();
#define VCT_ACC(ch) r.ch = fmaf(w6, t6.ch, r.ch); r.ch = fmaf(w7, t7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
                }
            }
#elif VCT_HALF_GATHER
            {   // lower z plane first, then the upper one: half the texel registers live at a time
                const float4 t0 = q[0], t1 = q[1], t2 = q[4], t3 = q[5];
                const float w0 = ab00 * c0, w1 = ab10 * c0, w2 = ab01 * c0, w3 = ab11 * c0;
#define VCT_ACC(ch) r.ch = w0 * t0.ch; r.ch = fmaf(w1, t1.ch, r.ch); r.ch = fmaf(w2, t2.ch, r.ch); r.ch = fmaf(w3, t3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float4 t4 = q[16], t5 = q[17], t6 = q[20], t7 = q[21];
                wave_sync();
                const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch) r.ch = fmaf(w4, t4.ch, r.ch); r.ch = fmaf(w5, t5.ch, r.ch); r.ch = fmaf(w6, t6.ch, r.ch); r.ch = fmaf(w7, t7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
#else
            const float4 t0 = q[0], t1 = q[1], t2 = q[4], t3 = q[5];
            const float4 t4 = q[16], t5 = q[17], t6 = q[20], t7 = q[21];
            wave_sync();
            if (VCT_LOAD_PRIO && PRIO && VCT_LOAD_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(0);
            const float w0 = ab00 * c0, w1 = ab10 * c0, w2 = ab01 * c0, w3 = ab11 * c0;
            const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch)                                                                           \
    r.ch = w0 * t0.ch;                                                                        \
    r.ch = fmaf(w1, t1.ch, r.ch); r.ch = fmaf(w2, t2.ch, r.ch); r.ch = fmaf(w3, t3.ch, r.ch); \
    r.ch = fmaf(w4, t4.ch, r.ch); r.ch = fmaf(w5, t5.ch, r.ch); r.ch = fmaf(w6, t6.ch, r.ch); \
    r.ch = fmaf(w7, t7.ch, r.ch);
            VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
#endif
#if VCT_TWO_BLOCKS
            }                              // a lane of the other block keeps what it has
#endif
        }
#if VCT_TWO_BLOCKS
        if (!two || pass == 1) break;
        if (VCT_STATS) ++ms.two_blocks;
        ax += ex; ay += ey; az += ez;
        dx -= ex; dy -= ey; dz -= ez;
        mine = mine2;
#endif
    } else {
      if (VCT_STATS) {
          ++ms.fallback; ms.fallback_lanes += (uint32_t)__popcll(am);
          int lo[3] = {act ? i0 : 0x7fffffff, act ? j0 : 0x7fffffff, act ? k0 : 0x7fffffff};
          int hi[3] = {act ? i0 : -0x7fffffff, act ? j0 : -0x7fffffff, act ? k0 : -0x7fffffff};
          for (int off = 32; off > 0; off >>= 1)
              for (int q = 0; q < 3; ++q) { lo[q] = min(lo[q], __shfl_xor(lo[q], off)); hi[q] = max(hi[q], __shfl_xor(hi[q], off)); }
          if (hi[0] - lo[0] <= 2 && hi[1] - lo[1] <= 2 && hi[2] - lo[2] <= 2) ++ms.fallback_fits;
          // greedy cover: blocks anchored at the first lane not yet covered
          unsigned long long pending = am;
          int nb = 0;
          while (pending != 0ull && nb < 16) {
              const int src = (int)__ffsll((long long)pending) - 1;
              const int bx = __builtin_amdgcn_readlane(i0, src) - 1, by = __builtin_amdgcn_readlane(j0, src) - 1,
                        bz = __builtin_amdgcn_readlane(k0, src) - 1;
              const bool in = (uint32_t)(i0 - bx) <= 2u && (uint32_t)(j0 - by) <= 2u && (uint32_t)(k0 - bz) <= 2u;
              pending &= ~ballot64(in);
              ++nb;
          }
          // the same range test inside 2x2-pixel quads (lanes l, l^1, l^8, l^9) and 4x4-pixel quadrants (+ l^2, l^16, ...)
          int qlo[3] = {act ? i0 : 0x7fffffff, act ? j0 : 0x7fffffff, act ? k0 : 0x7fffffff};
          int qhi[3] = {act ? i0 : -0x7fffffff, act ? j0 : -0x7fffffff, act ? k0 : -0x7fffffff};
          bool qany = act;
          auto widen = [&](int off) {
              for (int q = 0; q < 3; ++q) { qlo[q] = min(qlo[q], __shfl_xor(qlo[q], off)); qhi[q] = max(qhi[q], __shfl_xor(qhi[q], off)); }
              qany = qany || (__shfl_xor((int)qany, off) != 0);
          };
          widen(1); widen(8);
          const int span2 = max(max(qhi[0] - qlo[0], qhi[1] - qlo[1]), qhi[2] - qlo[2]);
          const bool lead2 = (lb.lane & 9) == 0;                      // one lane per quad
          ms.quads_live += (uint32_t)__popcll(ballot64(lead2 && qany));
          ms.quads_fit333 += (uint32_t)__popcll(ballot64(lead2 && qany && span2 <= 1));
          ms.quads_same += (uint32_t)__popcll(ballot64(lead2 && qany && span2 == 0));
          widen(2); widen(16);
          const int span4 = max(max(qhi[0] - qlo[0], qhi[1] - qlo[1]), qhi[2] - qlo[2]);
          const bool lead4 = (lb.lane & 27) == 0;                     // one lane per quadrant
          ms.quadrants_live += (uint32_t)__popcll(ballot64(lead4 && qany));
          ms.quadrants_fit444 += (uint32_t)__popcll(ballot64(lead4 && qany && span4 <= 2));
          ms.greedy_blocks += (uint32_t)nb;
          if (nb <= 2) ++ms.greedy_le2;
          if (nb <= 3) ++ms.greedy_le3;
          if (nb <= 4) ++ms.greedy_le4;
      }
      if (act) {
        uint32_t mx0, mx1, my0, my1, mz0, mz1;
        if (WRAP) {
            // (the instantiation with footprint records serves volumes that do NOT fit the caches: every sample is a per-lane
            // one and the memory pipe is what binds -- three more loads per sample cost it 1-3 %, VCT_CELLS_ARITH)
            constexpr bool use_lut = VCT_LANE_SPREAD_LUT != 0 && !(CELLS && VCT_CELLS_ARITH);
            if (use_lut) {
            // the dilated coordinates from the table the anchor path reads with scalar loads -- here one 4-byte VECTOR load
            // per axis (a 4 KiB table, cache resident) instead of ten vector instructions, half of them 4-cycle ones
            const uint32_t m4 = (uint32_t)m << 2;
            if (VCT_LANE_SPREAD_LUT == 2 && lb.lut_lds) {        // (A/B form: the table in LDS, filled per workgroup)
                const __attribute__((address_space(3))) char* ll = (const __attribute__((address_space(3))) char*)lb.lut_lds;
                mx0 = *(const __attribute__((address_space(3))) uint32_t*)(ll + (((uint32_t)i0 << 2) & m4)) >> 2;
                my0 = *(const __attribute__((address_space(3))) uint32_t*)(ll + (((uint32_t)j0 << 2) & m4)) >> 1;
                mz0 = *(const __attribute__((address_space(3))) uint32_t*)(ll + (((uint32_t)k0 << 2) & m4));
            } else {
            const char* lutb = (const char*)lb.lut_vec;      // (entries are spread3(i) << 2)
            mx0 = *(const uint32_t*)(lutb + (((uint32_t)i0 << 2) & m4)) >> 2;
            my0 = *(const uint32_t*)(lutb + (((uint32_t)j0 << 2) & m4)) >> 1;
            mz0 = *(const uint32_t*)(lutb + (((uint32_t)k0 << 2) & m4));
            }
            } else {
            mx0 = vct_spread3((uint32_t)i0 & (uint32_t)m);
            my0 = vct_spread3((uint32_t)j0 & (uint32_t)m) << 1;
            mz0 = vct_spread3((uint32_t)k0 & (uint32_t)m) << 2;
            }
            mx1 = ((mx0 | ~MX) + 1u) & MX;      // dilated increment, wraps at N
            my1 = ((my0 | ~MY) + 2u) & MY;
            mz1 = ((mz0 | ~MZ) + 4u) & MZ;
        } else {
            const int ci0 = min(max(i0, 0), m), ci1 = min(max(i0 + 1, 0), m);
            const int cj0 = min(max(j0, 0), m), cj1 = min(max(j0 + 1, 0), m);
            const int ck0 = min(max(k0, 0), m), ck1 = min(max(k0 + 1, 0), m);
            mx0 = vct_spread3((uint32_t)ci0); mx1 = vct_spread3((uint32_t)ci1);
            my0 = vct_spread3((uint32_t)cj0) << 1; my1 = vct_spread3((uint32_t)cj1) << 1;
            mz0 = vct_spread3((uint32_t)ck0) << 2; mz1 = vct_spread3((uint32_t)ck1) << 2;
        }
        constexpr bool hw_texels = VCT_HW_UNORM && !VCT_LUT && !VCT_PAIR_LOAD && !(VCT_CELLS && CELLS);
        if (VCT_CELLS_HW && VCT_HW_UNORM && VCT_CELLS && CELLS && WRAP && lv.off != 0u) {
            // EXPERIMENT: the footprint record read as eight texels of a texel buffer laid over the records (decoded by the
            // texture path; eight 4-byte requests into one 32-byte sector instead of two 16-byte ones + 32 decodes)
            const vct_v4i32 cbuf = level_texel_buffer((const uint32_t*)(cells + ((size_t)lv.off << 5)));
            const uint32_t ri = (mx0 | my0 | mz0) << 3;
            const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
            const float ab00 = a0 * b0, ab10 = a * b0, ab01 = a0 * b, ab11 = a * b;
            {
                const float4 f0 = texel_f32(cbuf, ri), f1 = texel_f32(cbuf, ri + 1u), f2 = texel_f32(cbuf, ri + 2u), f3 = texel_f32(cbuf, ri + 3u);
                const float w0 = ab00 * c0, w1 = ab10 * c0, w2 = ab01 * c0, w3 = ab11 * c0;
#define VCT_ACC(ch) r.ch = w0 * f0.ch; r.ch = fmaf(w1, f1.ch, r.ch); r.ch = fmaf(w2, f2.ch, r.ch); r.ch = fmaf(w3, f3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float4 f4 = texel_f32(cbuf, ri + 4u), f5 = texel_f32(cbuf, ri + 5u), f6 = texel_f32(cbuf, ri + 6u), f7 = texel_f32(cbuf, ri + 7u);
                const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch) r.ch = fmaf(w4, f4.ch, r.ch); r.ch = fmaf(w5, f5.ch, r.ch); r.ch = fmaf(w6, f6.ch, r.ch); r.ch = fmaf(w7, f7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
        } else
        if (hw_texels) {
            // eight typed-buffer loads: every texel arrives as four floats; one z plane at a time
            const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
            const float ab00 = a0 * b0, ab10 = a * b0, ab01 = a0 * b, ab11 = a * b;
#if VCT_LANE_PAIRS
            // two texels at a time (8 texel registers live): four dependent round trips, but no spill at 72 VGPRs
            {
                const uint32_t xy00 = mx0 | my0, xy10 = mx1 | my0;
                const float4 f0 = texel_f32(tb, xy00 | mz0), f1 = texel_f32(tb, xy10 | mz0);
                const float w0 = ab00 * c0, w1 = ab10 * c0;
#define VCT_ACC(ch) r.ch = w0 * f0.ch; r.ch = fmaf(w1, f1.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const uint32_t xy01 = mx0 | my1, xy11 = mx1 | my1;
                const float4 f2 = texel_f32(tb, xy01 | mz0), f3 = texel_f32(tb, xy11 | mz0);
                const float w2 = ab01 * c0, w3 = ab11 * c0;
#define VCT_ACC(ch) r.ch = fmaf(w2, f2.ch, r.ch); r.ch = fmaf(w3, f3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const uint32_t xy00 = mx0 | my0, xy10 = mx1 | my0;
                const float4 f4 = texel_f32(tb, xy00 | mz1), f5 = texel_f32(tb, xy10 | mz1);
                const float w4 = ab00 * c, w5 = ab10 * c;
#define VCT_ACC(ch) r.ch = fmaf(w4, f4.ch, r.ch); r.ch = fmaf(w5, f5.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const uint32_t xy01 = mx0 | my1, xy11 = mx1 | my1;
                const float4 f6 = texel_f32(tb, xy01 | mz1), f7 = texel_f32(tb, xy11 | mz1);
                const float w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch) r.ch = fmaf(w6, f6.ch, r.ch); r.ch = fmaf(w7, f7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
#else
#if VCT_LANE_224
            // phases of 2 + 2 + 4 texels: the first phase is where most else is still live (both planes' addresses, all weight
            // inputs), the last one where least is
            {
                const float4 f0 = texel_f32(tb, mx0 | my0 | mz0), f1 = texel_f32(tb, mx1 | my0 | mz0);
                const float w0 = ab00 * c0, w1 = ab10 * c0;
#define VCT_ACC(ch) r.ch = w0 * f0.ch; r.ch = fmaf(w1, f1.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float4 f2 = texel_f32(tb, mx0 | my1 | mz0), f3 = texel_f32(tb, mx1 | my1 | mz0);
                const float w2 = ab01 * c0, w3 = ab11 * c0;
#define VCT_ACC(ch) r.ch = fmaf(w2, f2.ch, r.ch); r.ch = fmaf(w3, f3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
#if VCT_LANE_HYBRID
            {   // EXPERIMENT: the upper plane through plain loads + the exact decode (4 B per lane and load through the data
                // return path instead of 16): balances the vector pipes against the texture path on per-lane-heavy scenes
                const uint32_t t4 = base[mx0 | my0 | mz1], t5 = base[mx1 | my0 | mz1];
                const uint32_t t6 = base[mx0 | my1 | mz1], t7 = base[mx1 | my1 | mz1];
                const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch, sh) r.ch = fmaf(w4, vct_unorm8_to_float((t4 >> sh) & 0xffu), r.ch); r.ch = fmaf(w5, vct_unorm8_to_float((t5 >> sh) & 0xffu), r.ch); \
                        r.ch = fmaf(w6, vct_unorm8_to_float((t6 >> sh) & 0xffu), r.ch); r.ch = fmaf(w7, vct_unorm8_to_float((t7 >> sh) & 0xffu), r.ch);
                VCT_ACC(x, 0) VCT_ACC(y, 8) VCT_ACC(z, 16) VCT_ACC(w, 24)
#undef VCT_ACC
            }
#else
            {
                const float4 f4 = texel_f32(tb, mx0 | my0 | mz1), f5 = texel_f32(tb, mx1 | my0 | mz1);
                const float4 f6 = texel_f32(tb, mx0 | my1 | mz1), f7 = texel_f32(tb, mx1 | my1 | mz1);
                if (VCT_LOAD_PRIO && PRIO) __builtin_amdgcn_s_setprio(0);
                const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch) r.ch = fmaf(w4, f4.ch, r.ch); r.ch = fmaf(w5, f5.ch, r.ch); r.ch = fmaf(w6, f6.ch, r.ch); r.ch = fmaf(w7, f7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
#endif
#elif VCT_LANE_RECOMPUTE_XY
            // (the x|y parts are formed again for the upper plane instead of being kept across the lower one)
            {
                const float4 f0 = texel_f32(tb, mx0 | my0 | mz0), f1 = texel_f32(tb, mx1 | my0 | mz0);
                const float4 f2 = texel_f32(tb, mx0 | my1 | mz0), f3 = texel_f32(tb, mx1 | my1 | mz0);
                const float w0 = ab00 * c0, w1 = ab10 * c0, w2 = ab01 * c0, w3 = ab11 * c0;
#define VCT_ACC(ch) r.ch = w0 * f0.ch; r.ch = fmaf(w1, f1.ch, r.ch); r.ch = fmaf(w2, f2.ch, r.ch); r.ch = fmaf(w3, f3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                uint32_t x0 = mx0, x1 = mx1, y0 = my0, y1 = my1;
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));      // (no common subexpression with the lower plane)
                const float4 f4 = texel_f32(tb, x0 | y0 | mz1), f5 = texel_f32(tb, x1 | y0 | mz1);
                const float4 f6 = texel_f32(tb, x0 | y1 | mz1), f7 = texel_f32(tb, x1 | y1 | mz1);
                const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch) r.ch = fmaf(w4, f4.ch, r.ch); r.ch = fmaf(w5, f5.ch, r.ch); r.ch = fmaf(w6, f6.ch, r.ch); r.ch = fmaf(w7, f7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
#else
            const uint32_t xy00 = mx0 | my0, xy10 = mx1 | my0, xy01 = mx0 | my1, xy11 = mx1 | my1;
            {
                const float4 f0 = texel_f32(tb, xy00 | mz0), f1 = texel_f32(tb, xy10 | mz0);
                const float4 f2 = texel_f32(tb, xy01 | mz0), f3 = texel_f32(tb, xy11 | mz0);
                const float w0 = ab00 * c0, w1 = ab10 * c0, w2 = ab01 * c0, w3 = ab11 * c0;
#define VCT_ACC(ch) r.ch = w0 * f0.ch; r.ch = fmaf(w1, f1.ch, r.ch); r.ch = fmaf(w2, f2.ch, r.ch); r.ch = fmaf(w3, f3.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
            __builtin_amdgcn_sched_barrier(0);      // (the upper plane's loads are issued after the lower plane is folded: 16 texel registers)
            {
                const float4 f4 = texel_f32(tb, xy00 | mz1), f5 = texel_f32(tb, xy10 | mz1);
                const float4 f6 = texel_f32(tb, xy01 | mz1), f7 = texel_f32(tb, xy11 | mz1);
                const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
#define VCT_ACC(ch) r.ch = fmaf(w4, f4.ch, r.ch); r.ch = fmaf(w5, f5.ch, r.ch); r.ch = fmaf(w6, f6.ch, r.ch); r.ch = fmaf(w7, f7.ch, r.ch);
                VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
            }
#endif
#endif
        } else {
        uint32_t t[8];
#define VCT_TEXEL(o) (base[(o)])
        if (VCT_CELLS && CELLS && WRAP && lv.off != 0u) {     // (CELLS instantiations are launched with records present)
            // footprint record of the lower corner: the 8 texels as 32 contiguous bytes (levels >= 1; Morton index * 32
            // = byte offset of the record, < 2^32 for every level but the first of a 2048^3 grid, which has none)
            const char* cb = cells + ((size_t)lv.off << 5);
            const uint32_t ro = (mx0 | my0 | mz0) << 5;
            const uint4 lo = *(const uint4*)(cb + ro);
            const uint4 hi = *(const uint4*)(cb + ro + 16u);
            t[0] = lo.x; t[1] = lo.y; t[2] = lo.z; t[3] = lo.w;
            t[4] = hi.x; t[5] = hi.y; t[6] = hi.z; t[7] = hi.w;
        } else {
#if VCT_PAIR_LOAD
        if (WRAP && (MX & 1u)) {            // (not the one-texel level: its x + 1 wraps onto x)
            const bool even = (mx0 & 1u) == 0u;
            const uint32_t mxe = mx0 & ~1u;
            const uint32_t yz[4] = {my0 | mz0, my1 | mz0, my0 | mz1, my1 | mz1};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 pr = *(const uint2*)(base + (mxe | yz[q]));       // (base: uint32_t*, index arithmetic)
                uint32_t hi = pr.y;
                if (!even) hi = VCT_TEXEL(mx1 | yz[q]);
                t[2 * q] = even ? pr.x : pr.y;
                t[2 * q + 1] = hi;
            }
        } else {
#endif
        t[0] = VCT_TEXEL(mx0 | my0 | mz0); t[1] = VCT_TEXEL(mx1 | my0 | mz0);
        t[2] = VCT_TEXEL(mx0 | my1 | mz0); t[3] = VCT_TEXEL(mx1 | my1 | mz0);
        t[4] = VCT_TEXEL(mx0 | my0 | mz1); t[5] = VCT_TEXEL(mx1 | my0 | mz1);
        t[6] = VCT_TEXEL(mx0 | my1 | mz1); t[7] = VCT_TEXEL(mx1 | my1 | mz1);
#if VCT_PAIR_LOAD
        }
#endif
        }
#undef VCT_TEXEL
        if (VCT_LOAD_PRIO && PRIO) __builtin_amdgcn_s_setprio(0);
        const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
        const float wg[8] = {(a0 * b0) * c0, (a * b0) * c0, (a0 * b) * c0, (a * b) * c0,
                             (a0 * b0) * c,  (a * b0) * c,  (a0 * b) * c,  (a * b) * c};
        constexpr bool L = (VCT_LUT & 2) != 0;
        r.x = wg[0] * unorm8_of<0, LOOSE>(lb.unorm, t[0], L);
        r.y = wg[0] * unorm8_of<8, LOOSE>(lb.unorm, t[0], L);
        r.z = wg[0] * unorm8_of<16, LOOSE>(lb.unorm, t[0], L);
        r.w = wg[0] * unorm8_of<24, LOOSE>(lb.unorm, t[0], L);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            r.x = fmaf(wg[i], unorm8_of<0, LOOSE>(lb.unorm, t[i], L), r.x);
            r.y = fmaf(wg[i], unorm8_of<8, LOOSE>(lb.unorm, t[i], L), r.y);
            r.z = fmaf(wg[i], unorm8_of<16, LOOSE>(lb.unorm, t[i], L), r.z);
            r.w = fmaf(wg[i], unorm8_of<24, LOOSE>(lb.unorm, t[i], L), r.w);
        }
        }
      }
    }
    return r;
}

// Anisotropic option (oracle/vct_oracle.h "anisotropic (directional) mip volumes"): a level >= 1 is
// sampled from the three directional chains the cone direction faces, weighted by dir^2.  The chain
// of an axis depends on the sign of that direction component, which is per lane: when the live
// lanes disagree the axis is sampled once per sign with the lanes split by mask (rare: a tile's
// cones are near-parallel), so sample_level always sees a wave-uniform chain.
struct AnisoCone {
    float wx, wy, wz;       // dir.x^2, dir.y^2, dir.z^2
    bool nx, ny, nz;        // component not >= 0: the chain pre-integrated towards -axis
};

template <bool WRAP, bool COOP>
__device__ __forceinline__ F4 sample_aniso(const VctTraceParams& p, const VctLevelRef lv, float ux,
                                           float uy, float uz, bool act, unsigned long long m,
                                           float4* __restrict__ blk, const LaneBlock& lb, const AnisoCone& ac,
                                           MarchStats& ms) {
    // chain pointer of direction d such that (pointer + lv.off) is the level's first texel
    auto chain_of = [&](int d) { return p.aniso + (size_t)d * p.aniso_stride - p.level_off[1]; };
    const unsigned long long mx = ballot64(ac.nx) & m, my = ballot64(ac.ny) & m, mz = ballot64(ac.nz) & m;
    F4 tx, ty, tz;
    bool done = false;
    if (COOP && WRAP) {
        // The six directional chains have the same geometry, so coordinates, anchor, cooperative
        // test, Morton index, LDS slot and trilinear weights are computed once for all of them; only
        // the texel, its decode and the 8-texel gather are per chain.  An axis whose live lanes
        // disagree on the sign (cones along a coordinate axis have components ~0 of either sign)
        // fetches both of its chains into two slabs and every lane gathers from the one its sign
        // selects.
        const int mm = lv.m;
        const float fN = lv.fN;
        const float u = fmaf(ux, fN, -0.5f), v = fmaf(uy, fN, -0.5f), w = fmaf(uz, fN, -0.5f);
        const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
        const float a = u - fu, b = v - fv, c = w - fw;
        const int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
        const int src = ((m >> 27) & 1ull) ? 27 : (int)__ffsll((long long)m) - 1;
        const int ax = __builtin_amdgcn_readlane(i0, src) - 1;
        const int ay = __builtin_amdgcn_readlane(j0, src) - 1;
        const int az = __builtin_amdgcn_readlane(k0, src) - 1;
        const int dx = i0 - ax, dy = j0 - ay, dz = k0 - az;
        const uint32_t far = max(max((uint32_t)dx, (uint32_t)dy), (uint32_t)dz);
        if ((ballot64(far > 2u) & m) == 0ull) {
            done = true;
            const uint32_t MX = lv.mask_x, MY = MX << 1, MZ = MX << 2;      // texel indices, as in sample_level
            const uint32_t m4 = (uint32_t)mm << 2;
            const SpreadLut lut = (SpreadLut)p.spread_lut;
            const uint32_t sax = spread_byte(lut, ax, m4) >> 2;
            const uint32_t say = spread_byte(lut, ay, m4) >> 1;
            const uint32_t saz = spread_byte(lut, az, m4);
            const uint32_t idx = (((sax | ~MX) + lb.sbx) & MX) | (((say | ~MY) + lb.sby) & MY) |
                                 (((saz | ~MZ) + lb.sbz) & MZ);
            const unsigned long long mneg[3] = {mx, my, mz};
            const bool lneg[3] = {ac.nx, ac.ny, ac.nz};
#if VCT_HW_UNORM && VCT_ANISO_HW
            // texels decoded by the texture path (sample_level): the up to six directional blocks arrive as float4s
            float4 fpos[3], fneg[3];
            const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < 3; ++k) {          // all block loads first (up to 6 in flight)
                fpos[k] = mneg[k] != m ? texel_f32(level_texel_buffer(chain_of(2 * k) + lv.off), idx) : zero4;          // some lane is >= 0
                fneg[k] = mneg[k] != 0ull ? texel_f32(level_texel_buffer(chain_of(2 * k + 1) + lv.off), idx) : zero4;   // some lane is < 0
            }
            auto bits4 = [](const float4& f) { return __float_as_uint(f.x) | __float_as_uint(f.y) | __float_as_uint(f.z) | __float_as_uint(f.w); };
#else
            uint32_t tpos[3], tneg[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {          // all block loads first (up to 6 in flight)
                tpos[k] = mneg[k] != m ? (chain_of(2 * k) + lv.off)[idx] : 0u;          // some lane is >= 0
                tneg[k] = mneg[k] != 0ull ? (chain_of(2 * k + 1) + lv.off)[idx] : 0u;   // some lane is < 0
            }
#endif
            const int slot = act ? (dz * 4 + dy) * 4 + dx : 0;
            const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
            const float ab00 = a0 * b0, ab10 = a * b0, ab01 = a0 * b, ab11 = a * b;
            const float w0 = ab00 * c0, w1 = ab10 * c0, w2 = ab01 * c0, w3 = ab11 * c0;
            const float w4 = ab00 * c, w5 = ab10 * c, w6 = ab01 * c, w7 = ab11 * c;
            float4* alt = p.aniso_alt_slab ? blk + p.aniso_alt_slab : blk;      // slab of the "towards -axis" chain
            F4 out3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                F4 r = {0.0f, 0.0f, 0.0f, 0.0f};
#if VCT_HW_UNORM && VCT_ANISO_HW
                if (ballot64((bits4(fpos[k]) | bits4(fneg[k])) != 0u) != 0ull) {
                    if (mneg[k] != m) blk[lb.lane] = fpos[k];
                    if (mneg[k] != 0ull) alt[lb.lane] = fneg[k];
#else
                if (ballot64((tpos[k] | tneg[k]) != 0u) != 0ull) {
                    auto dec = [](uint32_t t) {
                        float4 d;
                        d.x = unorm8(t & 0xffu); d.y = unorm8((t >> 8) & 0xffu);
                        d.z = unorm8((t >> 16) & 0xffu); d.w = unorm8(t >> 24);
                        return d;
                    };
                    if (mneg[k] != m) blk[lb.lane] = dec(tpos[k]);
                    if (mneg[k] != 0ull) alt[lb.lane] = dec(tneg[k]);
#endif
                    wave_sync();
                    const float4* q = (lneg[k] ? alt : blk) + slot;
                    const float4 t0 = q[0], t1 = q[1], t2 = q[4], t3v = q[5];
                    const float4 t4 = q[16], t5 = q[17], t6 = q[20], t7 = q[21];
                    wave_sync();
#define VCT_ACC(ch)                                                                            \
    r.ch = w0 * t0.ch;                                                                         \
    r.ch = fmaf(w1, t1.ch, r.ch); r.ch = fmaf(w2, t2.ch, r.ch); r.ch = fmaf(w3, t3v.ch, r.ch); \
    r.ch = fmaf(w4, t4.ch, r.ch); r.ch = fmaf(w5, t5.ch, r.ch); r.ch = fmaf(w6, t6.ch, r.ch);  \
    r.ch = fmaf(w7, t7.ch, r.ch);
                    VCT_ACC(x) VCT_ACC(y) VCT_ACC(z) VCT_ACC(w)
#undef VCT_ACC
                }
                out3[k] = r;
            }
            tx = out3[0]; ty = out3[1]; tz = out3[2];
        }
    }
    if (!done) {
        // per axis: lanes that disagree on the sign are served in two masked passes
        auto axis_sample = [&](int axis, bool neg, unsigned long long mn) -> F4 {
            if (mn == 0ull) return sample_level<WRAP, COOP>(chain_of(2 * axis), lv, ux, uy, uz, act, m, blk, lb, ms);
            if (mn == m) return sample_level<WRAP, COOP>(chain_of(2 * axis + 1), lv, ux, uy, uz, act, m, blk, lb, ms);
            const F4 a = sample_level<WRAP, COOP>(chain_of(2 * axis), lv, ux, uy, uz, act && !neg, m & ~mn, blk, lb, ms);
            const F4 b = sample_level<WRAP, COOP>(chain_of(2 * axis + 1), lv, ux, uy, uz, act && neg, mn, blk, lb, ms);
            return neg ? b : a;
        };
        tx = axis_sample(0, ac.nx, mx);
        ty = axis_sample(1, ac.ny, my);
        tz = axis_sample(2, ac.nz, mz);
    }
    F4 r;
    r.x = ac.wx * tx.x; r.x = fmaf(ac.wy, ty.x, r.x); r.x = fmaf(ac.wz, tz.x, r.x);
    r.y = ac.wx * tx.y; r.y = fmaf(ac.wy, ty.y, r.y); r.y = fmaf(ac.wz, tz.y, r.y);
    r.z = ac.wx * tx.z; r.z = fmaf(ac.wy, ty.z, r.z); r.z = fmaf(ac.wz, tz.z, r.z);
    r.w = ac.wx * tx.w; r.w = fmaf(ac.wy, ty.w, r.w); r.w = fmaf(ac.wz, tz.w, r.w);
    return r;
}

// trace.fs:82-107 with the pixel-independent step sequence read from `tab`.
// The step table is read through the constant address space: a wave-uniform index then becomes one
// scalar s_load_dwordx16 instead of vector loads + v_readfirstlane, and the entry of step k+1 is
// requested while step k is being marched.
typedef const __attribute__((address_space(4))) VctStep* StepTable;

__device__ __forceinline__ VctStep load_step(StepTable t, int k) {
    VctStep s;
    s.dist = t[k].dist; s.occ_rcp = t[k].occ_rcp; s.occ_den = t[k].occ_den; s.frac = t[k].frac;
    s.level = t[k].level; s.level2 = t[k].level2; s.two_levels = t[k].two_levels; s.omf = t[k].omf;
    s.l1.off = t[k].l1.off; s.l1.mask_x = t[k].l1.mask_x; s.l1.fN = t[k].l1.fN; s.l1.m = t[k].l1.m;
    s.l2.off = t[k].l2.off; s.l2.mask_x = t[k].l2.mask_x; s.l2.fN = t[k].l2.fN; s.l2.m = t[k].l2.m;
    return s;
}

// One march step with the table entry `st`; `live` = ballot of the lanes still marching.  A macro, not a lambda: the
// plain loop of the anisotropic march must compile exactly as it did before the unrolled form existed (a lambda cost it 7 %).
//   position: trace.fs:98 + :61-63  (q * 0.5f is exact, so fmaf(q, .5, .5) is the oracle's q*.5f + .5f; a coordinate below
//             div_const's 2^-100 domain gives |q| < 2^-26 and u = 0.5 either way)
//   sample:   textureLod = blend of the two levels (frac == 0: one level, decided in the table)
//   composite: trace.fs:100 (colour), :101 (occlusion), :102 (alpha), front to back
#define VCT_MARCH_STEP(st, act, live)                                                                        \
        if (VCT_LOAD_PRIO && PRIO && VCT_LOAD_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(1); \
        const float px = start.x + dir.x * st.dist; \
        const float py = start.y + dir.y * st.dist; \
        const float pz = start.z + dir.z * st.dist; \
        const float ux = fmaf(div_const<FASTDIV>(px, p.half_G_aux, p.half_G_rcp), 0.5f, 0.5f); \
        const float uy = fmaf(div_const<FASTDIV>(py, p.half_G_aux, p.half_G_rcp), 0.5f, 0.5f); \
        const float uz = fmaf(div_const<FASTDIV>(pz, p.half_G_aux, p.half_G_rcp), 0.5f, 0.5f); \
        F4 vc = (ANISO && st.level >= 1) ? sample_aniso<WRAP, COOP>(p, st.l1, ux, uy, uz, act, live, blk, lb, ac, ms) \
                                         : sample_level<WRAP, COOP, FASTDIV == 2, CELLS, PRIO>(p.chain, st.l1, ux, uy, uz, act, live, blk, lb, ms, p.cells_biased); \
        if (st.two_levels) { \
            const F4 t2 = ANISO ? sample_aniso<WRAP, COOP>(p, st.l2, ux, uy, uz, act, live, blk + 64, lb, ac, ms) \
                                : sample_level<WRAP, COOP, FASTDIV == 2, CELLS, PRIO>(p.chain, st.l2, ux, uy, uz, act, live, blk + 64, lb, ms, p.cells_biased); \
            const float g = st.omf;      /* 1 - frac, from the table */ \
            vc.x = fmaf(st.frac, t2.x, g * vc.x); \
            vc.y = fmaf(st.frac, t2.y, g * vc.y); \
            vc.z = fmaf(st.frac, t2.z, g * vc.z); \
            vc.w = fmaf(st.frac, t2.w, g * vc.w); \
        } \
        if (act) { \
            const float oma = 1.0f - alpha; \
            cr = fmaf(oma, vc.x, cr); \
            cg = fmaf(oma, vc.y, cg); \
            cb = fmaf(oma, vc.z, cb); \
            occ = occ + div_const<FASTDIV>(oma * vc.w, st.occ_den, st.occ_rcp); \
            alpha = fmaf(oma, vc.w, alpha); \
            ++steps; \
        }

template <bool WRAP, int FASTDIV, bool COOP, bool ANISO = false, bool CELLS = false, bool PRIO = false>
__device__ __forceinline__ F4 cone_march(const VctTraceParams& p, bool alive, F3 start, F3 dir,
                                         const VctStep* tab_global, int n,
                                         float4* __restrict__ blk, const LaneBlock& lb,
                                         int& steps_out, MarchStats& ms) {
    const StepTable tab = (StepTable)tab_global;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, alpha = 0.0f, occ = 0.0f;
    int steps = 0;
    AnisoCone ac = {0.0f, 0.0f, 0.0f, false, false, false};
    if (ANISO) {
        ac.wx = dir.x * dir.x; ac.wy = dir.y * dir.y; ac.wz = dir.z * dir.z;
        ac.nx = !(dir.x >= 0.0f); ac.ny = !(dir.y >= 0.0f); ac.nz = !(dir.z >= 0.0f);
    }
    const unsigned long long alive_mask = ballot64(alive);
    if constexpr (VCT_UNROLL2 && !ANISO) {
    // Two steps per loop iteration (not the anisotropic march: its body, twice, spills and ran 7x slower), the table entries ping-pong between two register sets: the entry of step k + 1 is
    // requested while step k is marched and is never copied (the rotating form below moves 12 SGPRs per step, and
    // the scalar pipe is the march's second bound).
    // (the step body goes through a lambda here: measured 2 % faster than the macro expanded in place, while the
    // plain loop below wants the macro expanded in place -- register allocation differs, the instructions do not)
    auto march_step = [&](const VctStep& st, const bool act, const unsigned long long live) { VCT_MARCH_STEP(st, act, live) };
    VctStep ea = load_step(tab, 0), eb = ea;
    for (int k = 0; k < n;) {
        {
            const bool act = alive && (alpha < p.max_alpha);     // trace.fs:94 (dist < MAX: table)
            const unsigned long long live = ballot64(alpha < p.max_alpha) & alive_mask;
            if (live == 0ull) break;
            if (VCT_STATS) { ++ms.wave_steps; ms.lane_steps += (uint32_t)__popcll(live); }
            eb = load_step(tab, k + 1 < n ? k + 1 : k);
            march_step(ea, act, live);
            if (++k >= n) break;
        }
        {
            const bool act = alive && (alpha < p.max_alpha);
            const unsigned long long live = ballot64(alpha < p.max_alpha) & alive_mask;
            if (live == 0ull) break;
            if (VCT_STATS) { ++ms.wave_steps; ms.lane_steps += (uint32_t)__popcll(live); }
            ea = load_step(tab, k + 1 < n ? k + 1 : k);
            march_step(eb, act, live);
            ++k;
        }
    }
    } else {
    VctStep nxt = load_step(tab, 0);
    for (int k = 0; k < n; ++k) {
        const bool act = alive && (alpha < p.max_alpha);     // trace.fs:94 (dist < MAX: table)
        const unsigned long long live = ballot64(alpha < p.max_alpha) & alive_mask;
        if (live == 0ull) break;
        if (VCT_STATS) { ++ms.wave_steps; ms.lane_steps += (uint32_t)__popcll(live); }
        const VctStep st = nxt;
        nxt = load_step(tab, k + 1 < n ? k + 1 : k);
        VCT_MARCH_STEP(st, act, live)
    }
    }
    steps_out = steps;
    return {cr, cg, cb, occ};
}

// EXPERIMENT (-DVCT_LOCKSTEP=1; round 3): the three diffuse cones of a wave marched in lockstep -- one
// table entry, one set of level constants per step for all three, three independent dependency chains.  Same bits.
// Result on MI355X: see profiles/experiments/README.md (the per-step work that is shared is scalar -- the table load
// and ~8 SALU of loop control -- while every vector instruction of a step depends on the cone's own position; the
// carried state triples: three accumulator sets + three directions).
#ifndef VCT_LOCKSTEP
#define VCT_LOCKSTEP 0
#endif
struct ConeAcc { float cr, cg, cb, alpha, occ; int steps; };
template <bool WRAP, int FASTDIV, bool COOP>
__device__ __forceinline__ void cone_march3(const VctTraceParams& p, bool alive, F3 start, const F3 dirs[3],
                                            const VctStep* tab_global, int n, float4* __restrict__ blk,
                                            const LaneBlock& lb, ConeAcc out[3], MarchStats& ms) {
    constexpr bool ANISO = false, CELLS = false, PRIO = false;
    const StepTable tab = (StepTable)tab_global;
    AnisoCone ac = {0.0f, 0.0f, 0.0f, false, false, false};
    ConeAcc c0 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0}, c1 = c0, c2 = c0;
    const unsigned long long alive_mask = ballot64(alive);
    auto one = [&](const VctStep& st, const F3 dir, ConeAcc& c) -> bool {
        float cr = c.cr, cg = c.cg, cb = c.cb, alpha = c.alpha, occ = c.occ;
        int steps = c.steps;
        const bool act = alive && (alpha < p.max_alpha);
        const unsigned long long live = ballot64(alpha < p.max_alpha) & alive_mask;
        if (live == 0ull) return false;
        if (VCT_STATS) { ++ms.wave_steps; ms.lane_steps += (uint32_t)__popcll(live); }
        VCT_MARCH_STEP(st, act, live)
        c.cr = cr; c.cg = cg; c.cb = cb; c.alpha = alpha; c.occ = occ; c.steps = steps;
        return true;
    };
    VctStep nxt = load_step(tab, 0);
    for (int k = 0; k < n; ++k) {
        const VctStep st = nxt;
        nxt = load_step(tab, k + 1 < n ? k + 1 : k);
        const bool a = one(st, dirs[0], c0);
        const bool b = one(st, dirs[1], c1);
        const bool c = one(st, dirs[2], c2);
        if (!(a || b || c)) break;
    }
    out[0] = c0; out[1] = c1; out[2] = c2;
}

__device__ __forceinline__ uint32_t pack_half2(float a, float b) {
    const __half ha = __float2half_rn(a), hb = __float2half_rn(b);
    return (uint32_t)__half_as_ushort(ha) | ((uint32_t)__half_as_ushort(hb) << 16);
}

__device__ __forceinline__ void flush_stats(const VctTraceParams& p, const MarchStats& ms, int lane) {
    if (VCT_STATS && p.stats && lane == 0) {
        const uint32_t v[16] = {ms.wave_steps, ms.lane_steps, ms.coop_zero, ms.coop_hit, ms.fallback,
                                ms.fallback_lanes, ms.fallback_fits, ms.greedy_blocks, ms.greedy_le2, ms.greedy_le3,
                                ms.greedy_le4, ms.quads_live, ms.quads_fit333, ms.quads_same, ms.quadrants_live,
                                ms.quadrants_fit444};
        for (int i = 0; i < 16; ++i)
            if (v[i]) atomicAdd(p.stats + i, (unsigned long long)v[i]);
    }
}

__constant__ float kConeDirs[18] = {0.0f, 0.0f, 1.0f,
                                    0.0f, 0.866025f, 0.5f,
                                    0.823639f, 0.267617f, 0.5f,
                                    0.509037f, -0.700629f, 0.5f,
                                    -0.509037f, -0.700629f, 0.5f,
                                    -0.823639f, 0.267617f, 0.5f};            // trace.fs:49-57
__constant__ float kConeWeights[6] = {0.25f, 0.15f, 0.15f, 0.15f, 0.15f, 0.15f};   // trace.fs:48

#ifndef VCT_XCD_MAP
#define VCT_XCD_MAP 16   // measured: 0 (round-robin) 0.764 ms, 1 (contiguous) 0.797, 16: 0.756, 60: 0.779, 240: 0.758
#endif
// Workgroups are dealt to XCDs round-robin by the dispatcher (block b -> XCD b % 8); the tile order is
// remapped so that every XCD works on short runs of neighbouring tiles (neighbouring tiles march
// through neighbouring voxels and share that XCD's L2) while the runs of all XCDs interleave over
// the frame -- one long contiguous run per XCD loses more to imbalance between cheap and expensive
// screen regions than it gains in locality (profiles/r01h_ab_xcd_map.txt).
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
#if VCT_XCD_MAP == 0
    (void)nblocks;
    return b;                                                    // tiles dealt round-robin to the XCDs
#elif VCT_XCD_MAP == 1
    const int per_xcd = nblocks >> 3;                            // one contiguous run of tiles per XCD
    return (b & 7) * per_xcd + (b >> 3);
#else
    // runs of VCT_XCD_MAP consecutive blocks per XCD: a permutation inside every full group of
    // 8*run blocks; the last partial group keeps identity
    const int run = VCT_XCD_MAP, group = 8 * run;
    const int g = b / group, local = b - g * group;
    return (g + 1) * group <= nblocks ? g * group + (local & 7) * run + (local >> 3) : b;
#endif
}

// tiles (waves) per workgroup: single-wave workgroups dispatch with the finest granularity, which
// balances the tail best (0.822 -> 0.800 ms against 4 waves; profiles/r01g_ab_waves_per_block.txt)
#ifndef VCT_WAVES_PER_BLOCK
#define VCT_WAVES_PER_BLOCK 1
#endif
#ifndef VCT_TRACE_MIN_WAVES
#define VCT_TRACE_MIN_WAVES 7     // waves per SIMD the register allocator must leave room for (<= 72 VGPRs).  A/B on the
                                  // final round-2 kernel (ms at 256^3/1080p): 4: 0.776, 5: 0.775, 6: 0.7235, 7: 0.695, 8: 0.719
                                  // (spills); with the two-plane gather 7: 0.683 (69 VGPRs, no scratch), 8: 0.689; gathering texel pairs: 7: 0.685, 8: 0.684
#endif

// One wave per tile, lane = pixel, the 7 cones in sequence; VCT_WAVES_PER_BLOCK horizontally
// adjacent tiles per workgroup.  Workgroups are dealt to XCDs round-robin by the dispatcher (block b -> XCD b % 8), so
// the tile order is remapped to give every XCD one contiguous run of tiles: neighbouring tiles
// march through neighbouring voxels and share that XCD's L2.
template <bool WRAP, int FASTDIV, bool COOP>
__global__ void __launch_bounds__(64 * VCT_WAVES_PER_BLOCK, VCT_TRACE_MIN_WAVES)
k_trace_tile(const VctTraceParams p) {
    __shared__ float4 lds_blk[VCT_WAVES_PER_BLOCK][2][64];
    VCT_LUT_DECL
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; readfirstlane tells the compiler, so tile indices, the tile's
    // G-buffer base and the LDS slab base live in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float4* blk = &lds_blk[wave][0][0];
    LaneBlock lb;
    VCT_LUT_FILL(lb)
#if VCT_LUT
    __syncthreads();
#endif

    const int ntiles = (p.tile_row1 - p.tile_row0) * p.tiles_x;
    const int vb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ti = vb * VCT_WAVES_PER_BLOCK + wave;
    if (ti >= ntiles) return;

    lb.lane = lane;
    lb.lut = (SpreadLut)p.spread_lut; lb.lut_vec = p.spread_lut;
    lb.sbx = vct_spread3((uint32_t)lane & 3u);     
    lb.sby = vct_spread3(((uint32_t)lane >> 2) & 3u) << 1;
    lb.sbz = vct_spread3((uint32_t)lane >> 4) << 2;
    MarchStats ms = {};

    const int tile = p.tile_row0 * p.tiles_x + ti;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    // Pixel / G-buffer addresses are re-derived from the lane id wherever they are needed (a few
    // integer ops) rather than carried in 64-bit VGPR pairs across the march loops, where they
    // would be spilled to scratch under the 80-VGPR budget.
    auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto pixel_index = [&](int l) {
        return (size_t)(ty * VCT_TILE + (l >> 3)) * p.width + (tx * VCT_TILE + (l & 7));
    };
    auto gbuf_ptr = [&](int l) {
        return p.gbuf + (size_t)tile * (VCT_GB_NPLANES * VCT_TILE_PIX) + l;
    };
    const int x = tx * VCT_TILE + (lane & 7), y = ty * VCT_TILE + (lane >> 3);
    // The G-buffer is read in three stages (cone frame, specular direction, composite) instead of
    // once up front: the 23 planes are only L1/L2 re-reads, while every VGPR kept live across the
    // march loops costs occupancy, and resident waves are what hides the sampler's latency chain.
    const float* gb = gbuf_ptr(lane);
#define VCT_GB(k) gb[(k) * VCT_TILE_PIX]
    const bool in_frame = (x < p.width) && (y < p.height);
    const bool alive = in_frame && !(VCT_GB(18) < 0.5f);            // trace.fs:171 discard

    F3 start, k0, k1, k2;
    {
        const F3 P = f3(VCT_GB(0), VCT_GB(1), VCT_GB(2)), Nw = f3(VCT_GB(3), VCT_GB(4), VCT_GB(5));
        const F3 T = f3(VCT_GB(6), VCT_GB(7), VCT_GB(8)), B = f3(VCT_GB(9), VCT_GB(10), VCT_GB(11));
        // trace.fs:175: inverse(transpose(mat3(T,B,N))) = columns (BxN, NxT, TxB) / det
        const F3 c0 = cross3(B, Nw), c1 = cross3(Nw, T), c2 = cross3(T, B);
        const float inv_det = div_rn(1.0f, dot3(T, c0));
        k0 = f3(c0.x * inv_det, c0.y * inv_det, c0.z * inv_det);
        k1 = f3(c1.x * inv_det, c1.y * inv_det, c1.z * inv_det);
        k2 = f3(c2.x * inv_det, c2.y * inv_det, c2.z * inv_det);
        start = f3(P.x + Nw.x * p.vs, P.y + Nw.y * p.vs, P.z + Nw.z * p.vs);       // :92
    }

    float ind[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int total = 0;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {                                           // :196-199
        const float ddx = kConeDirs[3 * i], ddy = kConeDirs[3 * i + 1], ddz = kConeDirs[3 * i + 2];
        F3 dir = f3(k0.x * ddx + k1.x * ddy + k2.x * ddz, k0.y * ddx + k1.y * ddy + k2.y * ddz,
                    k0.z * ddx + k1.z * ddy + k2.z * ddz);
        dir = normalize3(dir);
        int st;
        const F4 c = cone_march<WRAP, FASTDIV, COOP>(p, alive, start, dir, p.steps_diffuse,
                                                     p.n_diffuse, blk, lb, st, ms);
        total += st;
        const float wgt = kConeWeights[i];
        ind[0] = fmaf(wgt, c.x, ind[0]);
        ind[1] = fmaf(wgt, c.y, ind[1]);
        ind[2] = fmaf(wgt, c.z, ind[2]);
        ind[3] = fmaf(wgt, c.w, ind[3]);
        if (p.dbg_cones && alive) {
            float* d = p.dbg_cones + pixel_index(fresh_lane()) * 28 + 4 * i;
            d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
        }
        if (p.dbg_steps && in_frame) p.dbg_steps[pixel_index(fresh_lane()) * 7 + i] = (uint8_t)st;
    }

    // stage 2: specular cone along reflect(-E, N) with the bump normal           trace.fs:217-218
    const float* gb2 = gbuf_ptr(fresh_lane());   // a fresh pointer: re-read instead of keeping planes live
#undef VCT_GB
#define VCT_GB(k) gb2[(k) * VCT_TILE_PIX]
    F4 sc;
    {
        const F3 P = f3(VCT_GB(0), VCT_GB(1), VCT_GB(2));
        const F3 N = f3(VCT_GB(12), VCT_GB(13), VCT_GB(14));
        const F3 E = normalize3(f3(p.cam[0] - P.x, p.cam[1] - P.y, p.cam[2] - P.z));   // :181
        const F3 Rd = normalize3(reflect3(f3(E.x * -1.0f, E.y * -1.0f, E.z * -1.0f), N));  // :217
        int st6;
        sc = cone_march<WRAP, FASTDIV, COOP>(p, alive, start, Rd, p.steps_specular, p.n_specular,
                                             blk, lb, st6, ms);
        total += st6;
        if (p.dbg_cones && alive) {
            float* d = p.dbg_cones + pixel_index(fresh_lane()) * 28 + 24;
            d[0] = sc.x; d[1] = sc.y; d[2] = sc.z; d[3] = sc.w;
        }
        if (p.dbg_steps && in_frame) p.dbg_steps[pixel_index(fresh_lane()) * 7 + 6] = (uint8_t)st6;
    }

    // stage 3: composite                                                          trace.fs:179-227
    const float* gb3 = gbuf_ptr(fresh_lane());
#undef VCT_GB
#define VCT_GB(k) gb3[(k) * VCT_TILE_PIX]
    if (in_frame) {
        const F3 P = f3(VCT_GB(0), VCT_GB(1), VCT_GB(2));
        const F3 N = f3(VCT_GB(12), VCT_GB(13), VCT_GB(14));
        const float alb_r = VCT_GB(15), alb_g = VCT_GB(16), alb_b = VCT_GB(17), alb_a = VCT_GB(18);
        const float shadow = VCT_GB(22);
        const F3 L = normalize3(f3(p.light[0], p.light[1], p.light[2]));        // :179
        const F3 E = normalize3(f3(p.cam[0] - P.x, p.cam[1] - P.y, p.cam[2] - P.z));   // :181
        const float cos_theta = fmaxf(dot3(N, L), 0.0f);                        // :188
        const float direct_diffuse = shadow * cos_theta;                        // :192
        const float occlusion = 1.0f - ind[3];                                  // :201
        const float dr = (direct_diffuse + occlusion * ind[0]) * alb_r;         // :205
        const float dg = (direct_diffuse + occlusion * ind[1]) * alb_g;
        const float db = (direct_diffuse + occlusion * ind[2]) * alb_b;
        const F3 R = normalize3(reflect3(f3(L.x * -1.0f, L.y * -1.0f, L.z * -1.0f), N));   // :212
        const float spec = powf(fmaxf(dot3(E, R), 0.0f), p.shininess);          // :213
        const float direct_spec = spec * shadow;                                // :214
        const float spec_occ = 1.0f - sc.w;                                     // :221
        const float sr = (sc.x + spec_occ * direct_spec) * VCT_GB(19);          // :223
        const float sg = (sc.y + spec_occ * direct_spec) * VCT_GB(20);
        const float sb = (sc.z + spec_occ * direct_spec) * VCT_GB(21);
        const float ar = p.ambient * alb_r * occlusion;                         // :225
        const float ag = p.ambient * alb_g * occlusion;
        const float ab = p.ambient * alb_b * occlusion;
        float o0 = ar + dr + sr, o1 = ag + dg + sg, o2 = ab + db + sb, o3 = alb_a;   // :227
        if (!alive) {                                                           // VCT.h:156-159
            const float cc = p.ambient < 0.5f ? 0.5f : 1.0f;
            o0 = cc; o1 = cc; o2 = cc; o3 = 1.0f;
        }
        uint2 pk;
        pk.x = pack_half2(o0, o1);
        pk.y = pack_half2(o2, o3);
        *reinterpret_cast<uint2*>(p.out + pixel_index(fresh_lane()) * 4) = pk;
    }
#undef VCT_GB
    // executed-step count: wave reduction, stored into the tile's slot (a plain store: no atomic, nothing to clear)
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (lane == 0) p.tile_steps[tile] = (uint32_t)total;
    flush_stats(p, ms, lane);
}

// ---- the same trace with each tile split over 3 waves ------------------------------------------
// A wave that marches all 71 steps of a tile lives ~140 us; at the end of a launch the GPU drains
// for about half of that with ever fewer waves resident (measured: +55..66 us per launch, 7 % of a
// 1080p frame but 37 % of the 0.15 ms slab an 8-GPU rank traces).  Here a tile is a workgroup of 3
// waves -- diffuse cones 0-2, diffuse cones 3-5, the specular cone (21 / 21 / 29 steps) -- that
// leave their raw cone vec4s in LDS and exit; the last one to arrive gathers them in the oracle's
// order (the weighted sum is an fma chain over cones 0..5) and composites.  Same bits, waves one
// third as long, no wave ever waits on another.
#ifndef VCT_SPLIT
#define VCT_SPLIT 3               // waves per tile: 3 = {cones 0-2, cones 3-5, specular}, 4 = {0-1, 2-3, 4-5, specular}, 7 = one cone each
#endif
// A/B at 256^3 / 1080p (round 3, ms): 3 waves 0.615; 4 waves 0.661 and 7 waves 0.808 although they balance the waves
// better and fill the CU's 28 wave slots exactly -- every wave pays the G-buffer fetch + frame inversion again and
// a shorter wave amortises that start-up stall over fewer march steps; 2 waves {0-3, 4-5 + specular} 0.639.
#define VCT_CONES_PER_WAVE (6 / (VCT_SPLIT - 1))
static_assert(VCT_SPLIT == 3 || VCT_SPLIT == 4 || VCT_SPLIT == 7, "VCT_SPLIT must be 3, 4 or 7");
#ifndef VCT_ANISO_MIN_WAVES
#define VCT_ANISO_MIN_WAVES 5     // A/B (ms, 256^3 1080p): 4: 1.61, 5: 1.47, 6: 1.88 (spills), 7: 1.60
#endif

// (the anisotropic instantiation carries three samples' worth of state: it gets 128 VGPRs instead of
// spilling under the 80 of the default kernel)
// CELLS: the per-lane sampler reads footprint records (p.cells_biased != null; vct_set_footprint_records) -- an
// instantiation of its own, because the same code behind a run-time test cost the default kernel 2 % (0.608 -> 0.620 ms)
template <bool WRAP, int FASTDIV, bool ANISO, bool COMPACT = false, bool CELLS = false, bool PRIO = false>
__global__ void __launch_bounds__(64 * VCT_SPLIT, ANISO ? VCT_ANISO_MIN_WAVES : VCT_TRACE_MIN_WAVES)
k_trace_tile_split(const VctTraceParams p) {
    __shared__ float4 lds_blk[VCT_SPLIT][ANISO ? 4 : 2][64];   // per wave: level-1 slab, level-2 slab (+ their "-axis" slabs)
    __shared__ float4 lds_cone[7][64];
    __shared__ int lds_done;
    __shared__ int lds_steps;          // executed steps of the tile's waves, summed here; the last arriver stores the total
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float4* blk = &lds_blk[wave][0][0];
    VCT_LUT_DECL
    LaneBlock lb;
    VCT_LUT_FILL(lb)
#if VCT_LANE_SPREAD_LUT == 2
    __shared__ uint32_t lds_spread[1024];
    for (uint32_t i_ = threadIdx.x; i_ < 1024u; i_ += blockDim.x) lds_spread[i_] = p.spread_lut[i_];
    lb.lut_lds = (const __attribute__((address_space(3))) uint32_t*)lds_spread;
#endif
    if (threadIdx.x == 0) { lds_done = 0; lds_steps = 0; }
    __syncthreads();

    const int ntiles = p.ntiles;          // (host-computed: a division here is ~40 instructions per wave -- 1 % of the frame)
    const int ti = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (ti >= ntiles) return;
    // Live-pixel compaction (experiment, trace_variant 4): the wave's 64 pixels come from the compaction list -- the live
    // pixels of a 16x16 super-tile -- instead of being the launched tile's.  `cpix` = tile << 6 | pixel of the tile.
    constexpr bool compact = COMPACT;         // (a separate instantiation: the default kernel carries none of this)
    uint32_t cpix = 0u;
    if (compact) {
        if ((uint32_t)ti >= *p.vt_count) {
            if (threadIdx.x == 0) p.tile_steps[p.tile_row0 * p.tiles_x + ti] = 0u;
            return;
        }
        cpix = p.vt_pix[(size_t)ti * 64 + lane];
    }

    lb.lane = lane;
    lb.lut = (SpreadLut)p.spread_lut; lb.lut_vec = p.spread_lut;
    lb.sbx = vct_spread3((uint32_t)lane & 3u);     
    lb.sby = vct_spread3(((uint32_t)lane >> 2) & 3u) << 1;
    lb.sbz = vct_spread3((uint32_t)lane >> 4) << 2;
    MarchStats ms = {};

    // (interleaved slabs: traced row j of the launch is tile row row0 + j * stride; the plain launch pays no second division)
    int lrow = 0, tile0 = p.tile_row0 * p.tiles_x + ti;           // tile0: where the wave's step count is stored
    if (p.row_stride > 1) {
        lrow = ti / p.tiles_x;
        tile0 = (p.tile_row0 + lrow * p.row_stride) * p.tiles_x + (ti - lrow * p.tiles_x);
    }
    const bool cvalid = !compact || cpix != 0xffffffffu;
    const int tile = compact ? (cvalid ? (int)(cpix >> 6) : 0) : tile0;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    // tile row of the output (a packed slab holds its rows back to back)
    const int oy = p.pack_rows ? (p.row_stride > 1 ? lrow : ty - p.tile_row0) : ty;
    const int plane_ = compact ? (int)(cpix & 63u) : lane;        // this lane's pixel inside its tile
    auto fresh_lane = [&]() { int l = plane_; asm volatile("" : "+v"(l)); return l; };
    auto pixel_index = [&](int l) {
        return (size_t)(oy * VCT_TILE + (l >> 3)) * p.width + (tx * VCT_TILE + (l & 7));
    };
    auto gbuf_ptr = [&](int l) {
        return p.gbuf + (size_t)tile * (VCT_GB_NPLANES * VCT_TILE_PIX) + l;
    };
    const int x = tx * VCT_TILE + (plane_ & 7), y = ty * VCT_TILE + (plane_ >> 3);
    const float* gb = gbuf_ptr(plane_);
#define VCT_GB(k) gb[(k) * VCT_TILE_PIX]
    const bool in_frame = cvalid && (x < p.width) && (y < p.height);
    const bool alive = in_frame && !(VCT_GB(18) < 0.5f);            // trace.fs:171 discard
    int total = 0;
    if (wave < VCT_SPLIT - 1) {
        F3 start, k0, k1, k2;
        {
            const F3 P = f3(VCT_GB(0), VCT_GB(1), VCT_GB(2)), Nw = f3(VCT_GB(3), VCT_GB(4), VCT_GB(5));
            const F3 T = f3(VCT_GB(6), VCT_GB(7), VCT_GB(8)), B = f3(VCT_GB(9), VCT_GB(10), VCT_GB(11));
            // trace.fs:175: inverse(transpose(mat3(T,B,N))) = columns (BxN, NxT, TxB) / det
            const F3 c0 = cross3(B, Nw), c1 = cross3(Nw, T), c2 = cross3(T, B);
            const float inv_det = div_rn(1.0f, dot3(T, c0));
            k0 = f3(c0.x * inv_det, c0.y * inv_det, c0.z * inv_det);
            k1 = f3(c1.x * inv_det, c1.y * inv_det, c1.z * inv_det);
            k2 = f3(c2.x * inv_det, c2.y * inv_det, c2.z * inv_det);
            start = f3(P.x + Nw.x * p.vs, P.y + Nw.y * p.vs, P.z + Nw.z * p.vs);       // :92
        }
#if VCT_LOCKSTEP && VCT_SPLIT == 3
        if (!ANISO) {
            F3 dirs[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = wave * 3 + j;
                const float ddx = kConeDirs[3 * i], ddy = kConeDirs[3 * i + 1], ddz = kConeDirs[3 * i + 2];
                dirs[j] = normalize3(f3(k0.x * ddx + k1.x * ddy + k2.x * ddz, k0.y * ddx + k1.y * ddy + k2.y * ddz,
                                        k0.z * ddx + k1.z * ddy + k2.z * ddz));
            }
            ConeAcc acc3[3];
            cone_march3<WRAP, FASTDIV, true>(p, alive, start, dirs, p.steps_diffuse, p.n_diffuse, blk, lb, acc3, ms);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = wave * 3 + j;
                total += acc3[j].steps;
                lds_cone[i][lane] = make_float4(acc3[j].cr, acc3[j].cg, acc3[j].cb, acc3[j].occ);
                if (p.dbg_cones && alive) {
                    float* d = p.dbg_cones + pixel_index(fresh_lane()) * 28 + 4 * i;
                    d[0] = acc3[j].cr; d[1] = acc3[j].cg; d[2] = acc3[j].cb; d[3] = acc3[j].occ;
                }
                if (p.dbg_steps && in_frame) p.dbg_steps[pixel_index(fresh_lane()) * 7 + i] = (uint8_t)acc3[j].steps;
            }
        } else
#endif
#pragma unroll 1
        for (int i = wave * VCT_CONES_PER_WAVE; i < wave * VCT_CONES_PER_WAVE + VCT_CONES_PER_WAVE; ++i) {      // :196-199
            const float ddx = kConeDirs[3 * i], ddy = kConeDirs[3 * i + 1], ddz = kConeDirs[3 * i + 2];
            F3 dir = f3(k0.x * ddx + k1.x * ddy + k2.x * ddz, k0.y * ddx + k1.y * ddy + k2.y * ddz,
                        k0.z * ddx + k1.z * ddy + k2.z * ddz);
            dir = normalize3(dir);
            int st;
            const F4 c = cone_march<WRAP, FASTDIV, true, ANISO, CELLS, PRIO>(p, alive, start, dir, p.steps_diffuse,
                                                                p.n_diffuse, blk, lb, st, ms);
            total += st;
            lds_cone[i][lane] = make_float4(c.x, c.y, c.z, c.w);
            if (p.dbg_cones && alive) {
                float* d = p.dbg_cones + pixel_index(fresh_lane()) * 28 + 4 * i;
                d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
            }
            if (p.dbg_steps && in_frame) p.dbg_steps[pixel_index(fresh_lane()) * 7 + i] = (uint8_t)st;
        }
    } else {
        // the specular wave is the longest of a tile (29 march steps against 21) and the last ones of a launch are its
        // tail: raised issue priority lets them run ahead of the diffuse waves, which have the slack.  Pays when the
        // launch is a slab of a multi-GPU frame (8-way slabs of the 1080p frame: 0.1085 -> 0.1052 ms mean, 0.115 ->
        // 0.111 ms max), costs 0.5 % on the whole frame: the host sets it for launches of at most half the frame.
        // (the PRIO instantiation -- whole frames -- sets and resets the priority around every sample's loads instead)
        if (!PRIO && p.spec_prio) __builtin_amdgcn_s_setprio(1);
        // specular cone along reflect(-E, N) with the bump normal                 trace.fs:217-218
        const F3 P = f3(VCT_GB(0), VCT_GB(1), VCT_GB(2)), Nw = f3(VCT_GB(3), VCT_GB(4), VCT_GB(5));
        const F3 N = f3(VCT_GB(12), VCT_GB(13), VCT_GB(14));
        const F3 start = f3(P.x + Nw.x * p.vs, P.y + Nw.y * p.vs, P.z + Nw.z * p.vs);
        const F3 E = normalize3(f3(p.cam[0] - P.x, p.cam[1] - P.y, p.cam[2] - P.z));   // :181
        const F3 Rd = normalize3(reflect3(f3(E.x * -1.0f, E.y * -1.0f, E.z * -1.0f), N));  // :217
        int st6;
        const F4 sc = cone_march<WRAP, FASTDIV, true, ANISO, CELLS, PRIO>(p, alive, start, Rd, p.steps_specular,
                                                             p.n_specular, blk, lb, st6, ms);
        total += st6;
        lds_cone[6][lane] = make_float4(sc.x, sc.y, sc.z, sc.w);
        if (p.dbg_cones && alive) {
            float* d = p.dbg_cones + pixel_index(fresh_lane()) * 28 + 24;
            d[0] = sc.x; d[1] = sc.y; d[2] = sc.z; d[3] = sc.w;
        }
        if (p.dbg_steps && in_frame) p.dbg_steps[pixel_index(fresh_lane()) * 7 + 6] = (uint8_t)st6;
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (lane == 0) atomicAdd(&lds_steps, total);          // LDS: the last arriver below stores the tile's total
    flush_stats(p, ms, lane);

    // arrival: LDS operations of a wave are performed in order, so the cone values are in LDS before
    // the count is raised; whoever raises it to VCT_SPLIT sees all of them
    __threadfence_block();
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(&lds_done, 1);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != VCT_SPLIT - 1) return;
    __threadfence_block();
    if (lane == 0) p.tile_steps[tile0] = (uint32_t)lds_steps;     // one plain store per tile: no global atomic, nothing to clear

    // composite by the last wave                                                   trace.fs:179-227
    const float* gb3 = gbuf_ptr(fresh_lane());
#undef VCT_GB
#define VCT_GB(k) gb3[(k) * VCT_TILE_PIX]
    if (in_frame) {
        float ind[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float4 c = lds_cone[i][lane];
            const float wgt = kConeWeights[i];
            ind[0] = fmaf(wgt, c.x, ind[0]);
            ind[1] = fmaf(wgt, c.y, ind[1]);
            ind[2] = fmaf(wgt, c.z, ind[2]);
            ind[3] = fmaf(wgt, c.w, ind[3]);
        }
        const float4 sc = lds_cone[6][lane];
        const F3 P = f3(VCT_GB(0), VCT_GB(1), VCT_GB(2));
        const F3 N = f3(VCT_GB(12), VCT_GB(13), VCT_GB(14));
        const float alb_r = VCT_GB(15), alb_g = VCT_GB(16), alb_b = VCT_GB(17), alb_a = VCT_GB(18);
        const float shadow = VCT_GB(22);
        const F3 L = normalize3(f3(p.light[0], p.light[1], p.light[2]));        // :179
        const F3 E = normalize3(f3(p.cam[0] - P.x, p.cam[1] - P.y, p.cam[2] - P.z));   // :181
        const float cos_theta = fmaxf(dot3(N, L), 0.0f);                        // :188
        const float direct_diffuse = shadow * cos_theta;                        // :192
        const float occlusion = 1.0f - ind[3];                                  // :201
        const float dr = (direct_diffuse + occlusion * ind[0]) * alb_r;         // :205
        const float dg = (direct_diffuse + occlusion * ind[1]) * alb_g;
        const float db = (direct_diffuse + occlusion * ind[2]) * alb_b;
        const F3 R = normalize3(reflect3(f3(L.x * -1.0f, L.y * -1.0f, L.z * -1.0f), N));   // :212
        const float spec = powf(fmaxf(dot3(E, R), 0.0f), p.shininess);          // :213
        const float direct_spec = spec * shadow;                                // :214
        const float spec_occ = 1.0f - sc.w;                                     // :221
        const float sr = (sc.x + spec_occ * direct_spec) * VCT_GB(19);          // :223
        const float sg = (sc.y + spec_occ * direct_spec) * VCT_GB(20);
        const float sb = (sc.z + spec_occ * direct_spec) * VCT_GB(21);
        const float ar = p.ambient * alb_r * occlusion;                         // :225
        const float ag = p.ambient * alb_g * occlusion;
        const float ab = p.ambient * alb_b * occlusion;
        float o0 = ar + dr + sr, o1 = ag + dg + sg, o2 = ab + db + sb, o3 = alb_a;   // :227
        if (!alive) {                                                           // VCT.h:156-159
            const float cc = p.ambient < 0.5f ? 0.5f : 1.0f;
            o0 = cc; o1 = cc; o2 = cc; o3 = 1.0f;
        }
        uint2 pk;
        pk.x = pack_half2(o0, o1);
        pk.y = pack_half2(o2, o3);
        *reinterpret_cast<uint2*>(p.out + pixel_index(fresh_lane()) * 4) = pk;
    }
#undef VCT_GB
}

__device__ __forceinline__ uint32_t to_unorm8_dev(float f) {     // [GL] float -> unorm8, round to nearest
    const float s = f * 255.0f + 0.5f;
    if (!(s > 0.0f)) return 0u;
    if (s >= 255.0f) return 255u;
    return (uint32_t)(int)s;
}

// Second bounce (oracle/vct_oracle.h vcto_bounce), three kernels:
//   k_bounce_list   one wave per touched 8^3 brick (found through the voxelizer's slot table): copies the brick
//                   into the bounce chain (untouched voxels keep their bounce-0 value), compacts its occupied
//                   voxels (ballot + popcount through LDS) and appends them to a global voxel list, one
//                   reservation per workgroup of 16 bricks -- bricks stay contiguous in the list and follow
//                   each other in Morton order, so neighbouring entries are Morton-adjacent voxels;
//   k_bounce_march  one wave per 64 list entries, lane = voxel, the 6 diffuse cones marched with the
//                   same cone_march as the screen trace (voxels of a locally flat surface trace
//                   near-parallel cones, so the cooperative sampler applies);
//   k_bounce_bricks the same march per brick, only for bricks that did not fit the list.
template <class March>
__device__ __forceinline__ void bounce_voxels(const VctTraceParams& p, bool alive, size_t vox, int& total_out,
                                              March march) {
    const uint32_t* __restrict__ level0 = p.chain;          // level 0 starts the chain
    const float fV = (float)p.V;
    // attributes are pooled per touched brick; lanes without a voxel (alive == false) may point at a brick that
    // has no slot: they read slot 0 and their result is discarded
    const uint32_t slot = p.brick_slot[vox >> 9];
    const size_t pv = (size_t)(slot == VCT_NO_SLOT ? 0u : slot) * 512 + (vox & 511u);
    const uint32_t src = level0[vox], nq = p.attr_normal[pv], aq = p.attr_albedo[pv];
    const uint32_t mi = (uint32_t)vox;
    const int i = (int)vct_compact3(mi), j = (int)vct_compact3(mi >> 1), k = (int)vct_compact3(mi >> 2);
    const F3 P = f3((div_rn((float)i + 0.5f, fV) - 0.5f) * p.G, (div_rn((float)j + 0.5f, fV) - 0.5f) * p.G,
                    (div_rn((float)k + 0.5f, fV) - 0.5f) * p.G);
    const F3 nrm = normalize3(f3((float)((int)(nq & 0xffu) - 128), (float)((int)((nq >> 8) & 0xffu) - 128),
                                 (float)((int)((nq >> 16) & 0xffu) - 128)));
    const F3 helper = fabsf(nrm.y) < 0.9f ? f3(0.0f, 1.0f, 0.0f) : f3(1.0f, 0.0f, 0.0f);
    const F3 t = normalize3(cross3(helper, nrm));
    const F3 bt = cross3(nrm, t);
    const F3 start = f3(P.x + nrm.x * p.vs, P.y + nrm.y * p.vs, P.z + nrm.z * p.vs);
    float ind[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int total = 0;
#pragma unroll 1
    for (int c = 0; c < 6; ++c) {
        const float ddx = kConeDirs[3 * c], ddy = kConeDirs[3 * c + 1], ddz = kConeDirs[3 * c + 2];
        F3 dir = f3(t.x * ddx + bt.x * ddy + nrm.x * ddz, t.y * ddx + bt.y * ddy + nrm.y * ddz,
                    t.z * ddx + bt.z * ddy + nrm.z * ddz);
        dir = normalize3(dir);
        int st;
        const F4 cone = march(alive, start, dir, st);
        total += st;
        const float wgt = kConeWeights[c];
        ind[0] = fmaf(wgt, cone.x, ind[0]);
        ind[1] = fmaf(wgt, cone.y, ind[1]);
        ind[2] = fmaf(wgt, cone.z, ind[2]);
        ind[3] = fmaf(wgt, cone.w, ind[3]);
    }
    if (alive) {
        const float occlusion = 1.0f - ind[3];
        const uint32_t r = to_unorm8_dev(unorm8(src & 0xffu) + unorm8(aq & 0xffu) * (occlusion * ind[0]));
        const uint32_t g = to_unorm8_dev(unorm8((src >> 8) & 0xffu) + unorm8((aq >> 8) & 0xffu) * (occlusion * ind[1]));
        const uint32_t bl = to_unorm8_dev(unorm8((src >> 16) & 0xffu) + unorm8((aq >> 16) & 0xffu) * (occlusion * ind[2]));
        p.bounce_out[vox] = r | (g << 8) | (bl << 16) | (src & 0xff000000u);
    } else {
        total = 0;
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    total_out = total;
}

#define VCT_BOUNCE_SETUP                                                     \
    __shared__ float4 lds_blk[VCT_WAVES_PER_BLOCK][2][64];                   \
    const int lane = threadIdx.x & 63;                                       \
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); \
    float4* blk = &lds_blk[wave][0][0];                                      \
    VCT_LUT_DECL                                                             \
    LaneBlock lb;                                                            \
    VCT_LUT_FILL(lb)                                                         \
    if (VCT_LUT) __syncthreads();                                            \
    lb.lane = lane;                                                          \
    lb.lut = (SpreadLut)p.spread_lut; lb.lut_vec = p.spread_lut;                                        \
    lb.sbx = vct_spread3((uint32_t)lane & 3u);                               \
    lb.sby = vct_spread3(((uint32_t)lane >> 2) & 3u) << 1;                   \
    lb.sbz = vct_spread3((uint32_t)lane >> 4) << 2;                          \
    MarchStats ms = {};

// compaction of one brick into `list` (LDS); returns the number of occupied voxels
__device__ __forceinline__ int compact_brick(const VctTraceParams& p, uint32_t b, int lane, uint16_t* list) {
    // a brick without a slot holds nothing of the current mesh (the host refuses the bounce when the attributes
    // are stale, vct_capi.hip attrs_valid; this keeps the index in bounds regardless)
    const uint32_t slot = p.brick_slot[b];
    const bool has = slot != VCT_NO_SLOT;
    const uint32_t* __restrict__ src = p.chain + (size_t)b * 512 + lane;
    const uint32_t* __restrict__ nrm = p.attr_normal + (size_t)(has ? slot : 0u) * 512 + lane;
    // all sixteen loads of the brick in flight before the first ballot (one round trip, not eight)
    uint32_t t[8], a[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) { t[it] = src[it * 64]; a[it] = nrm[it * 64]; }
    int n = 0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        p.bounce_out[(size_t)b * 512 + it * 64 + lane] = t[it];
        const bool occ = (t[it] >> 24) != 0u && has && (a[it] & 0xffffffu) != 0x808080u;
        const unsigned long long m = ballot64(occ);
        if (occ) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(it * 64 + lane);
        n += __popcll(m);
    }
    return n;
}

// 16 waves per workgroup: the list positions of a workgroup's 16 bricks are reserved with ONE atomic on the list's
// counter.  One atomic per brick meant ~10^4 returning atomics on one address at 512^3, which the L2 executes one after
// the other (~12 ns each): they, not the bricks, were the 0.12-0.14 ms this kernel took (round 3).
#define VCT_BLIST_WAVES 16
__global__ void __launch_bounds__(64 * VCT_BLIST_WAVES)
k_bounce_list(const VctTraceParams p) {
    __shared__ uint16_t lds_list[VCT_BLIST_WAVES][512];
    __shared__ uint32_t lds_n[VCT_BLIST_WAVES];
    __shared__ uint32_t lds_base;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint16_t* list = &lds_list[wave][0];
    const uint32_t nwaves = gridDim.x * VCT_BLIST_WAVES;
    // (1) bricks the bounce chain still shows from an older pass and that hold nothing now are cleared.  The flags
    // are scanned 64 bricks at a time, one per lane (a 512^3 grid has 262,144 bricks, 2 % of them touched).
    const uint32_t nchunks = (p.nbricks + 63u) >> 6;
    for (uint32_t c = blockIdx.x * VCT_BLIST_WAVES + wave; c < nchunks; c += nwaves) {
        const uint32_t mine = c * 64u + (uint32_t)lane;
        const bool stale = mine < p.nbricks && p.brick_prev[mine] == 0u && p.bounce_seen[mine] != 0u;
        for (unsigned long long m = ballot64(stale); m != 0ull; m &= m - 1ull) {
            const uint32_t b = c * 64u + (uint32_t)(__ffsll((long long)m) - 1);
            for (int it = 0; it < 8; ++it) p.bounce_out[(size_t)b * 512 + it * 64 + lane] = 0u;
        }
    }
    // (2) the touched bricks, one wave each, found through the voxelizer's slot table (every brick level 0 can show
    // content in has a slot) instead of a scan of the grid
    for (uint32_t s0 = blockIdx.x * VCT_BLIST_WAVES; s0 < p.nslots; s0 += nwaves) {        // workgroup-uniform trip count
        const uint32_t sl = s0 + (uint32_t)wave;
        uint32_t b = 0u;
        int n = 0;
        if (sl < p.nslots) {
            b = p.slot_brick[sl];
            if (p.brick_prev[b]) n = compact_brick(p, b, lane, list);
        }
        if (lane == 0) lds_n[wave] = (uint32_t)n;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0u;
            for (int w = 0; w < VCT_BLIST_WAVES; ++w) total += lds_n[w];
            lds_base = total ? atomicAdd(p.bounce_list_count, total) : 0u;
        }
        __syncthreads();
        if (n != 0) {
            uint32_t off = lds_base;
            for (int w = 0; w < wave; ++w) off += lds_n[w];
            if (off + (uint32_t)n > p.bounce_list_cap) {
                // does not fit: k_bounce_bricks takes this brick (and resets the flag).  At most one brick straddles the
                // end of the list; it marks the entries it reserved there as empty, so the list needs no clear.
                if (lane == 0) p.brick_over[b] = 1u;
                for (uint32_t i = off + (uint32_t)lane; i < p.bounce_list_cap; i += 64u) p.bounce_list[i] = 0xffffffffu;
            } else {
                for (int i = lane; i < n; i += 64) p.bounce_list[off + i] = b * 512u + list[i];
            }
        }
        __syncthreads();           // lds_n / lds_base / the lists are reused by the next round
    }
}

#ifndef VCT_BOUNCE_MIN_WAVES
#define VCT_BOUNCE_MIN_WAVES 5     // the per-voxel frame + attribute state spills under the trace kernel's budget; A/B at 512^3
                                   // (bounce + mips, ms): 4: 0.495, 5: 0.489, 6: 0.500, 7: 0.504
#endif
template <bool WRAP, int FASTDIV>
__global__ void __launch_bounds__(64 * VCT_WAVES_PER_BLOCK, VCT_BOUNCE_MIN_WAVES)
k_bounce_march(const VctTraceParams p) {
    VCT_BOUNCE_SETUP
    const uint32_t n = min(*p.bounce_list_count, p.bounce_list_cap);
    const uint32_t nwaves = gridDim.x * VCT_WAVES_PER_BLOCK;
    unsigned long long wave_steps = 0;
    for (uint32_t w = blockIdx.x * VCT_WAVES_PER_BLOCK + wave; w * 64u < n; w += nwaves) {
        const uint32_t e = w * 64u + lane < n ? p.bounce_list[w * 64u + lane] : 0xffffffffu;
        const bool alive = e != 0xffffffffu;               // unwritten slots belong to overflowed bricks
        const uint32_t first = __builtin_amdgcn_readfirstlane(e);
        const size_t vox = alive ? e : (first != 0xffffffffu ? first : 0u);
        int total;
        bounce_voxels(p, alive, vox, total, [&](bool al, F3 start, F3 dir, int& st) {
            return cone_march<WRAP, FASTDIV, true>(p, al, start, dir, p.steps_diffuse, p.n_diffuse, blk, lb, st, ms);
        });
        wave_steps += (unsigned long long)total;
    }
    if (lane == 0 && wave_steps)
        atomicAdd(p.step_counter + ((blockIdx.x * VCT_WAVES_PER_BLOCK + wave) & (VCT_STEP_COUNTERS - 1)), wave_steps);
}

template <bool WRAP, int FASTDIV>
__global__ void __launch_bounds__(64 * VCT_WAVES_PER_BLOCK, VCT_BOUNCE_MIN_WAVES)
k_bounce_bricks(const VctTraceParams p) {
    VCT_BOUNCE_SETUP
    __shared__ uint16_t lds_list[VCT_WAVES_PER_BLOCK][512];
    uint16_t* list = &lds_list[wave][0];
    unsigned long long wave_steps = 0;
    const uint32_t nwaves = gridDim.x * VCT_WAVES_PER_BLOCK;
    // flags of the slots scanned 64 at a time, one per lane; a served brick's flag is reset for the next pass
    const uint32_t nchunks = (p.nslots + 63u) >> 6;
    for (uint32_t c = blockIdx.x * VCT_WAVES_PER_BLOCK + wave; c < nchunks; c += nwaves) {
        const uint32_t sl = c * 64u + (uint32_t)lane;
        const uint32_t mine = sl < p.nslots ? p.slot_brick[sl] : 0u;
        const bool over = sl < p.nslots && p.brick_over[mine] != 0u;
        if (over) p.brick_over[mine] = 0u;
        for (unsigned long long m = ballot64(over); m != 0ull; m &= m - 1ull) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)m) - 1);
            const int n = compact_brick(p, b, lane, list);
            wave_sync();
            for (int base = 0; base < n; base += 64) {
                const bool alive = base + lane < n;
                const size_t vox = (size_t)b * 512 + (alive ? list[base + lane] : list[base]);
                int total;
                bounce_voxels(p, alive, vox, total, [&](bool al, F3 start, F3 dir, int& st) {
                    return cone_march<WRAP, FASTDIV, true>(p, al, start, dir, p.steps_diffuse, p.n_diffuse, blk, lb, st, ms);
                });
                wave_steps += (unsigned long long)total;
            }
            wave_sync();
        }
    }
    if (lane == 0 && wave_steps)
        atomicAdd(p.step_counter + ((blockIdx.x * VCT_WAVES_PER_BLOCK + wave) & (VCT_STEP_COUNTERS - 1)), wave_steps);
}

// Live-pixel compaction (experiment, north_star "compact still-active cones"; profiles/experiments/README.md): one wave
// per 16x16-pixel super-tile of the launched rows gathers the live pixels (albedo.a >= 0.5, trace.fs:169-172) of its up
// to four 8x8 tiles into whole waves of 64 -- "virtual tiles" -- and gives the discarded pixels their clear colour
// (VCT.h:156-159) right away.  The trace then runs one workgroup per virtual tile.
__global__ void __launch_bounds__(256)
k_compact_tiles(const VctTraceParams p) {
    const int lane = threadIdx.x & 63;
    const int rows = p.tile_row1 - p.tile_row0;
    const int sx_n = (p.tiles_x + 1) >> 1, sy_n = (rows + 1) >> 1;
    const int st = blockIdx.x * (blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    if (st >= sx_n * sy_n) return;
    const int sy = st / sx_n, sx = st - sy * sx_n;
    unsigned long long m[4];
    int tiles[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tx = 2 * sx + (k & 1), ty = p.tile_row0 + 2 * sy + (k >> 1);
        const bool have = tx < p.tiles_x && ty < p.tile_row1;
        tiles[k] = have ? ty * p.tiles_x + tx : -1;
        bool live = false;
        if (have) {
            const int x = tx * VCT_TILE + (lane & 7), y = ty * VCT_TILE + (lane >> 3);
            const bool in_frame = x < p.width && y < p.height;
            const float a = p.gbuf[(size_t)tiles[k] * (VCT_GB_NPLANES * VCT_TILE_PIX) + 18 * VCT_TILE_PIX + lane];
            live = in_frame && !(a < 0.5f);
            if (in_frame && !live) {
                const float cc = p.ambient < 0.5f ? 0.5f : 1.0f;
                uint2 pk;
                pk.x = pack_half2(cc, cc);
                pk.y = pack_half2(cc, 1.0f);
                *reinterpret_cast<uint2*>(p.out + ((size_t)y * p.width + x) * 4) = pk;
            }
        }
        m[k] = ballot64(live);
    }
    const int total = (int)(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
    const int nvt = (total + 63) >> 6;
    uint32_t base = 0u;
    if (lane == 0 && nvt) base = atomicAdd(p.vt_count, (uint32_t)nvt);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    int n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((m[k] >> lane) & 1ull)
            p.vt_pix[(size_t)base * 64 + n + __popcll(m[k] & ((1ull << lane) - 1ull))] = ((uint32_t)tiles[k] << 6) | (uint32_t)lane;
        n += (int)__popcll(m[k]);
    }
    for (int i = total + lane; i < nvt * 64; i += 64) p.vt_pix[(size_t)base * 64 + i] = 0xffffffffu;
}

template <bool WRAP, int FASTDIV, bool COOP>
hipError_t launch(const VctTraceParams& p, int blocks, hipStream_t s) {
    hipLaunchKernelGGL((k_trace_tile<WRAP, FASTDIV, COOP>), dim3(blocks),
                       dim3(64 * VCT_WAVES_PER_BLOCK), 0, s, p);
    return hipGetLastError();
}

template <bool WRAP, int FASTDIV>
hipError_t launch_v(const VctTraceParams& p, int variant, int ntiles, hipStream_t s) {
    if (!p.aniso && (variant == 1 || variant == 2)) {      // the anisotropic option exists in the default kernel only
        const int nblocks = (ntiles + VCT_WAVES_PER_BLOCK - 1) / VCT_WAVES_PER_BLOCK;
        const int blocks = ((nblocks + 7) / 8) * 8;     // whole rounds of the 8 XCDs
        return variant == 1 ? launch<WRAP, FASTDIV, false>(p, blocks, s) : launch<WRAP, FASTDIV, true>(p, blocks, s);
    }
    const int blocks = ((ntiles + 7) / 8) * 8;
    if (p.vt_pix) {         // variant 4: compaction pre-pass (the counter was zeroed by the caller)
        const int rows = p.tile_row1 - p.tile_row0;
        const int nst = ((p.tiles_x + 1) >> 1) * ((rows + 1) >> 1);
        hipLaunchKernelGGL(k_compact_tiles, dim3((nst + 3) / 4), dim3(256), 0, s, p);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // variant 3: the default kernel with the one-multiply decode and reciprocal-multiply divisions -- never the default,
    // not bit-exact; it exists to price the exactness (bench.py exactness_tax)
    if (variant == 3 && !p.aniso) {
        hipLaunchKernelGGL((k_trace_tile_split<WRAP, 2, false>), dim3(blocks), dim3(64 * VCT_SPLIT), 0, s, p);
        return hipGetLastError();
    }
    if (p.vt_pix) {
        hipLaunchKernelGGL((k_trace_tile_split<WRAP, FASTDIV, false, true>), dim3(blocks), dim3(64 * VCT_SPLIT), 0, s, p);
        return hipGetLastError();
    }
    if (p.aniso) hipLaunchKernelGGL((k_trace_tile_split<WRAP, FASTDIV, true>), dim3(blocks), dim3(64 * VCT_SPLIT), 0, s, p);
    else if (VCT_CELLS && WRAP && p.cells_biased)
        hipLaunchKernelGGL((k_trace_tile_split<WRAP, FASTDIV, false, false, true>), dim3(blocks), dim3(64 * VCT_SPLIT), 0, s, p);
    else if (VCT_LOAD_PRIO && !p.spec_prio)     // a whole frame (or most of one): issue priority around the samples' loads (sample_level)
        hipLaunchKernelGGL((k_trace_tile_split<WRAP, FASTDIV, false, false, false, true>), dim3(blocks), dim3(64 * VCT_SPLIT), 0, s, p);
    else hipLaunchKernelGGL((k_trace_tile_split<WRAP, FASTDIV, false>), dim3(blocks), dim3(64 * VCT_SPLIT), 0, s, p);
    return hipGetLastError();
}

// every fp32 bit pattern: div_const<true> against the IEEE divide
__global__ void __launch_bounds__(256)
k_divide_selftest(float d, float r, float aux, unsigned long long* mismatches) {
    unsigned long long bad = 0;
    const unsigned long long total = 1ull << 32;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        if (!(fabsf(x) <= 3.0e38f)) continue;                 // inf / nan
        if (fabsf(x) < VCT_DIV_TINY && (uint32_t)i != 0u) continue;   // outside the documented domain
        const float want = x / d;
        if (want != 0.0f && fabsf(want) < 1.17549435e-38f) continue;   // subnormal quotient
        if (fabsf(want) > 3.0e38f) continue;
        const float got = div_const<1>(x, aux, r);
        if (__float_as_uint(got) != __float_as_uint(want)) { ++bad; mismatches[1] = i; }
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

// The texel buffer's conversion against the exact decode, texel by texel: out[0] = channels that differ, out[1] = the
// first offending texel word (vct_selftest_texel_buffer; vct_create runs it once per process)
__global__ void __launch_bounds__(256)
k_texel_buffer_selftest(const uint32_t* texels, uint32_t n, unsigned long long* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 d = texel_f32(level_texel_buffer(texels), i);
    const uint32_t t = texels[i];
    const float want[4] = {vct_unorm8_to_float(t & 0xffu), vct_unorm8_to_float((t >> 8) & 0xffu),
                           vct_unorm8_to_float((t >> 16) & 0xffu), vct_unorm8_to_float(t >> 24)};
    const float got[4] = {d.x, d.y, d.z, d.w};
    for (int ch = 0; ch < 4; ++ch)
        if (__float_as_uint(got[ch]) != __float_as_uint(want[ch])) { atomicAdd(&out[0], 1ull); out[1] = t; }
}

}  // namespace

hipError_t vct_launch_texel_buffer_selftest(const uint32_t* texels, uint32_t n, unsigned long long* out, hipStream_t s) {
    hipLaunchKernelGGL(k_texel_buffer_selftest, dim3((n + 255u) / 256u), dim3(256), 0, s, texels, n, out);
    return hipGetLastError();
}

hipError_t vct_launch_divide_selftest(float d, unsigned long long* mismatches, hipStream_t s) {
    const float r = 1.0f / d;
    hipLaunchKernelGGL(k_divide_selftest, dim3(256 * 16), dim3(256), 0, s, d, r, vct_div_aux(d, r), mismatches);
    return hipGetLastError();
}

template <bool WRAP, int FASTDIV>
hipError_t launch_bounce(const VctTraceParams& p, hipStream_t s) {
    uint32_t blocks = (p.nbricks + VCT_WAVES_PER_BLOCK - 1) / VCT_WAVES_PER_BLOCK;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    const dim3 block(64 * VCT_WAVES_PER_BLOCK);
    uint32_t lblocks = (p.nslots + VCT_BLIST_WAVES - 1) / VCT_BLIST_WAVES;               // one wave per touched brick
    if (lblocks > 256u * 8u) lblocks = 256u * 8u;
    if (lblocks < 64u) lblocks = 64u;
    hipLaunchKernelGGL(k_bounce_list, dim3(lblocks), dim3(64 * VCT_BLIST_WAVES), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_bounce_march<WRAP, FASTDIV>), dim3(256 * 24), block, 0, s, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_bounce_bricks<WRAP, FASTDIV>), dim3(blocks), block, 0, s, p);
    return hipGetLastError();
}

hipError_t vct_launch_bounce(const VctTraceParams& p, hipStream_t s) {
    if (p.nbricks == 0) return hipSuccess;
    if (p.wrap_repeat) return p.fast_div ? launch_bounce<true, true>(p, s) : launch_bounce<true, false>(p, s);
    return p.fast_div ? launch_bounce<false, true>(p, s) : launch_bounce<false, false>(p, s);
}

// variant 0 (default): cooperative sampler, each tile split over 3 waves; 2: cooperative sampler, one
// wave per tile; 1: per-lane sampler only, one wave per tile.
hipError_t vct_launch_trace(const VctTraceParams& p, int variant, hipStream_t s) {
    const int rstride = p.row_stride > 1 ? p.row_stride : 1;
    const int ntiles = ((p.tile_row1 - p.tile_row0 + rstride - 1) / rstride) * p.tiles_x;
    if (ntiles <= 0) return hipSuccess;
    if ((rstride > 1 || p.pack_rows) && (variant == 1 || variant == 2 || variant == 4)) return hipErrorInvalidValue;   // the default kernel only
    if (p.wrap_repeat)
        return p.fast_div ? launch_v<true, true>(p, variant, ntiles, s)
                          : launch_v<true, false>(p, variant, ntiles, s);
    return p.fast_div ? launch_v<false, true>(p, variant, ntiles, s)
                      : launch_v<false, false>(p, variant, ntiles, s);
}
