// vct_trace.hip -- per-pixel cone trace (6 diffuse + 1 specular) through the Morton brick chain.
//
// Replaces the fragment stage of the reference's main draw: S/VoxelConeTracing.fs:59-66
// (SampleVoxels), :82-107 (Voxel_Cone_Tracing), :165-228 (gather + composite), with the driver's
// textureLod (trilinear x 2 levels, GL_REPEAT, texel centres -- OpenGL 4.3 core)
// written out by hand because there is no texture unit behind a HIP pointer.
//
// Arithmetic contract: the march is operation-for-operation the scalar oracle's (fp32, explicit
// fmaf only where the oracle has one, IEEE divide/sqrt), so per-cone step counts and the raw cone
// vec4s are bit-identical; this file must be compiled with -ffp-contract=off.
//
// Mapping: one wavefront per 8x8 screen tile, lane = pixel.  Gather-bound (no MFMA): the
// dominant cost is the 16 texel fetches per step, so the work is organised around where those
// fetches are served from -- the coarse tail of the chain is staged once per workgroup in LDS,
// fine levels come through L1/L2/Infinity Cache from the Morton chain.
#include <hip/hip_fp16.h>

#include "vct_internal.h"

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

// unorm8 -> float, bit-identical to (float)c / 255.0f for every c in [0,255]:
// c * RN(1/255) misses for 126 of the 256 bytes; the two-term product below never does
// (checked exhaustively in tests/test_host_logic.py::test_unorm8_decode_exact).
__device__ __forceinline__ float unorm8(uint32_t c) {
    const float f = (float)c;
    return fmaf(f, 0x1.010102p-8f, f * -0x1.fdfdfep-33f);
}

struct Sampler {
    const uint32_t* chain;
    const uint32_t* lds;        // staged tail, texel 0 = first texel of level lds_first
    uint32_t lds_base_off;      // level_off[lds_first]
    int lds_first;
    int V;
    int wrap;
};

// [GL] tri(level): trilinear, texel centres, REPEAT (or clamp).  `level` is wave-uniform.
__device__ __forceinline__ F4 tri_sample(const Sampler& s, const uint32_t* __restrict__ level_off,
                                         int level, float ux, float uy, float uz) {
    const int N = s.V >> level;
    const float fN = (float)N;
    const float u = ux * fN - 0.5f, v = uy * fN - 0.5f, w = uz * fN - 0.5f;
    const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    const float a = u - fu, b = v - fv, c = w - fw;
    int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
    int i1 = i0 + 1, j1 = j0 + 1, k1 = k0 + 1;
    const int m = N - 1;
    if (s.wrap) {
        i0 &= m; i1 &= m; j0 &= m; j1 &= m; k0 &= m; k1 &= m;
    } else {
        i0 = min(max(i0, 0), m); i1 = min(max(i1, 0), m);
        j0 = min(max(j0, 0), m); j1 = min(max(j1, 0), m);
        k0 = min(max(k0, 0), m); k1 = min(max(k1, 0), m);
    }
    const uint32_t mx0 = vct_spread3((uint32_t)i0), mx1 = vct_spread3((uint32_t)i1);
    const uint32_t my0 = vct_spread3((uint32_t)j0) << 1, my1 = vct_spread3((uint32_t)j1) << 1;
    const uint32_t mz0 = vct_spread3((uint32_t)k0) << 2, mz1 = vct_spread3((uint32_t)k1) << 2;
    uint32_t t[8];
    if (level >= s.lds_first) {
        const uint32_t* base = s.lds + (level_off[level] - s.lds_base_off);
        t[0] = base[mx0 | my0 | mz0]; t[1] = base[mx1 | my0 | mz0];
        t[2] = base[mx0 | my1 | mz0]; t[3] = base[mx1 | my1 | mz0];
        t[4] = base[mx0 | my0 | mz1]; t[5] = base[mx1 | my0 | mz1];
        t[6] = base[mx0 | my1 | mz1]; t[7] = base[mx1 | my1 | mz1];
    } else {
        const uint32_t* base = s.chain + level_off[level];
        t[0] = base[mx0 | my0 | mz0]; t[1] = base[mx1 | my0 | mz0];
        t[2] = base[mx0 | my1 | mz0]; t[3] = base[mx1 | my1 | mz0];
        t[4] = base[mx0 | my0 | mz1]; t[5] = base[mx1 | my0 | mz1];
        t[6] = base[mx0 | my1 | mz1]; t[7] = base[mx1 | my1 | mz1];
    }
    const float a0 = 1.0f - a, b0 = 1.0f - b, c0 = 1.0f - c;
    const float wg[8] = {(a0 * b0) * c0, (a * b0) * c0, (a0 * b) * c0, (a * b) * c0,
                         (a0 * b0) * c,  (a * b0) * c,  (a0 * b) * c,  (a * b) * c};
    F4 acc;
    acc.x = wg[0] * unorm8(t[0] & 0xffu);
    acc.y = wg[0] * unorm8((t[0] >> 8) & 0xffu);
    acc.z = wg[0] * unorm8((t[0] >> 16) & 0xffu);
    acc.w = wg[0] * unorm8(t[0] >> 24);
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        acc.x = fmaf(wg[i], unorm8(t[i] & 0xffu), acc.x);
        acc.y = fmaf(wg[i], unorm8((t[i] >> 8) & 0xffu), acc.y);
        acc.z = fmaf(wg[i], unorm8((t[i] >> 16) & 0xffu), acc.z);
        acc.w = fmaf(wg[i], unorm8(t[i] >> 24), acc.w);
    }
    return acc;
}

// trace.fs:82-107 with the pixel-independent step sequence read from `tab`.
__device__ __forceinline__ F4 cone_march(const VctTraceParams& p, const Sampler& s, bool alive,
                                         F3 start, F3 dir, const VctStep* __restrict__ tab, int n,
                                         int& steps_out) {
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, alpha = 0.0f, occ = 0.0f;
    int steps = 0;
    for (int k = 0; k < n; ++k) {
        const bool act = alive && (alpha < p.max_alpha);     // trace.fs:94 (dist < MAX: table)
        if (!__any(act)) break;
        const VctStep st = tab[k];
        if (act) {
            // trace.fs:98 + :61-63
            const float px = start.x + dir.x * st.dist;
            const float py = start.y + dir.y * st.dist;
            const float pz = start.z + dir.z * st.dist;
            const float ux = div_rn(px, p.half_G) * 0.5f + 0.5f;
            const float uy = div_rn(py, p.half_G) * 0.5f + 0.5f;
            const float uz = div_rn(pz, p.half_G) * 0.5f + 0.5f;
            F4 vc = tri_sample(s, p.level_off, st.level, ux, uy, uz);
            if (st.two_levels && st.frac != 0.0f) {
                const F4 t2 = tri_sample(s, p.level_off, st.level2, ux, uy, uz);
                const float g = 1.0f - st.frac;
                vc.x = fmaf(st.frac, t2.x, g * vc.x);
                vc.y = fmaf(st.frac, t2.y, g * vc.y);
                vc.z = fmaf(st.frac, t2.z, g * vc.z);
                vc.w = fmaf(st.frac, t2.w, g * vc.w);
            }
            const float oma = 1.0f - alpha;
            cr = fmaf(oma, vc.x, cr);                                  // :100
            cg = fmaf(oma, vc.y, cg);
            cb = fmaf(oma, vc.z, cb);
            occ = occ + div_rn(oma * vc.w, st.occ_den);                // :101
            alpha = fmaf(oma, vc.w, alpha);                            // :102
            ++steps;
        }
    }
    steps_out = steps;
    return {cr, cg, cb, occ};
}

__device__ __forceinline__ uint32_t pack_half2(float a, float b) {
    const __half ha = __float2half_rn(a), hb = __float2half_rn(b);
    return (uint32_t)__half_as_ushort(ha) | ((uint32_t)__half_as_ushort(hb) << 16);
}

__constant__ float kConeDirs[18] = {0.0f, 0.0f, 1.0f,
                                    0.0f, 0.866025f, 0.5f,
                                    0.823639f, 0.267617f, 0.5f,
                                    0.509037f, -0.700629f, 0.5f,
                                    -0.509037f, -0.700629f, 0.5f,
                                    -0.823639f, 0.267617f, 0.5f};            // trace.fs:49-57
__constant__ float kConeWeights[6] = {0.25f, 0.15f, 0.15f, 0.15f, 0.15f, 0.15f};   // trace.fs:48

// One wave per tile, lane = pixel, the 7 cones in sequence.  Workgroups are persistent over a
// strided tile range so the LDS copy of the coarse levels is paid once per workgroup.
__global__ void __launch_bounds__(1024)
k_trace_tile(const VctTraceParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;

    // stage the coarse tail of the chain (levels >= lds_first_level) in LDS
    Sampler s;
    s.chain = p.chain;
    s.lds = lds;
    s.lds_first = p.lds_first_level;
    s.V = p.V;
    s.wrap = p.wrap_repeat;
    s.lds_base_off = 0;
    if (p.lds_first_level < p.nlev) {
        s.lds_base_off = p.level_off[p.lds_first_level];
        const uint32_t end = p.level_off[p.nlev - 1] + 1u;
        const uint32_t count = end - s.lds_base_off;
        const uint32_t* src = p.chain + s.lds_base_off;
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) lds[i] = src[i];
        __syncthreads();
    }

    const int ntiles = (p.tile_row1 - p.tile_row0) * p.tiles_x;
    unsigned long long wave_steps = 0;
    for (int ti = blockIdx.x * waves_per_block + wave; ti < ntiles;
         ti += gridDim.x * waves_per_block) {
        const int tile = p.tile_row0 * p.tiles_x + ti;
        const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const int x = tx * VCT_TILE + (lane & 7), y = ty * VCT_TILE + (lane >> 3);
        const float* gb = p.gbuf + (size_t)tile * (VCT_GB_NPLANES * VCT_TILE_PIX) + lane;
        float g[VCT_GB_NPLANES];
#pragma unroll
        for (int k = 0; k < VCT_GB_NPLANES; ++k) g[k] = gb[k * VCT_TILE_PIX];

        const bool in_frame = (x < p.width) && (y < p.height);
        const bool alive = in_frame && !(g[18] < 0.5f);                 // trace.fs:171 discard
        const F3 P = f3(g[0], g[1], g[2]), Nw = f3(g[3], g[4], g[5]);
        const F3 T = f3(g[6], g[7], g[8]), B = f3(g[9], g[10], g[11]);
        const F3 N = f3(g[12], g[13], g[14]);
        const float shadow = g[22];

        // trace.fs:175: inverse(transpose(mat3(T,B,N))) = columns (BxN, NxT, TxB) / det
        const F3 c0 = cross3(B, Nw), c1 = cross3(Nw, T), c2 = cross3(T, B);
        const float inv_det = div_rn(1.0f, dot3(T, c0));
        const F3 k0 = f3(c0.x * inv_det, c0.y * inv_det, c0.z * inv_det);
        const F3 k1 = f3(c1.x * inv_det, c1.y * inv_det, c1.z * inv_det);
        const F3 k2 = f3(c2.x * inv_det, c2.y * inv_det, c2.z * inv_det);

        const F3 L = normalize3(f3(p.light[0], p.light[1], p.light[2]));        // :179
        const F3 E = normalize3(f3(p.cam[0] - P.x, p.cam[1] - P.y, p.cam[2] - P.z));   // :181
        const float cos_theta = fmaxf(dot3(N, L), 0.0f);                        // :188
        const float direct_diffuse = shadow * cos_theta;                        // :192
        const F3 start = f3(P.x + Nw.x * p.vs, P.y + Nw.y * p.vs, P.z + Nw.z * p.vs);   // :92

        float ind[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int nsteps[7];
        const size_t pix = (size_t)y * p.width + x;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) {                                           // :196-199
            const float dx = kConeDirs[3 * i], dy = kConeDirs[3 * i + 1], dz = kConeDirs[3 * i + 2];
            F3 dir = f3(k0.x * dx + k1.x * dy + k2.x * dz, k0.y * dx + k1.y * dy + k2.y * dz,
                        k0.z * dx + k1.z * dy + k2.z * dz);
            dir = normalize3(dir);
            int st;
            const F4 c = cone_march(p, s, alive, start, dir, p.steps_diffuse, p.n_diffuse, st);
            nsteps[i] = st;
            const float wgt = kConeWeights[i];
            ind[0] = fmaf(wgt, c.x, ind[0]);
            ind[1] = fmaf(wgt, c.y, ind[1]);
            ind[2] = fmaf(wgt, c.z, ind[2]);
            ind[3] = fmaf(wgt, c.w, ind[3]);
            if (p.dbg_cones && alive) {
                float* d = p.dbg_cones + pix * 28 + 4 * i;
                d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
            }
        }
        const float occlusion = 1.0f - ind[3];                                  // :201
        const float dr = (direct_diffuse + occlusion * ind[0]) * g[15];         // :205
        const float dg = (direct_diffuse + occlusion * ind[1]) * g[16];
        const float db = (direct_diffuse + occlusion * ind[2]) * g[17];

        const F3 R = normalize3(reflect3(f3(L.x * -1.0f, L.y * -1.0f, L.z * -1.0f), N));   // :212
        const float spec = powf(fmaxf(dot3(E, R), 0.0f), p.shininess);          // :213
        const float direct_spec = spec * shadow;                                // :214
        const F3 Rd = normalize3(reflect3(f3(E.x * -1.0f, E.y * -1.0f, E.z * -1.0f), N));  // :217
        int st6;
        const F4 sc = cone_march(p, s, alive, start, Rd, p.steps_specular, p.n_specular, st6);
        nsteps[6] = st6;
        if (p.dbg_cones && alive) {
            float* d = p.dbg_cones + pix * 28 + 24;
            d[0] = sc.x; d[1] = sc.y; d[2] = sc.z; d[3] = sc.w;
        }
        const float spec_occ = 1.0f - sc.w;                                     // :221
        const float sr = (sc.x + spec_occ * direct_spec) * g[19];               // :223
        const float sg = (sc.y + spec_occ * direct_spec) * g[20];
        const float sb = (sc.z + spec_occ * direct_spec) * g[21];
        const float ar = p.ambient * g[15] * occlusion;                         // :225
        const float ag = p.ambient * g[16] * occlusion;
        const float ab = p.ambient * g[17] * occlusion;

        float o0 = ar + dr + sr, o1 = ag + dg + sg, o2 = ab + db + sb, o3 = g[18];   // :227
        if (!alive) {                                                           // VCT.h:156-159
            const float cc = p.ambient < 0.5f ? 0.5f : 1.0f;
            o0 = cc; o1 = cc; o2 = cc; o3 = 1.0f;
        }
        int total = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) total += alive ? nsteps[i] : 0;
        if (in_frame) {
            uint2 pk;
            pk.x = pack_half2(o0, o1);
            pk.y = pack_half2(o2, o3);
            *reinterpret_cast<uint2*>(p.out + pix * 4) = pk;
            if (p.dbg_steps) {
                uint8_t* d = p.dbg_steps + pix * 7;
#pragma unroll
                for (int i = 0; i < 7; ++i) d[i] = alive ? (uint8_t)nsteps[i] : (uint8_t)0;
            }
        }
        // wave-level reduction of the executed step count
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
        wave_steps += (unsigned long long)total;
    }
    if (lane == 0 && wave_steps) atomicAdd(p.step_counter, wave_steps);
}

}  // namespace

hipError_t vct_launch_trace(const VctTraceParams& p, int variant, hipStream_t s) {
    (void)variant;
    const int ntiles = (p.tile_row1 - p.tile_row0) * p.tiles_x;
    if (ntiles <= 0) return hipSuccess;
    size_t lds_bytes = 0;
    if (p.lds_first_level < p.nlev)
        lds_bytes = 4u * (size_t)(p.level_off[p.nlev - 1] + 1u - p.level_off[p.lds_first_level]);
    // wide workgroups when the staged tail is large (one workgroup per CU), else 256 threads
    const int threads = lds_bytes > 40 * 1024 ? 1024 : 256;
    const int waves = threads / 64;
    int blocks = (ntiles + waves - 1) / waves;
    const int max_blocks = lds_bytes > 40 * 1024 ? 256 : 256 * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_trace_tile),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_trace_tile, dim3(blocks), dim3(threads), lds_bytes, s, p);
    return hipGetLastError();
}
