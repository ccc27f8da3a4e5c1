// vct_voxelize.hip -- triangle voxelization + light injection into level 0 of the brick chain.
//
// Replaces the reference's voxelization draw (VCT.h:213-245): S/Voxelization.vs:15-22 (world
// position, shadow coordinate), S/Voxelization.gs:22-51 (dominant axis), S/Voxelization.fs:18-89
// (PCF, voxel index, imageStore of albedo*shadow).  North-star mode: conservative triangle /
// voxel-box overlap (Schwarz & Seidel 2010) instead of pixel-centre raster, and instead of the
// racy last-writer imageStore every fragment adds its unorm8 value into per-voxel 64-bit integer
// accumulators (sumR|sumG, sumB|count), which is exact and order-independent; vct_launch_resolve
// turns them into the rounded mean.  fp32 operation order mirrors the scalar oracle
// (compile with -ffp-contract=off).
//
// Work distribution: one thread per triangle for triangles whose voxel bounding box is small;
// larger ones are deferred to a second pass with one workgroup per triangle.
#include "vct_internal.h"

namespace {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float comp(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

__device__ __forceinline__ F3 xform_point(const float* m, F3 p) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}

// [GL] bilinear, clamp-to-edge depth fetch (VCT.h:93-96)
__device__ __forceinline__ float shadow_tex(const float* __restrict__ depth, int S, float u, float v) {
    const float fS = (float)S;
    const float x = u * fS - 0.5f, y = v * fS - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const float top = (float)(S - 1);
    auto cl = [&](float f) -> int {
        if (!(f > 0.0f)) return 0;
        if (f >= top) return S - 1;
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

// vox.fs:18-52 (count of passing taps; caller divides by 25)
__device__ __forceinline__ int pcf25(const float* __restrict__ depth, int S, F3 c, float bias) {
    const float inv = __fdiv_rn(1.0f, (float)S);
    int count = 0;
    for (int x = -2; x <= 2; ++x)
        for (int y = -2; y <= 2; ++y) {
            const float ox = inv * (float)x, oy = inv * (float)y;
            const float closest = shadow_tex(depth, S, c.x + ox, c.y + oy);
            if (c.z - bias <= closest) ++count;
        }
    return count;
}

__device__ __forceinline__ uint32_t to_unorm8(float f) {
    const float s = f * 255.0f + 0.5f;
    if (!(s > 0.0f)) return 0u;
    if (s >= 255.0f) return 255u;
    return (uint32_t)(int)s;
}

struct TriSetup {
    F3 g[3];     // voxel-space vertices
    F3 dc[3];    // shadow coordinates
    F3 n;
    float d1, d2;
    float ne[3][3][2];
    float de[3][3];
    int lo[3], hi[3];
    int ua, ub;
    float area;
    float alb[3];
    bool valid;
};

__device__ __forceinline__ void setup_tri(const VctVoxParams& p, int t, TriSetup& r) {
    F3 w[3];
    const float fV = (float)p.V;
    for (int k = 0; k < 3; ++k) {
        const float* q = p.pos + (size_t)t * 9 + 3 * k;
        w[k] = {q[0] * p.model_scale, q[1] * p.model_scale, q[2] * p.model_scale};   // vox.vs:21
        const F3 d = xform_point(p.light_vp, w[k]);                                  // vox.vs:18
        r.dc[k] = {d.x * 0.5f + 0.5f, d.y * 0.5f + 0.5f, d.z * 0.5f + 0.5f};          // vox.vs:19
        r.g[k] = {(__fdiv_rn(w[k].x, p.G) + 0.5f) * fV, (__fdiv_rn(w[k].y, p.G) + 0.5f) * fV,
                  (__fdiv_rn(w[k].z, p.G) + 0.5f) * fV};
    }
    // vox.gs:24-39 dominant axis
    const F3 e1 = sub3(w[0], w[1]), e2 = sub3(w[2], w[0]);
    F3 nn = cross3(e1, e2);
    const float len = __builtin_sqrtf(dot3(nn, nn));
    const float nx = fabsf(__fdiv_rn(nn.x, len)), ny = fabsf(__fdiv_rn(nn.y, len)),
                nz = fabsf(__fdiv_rn(nn.z, len));
    int axis;
    if (nx >= ny && nx >= nz) axis = 1;
    else if (ny >= nx && ny >= nz) axis = 2;
    else axis = 3;

    const F3 e[3] = {sub3(r.g[1], r.g[0]), sub3(r.g[2], r.g[1]), sub3(r.g[0], r.g[2])};
    r.n = cross3(e[0], e[1]);
    r.valid = !((r.n.x == 0.0f && r.n.y == 0.0f && r.n.z == 0.0f) || r.n.x != r.n.x ||
                r.n.y != r.n.y || r.n.z != r.n.z);
    const F3 cp = {r.n.x > 0.0f ? 1.0f : 0.0f, r.n.y > 0.0f ? 1.0f : 0.0f, r.n.z > 0.0f ? 1.0f : 0.0f};
    r.d1 = dot3(r.n, sub3(cp, r.g[0]));
    r.d2 = dot3(r.n, sub3(sub3(F3{1.0f, 1.0f, 1.0f}, cp), r.g[0]));
    const float nsel[3] = {r.n.z, r.n.x, r.n.y};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const int a0 = pl, a1 = (pl + 1) % 3;    // xy, yz, zx
        const float sg = nsel[pl] >= 0.0f ? 1.0f : -1.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float ea = comp(e[i], a0), eb = comp(e[i], a1);
            const float na = -eb * sg, nb = ea * sg;
            r.ne[pl][i][0] = na;
            r.ne[pl][i][1] = nb;
            const float va = comp(r.g[i], a0), vb = comp(r.g[i], a1);
            r.de[pl][i] = -(na * va + nb * vb) + fmaxf(0.0f, na) + fmaxf(0.0f, nb);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float mn = fminf(fminf(comp(r.g[0], a), comp(r.g[1], a)), comp(r.g[2], a));
        const float mx = fmaxf(fmaxf(comp(r.g[0], a), comp(r.g[1], a)), comp(r.g[2], a));
        r.lo[a] = max((int)floorf(mn), 0);
        r.hi[a] = min((int)floorf(mx), p.V - 1);
    }
    r.ua = axis == 1 ? 1 : 0;
    r.ub = axis == 3 ? 1 : 2;
    const float ax0 = comp(r.g[0], r.ua), ay0 = comp(r.g[0], r.ub);
    const float ax1 = comp(r.g[1], r.ua), ay1 = comp(r.g[1], r.ub);
    const float ax2 = comp(r.g[2], r.ua), ay2 = comp(r.g[2], r.ub);
    r.area = (ax1 - ax0) * (ay2 - ay0) - (ax2 - ax0) * (ay1 - ay0);
    const float* alb = p.albedo + 4 * (size_t)p.material[t];
    r.alb[0] = alb[0]; r.alb[1] = alb[1]; r.alb[2] = alb[2];
}

__device__ __forceinline__ bool overlap(const TriSetup& c, int i, int j, int k) {
    const F3 pp = {(float)i, (float)j, (float)k};
    const float np = dot3(c.n, pp);
    if ((np + c.d1) * (np + c.d2) > 0.0f) return false;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const float pa = comp(pp, pl), pb = comp(pp, (pl + 1) % 3);
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (c.ne[pl][e][0] * pa + c.ne[pl][e][1] * pb + c.de[pl][e] < 0.0f) return false;
    }
    return true;
}

__device__ __forceinline__ void fragment(const VctVoxParams& p, const TriSetup& r, int i, int j,
                                         int k) {
    if (!overlap(r, i, j, k)) return;
    const F3 ctr = {(float)i + 0.5f, (float)j + 0.5f, (float)k + 0.5f};
    const float cx = comp(ctr, r.ua), cy = comp(ctr, r.ub);
    const float ax0 = comp(r.g[0], r.ua), ay0 = comp(r.g[0], r.ub);
    const float ax1 = comp(r.g[1], r.ua), ay1 = comp(r.g[1], r.ub);
    const float ax2 = comp(r.g[2], r.ua), ay2 = comp(r.g[2], r.ub);
    float b0 = __fdiv_rn((ax1 - cx) * (ay2 - cy) - (ax2 - cx) * (ay1 - cy), r.area);
    float b1 = __fdiv_rn((ax2 - cx) * (ay0 - cy) - (ax0 - cx) * (ay2 - cy), r.area);
    b0 = fminf(fmaxf(b0, 0.0f), 1.0f);
    b1 = fminf(fmaxf(b1, 0.0f), 1.0f);
    const float sum = b0 + b1;
    if (sum > 1.0f) { b0 = __fdiv_rn(b0, sum); b1 = __fdiv_rn(b1, sum); }
    const float b2 = fmaxf(1.0f - b0 - b1, 0.0f);
    float sh = 1.0f;
    if (p.shadow) {
        const F3 dc = {b0 * r.dc[0].x + b1 * r.dc[1].x + b2 * r.dc[2].x,
                       b0 * r.dc[0].y + b1 * r.dc[1].y + b2 * r.dc[2].y,
                       b0 * r.dc[0].z + b1 * r.dc[1].z + b2 * r.dc[2].z};
        sh = __fdiv_rn((float)pcf25(p.shadow, p.shadow_size, dc, 0.002f), 25.0f);   // vox.fs:46
    }
    const unsigned long long cr = to_unorm8(r.alb[0] * sh), cg = to_unorm8(r.alb[1] * sh),
                             cb = to_unorm8(r.alb[2] * sh);                          // vox.fs:88
    unsigned long long* a = p.acc + 2 * (size_t)vct_morton3((uint32_t)i, (uint32_t)j, (uint32_t)k);
    atomicAdd(a, cr | (cg << 32));
    atomicAdd(a + 1, cb | (1ull << 32));
}

#define VCT_SMALL_BBOX 27

__global__ void __launch_bounds__(256)
k_voxelize_small(const VctVoxParams p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.ntri) return;
    TriSetup r;
    setup_tri(p, t, r);
    if (!r.valid) return;
    const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return;
    const long long cnt = (long long)nx * ny * nz;
    if (cnt > VCT_SMALL_BBOX) {
        const int slot = atomicAdd(p.big_count, 1);
        p.big_list[slot] = t;
        return;
    }
    for (int k = r.lo[2]; k <= r.hi[2]; ++k)
        for (int j = r.lo[1]; j <= r.hi[1]; ++j)
            for (int i = r.lo[0]; i <= r.hi[0]; ++i) fragment(p, r, i, j, k);
}

__global__ void __launch_bounds__(256)
k_voxelize_big(const VctVoxParams p) {
    const int nbig = *p.big_count;
    for (int b = blockIdx.x; b < nbig; b += gridDim.x) {
        const int t = p.big_list[b];
        TriSetup r;
        setup_tri(p, t, r);
        const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
        const long long cnt = (long long)nx * ny * nz;
        for (long long v = threadIdx.x; v < cnt; v += blockDim.x) {
            const int i = r.lo[0] + (int)(v % nx);
            const int j = r.lo[1] + (int)((v / nx) % ny);
            const int k = r.lo[2] + (int)(v / ((long long)nx * ny));
            fragment(p, r, i, j, k);
        }
    }
}

// accumulators -> RGBA8 level 0 (Morton), rounded mean, a = 255 where any fragment landed
__global__ void __launch_bounds__(256)
k_resolve(const unsigned long long* __restrict__ acc, uint32_t* __restrict__ level0, size_t nvox) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox;
         v += (size_t)gridDim.x * blockDim.x) {
        const ulonglong2 a = reinterpret_cast<const ulonglong2*>(acc)[v];
        const uint32_t c = (uint32_t)(a.y >> 32);
        uint32_t out = 0;
        if (c) {
            const uint32_t h = c >> 1;
            const uint32_t r = ((uint32_t)a.x + h) / c, g = ((uint32_t)(a.x >> 32) + h) / c,
                           b = ((uint32_t)a.y + h) / c;
            out = r | (g << 8) | (b << 16) | 0xff000000u;
        }
        level0[v] = out;
    }
}

}  // namespace

hipError_t vct_launch_voxelize(const VctVoxParams& p, hipStream_t s) {
    if (p.ntri <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_voxelize_small, dim3((p.ntri + 255) / 256), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_voxelize_big, dim3(256 * 8), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t vct_launch_resolve(const unsigned long long* acc, uint32_t* level0, int V, int mode,
                              hipStream_t s) {
    (void)mode;
    const size_t nvox = (size_t)V * V * V;
    size_t blocks = (nvox + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, s, acc, level0, nvox);
    return hipGetLastError();
}
