// vct_volume.hip -- layout conversion and mip build over the Morton brick chain.
//
// vct_launch_build_mips replaces glGenerateMipmap(GL_TEXTURE_3D) (VCT.h:126,248): 2x2x2 box
// mean per channel, rounded to the nearest unorm8 and requantised level by level
// (OpenGL 4.3 core mipmap generation).  In Morton order the 8 children of a
// texel are 32 contiguous bytes, so one wave turns a 2 KiB brick (512 texels) into 64 + 8 + 1
// texels of the next three levels: three levels per launch, pure streaming, HBM-bound.
#include "vct_internal.h"

namespace {

__global__ void k_linear_to_morton(const uint32_t* __restrict__ lin, uint32_t* __restrict__ mor,
                                   int N, int shift) {
    const size_t total = (size_t)N * N * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(i & (size_t)(N - 1));
        const uint32_t y = (uint32_t)((i >> shift) & (size_t)(N - 1));
        const uint32_t z = (uint32_t)(i >> (2 * shift));
        mor[vct_morton3(x, y, z)] = lin[i];
    }
}

__global__ void k_morton_to_linear(const uint32_t* __restrict__ mor, uint32_t* __restrict__ lin,
                                   int N, int shift) {
    const size_t total = (size_t)N * N * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(i & (size_t)(N - 1));
        const uint32_t y = (uint32_t)((i >> shift) & (size_t)(N - 1));
        const uint32_t z = (uint32_t)(i >> (2 * shift));
        lin[i] = mor[vct_morton3(x, y, z)];
    }
}

__global__ void __launch_bounds__(256)
k_build_cells(const uint32_t* __restrict__ level, uint4* __restrict__ cells, uint32_t m, uint32_t count) {
    const uint32_t MX = vct_spread3(m), MY = MX << 1, MZ = MX << 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)i;
        const uint32_t x0 = c & MX, y0 = c & MY, z0 = c & MZ;
        const uint32_t x1 = ((x0 | ~MX) + 1u) & MX, y1 = ((y0 | ~MY) + 2u) & MY, z1 = ((z0 | ~MZ) + 4u) & MZ;   // dilated + 1, wraps
        uint4 lo, hi;
        lo.x = level[x0 | y0 | z0]; lo.y = level[x1 | y0 | z0]; lo.z = level[x0 | y1 | z0]; lo.w = level[x1 | y1 | z0];
        hi.x = level[x0 | y0 | z1]; hi.y = level[x1 | y0 | z1]; hi.z = level[x0 | y1 | z1]; hi.w = level[x1 | y1 | z1];
        cells[2 * i] = lo;
        cells[2 * i + 1] = hi;
    }
}

// mean of 8 RGBA8 texels per channel, (sum + 4) >> 3, on two 16-bit-lane SWAR accumulators
__device__ __forceinline__ uint32_t box8(uint32_t rb, uint32_t ga) {
    rb = ((rb + 0x00040004u) >> 3) & 0x00ff00ffu;
    ga = ((ga + 0x00040004u) >> 3) & 0x00ff00ffu;
    return rb | (ga << 8);
}

// src: level L (count texels), dst1..3: levels L+1..L+3 (nout of them exist).
// Sparse form (level 0 written by the voxelizer): wave w reduces exactly the 8^3 brick w, so bricks
// that hold nothing now (`now`) and held nothing when the mips were last built (`seen`) are
// skipped -- their three ancestors are already zero.
__device__ __forceinline__ void mip3_thread(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst1,
                                            uint32_t* __restrict__ dst2, uint32_t* __restrict__ dst3, uint32_t n1,
                                            int nout, uint32_t t) {
    uint32_t q = 0;
    if (t < n1) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src) + 2 * (size_t)t;
        const uint4 a = s4[0], b = s4[1];
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t rb = 0, ga = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            rb += v[i] & 0x00ff00ffu;
            ga += (v[i] >> 8) & 0x00ff00ffu;
        }
        q = box8(rb, ga);
        dst1[t] = q;
    }
    if (nout < 2) return;
    uint32_t rb = q & 0x00ff00ffu, ga = (q >> 8) & 0x00ff00ffu;
    rb += __shfl_xor(rb, 1); ga += __shfl_xor(ga, 1);
    rb += __shfl_xor(rb, 2); ga += __shfl_xor(ga, 2);
    rb += __shfl_xor(rb, 4); ga += __shfl_xor(ga, 4);
    const uint32_t q2 = box8(rb, ga);
    const uint32_t n2 = n1 >> 3;
    if ((t & 7u) == 0u && (t >> 3) < n2) dst2[t >> 3] = q2;
    if (nout < 3) return;
    rb = q2 & 0x00ff00ffu; ga = (q2 >> 8) & 0x00ff00ffu;
    rb += __shfl_xor(rb, 8); ga += __shfl_xor(ga, 8);
    rb += __shfl_xor(rb, 16); ga += __shfl_xor(ga, 16);
    rb += __shfl_xor(rb, 32); ga += __shfl_xor(ga, 32);
    const uint32_t n3 = n2 >> 3;
    if ((t & 63u) == 0u && (t >> 6) < n3) dst3[t >> 6] = box8(rb, ga);
}

__global__ void __launch_bounds__(256)
k_mip3(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst1, uint32_t* __restrict__ dst2,
       uint32_t* __restrict__ dst3, uint32_t count, int nout, const uint32_t* __restrict__ now,
       uint32_t* __restrict__ seen) {
    const uint32_t n1 = count >> 3;
    const uint32_t nthreads_needed = (n1 + 63u) & ~63u;   // whole waves so shuffles are defined
    if (now) {
        // sparse level 0: a wave serves a brick (64 parents).  The brick flags are read 64 at a time, one brick per lane
        // (spread over the grid: lane * nchunks + chunk), and the wave then takes the flagged ones in turn -- one brick per
        // wave iteration meant 64 dependent flag loads per wave for the 2 M bricks of a 1024^3 grid (round 3).
        // (only for the largest grids -- above 2^19 bricks; smaller ones keep one brick per wave iteration, which
        // spreads their few touched bricks over all waves: 256^3 0.011 vs 0.020 ms, 1024^3 0.14 vs 0.053 ms)
        const int lane = threadIdx.x & 63;
        const uint32_t nbricks = nthreads_needed >> 6;
        const bool scan64 = nbricks > 524288u;
        const uint32_t group = scan64 ? 64u : 1u, nchunks = (nbricks + group - 1u) / group;
        const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
        for (uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nchunks; c += waves) {
            const uint32_t mine = scan64 ? (uint32_t)lane * nchunks + c : c;
            const bool in = mine < nbricks && (scan64 || lane == 0);
            const uint32_t cur = in ? now[mine] : 0u, old = in ? seen[mine] : 0u;
            const bool live = in && (cur | old) != 0u;
            if (live) seen[mine] = cur;
            for (unsigned long long todo = __builtin_amdgcn_ballot_w64(live); todo != 0ull; todo &= todo - 1ull) {
                const uint32_t b = scan64 ? (uint32_t)(__ffsll((long long)todo) - 1) * nchunks + c : c;
                mip3_thread(src, dst1, dst2, dst3, n1, nout, b * 64u + (uint32_t)lane);
            }
        }
        return;
    }
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nthreads_needed; t += gridDim.x * blockDim.x)
        mip3_thread(src, dst1, dst2, dst3, n1, nout, t);
}

// Directional (anisotropic) mip level: one thread per (parent texel, direction).  Definition:
// oracle/vct_oracle.h "anisotropic (directional) mip volumes".  In Morton order the 8 children of
// parent m are src[8m .. 8m+7], child t at (x,y,z) = (t&1, (t>>1)&1, t>>2).
__global__ void __launch_bounds__(256)
k_mip_aniso(const uint32_t* __restrict__ level0, uint32_t* __restrict__ aniso, uint32_t stride,
            uint32_t src_off, uint32_t dst_off, uint32_t nparents, int from_level0) {
    const uint32_t total = nparents * 6u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t d = i / nparents, m = i - d * nparents;
        const int axis = (int)(d >> 1);
        const bool toward_plus = (d & 1u) == 0u;
        const uint32_t* src = from_level0 ? level0 + 8 * (size_t)m
                                          : aniso + (size_t)d * stride + src_off + 8 * (size_t)m;
        uint32_t c[8];
        const uint4 lo = reinterpret_cast<const uint4*>(src)[0], hi = reinterpret_cast<const uint4*>(src)[1];
        c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w; c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
        const int oa = axis == 0 ? 1 : 0, ob = axis == 2 ? 1 : 2;       // the two other axes, lower first
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int base = ((pr & 1) << oa) | ((pr >> 1) << ob);
            const uint32_t t0 = c[base], t1 = c[base | (1 << axis)];    // lower / higher along the axis
            const uint32_t F = toward_plus ? t0 : t1, B = toward_plus ? t1 : t0;
            const float oma = 1.0f - vct_unorm8_to_float(F >> 24);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float comp = fmaf(oma, vct_unorm8_to_float((B >> (8 * ch)) & 0xffu),
                                        vct_unorm8_to_float((F >> (8 * ch)) & 0xffu));
                acc[ch] = pr == 0 ? comp : acc[ch] + comp;
            }
        }
        uint32_t out = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) out |= vct_float_to_unorm8(acc[ch] * 0.25f) << (8 * ch);
        aniso[(size_t)d * stride + dst_off + m] = out;
    }
}

// planes [23][h*w] -> tiled [tile][23][64]; pixels outside the frame are zero (albedo.a = 0).
__global__ void k_tile_gbuffer(const float* __restrict__ planes, float* __restrict__ tiled, int w,
                               int h, int tiles_x, int tiles_y) {
    const size_t total = (size_t)tiles_x * tiles_y * VCT_GB_NPLANES * VCT_TILE_PIX;
    const size_t npix = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t r = i >> 6;
        const int plane = (int)(r % VCT_GB_NPLANES);
        const size_t tile = r / VCT_GB_NPLANES;
        const int ty = (int)(tile / tiles_x), tx = (int)(tile - (size_t)ty * tiles_x);
        const int x = tx * VCT_TILE + (lane & 7), y = ty * VCT_TILE + (lane >> 3);
        float v = 0.0f;
        if (x < w && y < h) v = planes[(size_t)plane * npix + (size_t)y * w + x];
        tiled[i] = v;
    }
}

inline int grid_for(size_t n, int threads) {
    size_t b = (n + threads - 1) / threads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

hipError_t vct_launch_linear_to_morton(const uint32_t* lin, uint32_t* mor, int N, hipStream_t s) {
    const size_t n = (size_t)N * N * N;
    hipLaunchKernelGGL(k_linear_to_morton, dim3(grid_for(n, 256)), dim3(256), 0, s, lin, mor, N,
                       vct_ilog2(N));
    return hipGetLastError();
}

hipError_t vct_launch_morton_to_linear(const uint32_t* mor, uint32_t* lin, int N, hipStream_t s) {
    const size_t n = (size_t)N * N * N;
    hipLaunchKernelGGL(k_morton_to_linear, dim3(grid_for(n, 256)), dim3(256), 0, s, mor, lin, N,
                       vct_ilog2(N));
    return hipGetLastError();
}

hipError_t vct_launch_build_mips(uint32_t* chain, int V, const uint32_t* bricks_now, uint32_t* bricks_seen,
                                 hipStream_t s) {
    const int nlev = vct_ilog2(V) + 1;
    for (int L = 0; L + 1 < nlev; L += 3) {
        const int nout = (nlev - 1 - L) < 3 ? (nlev - 1 - L) : 3;
        const uint32_t n = (uint32_t)(V >> L);
        const uint32_t count = n * n * n;
        uint32_t* src = chain + vct_level_offset(V, L);
        uint32_t* d1 = chain + vct_level_offset(V, L + 1);
        uint32_t* d2 = nout >= 2 ? chain + vct_level_offset(V, L + 2) : nullptr;
        uint32_t* d3 = nout >= 3 ? chain + vct_level_offset(V, L + 3) : nullptr;
        const size_t threads_needed = ((size_t)(count >> 3) + 63) & ~(size_t)63;
        const bool sparse = L == 0 && bricks_now && bricks_seen && V >= 8;
        hipLaunchKernelGGL(k_mip3, dim3(grid_for(threads_needed, 256)), dim3(256), 0, s, src, d1,
                           d2, d3, count, nout, sparse ? bricks_now : nullptr, sparse ? bricks_seen : nullptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Footprint records ("cells") of the levels >= 1: record c of a level holds the 8 texels of the trilinear footprint
// anchored at texel c -- (i + dx) & m, (j + dy) & m, (k + dz) & m, dx fastest -- as 32 contiguous bytes, Morton-indexed
// like the texels (GL_REPEAT folded in).  A per-lane level sample is then ONE 32-byte fetch instead of eight 4-byte
// ones from two to four cache lines.  8 x the bytes of those levels = 1.14 x level 0 (vct_trace.hip sample_level).
hipError_t vct_launch_build_cells(const uint32_t* chain, uint4* cells, int V, hipStream_t s) {
    const int nlev = vct_ilog2(V) + 1;
    const size_t off1 = vct_level_offset(V, 1);
    for (int l = 1; l < nlev; ++l) {
        const uint32_t n = (uint32_t)(V >> l);
        const uint32_t count = n * n * n;
        const size_t off = vct_level_offset(V, l);
        hipLaunchKernelGGL(k_build_cells, dim3(grid_for(count, 256)), dim3(256), 0, s, chain + off,
                           cells + 2 * (off - off1), n - 1u, count);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t vct_launch_build_mips_aniso(const uint32_t* level0, uint32_t* aniso, int V, hipStream_t s) {
    const int nlev = vct_ilog2(V) + 1;
    const uint32_t V3 = (uint32_t)vct_level_offset(V, 1);
    const uint32_t stride = (uint32_t)vct_level_offset(V, nlev) - V3;
    for (int l = 1; l < nlev; ++l) {
        const uint32_t n = (uint32_t)(V >> l);
        const uint32_t nparents = n * n * n;
        const uint32_t dst_off = (uint32_t)vct_level_offset(V, l) - V3;
        const uint32_t src_off = l == 1 ? 0u : (uint32_t)vct_level_offset(V, l - 1) - V3;
        hipLaunchKernelGGL(k_mip_aniso, dim3(grid_for((size_t)nparents * 6, 256)), dim3(256), 0, s, level0,
                           aniso, stride, src_off, dst_off, nparents, l == 1 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t vct_launch_tile_gbuffer(const float* planes_linear, float* tiled, int w, int h,
                                   hipStream_t s) {
    const int tx = (w + VCT_TILE - 1) / VCT_TILE, ty = (h + VCT_TILE - 1) / VCT_TILE;
    const size_t n = (size_t)tx * ty * VCT_GB_NPLANES * VCT_TILE_PIX;
    hipLaunchKernelGGL(k_tile_gbuffer, dim3(grid_for(n, 256)), dim3(256), 0, s, planes_linear, tiled,
                       w, h, tx, ty);
    return hipGetLastError();
}
