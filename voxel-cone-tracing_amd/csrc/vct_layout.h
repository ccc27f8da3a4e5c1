// vct_layout.h -- HBM layout of the radiance mip chain.
//
// The reference keeps the volume as a driver-tiled GL_RGBA8 image3D (VCT.h:110-126).  Here every
// level is stored in 3-D Morton (Z-curve) order, 4 B per texel (r | g<<8 | b<<16 | a<<24):
//   * any aligned 8x8x8 brick is one contiguous 2 KiB run (16 x 128 B lines), bricks themselves
//     follow each other in Morton order;
//   * the 2x2x2 children of a parent texel are 32 contiguous bytes, so a mip level is a pure
//     streaming reduction of the level below it;
//   * a 128 B line holds a 4x4x2 texel block, a 64 B half-line 4x2x2: a trilinear footprint
//     touches ~2.3 lines on average instead of 4 in a row-major volume.
// Levels are concatenated from level 0 (finest); level k starts at texel offset
// sum_{j<k} (V>>j)^3 -- the same offsets as the linear chain of the C ABI.
#ifndef VCT_LAYOUT_H_
#define VCT_LAYOUT_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define VCT_HD __host__ __device__ __forceinline__
#else
#define VCT_HD inline
#endif

#define VCT_MAX_LEVELS 11   // V <= 1024

// spread the low 10 bits of x so that bit i lands on bit 3*i
VCT_HD uint32_t vct_spread3(uint32_t x) {
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

VCT_HD uint32_t vct_compact3(uint32_t x) {
    x &= 0x09249249u;
    x = (x | (x >> 2)) & 0x030c30c3u;
    x = (x | (x >> 4)) & 0x0300f00fu;
    x = (x | (x >> 8)) & 0x030000ffu;
    x = (x | (x >> 16)) & 0x3ffu;
    return x;
}

VCT_HD uint32_t vct_morton3(uint32_t x, uint32_t y, uint32_t z) {
    return vct_spread3(x) | (vct_spread3(y) << 1) | (vct_spread3(z) << 2);
}

VCT_HD int vct_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

VCT_HD uint64_t vct_level_offset(int V, int level) {
    uint64_t off = 0;
    for (int l = 0; l < level; ++l) {
        const uint64_t n = (uint64_t)(V >> l);
        off += n * n * n;
    }
    return off;
}

#endif
