// ON THE GPU BOX: does the buffer path's UNORM8 -> fp32 conversion (tbuffer_load_format_xyzw, BUF_DATA_FORMAT_8_8_8_8 /
// BUF_NUM_FORMAT_UNORM) return exactly (float)c / 255.0f for every byte?  If it does, a texel's four channels arrive decoded
// and the trace's exact decode (cvt + mul + fma per channel) is not needed.
//   hipcc --offload-arch=gfx950 -O2 -o tools/unorm_probe tools/unorm_probe.hip && tools/unorm_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// the same load as a compiler intrinsic (format from the descriptor): what the trace kernel uses, so that the compiler
// tracks the load's completion itself (composable_kernel declares its buffer loads the same way)
__device__ v4f vct_buffer_load_format(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");

// structured form: stride 4, the texel INDEX in vindex -- a 4 GiB level (2^30 texels) is addressed in full
__device__ v4f vct_struct_buffer_load_format(v4i rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.format.v4f32");

__global__ void k_probe_struct(const uint32_t* texels, float* out, int n, uint32_t big_index, float* big_out, uint32_t records = 0x40000000u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = (uint64_t)texels;
    v4i rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(base >> 32) | (4u << 16)));      // stride 4 bytes
    rsrc.z = __builtin_amdgcn_readfirstlane((int)records);                                    // records (texels)
    rsrc.w = 0x50fac;
    const v4f r = vct_struct_buffer_load_format(rsrc, i < n ? i : 0, 0, 0, 0);
    if (i < n) { out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w; }
    if (big_out && i == 0) {
        const v4f q = vct_struct_buffer_load_format(rsrc, (int)big_index, 0, 0, 0);
        big_out[0] = q.x; big_out[1] = q.y; big_out[2] = q.z; big_out[3] = q.w;
    }
}

__global__ void k_probe_intrinsic(const uint32_t* texels, float* out, int n, uint32_t big_off, float* big_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = (uint64_t)texels;
    v4i rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    rsrc.z = (int)0xffffffffu;
    rsrc.w = 0x50fac;        // DST_SEL x,y,z,w = R,G,B,A | NUM_FORMAT_UNORM << 12 | DATA_FORMAT_8_8_8_8 << 15
    const v4f r = vct_buffer_load_format(rsrc, (int)((uint32_t)(i < n ? i : 0) * 4u), 0, 0);
    if (i < n) { out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w; }
    if (big_out && i == 0) {        // the last texel of a 4 GiB level: byte offset 2^32 - 4 against num_records 2^32 - 1
        const v4f q = vct_buffer_load_format(rsrc, (int)big_off, 0, 0);
        big_out[0] = q.x; big_out[1] = q.y; big_out[2] = q.z; big_out[3] = q.w;
    }
}

__global__ void k_probe(const uint32_t* texels, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = (uint64_t)texels;
    v4i rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));          // stride 0, no swizzle
    rsrc.z = __builtin_amdgcn_readfirstlane((int)0xffffffffu);                       // num_records (bytes)
    rsrc.w = __builtin_amdgcn_readfirstlane((int)0x00027000);                        // type 0 buffer; format comes from the instruction
    const uint32_t off = (uint32_t)(i < n ? i : 0) * 4u;
    v4f r;
    asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\ts_waitcnt vmcnt(0)"
                 : "=v"(r) : "v"(off), "s"(rsrc) : "memory");
    if (i < n) { out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w; }
}

int main() {
    const int n = 256;
    std::vector<uint32_t> h(n);
    for (int c = 0; c < n; ++c) h[c] = (uint32_t)c | ((uint32_t)(255 - c) << 8) | ((uint32_t)((c * 7) & 255) << 16) | ((uint32_t)((c * 13 + 5) & 255) << 24);
    uint32_t* d; float* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(4), dim3(64), 0, 0, d, o, n);
    std::vector<float> r(n * 4);
    if (hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    int bad = 0, seen[256] = {0};
    for (int i = 0; i < n; ++i)
        for (int ch = 0; ch < 4; ++ch) {
            const uint32_t c = (h[i] >> (8 * ch)) & 255u;
            const float want = (float)c / 255.0f, got = r[4 * i + ch];
            seen[c] = 1;
            if (memcmp(&want, &got, 4) != 0) {
                if (bad < 12) { uint32_t a, b; memcpy(&a, &want, 4); memcpy(&b, &got, 4); printf("byte %3u channel %d: want %.9g (%08x) got %.9g (%08x)\n", c, ch, want, a, got, b); }
                ++bad;
            }
        }
    // the intrinsic form
    hipMemset(o, 0xff, n * 16);
    hipLaunchKernelGGL(k_probe_intrinsic, dim3(4), dim3(64), 0, 0, d, o, n, 0u, (float*)nullptr);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int i = 0; i < n; ++i)
        for (int ch = 0; ch < 4; ++ch) {
            const float want = (float)((h[i] >> (8 * ch)) & 255u) / 255.0f;
            if (memcmp(&want, &r[4 * i + ch], 4) != 0) ++bad2;
        }
    printf("the same through llvm.amdgcn.raw.buffer.load.format.v4f32 (format in the descriptor): %d differ\n", bad2);
    bad += bad2;
    // a 4 GiB level (1024^3 level 0): its last texel sits at byte offset 2^32 - 4
    {
        uint8_t* big = nullptr; float* bo = nullptr;
        const size_t G4 = (size_t)1 << 32;
        if (hipMalloc(&big, G4) == hipSuccess && hipMalloc(&bo, 16) == hipSuccess) {
            const uint32_t last = 0x80ff4001u;      // bytes 0x01, 0x40, 0xff, 0x80
            hipMemcpy(big + G4 - 4, &last, 4, hipMemcpyHostToDevice);
            hipMemset(bo, 0, 16);
            hipLaunchKernelGGL(k_probe_intrinsic, dim3(1), dim3(64), 0, 0, (const uint32_t*)big, o, 1, 0xfffffffcu, bo);
            float q[4];
            hipMemcpy(q, bo, 16, hipMemcpyDeviceToHost);
            const float w[4] = {1.0f / 255.0f, 64.0f / 255.0f, 1.0f, 128.0f / 255.0f};
            const bool ok = memcmp(q, w, 16) == 0;
            printf("texel at byte offset 2^32 - 4 of a 4 GiB buffer (num_records 0xffffffff): %s (%g %g %g %g)\n", ok ? "read correctly" : "NOT read (range check)", q[0], q[1], q[2], q[3]);
            // (informational: the raw form's range check is why the kernel uses the structured form)
            hipMemset(bo, 0, 16);
            hipLaunchKernelGGL(k_probe_struct, dim3(1), dim3(64), 0, 0, (const uint32_t*)big, o, 1, 0x3fffffffu, bo);
            hipMemcpy(q, bo, 16, hipMemcpyDeviceToHost);
            const bool ok2 = memcmp(q, w, 16) == 0;
            printf("the same texel through the structured form (stride 4, index 2^30 - 1, 2^30 records): %s (%g %g %g %g)\n", ok2 ? "read correctly" : "NOT read", q[0], q[1], q[2], q[3]);
            if (!ok2) ++bad;
        } else printf("4 GiB allocation failed: range check not probed\n");
        if (big) hipFree(big);
        if (bo) hipFree(bo);
    }
    // beyond 2^30 records: the packed buffer of a scene's material textures is indexed by 32-bit texel offsets, so the
    // structured form has to reach any of 2^32 - 1 texels (16 GiB) with num_records 0xffffffff
    {
        uint8_t* big = nullptr; float* bo = nullptr;
        const size_t G16 = (size_t)1 << 34;
        if (hipMalloc(&big, G16) == hipSuccess && hipMalloc(&bo, 16) == hipSuccess) {
            const uint32_t idx[4] = {0x40000005u, 0x80000007u, 0xc0000009u, 0xfffffffeu};
            for (int k = 0; k < 4; ++k) {
                const uint32_t tex = 0x80ff4001u + (uint32_t)k;      // bytes 0x01 + k, 0x40, 0xff, 0x80
                hipMemcpy(big + (size_t)idx[k] * 4, &tex, 4, hipMemcpyHostToDevice);
                hipMemset(bo, 0, 16);
                hipLaunchKernelGGL(k_probe_struct, dim3(1), dim3(64), 0, 0, (const uint32_t*)big, o, 1, idx[k], bo, 0xffffffffu);
                float q[4];
                hipMemcpy(q, bo, 16, hipMemcpyDeviceToHost);
                const float w[4] = {(float)(1 + k) / 255.0f, 64.0f / 255.0f, 1.0f, 128.0f / 255.0f};
                const bool ok = memcmp(q, w, 16) == 0;
                printf("structured form, 2^32 - 1 records, texel index 0x%08x of a 16 GiB buffer: %s (%g %g %g %g)\n", idx[k], ok ? "read correctly" : "NOT read", q[0], q[1], q[2], q[3]);
                if (!ok) ++bad;
            }
        } else printf("16 GiB allocation failed: indices beyond 2^30 not probed\n");
        if (big) hipFree(big);
        if (bo) hipFree(bo);
    }
    hipMemset(o, 0xff, n * 16);
    hipLaunchKernelGGL(k_probe_struct, dim3(4), dim3(64), 0, 0, d, o, n, 0u, (float*)nullptr);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad3 = 0;
    for (int i = 0; i < n; ++i)
        for (int ch = 0; ch < 4; ++ch) {
            const float want = (float)((h[i] >> (8 * ch)) & 255u) / 255.0f;
            if (memcmp(&want, &r[4 * i + ch], 4) != 0) ++bad3;
        }
    printf("the same through llvm.amdgcn.struct.buffer.load.format.v4f32 (stride 4, idxen): %d differ\n", bad3);
    bad += bad3;
    int cov = 0; for (int c = 0; c < 256; ++c) cov += seen[c];
    printf("unorm8 -> fp32 through tbuffer_load_format_xyzw: %d of %d channel values differ from (float)c / 255.0f (bytes covered: %d of 256)\n", bad, n * 4, cov);
    return bad ? 1 : 0;
}
