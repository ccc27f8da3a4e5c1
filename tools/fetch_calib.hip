// fetch_calib.hip -- known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 for the access widths the trace kernel uses (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads
// half of a 16 B/lane stream; other widths are uncalibrated).  Diagnostic tool, not product code.
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void read_dword(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read_dwordx4(const uint4* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// plane-strided tile read like the G-buffer prologue: 23 planes x 64 lanes x 4 B per wave
__global__ void read_tiles(const float* __restrict__ p, size_t ntiles, uint32_t* out) {
    float acc = 0.0f;
    const int lane = threadIdx.x & 63;
    for (size_t t = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < ntiles;
         t += (size_t)gridDim.x * (blockDim.x >> 6))
        for (int k = 0; k < 23; ++k) acc += p[t * 23 * 64 + k * 64 + lane];
    if (acc == 1234.5f) out[0] = 1;
}
__global__ void write_dwordx2(uint2* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint2((uint32_t)i, 7u);
}

// Scattered 32-byte records (round 5: the access pattern of the trace kernel's footprint records): every lane
// reads one 32-byte-aligned record (two dwordx4) at a pseudo-random place of a buffer far larger than the caches.  Useful
// bytes = 32 per access; what the memory system fetches per access (a 64-byte or a 128-byte request) is what the counters
// and the kernel's own time -- accesses x granule / time against the HBM peak -- have to tell.
__global__ void gather32(const uint4* __restrict__ p, size_t nrec, size_t naccess, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < naccess; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t h = (i + 1) * 0x9e3779b97f4a7c15ull;
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        const size_t r = (size_t)(h % nrec);
        const uint4 a = p[2 * r], b = p[2 * r + 1];
        acc += a.x + a.w + b.x + b.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'g') {          // tools/fetch_calib gather: the scattered-record pattern alone
        const size_t gb = 8ull << 30, nrec = gb / 32, naccess = 1ull << 27;     // 8 GiB buffer, 4 GiB of useful bytes
        uint32_t *gbuf, *gout;
        if (hipMalloc(&gbuf, gb) != hipSuccess || hipMalloc(&gout, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(gbuf, 1, gb);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(gather32, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)gbuf, nrec, naccess, gout);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("gather32: %zu accesses of 32 B in %.3f ms = %.1f G accesses/s; useful %.2f TB/s; x64 B %.2f TB/s; x128 B %.2f TB/s\n",
                   naccess, ms, naccess / ms / 1e6, naccess * 32.0 / ms / 1e9, naccess * 64.0 / ms / 1e9, naccess * 128.0 / ms / 1e9);
        }
        return 0;
    }
    const size_t bytes = 1ull << 30;     // 1 GiB: 4x the Infinity Cache
    uint32_t *buf, *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_dword, dim3(4096), dim3(256), 0, 0, buf, bytes / 4, out);
        hipLaunchKernelGGL(read_dwordx4, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(read_tiles, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / (23 * 64 * 4), out);
        hipLaunchKernelGGL(write_dwordx2, dim3(4096), dim3(256), 0, 0, (uint2*)buf, bytes / 8);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes (read_tiles: %zu)\n", bytes, (bytes / (23 * 64 * 4)) * 23 * 64 * 4);
    return 0;
}
