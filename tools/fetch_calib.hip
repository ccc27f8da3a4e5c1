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

int main() {
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
