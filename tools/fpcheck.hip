// Dev tool: checks that the GPU's fp32 primitives used by the trace kernel round like the host's.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
struct F3 { float x, y, z; };
__host__ __device__ inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline F3 norm3(F3 a) { float l = __builtin_sqrtf(dot3(a, a)); return {a.x / l, a.y / l, a.z / l}; }
__global__ void k(const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    F3 a = {in[3*i], in[3*i+1], in[3*i+2]};
    float d = dot3(a, a);
    float s = __builtin_sqrtf(d);
    F3 nn = norm3(a);
    out[8*i+0] = d; out[8*i+1] = s; out[8*i+2] = a.x / s; out[8*i+3] = nn.x; out[8*i+4] = nn.y; out[8*i+5] = nn.z;
    out[8*i+6] = 4.0f - a.y; out[8*i+7] = a.x * a.y - a.z * a.x;
}
int main() {
    const int n = 1 << 20; std::vector<float> in(3*n), out(8*n);
    srand(1); for (auto& v : in) v = ((float)rand() / RAND_MAX - 0.5f) * 140.0f;
    float *di, *dout; hipMalloc(&di, in.size()*4); hipMalloc(&dout, out.size()*4);
    hipMemcpy(di, in.data(), in.size()*4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, di, dout, n);
    hipMemcpy(out.data(), dout, out.size()*4, hipMemcpyDeviceToHost);
    long bad[8] = {0};
    for (int i = 0; i < n; ++i) {
        F3 a = {in[3*i], in[3*i+1], in[3*i+2]};
        volatile float d = dot3(a, a); volatile float s = sqrtf(d); F3 nn = norm3(a);
        float ref[8] = {d, s, a.x / s, nn.x, nn.y, nn.z, 4.0f - a.y, a.x * a.y - a.z * a.x};
        for (int j = 0; j < 8; ++j) if (memcmp(&ref[j], &out[8*i+j], 4)) bad[j]++;
    }
    const char* names[8] = {"dot", "sqrt", "div", "nx", "ny", "nz", "sub", "mulsub"};
    for (int j = 0; j < 8; ++j) printf("%s mismatches %ld / %d\n", names[j], bad[j], n);
    return 0;
}
