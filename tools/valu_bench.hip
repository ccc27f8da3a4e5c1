// valu_bench.hip -- issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the trace
// kernel is made of, measured on the GPU box.  Diagnostic tool, not product code.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench && tools/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x

#define KERNEL(name, ASMSTR)                                                                  \
    __global__ void __launch_bounds__(256) k_##name(float* out, int iters) {                  \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,           \
              a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                          \
        float b = out[0], c = out[1];                                                         \
        for (int i = 0; i < iters; ++i) {                                                     \
            REP16(asm volatile(ASMSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4),     \
                               "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)               \
        }                                                                                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[2] = a0;                 \
    }

// 8 independent instructions per asm block
#define OP8(fmt) fmt("%0") fmt("%1") fmt("%2") fmt("%3") fmt("%4") fmt("%5") fmt("%6") fmt("%7")
#define FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define MUL(r) "v_mul_f32 " r ", " r ", %8\n"
#define ADD(r) "v_add_f32 " r ", " r ", %8\n"
#define CVTUB(r) "v_cvt_f32_ubyte1 " r ", " r "\n"
#define CVTI(r) "v_cvt_i32_f32 " r ", " r "\n"
#define FLOOR(r) "v_floor_f32 " r ", " r "\n"
#define FRACT(r) "v_fract_f32 " r ", " r "\n"
#define RCP(r) "v_rcp_f32 " r ", " r "\n"
#define AND(r) "v_and_b32 " r ", " r ", %8\n"
#define LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n"
#define ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define OR3(r) "v_or3_b32 " r ", " r ", %8, %9\n"
#define BFE(r) "v_bfe_u32 " r ", " r ", 3, 8\n"
#define PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define MULLO(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define CVTPK(r) "v_cvt_pkrtz_f16_f32 " r ", " r ", %8\n"
#define LDEXP(r) "v_ldexp_f32 " r ", " r ", %8\n"
#define DIVFIX(r) "v_div_fixup_f32 " r ", " r ", %8, %9\n"
#define MED3(r) "v_med3_f32 " r ", " r ", %8, %9\n"
#define MAX(r) "v_max_f32 " r ", " r ", %8\n"
#define CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define DOT4(r) "v_dot4_u32_u8 " r ", " r ", %8, %9\n"

KERNEL(fma, OP8(FMA))
KERNEL(mul, OP8(MUL))
KERNEL(add, OP8(ADD))
KERNEL(cvt_ubyte, OP8(CVTUB))
KERNEL(cvt_i32, OP8(CVTI))
KERNEL(floor, OP8(FLOOR))
KERNEL(fract, OP8(FRACT))
KERNEL(rcp, OP8(RCP))
KERNEL(and, OP8(AND))
KERNEL(lshl_or, OP8(LSHLOR))
KERNEL(and_or, OP8(ANDOR))
KERNEL(add3, OP8(ADD3))
KERNEL(or3, OP8(OR3))
KERNEL(bfe, OP8(BFE))
KERNEL(perm, OP8(PERM))
KERNEL(mad24, OP8(MAD24))
KERNEL(mul_lo, OP8(MULLO))
KERNEL(cvt_pkrtz, OP8(CVTPK))
KERNEL(ldexp, OP8(LDEXP))
KERNEL(div_fixup, OP8(DIVFIX))
KERNEL(med3, OP8(MED3))
KERNEL(max, OP8(MAX))
KERNEL(cndmask, OP8(CNDMASK))
KERNEL(dot4, OP8(DOT4))

// packed fp32: 4 independent register pairs
__global__ void __launch_bounds__(256) k_pk_fma(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    f2 b = {out[0], out[1]}, c = {out[1], out[0]};
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\n"
                           "v_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                           "v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\n"
                           "v_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    }
    f2 s = a0 + a1 + a2 + a3;
    if (s.x + s.y == 12345.678f) out[2] = s.x;
}
__global__ void __launch_bounds__(256) k_pk_mul(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f;
    f2 b = {out[0], out[1]};
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\n"
                           "v_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n"
                           "v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\n"
                           "v_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    }
    f2 s = a0 + a1 + a2 + a3;
    if (s.x + s.y == 12345.678f) out[2] = s.x;
}

// scalar ALU: 8 independent SGPR chains per block (is the scalar unit a co-bottleneck of a kernel with half as
// many SALU as VALU instructions?)
#define SKERNEL(name, ASMSTR)                                                                  \
    __global__ void __launch_bounds__(256) k_##name(float* out, int iters) {                   \
        int a0 = iters, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,                    \
            a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                             \
        const int b = iters | 3;                                                               \
        for (int i = 0; i < iters; ++i) {                                                      \
            REP16(asm volatile(ASMSTR : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4),      \
                               "+s"(a5), "+s"(a6), "+s"(a7) : "s"(b) : "scc");)                \
        }                                                                                      \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123456789) out[2] = 1.0f;                 \
    }
#define SOP8(fmt) fmt("%0") fmt("%1") fmt("%2") fmt("%3") fmt("%4") fmt("%5") fmt("%6") fmt("%7")
#define SADD(r) "s_add_i32 " r ", " r ", %8\n"
#define SAND(r) "s_and_b32 " r ", " r ", %8\n"
#define SMUL(r) "s_mul_i32 " r ", " r ", %8\n"
#define SLSHL(r) "s_lshl_b32 " r ", " r ", 1\n"
SKERNEL(s_add, SOP8(SADD))
SKERNEL(s_and, SOP8(SAND))
SKERNEL(s_mul, SOP8(SMUL))
SKERNEL(s_lshl, SOP8(SLSHL))
// two VALU + one SALU interleaved: does the scalar instruction cost VALU issue time?
__global__ void __launch_bounds__(256) k_mix21(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = out[0], c = out[1];
    int s0 = iters, s1 = s0 + 1;
    const int sb = iters | 3;
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_fma_f32 %0, %0, %6, %7\nv_fma_f32 %1, %1, %6, %7\ns_add_i32 %4, %4, %8\n"
                           "v_fma_f32 %2, %2, %6, %7\nv_fma_f32 %3, %3, %6, %7\ns_and_b32 %5, %5, %8\n"
                           "v_fma_f32 %0, %0, %6, %7\nv_fma_f32 %1, %1, %6, %7\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(b), "v"(c), "s"(sb) : "scc");)
    }
    if (a0 + a1 + a2 + a3 == 12345.678f || s0 + s1 == 123456789) out[2] = a0;
}

template <typename K>
static void run(const char* name, K kern, float* d, int waves_per_simd) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * 8 * waves_per_simd;
    // assume 2.4 GHz; report ns per instruction per SIMD and cycles at 2.4 GHz
    const double ns = ms * 1e6 / instr_per_simd;
    printf("%-12s waves/SIMD=%d  %.3f ms  %.3f ns/instr/SIMD  = %.2f cyc @2.4GHz\n", name,
           waves_per_simd, ms, ns, ns * 2.4);
}

int main() {
    float* d;
    hipMalloc(&d, 1024);
    hipMemset(d, 0, 1024);
#define RUN(n) run(#n, k_##n, d, 1); run(#n, k_##n, d, 4); run(#n, k_##n, d, 8);
    RUN(fma) RUN(pk_fma) RUN(mul) RUN(pk_mul) RUN(add) RUN(cvt_ubyte) RUN(cvt_i32) RUN(floor) RUN(fract)
    RUN(rcp) RUN(and) RUN(lshl_or) RUN(and_or) RUN(add3) RUN(or3) RUN(bfe) RUN(perm) RUN(mad24)
    RUN(mul_lo) RUN(cvt_pkrtz) RUN(ldexp) RUN(div_fixup) RUN(med3) RUN(max) RUN(cndmask) RUN(dot4)
    RUN(s_add) RUN(s_and) RUN(s_mul) RUN(s_lshl)
    // mix21: 6 VALU + 2 SALU per block of 8: printed per instruction of the 8
    RUN(mix21)
    return 0;
}
