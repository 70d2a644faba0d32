// Issue cost (cycles per wave-instruction, one wave alone on its SIMD) of the VALU instructions a field-element conversion can be
// built from on gfx950: the 32 x 32 + 64 integer multiply-add against the FP64 FMA and its helpers.  s_memtime around 16 x 32
// independent instructions.   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/_bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
#define PROBE(name, body)                                                                                          \
    __global__ void name(unsigned long long *out, double *sink) {                                                  \
        double d0 = threadIdx.x * 1.5 + 1, d1 = d0 + 2, d2 = d0 + 3, d3 = d0 + 4, d4 = 1.25, d5 = 3.5;              \
        unsigned v0 = threadIdx.x * 77u + 5u, v1 = v0 + 3u, v2 = v0 ^ 0x1234u, v3 = v0 * 3u;                         \
        unsigned long long q0 = v0, q1 = v1, q2 = v2, q3 = v3;                                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
        for (int i = 0; i < 16; ++i) { asm volatile(REP32(body) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(d4), "v"(d5) : "vcc"); } \
        asm volatile("s_nop 0" ::: "memory");                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                                      \
        sink[threadIdx.x] = d0 + d1 + d2 + d3 + v0 + v1 + v2 + v3 + (double)(q0 + q1 + q2 + q3);                     \
    }
// four independent chains per body so that latency does not serialise
PROBE(k_fma64, "v_fma_f64 %0, %12, %13, %0\n v_fma_f64 %1, %12, %13, %1\n v_fma_f64 %2, %12, %13, %2\n v_fma_f64 %3, %12, %13, %3\n")
PROBE(k_add64, "v_add_f64 %0, %12, %0\n v_add_f64 %1, %12, %1\n v_add_f64 %2, %12, %2\n v_add_f64 %3, %12, %3\n")
PROBE(k_mad64, "v_mad_u64_u32 %8, vcc, %4, %5, %8\n v_mad_u64_u32 %9, vcc, %4, %5, %9\n v_mad_u64_u32 %10, vcc, %4, %5, %10\n v_mad_u64_u32 %11, vcc, %4, %5, %11\n")
PROBE(k_mullo, "v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %4\n")
PROBE(k_mul24, "v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %4\n")
PROBE(k_mad24, "v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %4\n v_mad_u32_u24 %7, %7, %4, %5\n")
PROBE(k_add32, "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %4\n")
PROBE(k_lshladd64, "v_lshl_add_u64 %8, %9, 0, %8\n v_lshl_add_u64 %9, %10, 0, %9\n v_lshl_add_u64 %10, %11, 0, %10\n v_lshl_add_u64 %11, %8, 0, %11\n")
PROBE(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7\n")
PROBE(k_cvt_u32_f64, "v_cvt_u32_f64 %4, %0\n v_cvt_u32_f64 %5, %1\n v_cvt_u32_f64 %6, %2\n v_cvt_u32_f64 %7, %3\n")
PROBE(k_floor64, "v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3\n")
PROBE(k_mul64f, "v_mul_f64 %0, %12, %0\n v_mul_f64 %1, %12, %1\n v_mul_f64 %2, %12, %2\n v_mul_f64 %3, %12, %3\n")
PROBE(k_fma32, "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %4\n v_fma_f32 %7, %7, %4, %5\n")
int main() {
    unsigned long long *d; double *s;
    hipMalloc(&d, 8); hipMalloc(&s, 64 * 8);
#define RUNP(k) { for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s); unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); std::printf("%-16s %6.2f memtime ticks per instruction (%llu / 2048)\n", #k, (double)h / 2048.0, h); }
    RUNP(k_add32) RUNP(k_fma32) RUNP(k_mad24) RUNP(k_mul24) RUNP(k_mullo) RUNP(k_mad64) RUNP(k_lshladd64) RUNP(k_fma64) RUNP(k_add64) RUNP(k_mul64f) RUNP(k_floor64) RUNP(k_cvt_f64_u32) RUNP(k_cvt_u32_f64)
    return 0;
}
