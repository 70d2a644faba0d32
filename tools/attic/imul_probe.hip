// Probe: cycles per wave-instruction of the 32-bit integer multiply flavours on gfx950 (one wave per SIMD, s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef uint32_t u32;
#define N 256
template <int MODE>
__global__ __launch_bounds__(64) void k(u64 *out, u32 seed) {
    u32 a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
    u64 acc0 = 0, acc1 = 1, acc2 = 2, acc3 = 3; u32 lo0 = 0, lo1 = 1, lo2 = 2, lo3 = 3, hi0 = 0, hi1 = 0, hi2 = 0, hi3 = 0;
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) {  // 4 independent v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 1) {  // 4 dependent v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0"
                         : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 2) {  // 4 x v_mul_lo_u32 independent
            asm volatile("v_mul_lo_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %5\n v_mul_lo_u32 %2, %4, %5\n v_mul_lo_u32 %3, %4, %5"
                         : "=v"(lo0), "=v"(lo1), "=v"(lo2), "=v"(lo3) : "v"(a), "v"(b));
        } else if (MODE == 3) {  // 4 x v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %4, %5\n v_mul_hi_u32 %1, %4, %5\n v_mul_hi_u32 %2, %4, %5\n v_mul_hi_u32 %3, %4, %5"
                         : "=v"(hi0), "=v"(hi1), "=v"(hi2), "=v"(hi3) : "v"(a), "v"(b));
        } else if (MODE == 4) {  // 4 x v_mad_u32_u24
            asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3"
                         : "+v"(lo0), "+v"(lo1), "+v"(lo2), "+v"(lo3) : "v"(a), "v"(b));
        } else if (MODE == 5) {  // 4 x v_add_co_u32 + v_addc (reference cheap op)
            asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %5, %2\n v_add_u32 %3, %5, %3"
                         : "+v"(lo0), "+v"(lo1), "+v"(lo2), "+v"(lo3) : "v"(a), "v"(b));
        } else if (MODE == 6) {  // v_mad_u64_u32 + s_nop 1 + v_addc (the chain kernel's mac) x4 independent
            asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n s_nop 1\n v_addc_co_u32_e64 %4, s[20:21], 0, %4, s[20:21]\n"
                         "v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n s_nop 1\n v_addc_co_u32_e64 %5, s[22:23], 0, %5, s[22:23]\n"
                         "v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n s_nop 1\n v_addc_co_u32_e64 %6, s[24:25], 0, %6, s[24:25]\n"
                         "v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n s_nop 1\n v_addc_co_u32_e64 %7, s[26:27], 0, %7, s[26:27]"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(hi0), "+v"(hi1), "+v"(hi2), "+v"(hi3) : "v"(a), "v"(b)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 7) {  // f64 fma x4 independent (double-precision multiplier as an integer engine?)
            asm volatile("v_fma_f64 %0, %4, %4, %0\n v_fma_f64 %1, %4, %4, %1\n v_fma_f64 %2, %4, %4, %2\n v_fma_f64 %3, %4, %4, %3"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(acc0));
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc0 + acc1 + acc2 + acc3 + lo0 + lo1 + lo2 + lo3 + hi0 + hi1 + hi2 + hi3; }
}
template <int MODE> void run(const char *name, int blocks) {
    u64 *d; hipMalloc(&d, blocks * 16); u64 h[2];
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 12345u); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 12345u); hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s blocks %5d: %.1f cycles per instruction (4 per iteration)\n", name, blocks, (double)h[0] / (N * 4));
    hipFree(d);
}
int main() {
    for (int blocks : {1, 4096}) {
        run<0>("v_mad_u64_u32 x4 independent", blocks);
        run<1>("v_mad_u64_u32 x4 dependent", blocks);
        run<2>("v_mul_lo_u32 x4", blocks);
        run<3>("v_mul_hi_u32 x4", blocks);
        run<4>("v_mad_u32_u24 x4", blocks);
        run<5>("v_add_u32 x4", blocks);
        run<6>("mad_u64_u32 + s_nop 1 + addc, x4 independent", blocks);
        run<7>("v_fma_f64 x4 independent", blocks);
    }
    return 0;
}
