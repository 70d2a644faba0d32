// Cost of one field-element conversion (x -> x * 2^256 mod p) for a wave alone on its SIMD, constants in kernel arguments (SGPRs):
// the radix-2^30 integer product (mont30), the radix-2^32 one (mont_short).   s_memtime around 32 dependent conversions.
//   hipcc -O3 --offload-arch=gfx950 -Ihalo2_rsa_amd/csrc -Iinclude tools/mont_probe.hip -o tools/_bin/mont_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "h2r_field.hpp"
#include "h2r_layout.hpp"
using namespace h2r;
template <int V>
__global__ void probe(MontK mk, unsigned long long *out, unsigned *sink) {
    unsigned x[5] = {threadIdx.x * 2654435761u, threadIdx.x + 77u, threadIdx.x ^ 0xdeadbeefu, threadIdx.x * 31u, threadIdx.x & 31u};
    unsigned t[8];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 32; ++i) {
        if constexpr (V == 0) mont_bits<150, 5>(x, mk, t);
        else if constexpr (V == 1) mont_short<5>(x, mk.bk[5], mk.p, mk.n0inv, t);
        else if constexpr (V == 2) { unsigned y[3] = {x[0], x[1], x[2] & 63u}; mont_bits<90, 3>(y, mk, t); }
        else if constexpr (V == 3) { unsigned y[1] = {x[0] & 255u}; mont_bits<8, 1>(y, mk, t); }
#ifdef H2R_HAVE_MONT24F
        else if constexpr (V == 4) mont24f_bits<144, 5>(x, mk, t);
        else if constexpr (V == 5) { unsigned y[3] = {x[0], x[1], x[2] & 63u}; mont24f_bits<72, 3>(y, mk, t); }
        else if constexpr (V == 6) { unsigned y[1] = {x[0] & 255u}; mont24f_bits<24, 1>(y, mk, t); }
#endif
        x[0] ^= t[0]; x[1] ^= t[3]; x[2] ^= t[5]; x[3] ^= t[7]; x[4] = (x[4] ^ t[1]) & 31u;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = x[0] + x[1] + x[2] + x[3] + x[4];
}
int main() {
    unsigned long long *d; unsigned *s;
    hipMalloc(&d, 8); hipMalloc(&s, 64 * 4);
    u64 p[4]; field_modulus(H2R_FIELD_BN254_FR, p);
    MontK mk; montk_init(p, &mk);
#define RUNP(V, name) { for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<V>, dim3(1), dim3(64), 0, 0, mk, d, s); unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); std::printf("%-44s %7.1f ticks per conversion\n", name, (double)h / 32.0); }
    RUNP(0, "mont30, 5 digits (133-bit accumulator)") RUNP(1, "mont_short, 5 words") RUNP(2, "mont30, 3 digits (limb, carry)") RUNP(3, "mont30, 1 digit (sub-limb)")
#ifdef H2R_HAVE_MONT24F
    RUNP(4, "mont24f (FP64), 6 digits (accumulator)") RUNP(5, "mont24f, 3 digits") RUNP(6, "mont24f, 1 digit")
#endif
    return 0;
}
