// Does it matter WHICH XCD writes WHERE?  A fill of S bytes in 256 KB pieces, one piece per workgroup (non-temporal 16-byte
// stores), with two workgroup -> piece mappings: identity (the 8 XCDs, blockIdx % 8, sweep the buffer together) and
// contiguous eighths (every XCD fills its own eighth).  Prints TB/s for several buffer sizes.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/xcd_fill_probe tools/xcd_fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr size_t PIECE = 256 << 10;

template <int MODE>   // 0 identity, 1 eighths, 2 chunks of 256 pieces per XCD
__global__ void __launch_bounds__(256) fill(v4u *p, unsigned v) {
    extern __shared__ unsigned dyn[];
    unsigned b = blockIdx.x;
    const unsigned n = gridDim.x, n8 = n >> 3;
    if (MODE == 1 && b < (n8 << 3)) b = (b & 7) * n8 + (b >> 3);
    if (MODE == 2) { const unsigned x = b & 7, q = b >> 3, ch = 256, full = n8 / ch * ch; if (b < (n8 << 3) && q < full) b = (q / ch) * (8 * ch) + x * ch + q % ch; }
    v4u *dst = p + (size_t)b * (PIECE / 16);
    for (unsigned i = threadIdx.x; i < PIECE / 16; i += 256) {
        const v4u x = {v, i, b, v};
        __builtin_nontemporal_store(x, dst + i);
    }
}

int main(int argc, char **argv) {
    const double sizes_gb[] = {1.25, 2.5, 5, 10, 20, 40, 80};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (double gb : sizes_gb) {
        const size_t bytes = (size_t)(gb * (1ull << 30)) / PIECE * PIECE;
        v4u *buf; if (hipMalloc(&buf, bytes) != hipSuccess) { std::printf("%.2f GB: alloc failed\n", gb); break; }
        CK(hipMemset(buf, 0, bytes));
        const unsigned grid = (unsigned)(bytes / PIECE);
        for (int lds : {0, 45000}) {
            double tb[3];
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e9f, sum = 0; const int reps = 6;
                for (int r = 0; r < reps + 1; ++r) {
                    CK(hipEventRecord(a, st));
                    if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(grid), dim3(256), lds, st, buf, (unsigned)r);
                    else if (mode == 1) hipLaunchKernelGGL(fill<1>, dim3(grid), dim3(256), lds, st, buf, (unsigned)r);
                    else hipLaunchKernelGGL(fill<2>, dim3(grid), dim3(256), lds, st, buf, (unsigned)r);
                    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (r) { sum += ms; if (ms < best) best = ms; }
                }
                tb[mode] = bytes / (sum / reps * 1e-3) / 1e12;
            }
            std::printf("%6.2f GB, %5d B LDS per workgroup: identity %.2f TB/s   eighths %.2f TB/s   chunks of 256 %.2f TB/s\n", gb, lds, tb[0], tb[1], tb[2]);
        }
        CK(hipFree(buf));
    }
    return 0;
}
