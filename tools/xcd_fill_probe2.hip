// Second XCD-mapping probe: the distance D between the 8 concurrent XCD store streams, scanned.  The buffer is walked in
// groups of 8*D bytes; inside a group XCD x (blockIdx % 8) fills the D bytes at offset x*D, 256 KB per workgroup.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/xcd_fill_probe2 tools/xcd_fill_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr size_t PIECE = 256 << 10;

__global__ void __launch_bounds__(256) fill(v4u *p, unsigned ch, unsigned v) {   // ch = D in pieces (0: identity)
    unsigned b = blockIdx.x;
    const unsigned n = gridDim.x, n8 = n >> 3;
    if (ch) { const unsigned x = b & 7, q = b >> 3, full = n8 / ch * ch; if (b < (n8 << 3) && q < full) b = (q / ch) * (8 * ch) + x * ch + q % ch; }
    v4u *dst = p + (size_t)b * (PIECE / 16);
    for (unsigned i = threadIdx.x; i < PIECE / 16; i += 256) {
        const v4u x = {v, i, b, v};
        __builtin_nontemporal_store(x, dst + i);
    }
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 20.0;
    const size_t bytes = (size_t)(gb * (1ull << 30)) / (8 * PIECE) * (8 * PIECE);
    v4u *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(bytes / PIECE), n8 = grid / 8;
    std::vector<unsigned> chs = {0};
    for (unsigned c = 1; c <= n8; c *= 2) { chs.push_back(c); if (c >= 4 && c / 2 * 3 <= n8) chs.push_back(c / 2 * 3); if (c >= 8 && c / 4 * 5 <= n8) chs.push_back(c / 4 * 5); }
    chs.push_back(n8);
    std::printf("buffer %.2f GB at %p, %u pieces of 256 KB\n", bytes / 1073741824.0, (void *)buf, grid);
    for (unsigned ch : chs) {
        float sum = 0; const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, st, buf, ch, (unsigned)r);
            CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r) sum += ms;
        }
        std::printf("D = %9.2f MB (%6u pieces)%s: %.2f TB/s\n", ch * (double)PIECE / 1048576.0, ch, ch == 0 ? " identity" : (ch == n8 ? " eighths" : ""), bytes / (sum / reps * 1e-3) / 1e12);
    }
    return 0;
}
