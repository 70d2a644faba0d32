// What does the boundary between two kernels on one stream cost on this part, and what does it depend on?
// Launches pairs of store kernels back to back with dispatch-stamped events (hipExtLaunchKernelGGL start/stop, as the
// library does) and prints stop(k) -> start(k+1).  Variables: bytes written per kernel, store policy, grid size
// (workgroups), dynamic LDS request, whether the stop event is attached at all.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/boundary_probe tools/boundary_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int POLICY>   // 0 plain, 1 non-temporal
__global__ void __launch_bounds__(256) fill(uint4 *p_, size_t n16, unsigned v) {
    v4u *p = reinterpret_cast<v4u *>(p_);
    extern __shared__ unsigned dyn[];
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const v4u x = {v, (unsigned)i, v, v};
        if (POLICY == 1) __builtin_nontemporal_store(x, p + i); else p[i] = x;
    }
}

struct Cfg { const char *name; size_t bytes; int policy; unsigned grid; unsigned lds; bool events; };

int main() {
    const size_t cap = (size_t)1280 << 20;
    uint4 *buf[2];
    CK(hipMalloc(&buf[0], cap)); CK(hipMalloc(&buf[1], cap));
    CK(hipMemset(buf[0], 0, cap)); CK(hipMemset(buf[1], 0, cap));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 12;
    std::vector<hipEvent_t> a(N), b(N);
    for (int i = 0; i < N; ++i) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); }
    const Cfg cfgs[] = {
        {"empty (0 B), 768 wg", 0, 1, 768, 0, true},
        {"empty (0 B), 4864 wg, 45 KB LDS", 0, 1, 4864, 45000, true},
        {"64 MB nt, 768 wg", (size_t)64 << 20, 1, 768, 0, true},
        {"1.25 GB nt, 768 wg", cap, 1, 768, 0, true},
        {"1.25 GB nt, 4864 wg", cap, 1, 4864, 0, true},
        {"1.25 GB nt, 4864 wg, 45 KB LDS", cap, 1, 4864, 45000, true},
        {"1.25 GB nt, 38912 wg, 45 KB LDS", cap, 1, 38912, 45000, true},
        {"1.25 GB plain, 768 wg", cap, 0, 768, 0, true},
        {"1.25 GB plain, 4864 wg, 45 KB LDS", cap, 0, 4864, 45000, true},
    };
    for (const Cfg &c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < N; ++i) {
                uint4 *p = buf[i & 1];
                if (c.policy) hipExtLaunchKernelGGL((fill<1>), dim3(c.grid), dim3(256), c.lds, st, a[i], b[i], 0, p, c.bytes / 16, (unsigned)i);
                else hipExtLaunchKernelGGL((fill<0>), dim3(c.grid), dim3(256), c.lds, st, a[i], b[i], 0, p, c.bytes / 16, (unsigned)i);
            }
            CK(hipStreamSynchronize(st));
        }
        float dur = 0, gap = 0, gmin = 1e9f, gmax = 0;
        for (int i = 2; i < N; ++i) { float d; CK(hipEventElapsedTime(&d, a[i], b[i])); dur += d; }
        for (int i = 2; i + 1 < N; ++i) { float g; CK(hipEventElapsedTime(&g, b[i], a[i + 1])); gap += g; if (g < gmin) gmin = g; if (g > gmax) gmax = g; }
        // whole-train time: start of kernel 2 to stop of the last one
        float train; CK(hipEventElapsedTime(&train, a[2], b[N - 1]));
        std::printf("%-40s kernel %.1f us  gap %.1f us (min %.1f max %.1f)  period %.1f us  %.2f TB/s per kernel, %.2f TB/s sustained\n", c.name,
                    1e3 * dur / (N - 2), 1e3 * gap / (N - 3), 1e3 * gmin, 1e3 * gmax, 1e3 * train / (N - 2),
                    c.bytes / (1e9 * dur / (N - 2)), c.bytes * (double)(N - 2) / (1e9 * train));
    }
    // the same train without per-kernel events: only the first start and the last stop are stamped
    for (int lds : {0, 45000}) {
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < N; ++i) {
                hipEvent_t ea = i == 2 ? a[0] : nullptr, eb = i == N - 1 ? b[0] : nullptr;
                hipExtLaunchKernelGGL((fill<1>), dim3(4864), dim3(256), lds, st, ea, eb, 0, buf[i & 1], cap / 16, (unsigned)i);
            }
            CK(hipStreamSynchronize(st));
        }
        float train; CK(hipEventElapsedTime(&train, a[0], b[0]));
        std::printf("1.25 GB nt, 4864 wg, %d B LDS, no per-kernel events: period %.1f us, %.2f TB/s sustained\n", lds, 1e3 * train / (N - 2), cap * (double)(N - 2) / (1e9 * train));
    }
    // plain launches (hipLaunchKernelGGL), bracketed by two hipEventRecord markers
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a[0], st));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL((fill<1>), dim3(4864), dim3(256), 45000, st, buf[i & 1], cap / 16, (unsigned)i);
        CK(hipEventRecord(b[0], st));
        CK(hipStreamSynchronize(st));
    }
    float train; CK(hipEventElapsedTime(&train, a[0], b[0]));
    std::printf("1.25 GB nt, 4864 wg, 45 KB LDS, hipLaunchKernelGGL x %d between two markers: period %.1f us, %.2f TB/s sustained\n", N, 1e3 * train / N, cap * (double)N / (1e9 * train));
    return 0;
}
