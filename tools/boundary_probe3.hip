// Third part of the kernel-boundary study: ONE stream, overlap through hipExtAnyOrderLaunch.
//   queue order per step:  [spin  ("chain" k+1), barrier bit set]  [fill ("record" k), launched any-order = barrier bit clear]
// The spin kernel waits for everything queued before it (the previous fill included); the fill kernel starts as soon as the
// spin kernel's workgroups are dispatched and runs next to it.  Prints fill stop(k) -> fill start(k+1) and whether the two
// kernels really overlap.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/boundary_probe3 tools/boundary_probe3.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fill(v4u *p, size_t n16, unsigned v) {
    extern __shared__ unsigned dyn[];
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const v4u x = {v, (unsigned)i, v, v};
        __builtin_nontemporal_store(x, p + i);
    }
}
__global__ void __launch_bounds__(256) spin(unsigned ticks, unsigned *sink) {   // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) acc += 1;
    if (acc == 0xffffffffu) *sink = acc;
}
__global__ void tiny(unsigned *sink) { if (threadIdx.x == 9999) *sink = 1; }

int main() {
    const size_t cap = (size_t)1280 << 20;
    v4u *buf[2]; unsigned *sink;
    CK(hipMalloc(&buf[0], cap)); CK(hipMalloc(&buf[1], cap)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(buf[0], 0, cap)); CK(hipMemset(buf[1], 0, cap));
    const int N = 14;
    std::vector<hipEvent_t> a(N), b(N), ca(N), cb(N);
    for (int i = 0; i < N; ++i) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); CK(hipEventCreate(&ca[i])); CK(hipEventCreate(&cb[i])); }
    struct V { const char *name; bool any_fill; bool with_tiny; unsigned spin_us; bool events_on_spin; };
    const V vs[] = {
        {"spin B, fill B (fully serial reference)", false, false, 130, true},
        {"spin B, fill any-order", true, false, 130, true},
        {"spin B, tiny any-order, fill any-order", true, true, 130, true},
        {"spin B (no events on spin), fill any-order", true, false, 130, false},
        {"spin 10 us B, fill any-order", true, false, 10, true},
    };
    hipStream_t S; CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    for (const V &v : vs) {
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < N; ++i) {
                hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, S, v.events_on_spin ? ca[i] : nullptr, v.events_on_spin ? cb[i] : nullptr, 0, v.spin_us * 100, sink);
                if (v.with_tiny) hipExtLaunchKernelGGL(tiny, dim3(1024), dim3(64), 0, S, nullptr, nullptr, hipExtAnyOrderLaunch, sink);
                hipExtLaunchKernelGGL(fill, dim3(4864), dim3(256), 45000, S, a[i], b[i], v.any_fill ? hipExtAnyOrderLaunch : 0, buf[i & 1], cap / 16, (unsigned)i);
            }
            CK(hipStreamSynchronize(S));
        }
        float dur = 0, gap = 0, gmin = 1e9f, gmax = 0, train, lead = 0, sdur = 0;
        for (int i = 3; i < N; ++i) { float d; CK(hipEventElapsedTime(&d, a[i], b[i])); dur += d; }
        for (int i = 3; i + 1 < N; ++i) { float g; CK(hipEventElapsedTime(&g, b[i], a[i + 1])); gap += g; if (g < gmin) gmin = g; if (g > gmax) gmax = g; }
        if (v.events_on_spin) for (int i = 3; i < N; ++i) { float d; CK(hipEventElapsedTime(&d, ca[i], a[i])); lead += d; CK(hipEventElapsedTime(&d, ca[i], cb[i])); sdur += d; }
        CK(hipEventElapsedTime(&train, a[3], b[N - 1]));
        std::printf("%-48s fill %.1f us  gap %.1f us (min %.1f max %.1f)  period %.1f us  spin start -> fill start %.1f us, spin %.1f us\n", v.name,
                    1e3 * dur / (N - 3), 1e3 * gap / (N - 4), 1e3 * gmin, 1e3 * gmax, 1e3 * train / (N - 3), 1e3 * lead / (N - 3), 1e3 * sdur / (N - 3));
    }
    return 0;
}
