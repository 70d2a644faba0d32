// Second part of the kernel-boundary study: the pipeline's queue pattern with stand-in kernels.
//   stream A ("record"): a train of 1.25 GB non-temporal fills, each with dispatch-stamped start/stop events
//   stream B ("chain"):  a train of ~130 us spin kernels (1024 workgroups)
// and the cross-queue waits the pipeline uses: A's kernel k waits for B's kernel k (hipStreamWaitEvent), B's kernel k+2
// waits for A's kernel k (the lazy join).  Prints A's stop(k) -> start(k+1) for every variant.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/boundary_probe2 tools/boundary_probe2.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fill(v4u *p, size_t n16, unsigned v, const unsigned long long *gate = nullptr, unsigned long long want = 0, unsigned *late = nullptr) {
    extern __shared__ unsigned dyn[];
    if (gate) {   // device-side dependency: wait until the other stream has published `want`
        if (threadIdx.x == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool waited = false;
            while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                waited = true;
                __builtin_amdgcn_s_sleep(16);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) break;   // 1 s
            }
            if (waited && blockIdx.x == 0) atomicAdd(late, 1u);
        }
        __syncthreads();
    }
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

int main() {
    const size_t cap = (size_t)1280 << 20;
    v4u *buf[2]; unsigned *sink;
    CK(hipMalloc(&buf[0], cap)); CK(hipMalloc(&buf[1], cap)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(buf[0], 0, cap)); CK(hipMemset(buf[1], 0, cap));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int N = 14;
    std::vector<hipEvent_t> a(N), b(N), ca(N), cb(N);
    for (int i = 0; i < N; ++i) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); CK(hipEventCreate(&ca[i])); CK(hipEventCreate(&cb[i])); }
    unsigned long long *gate; unsigned *late;
    CK(hipMalloc(&gate, 256)); CK(hipMemset(gate, 0, 256)); late = reinterpret_cast<unsigned *>(gate + 8);
    unsigned long long seq = 0;
    struct V { const char *name; int prioA; bool chain; bool a_waits_b; bool b_waits_a; bool host_paced; unsigned spin_us; bool gated = false; };
    const V vs[] = {
        {"A alone, default priority", 0, false, false, false, false, 130},
        {"A alone, lowest priority", 1, false, false, false, false, 130},
        {"A low + B spinning, no waits", 1, true, false, false, false, 130},
        {"A low + B, A waits B (record waits chain)", 1, true, true, false, false, 130},
        {"A low + B, B waits A (lazy join only)", 1, true, false, true, false, 130},
        {"A low + B, both waits (the pipeline)", 1, true, true, true, false, 130},
        {"A default + B, both waits", 0, true, true, true, false, 130},
        {"A high + B, both waits", 2, true, true, true, false, 130},
        {"A low + B, both waits, B spins 10 us", 1, true, true, true, false, 10},
        {"A low + B, A waits B, B paced by the HOST (hipEventSynchronize)", 1, true, true, false, true, 130},
        {"A low + B, B waits A, A GATED on a value B writes (hipStreamWriteValue64)", 1, true, false, true, false, 130, true},
        {"A low + B, B waits A, A GATED, B spins 400 us (A must really wait)", 1, true, false, true, false, 400, true},
        {"A low + B, both waits (again)", 1, true, true, true, false, 130},
    };
    for (const V &v : vs) {
        hipStream_t A, B;
        CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, v.prioA == 1 ? lo : (v.prioA == 2 ? hi : 0)));
        CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < N; ++i) {
                if (v.chain) {
                    if (v.b_waits_a && i >= 2) CK(hipStreamWaitEvent(B, b[i - 2], 0));
                    if (v.host_paced && i >= 2) CK(hipEventSynchronize(b[i - 2]));
                    hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, B, ca[i], cb[i], 0, v.spin_us * 100, sink);
                    if (v.a_waits_b) CK(hipStreamWaitEvent(A, cb[i], 0));
                    if (v.gated) CK(hipStreamWriteValue64(B, gate, ++seq, 0));
                }
                hipExtLaunchKernelGGL(fill, dim3(4864), dim3(256), 45000, A, a[i], b[i], 0, buf[i & 1], cap / 16, (unsigned)i,
                                      (const unsigned long long *)(v.gated ? gate : nullptr), seq, late);
            }
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        }
        float dur = 0, gap = 0, gmin = 1e9f, gmax = 0, train;
        for (int i = 3; i < N; ++i) { float d; CK(hipEventElapsedTime(&d, a[i], b[i])); dur += d; }
        for (int i = 3; i + 1 < N; ++i) { float g; CK(hipEventElapsedTime(&g, b[i], a[i + 1])); gap += g; if (g < gmin) gmin = g; if (g > gmax) gmax = g; }
        CK(hipEventElapsedTime(&train, a[3], b[N - 1]));
        unsigned nl = 0; CK(hipMemcpy(&nl, late, 4, hipMemcpyDeviceToHost)); CK(hipMemset(late, 0, 4));
        if (v.gated) std::printf("  (kernels whose first workgroup had to wait: %u of %d)\n", nl, 2 * N);
        std::printf("%-66s kernel %.1f us  gap %.1f us (min %.1f max %.1f)  period %.1f us\n", v.name, 1e3 * dur / (N - 3), 1e3 * gap / (N - 4), 1e3 * gmin, 1e3 * gmax, 1e3 * train / (N - 3));
        CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B));
    }
    return 0;
}
