// Pair relations between physical chunks: n chunks of `chunk_mb` (hipMemCreate, each mapped on its own), concurrent pair fill of
// (chunk i, chunk j) -- workgroup 2k writes 64 KB piece k of i, workgroup 2k + 1 piece k of j -- for i in a few references, every j.
// Prints the time matrix rows and the three-level histogram.  usage: chunk_relation_probe [chunk_mb = 256] [n = 96] [refs = 4]
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/chunk_relation_probe tools/chunk_relation_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned long long u64;
constexpr u64 CHUNK = 65536, BLK = 4096;
__device__ __forceinline__ void st16(void *p, u64 a, u64 b) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    v2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
}
__global__ __launch_bounds__(256) void fill(char *A, char *B, u64 pieces) {   // B == nullptr: every workgroup writes A; the grid walks `pieces` pieces per stream
    const u64 t = threadIdx.x;
    for (u64 w = blockIdx.x; w < (B ? 2 : 1) * pieces; w += gridDim.x) {
        const u64 c = B ? w >> 1 : w;
        char *p = ((B && (w & 1)) ? B : A) + c * CHUNK;
        for (u64 k = 0; k < CHUNK / BLK; ++k) st16(p + k * BLK + t * 16, c, k);
    }
}
static float time_fill(char *A, char *B, u64 bytes_each, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 pieces = bytes_each / CHUNK;
    const unsigned grid = (unsigned)std::min<u64>((B ? 2 : 1) * pieces, 256 * 8 * 4);
    hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, A, B, pieces);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, A, B, pieces);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}
int main(int argc, char **argv) {
    const u64 chunk = (u64)(argc > 1 ? std::atoi(argv[1]) : 256) << 20;
    const u64 n = argc > 2 ? std::atoi(argv[2]) : 96;
    const u64 refs = argc > 3 ? std::atoi(argv[3]) : 4;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    std::vector<char *> va(n);
    for (u64 i = 0; i < n; ++i) {
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemAddressReserve((void **)&va[i], chunk, 0, nullptr, 0));
        CK(hipMemMap(va[i], chunk, 0, h[i], 0));
        CK(hipMemSetAccess(va[i], chunk, &acc, 1));
    }
    std::printf("%llu chunks of %llu MB; pair fill = 2 x %llu MB, TB/s\n", n, chunk >> 20, chunk >> 20);
    std::printf("alone (TB/s):");
    for (u64 j = 0; j < n; ++j) std::printf(" %.2f", chunk / time_fill(va[j], nullptr, chunk, 4) / 1e9);
    std::printf("\nhalves of one chunk, concurrently (TB/s):");
    for (u64 j = 0; j < std::min<u64>(n, 32); ++j) std::printf(" %.2f", chunk / time_fill(va[j], va[j] + chunk / 2, chunk / 2, 4) / 1e9);
    std::printf("\n");
    std::vector<std::vector<float>> M(refs, std::vector<float>(n, 0.f));
    for (u64 i = 0; i < refs; ++i) {
        std::printf("ref %llu:", i);
        for (u64 j = 0; j < n; ++j) {
            if (j == i) { std::printf("  -- "); continue; }
            M[i][j] = (float)(2.0 * chunk / time_fill(va[i], va[j], chunk, 4) / 1e9);
            std::printf(" %.2f", M[i][j]);
        }
        std::printf("\n");
    }
    // is the relation a class function?  level of (i, j) from the levels of (0, i) and (0, j)
    auto lvl = [](float tbs) { return tbs > 6.3f ? 2 : (tbs > 4.3f ? 1 : 0); };   // 2 fast, 1 normal, 0 slow
    int hist[3] = {0, 0, 0};
    for (u64 j = 1; j < n; ++j) hist[lvl(M[0][j])]++;
    std::printf("levels vs ref 0: slow %d normal %d fast %d\n", hist[0], hist[1], hist[2]);
    for (u64 i = 1; i < refs; ++i) {
        int agree[3][3] = {{0}};
        for (u64 j = 1; j < n; ++j) if (j != i) agree[lvl(M[0][j])][lvl(M[i][j])]++;
        std::printf("ref %llu is %s vs ref 0; table [level vs ref 0][level vs ref %llu]:", i, lvl(M[0][i]) == 2 ? "FAST" : lvl(M[0][i]) == 1 ? "normal" : "SLOW", i);
        for (int a = 0; a < 3; ++a) std::printf("  [%d %d %d]", agree[a][0], agree[a][1], agree[a][2]);
        std::printf("\n");
    }
    return 0;
}
