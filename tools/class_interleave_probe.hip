// Placement classes, the constructive test.  Physical chunks (hipMemCreate, `chunk_mb` each) are classified ONE BY ONE against reference
// chunks by a concurrent pair fill (different class: ~6.9 TB/s, same class: ~5.3 TB/s -- tools/chunk_relation_probe.hip), then regions of
// the lookup column size are STITCHED from them with the virtual-memory API: class X only, class Y only, X / Y alternating in runs of
// 1, 2, 4 ... chunks.  One store stream is timed on each, in two block orders: linear (the resident workgroups cover ONE contiguous
// window) and XCD-contiguous (eight windows an eighth of the region apart -- the product kernels' order).
// usage: class_interleave_probe [chunk_mb = 32] [total_gb = 40] [region_gb = 5]
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/class_interleave_probe tools/class_interleave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned long long u64;
constexpr u64 PIECE = 65536, BLK = 4096;
__device__ __forceinline__ void st16(void *p, u64 a, u64 b) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    v2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
}
// `passes` sweeps over `pieces` 64 KB pieces per stream; B != nullptr: work item 2k is piece k of A, 2k + 1 piece k of B.
// xcd: work item w of a sweep is taken by block order such that the blocks of one XCD (blockIdx % 8) cover a contiguous eighth.
__global__ __launch_bounds__(256) void fill(char *A, char *B, u64 pieces, u64 passes, int xcd) {
    const u64 t = threadIdx.x, items = (B ? 2 : 1) * pieces;
    for (u64 ps = 0; ps < passes; ++ps)
        for (u64 b = blockIdx.x; b < items; b += gridDim.x) {
            u64 w = b;
            if (xcd) { const u64 per = (items + 7) / 8; w = (b & 7) * per + (b >> 3); if (w >= items) continue; }
            const u64 c = B ? w >> 1 : w;
            char *p = ((B && (w & 1)) ? B : A) + c * PIECE;
            for (u64 k = 0; k < PIECE / BLK; ++k) st16(p + k * BLK + t * 16, c, k + ps);
        }
}
static float time_fill(char *A, char *B, u64 bytes_each, u64 passes, int xcd = 0) {   // ms per pass
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 pieces = bytes_each / PIECE;
    const unsigned grid = (unsigned)std::min<u64>((B ? 2 : 1) * pieces, 256 * 8);
    hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, A, B, pieces, 1, xcd);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, A, B, pieces, passes, xcd);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / passes;
}
int main(int argc, char **argv) {
    const u64 chunk = (u64)(argc > 1 ? std::atoi(argv[1]) : 32) << 20;
    const u64 total = (u64)(argc > 2 ? std::atoi(argv[2]) : 40) << 30;
    const u64 region = (u64)(argc > 3 ? std::atoi(argv[3]) : 5) << 30;
    const u64 n = total / chunk;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    char *va = nullptr;
    CK(hipMemAddressReserve((void **)&va, total, 0, nullptr, 0));
    for (u64 i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemMap(va + i * chunk, chunk, 0, h[i], 0)); }
    CK(hipMemSetAccess(va, total, &acc, 1));
    std::printf("%llu chunks of %llu MB (%llu GB)\n", n, chunk >> 20, total >> 30);
    const u64 passes = std::max<u64>(4, (512ull << 20) / chunk);   // >= 1 GB written per measurement
    auto pair_tbs = [&](u64 i, u64 j) { return (float)(2.0 * chunk / time_fill(va + i * chunk, va + j * chunk, chunk, passes) / 1e9); };
    // class of every chunk: against chunk 0; then a chunk of the other class as the second reference settles the doubtful ones
    std::vector<float> r0(n, 0.f), r1(n, 0.f);
    for (u64 j = 1; j < n; ++j) r0[j] = pair_tbs(0, j);
    float lo = 1e9f, hi = 0.f;
    for (u64 j = 1; j < n; ++j) { lo = std::min(lo, r0[j]); hi = std::max(hi, r0[j]); }
    std::printf("pair fill with chunk 0: %.2f .. %.2f TB/s\n", lo, hi);
    if (hi / lo < 1.12f) { std::printf("one class only in this range: nothing to interleave\n"); return 0; }
    const float thr = 0.5f * (lo + hi);
    u64 gy = 0;
    for (u64 j = 1; j < n; ++j) if (r0[j] > thr + 0.25f * (hi - thr)) { gy = j; break; }
    for (u64 j = 0; j < n; ++j) if (j != gy) r1[j] = pair_tbs(gy, j);
    std::vector<int> cls(n, 0);   // 0 = chunk 0's class (X), 1 = the other (Y), -1 = the two references disagree
    int n_bad = 0;
    for (u64 j = 0; j < n; ++j) {
        const int by0 = j == 0 ? 0 : (r0[j] > thr ? 1 : 0), by1 = j == gy ? 1 : (r1[j] > thr ? 0 : 1);
        cls[j] = by0 == by1 ? by0 : -1;
        if (j == 0) cls[j] = by1 == 0 ? 0 : -1;
        if (j == gy) cls[j] = by0 == 1 ? 1 : -1;
        n_bad += cls[j] < 0;
    }
    std::printf("classes in allocation order (X = chunk 0's, Y = chunk %llu's, ? = references disagree): ", gy);
    for (u64 j = 0; j < n; ++j) std::printf("%c", cls[j] < 0 ? '?' : (cls[j] ? 'Y' : 'X'));
    std::printf("\n%d doubtful\n", n_bad);
    std::vector<u64> X, Y;
    for (u64 j = 0; j < n; ++j) { if (cls[j] == 0) X.push_back(j); else if (cls[j] == 1) Y.push_back(j); }
    const u64 rc = region / chunk;
    std::printf("chunks: %zu X, %zu Y; region = %llu chunks\n", X.size(), Y.size(), rc);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(va, total));
    auto stitch = [&](const std::vector<u64> &ids) -> char * {
        char *p = nullptr;
        CK(hipMemAddressReserve((void **)&p, ids.size() * chunk, 0, nullptr, 0));
        for (u64 k = 0; k < ids.size(); ++k) CK(hipMemMap(p + k * chunk, chunk, 0, h[ids[k]], 0));
        CK(hipMemSetAccess(p, ids.size() * chunk, &acc, 1));
        return p;
    };
    auto unstitch = [&](char *p, u64 cnt) { CK(hipDeviceSynchronize()); CK(hipMemUnmap(p, cnt * chunk)); CK(hipMemAddressFree(p, cnt * chunk)); };
    auto report = [&](const char *what, const std::vector<u64> &ids) {
        char *p = stitch(ids);
        const u64 bytes = ids.size() * chunk;
        const float tl = time_fill(p, nullptr, bytes, 3, 0), tx = time_fill(p, nullptr, bytes, 3, 1);
        std::printf("%-52s linear %.3f ms %.2f TB/s   XCD-contiguous %.3f ms %.2f TB/s\n", what, tl, bytes / tl / 1e9, tx, bytes / tx / 1e9);
        unstitch(p, ids.size());
    };
    if (X.size() >= rc) report("class X only", std::vector<u64>(X.begin(), X.begin() + rc));
    if (Y.size() >= rc) report("class Y only", std::vector<u64>(Y.begin(), Y.begin() + rc));
    if (X.size() >= rc / 2 + 1 && Y.size() >= rc / 2 + 1) {
        for (u64 run = 1; run * chunk <= (1ull << 30); run *= 2) {
            if (run * chunk < (16ull << 20) && run != 1) continue;
            std::vector<u64> ids;
            u64 ix = 0, iy = 0;
            while (ids.size() < rc) {
                for (u64 k = 0; k < run && ids.size() < rc && ix < X.size(); ++k) ids.push_back(X[ix++]);
                for (u64 k = 0; k < run && ids.size() < rc && iy < Y.size(); ++k) ids.push_back(Y[iy++]);
                if (ix >= X.size() && iy >= Y.size()) break;
            }
            if (ids.size() < rc) break;
            char what[96];
            std::snprintf(what, sizeof what, "X / Y alternating in runs of %llu MB", run * chunk >> 20);
            report(what, ids);
        }
        // first half X, second half Y (what the XCD-contiguous order turns into four windows of each)
        std::vector<u64> ids(X.begin(), X.begin() + rc / 2);
        ids.insert(ids.end(), Y.begin(), Y.begin() + (rc - rc / 2));
        report("first half X, second half Y", ids);
    }
    return 0;
}
