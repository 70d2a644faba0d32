// Which part of the pair fill's access pattern carries the class effect?  Two 1 GB chunks of different classes (X, Y) and two of the same
// class (X, X'), found by the pair fill; then 2 GB are written by a persistent grid (2,048 workgroups) whose work item -> (stream, 64 KB piece)
// mapping varies:
//   P1  item w: stream w & 1, piece w >> 1                        (the pair fill: neighbouring workgroups = XCDs alternate streams, pieces ascend)
//   P2  stream (w >> 3) & 1, piece ((w >> 4) << 3) | (w & 7)      (all eight XCDs write one stream for a 512 KB run, then the other)
//   P3  stream = XCD < 4 (w & 7 < 4), pieces ascending per stream (XCDs 0-3 write A, 4-7 write B)
//   P4  stream = XCD parity, every XCD its own contiguous quarter of its stream
//   P5  stream = XCD < 4, every XCD its own contiguous quarter   (the XCD-contiguous order over a region whose halves are A and B)
//   P6  first all of A, then all of B                             (one stream at a time)
//   P7  stream w & 1, piece (w >> 1) for A but B walked from its END (no equal offsets in flight)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/pair_pattern_probe tools/pair_pattern_probe.hip
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
__global__ __launch_bounds__(256) void fill(char *A, char *B, u64 pieces, int mode) {
    const u64 t = threadIdx.x, items = 2 * pieces;
    for (u64 w = blockIdx.x; w < items; w += gridDim.x) {
        u64 s, c;
        const u64 x = w & 7, g = w >> 3;           // XCD of the block (the grid is a multiple of 8), its g-th item
        switch (mode) {
            case 1: s = w & 1; c = w >> 1; break;
            case 2: s = (w >> 3) & 1; c = ((w >> 4) << 3) | (w & 7); break;
            case 3: s = x < 4 ? 0 : 1; c = g * 4 + (x & 3); break;
            case 4: s = x & 1; c = (x >> 1) * (pieces / 4) + g; break;
            case 5: s = x < 4 ? 0 : 1; c = (x & 3) * (pieces / 4) + g; break;
            case 6: s = w < pieces ? 0 : 1; c = w < pieces ? w : w - pieces; break;
            default: s = w & 1; c = s ? pieces - 1 - (w >> 1) : (w >> 1); break;
        }
        if (c >= pieces) continue;
        char *p = (s ? B : A) + c * PIECE;
        for (u64 k = 0; k < PIECE / BLK; ++k) st16(p + k * BLK + t * 16, c, k);
    }
}
static float tbs(char *A, char *B, u64 bytes_each, int mode) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 pieces = bytes_each / PIECE;
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, B, pieces, mode);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, B, pieces, mode);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (float)(2.0 * bytes_each / (ms / 3) / 1e9);
}
int main(int argc, char **argv) {
    const u64 GB = 1ull << 30, n = argc > 1 ? std::atoi(argv[1]) : 48;
    std::vector<char *> c(n);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (u64 i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], GB, &prop, 0)); CK(hipMemAddressReserve((void **)&c[i], GB, 0, nullptr, 0)); CK(hipMemMap(c[i], GB, 0, h[i], 0)); CK(hipMemSetAccess(c[i], GB, &acc, 1)); }
    std::vector<float> r(n, 0.f);
    std::printf("pair fill (P1) of GB 0 + GB j:");
    for (u64 j = 1; j < n; ++j) { r[j] = tbs(c[0], c[j], GB, 1); std::printf(" %.2f", r[j]); }
    std::printf("\n");
    u64 jy = 0, jx = 0;
    for (u64 j = 1; j < n; ++j) { if (!jy && r[j] > 6.5f) jy = j; if (!jx && r[j] < 5.8f) jx = j; }
    if (!jy || !jx) { std::printf("no pair of each kind among these chunks\n"); return 0; }
    std::printf("different classes: GB 0 + GB %llu; same class: GB 0 + GB %llu\n%-4s %12s %12s %12s\n", jy, jx, "", "X + Y", "X + X'", "Y + X (swapped)");
    for (int m = 1; m <= 7; ++m) std::printf("P%d   %9.2f    %9.2f    %9.2f   TB/s\n", m, tbs(c[0], c[jy], GB, m), tbs(c[0], c[jx], GB, m), tbs(c[jy], c[0], GB, m));
    // the same two chunks behind ONE virtual range ([X | Y]): does the separate reservation matter?
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(c[0], GB)); CK(hipMemUnmap(c[jy], GB)); CK(hipMemUnmap(c[jx], GB));
    char *v = nullptr;
    CK(hipMemAddressReserve((void **)&v, 3 * GB, 0, nullptr, 0));
    CK(hipMemMap(v, GB, 0, h[0], 0)); CK(hipMemMap(v + GB, GB, 0, h[jy], 0)); CK(hipMemMap(v + 2 * GB, GB, 0, h[jx], 0));
    CK(hipMemSetAccess(v, 3 * GB, &acc, 1));
    std::printf("one virtual range [X | Y | X']:\n");
    for (int m : {1, 5, 6}) std::printf("P%d   %9.2f    %9.2f   TB/s\n", m, tbs(v, v + GB, GB, m), tbs(v, v + 2 * GB, GB, m));
    // regions of 6 GB stitched from 1 GB chunks, ONE store stream: linear order (one 128 MB window in flight) and XCD-contiguous order
    // (eight windows, an eighth of the region apart)
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(v, 3 * GB)); CK(hipMemAddressFree(v, 3 * GB));
    std::vector<u64> X{0, jx}, Y{jy};
    for (u64 j = 1; j < n && (X.size() < 6 || Y.size() < 6); ++j) {
        if (j == jx || j == jy) continue;
        if (r[j] > 6.5f && Y.size() < 6) Y.push_back(j); else if (r[j] < 5.8f && X.size() < 6) X.push_back(j);
    }
    std::printf("X chunks:"); for (u64 k : X) std::printf(" %llu", k); std::printf("   Y chunks:"); for (u64 k : Y) std::printf(" %llu", k); std::printf("\n");
    if (X.size() >= 6 && Y.size() >= 6) {
        for (u64 k : X) if (k != 0 && k != jx) CK(hipMemUnmap(c[k], GB));
        for (u64 k : Y) if (k != jy) CK(hipMemUnmap(c[k], GB));
        auto region = [&](const char *what, std::vector<u64> ids) {
            char *q = nullptr;
            CK(hipMemAddressReserve((void **)&q, ids.size() * GB, 0, nullptr, 0));
            for (u64 k = 0; k < ids.size(); ++k) CK(hipMemMap(q + k * GB, GB, 0, h[ids[k]], 0));
            CK(hipMemSetAccess(q, ids.size() * GB, &acc, 1));
            const u64 half = ids.size() * GB / 2;
            // one stream over the whole region = modes 6 (linear: A then B, A | B adjacent) and 5 (XCD-contiguous eighths) with A = first half, B = second half
            std::printf("%-28s linear %.2f   XCD-contiguous %.2f   alternate workgroups between halves %.2f  TB/s\n", what, tbs(q, q + half, half, 6), tbs(q, q + half, half, 5), tbs(q, q + half, half, 1));
            CK(hipDeviceSynchronize()); CK(hipMemUnmap(q, ids.size() * GB)); CK(hipMemAddressFree(q, ids.size() * GB));
        };
        region("X X X X X X", {X[0], X[1], X[2], X[3], X[4], X[5]});
        region("Y Y Y Y Y Y", {Y[0], Y[1], Y[2], Y[3], Y[4], Y[5]});
        region("X X X Y Y Y", {X[0], X[1], X[2], Y[0], Y[1], Y[2]});
        region("X Y X Y X Y", {X[0], Y[0], X[1], Y[1], X[2], Y[2]});
        region("X X Y Y X X", {X[0], X[1], Y[0], Y[1], X[2], X[3]});
    }
    return 0;
}
