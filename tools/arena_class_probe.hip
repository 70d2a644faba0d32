// Does the CLASS PATTERN of a trace region's physical chunks decide how fast the record kernel writes it?  (profiles/r06_placement_counters.txt, 4.)
// 256 MB chunks (hipMemCreate) are classified against reference chunks by the concurrent pair fill; regions of config 2's trace size (1,024 elements:
// five chunks) are stitched in chosen patterns -- one class only, alternating, halves -- and the library's record kernel runs on each
// (h2r_pow_mod_fixed_exp_batch, per-kernel events from h2r_profile_*).  Run with the developer library and no internal overlap so that a call is ONE
// chain kernel + ONE record kernel:   LD_PRELOAD=halo2_rsa_amd/lib/variants/devknobs.so H2R_PLAIN_OVERLAP=0 tools/_bin/arena_class_probe
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/arena_class_probe.hip -o tools/_bin/arena_class_probe -Lhalo2_rsa_amd/lib -lh2r -Wl,-rpath,$PWD/halo2_rsa_amd/lib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
#include "h2r.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned long long u64;
constexpr u64 PIECE = 65536, BLK = 4096;
__device__ __forceinline__ void st16(void *p, u64 a, u64 b) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    v2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(p));
}
__global__ __launch_bounds__(256) void fill(char *A, char *B, u64 pieces, u64 passes) {
    const u64 t = threadIdx.x;
    for (u64 ps = 0; ps < passes; ++ps)
        for (u64 w = blockIdx.x; w < 2 * pieces; w += gridDim.x) {
            char *p = ((w & 1) ? B : A) + (w >> 1) * PIECE;
            for (u64 k = 0; k < PIECE / BLK; ++k) st16(p + k * BLK + t * 16, w, k + ps);
        }
}
static float pair_tbs(char *A, char *B, u64 bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, B, bytes / PIECE, 1);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, B, bytes / PIECE, 4);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (float)(2.0 * bytes / (ms / 4) / 1e9);
}
static u64 rs = 88172645463325252ull;
static u64 rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
int main(int argc, char **argv) {
    const u64 chunk = 256ull << 20, n = argc > 1 ? std::atoi(argv[1]) : 96;
    const u64 batch = 1024; const unsigned L = 32;
    h2r_params pr; std::memset(&pr, 0, sizeof pr); pr.limb_width = 64; pr.bits_len = 2048; pr.field = H2R_FIELD_BN254_FR; pr.device = 0;
    h2r_ctx *ctx = nullptr; if (h2r_ctx_create(&pr, &ctx)) { std::printf("ctx\n"); return 1; }
    const unsigned char e_le[3] = {1, 0, 1};
    h2r_pow_layout pl; h2r_pow_fixed_layout(ctx, e_le, 3, &pl);
    const u64 tbytes = batch * pl.elem_stride, rc = (tbytes + chunk - 1) / chunk;
    std::printf("trace of %llu elements: %llu bytes = %llu chunks of 256 MB; %u mul_mods per element\n", batch, tbytes, rc, pl.num_mul_mods);
    std::vector<u64> hx(batch * L), hn(batch * L);
    for (u64 i = 0; i < batch; ++i) { for (unsigned k = 0; k < L; ++k) { hn[i * L + k] = rnd(); hx[i * L + k] = rnd(); } hn[i * L] |= 1; hn[i * L + L - 1] |= 1ull << 63; hx[i * L + L - 1] &= ~(1ull << 63); }
    void *dx, *dn, *dout, *dws; unsigned char *dst;
    CK(hipMalloc(&dx, hx.size() * 8)); CK(hipMalloc(&dn, hn.size() * 8)); CK(hipMalloc(&dout, hx.size() * 8)); CK(hipMalloc(&dws, h2r_workspace_bytes(ctx, batch, pl.num_mul_mods))); CK(hipMalloc((void **)&dst, batch));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), hn.size() * 8, hipMemcpyHostToDevice));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(n); std::vector<char *> va(n);
    for (u64 i = 0; i < n; ++i) { CK(hipMemCreate(&h[i], chunk, &prop, 0)); CK(hipMemAddressReserve((void **)&va[i], chunk, 0, nullptr, 0)); CK(hipMemMap(va[i], chunk, 0, h[i], 0)); CK(hipMemSetAccess(va[i], chunk, &acc, 1)); }
    std::vector<float> r0(n, 0.f);
    for (u64 j = 1; j < n; ++j) r0[j] = pair_tbs(va[0], va[j], chunk);
    float lo = 1e9f, hi = 0.f; for (u64 j = 1; j < n; ++j) { lo = std::min(lo, r0[j]); hi = std::max(hi, r0[j]); }
    std::printf("pair fill with chunk 0: %.2f .. %.2f TB/s\n", lo, hi);
    if (hi / lo < 1.12f) { std::printf("one class only among these chunks\n"); return 0; }
    const float thr = 0.5f * (lo + hi);
    u64 gy = 0; for (u64 j = 1; j < n; ++j) if (r0[j] > thr) { gy = j; break; }
    std::vector<int> cls(n, 0);
    std::string pat;
    for (u64 j = 0; j < n; ++j) {
        const int by0 = j == 0 ? 0 : (r0[j] > thr ? 1 : 0);
        const float r1 = j == gy ? 0.f : pair_tbs(va[gy], va[j], chunk);
        const int by1 = j == gy ? 1 : (r1 > thr ? 0 : 1);
        cls[j] = by0 == by1 ? by0 : -1;
        pat += cls[j] < 0 ? '?' : (cls[j] ? 'Y' : 'X');
    }
    std::printf("classes in allocation order: %s\n", pat.c_str());
    std::vector<u64> X, Y; for (u64 j = 0; j < n; ++j) { if (cls[j] == 0) X.push_back(j); else if (cls[j] == 1) Y.push_back(j); }
    std::printf("%zu X, %zu Y\n", X.size(), Y.size());
    CK(hipDeviceSynchronize());
    for (u64 i = 0; i < n; ++i) CK(hipMemUnmap(va[i], chunk));
    h2r_profile_enable(256);
    auto run_region = [&](const char *what, const std::vector<u64> &ids) {
        char *p = nullptr; CK(hipMemAddressReserve((void **)&p, ids.size() * chunk, 0, nullptr, 0));
        for (u64 k = 0; k < ids.size(); ++k) CK(hipMemMap(p + k * chunk, chunk, 0, h[ids[k]], 0));
        CK(hipMemSetAccess(p, ids.size() * chunk, &acc, 1));
        float best = 1e9f, sum = 0.f; int cnt = 0;
        for (int rep = 0; rep < 5; ++rep) {
            h2r_profile_enable(0); h2r_profile_enable(64);
            if (h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, p, dout, dst, dws, nullptr)) { std::printf("pow failed\n"); std::exit(1); }
            CK(hipDeviceSynchronize());
            float ms[64]; unsigned c = 0; h2r_profile_read(H2R_KERNEL_TRACE, ms, 64, &c);
            float tot = 0; for (unsigned k = 0; k < c && k < 64; ++k) tot += ms[k];
            if (rep) { best = std::min(best, tot); sum += tot; ++cnt; }
            if (rep == 1 && c != 1) std::printf("  (%u record launches per call)\n", c);
        }
        const double algo = (double)batch * pl.num_mul_mods * 64338.0;
        std::printf("%-26s record kernel %.4f ms avg, %.4f best = %.2f TB/s (%.3f of the peak)\n", what, sum / cnt, best, algo / (sum / cnt) / 1e9, algo / (sum / cnt) / 8e9);
        CK(hipDeviceSynchronize()); CK(hipMemUnmap(p, ids.size() * chunk)); CK(hipMemAddressFree(p, ids.size() * chunk));
    };
    auto pick = [&](const char *pattern, u64 offset) {
        std::vector<u64> ids; u64 ix = offset, iy = offset;
        for (const char *c = pattern; *c; ++c) { if (*c == 'X') { if (ix >= X.size()) return ids; ids.push_back(X[ix++]); } else { if (iy >= Y.size()) return ids; ids.push_back(Y[iy++]); } }
        return ids;
    };
    const char *pats[] = {"XXXXX", "YYYYY", "XYXYX", "YXYXY", "XXYYX", "XXXYY", "YYXXX", "XYYYX", "XXXXY", "XYXXX"};
    for (int round = 0; round < 2; ++round)
        for (const char *pt : pats) {
            std::vector<u64> ids = pick(pt, round * 5);
            if (ids.size() != std::strlen(pt)) { std::printf("%-26s not enough chunks of a class\n", pt); continue; }
            char nm[64]; std::snprintf(nm, sizeof nm, "%s (round %d)", pt, round);
            run_region(nm, ids);
        }
    // plain hipMalloc regions for comparison
    for (int k = 0; k < 3; ++k) { char *p; CK(hipMalloc((void **)&p, tbytes)); 
        float sum = 0; for (int rep = 0; rep < 4; ++rep) { h2r_profile_enable(0); h2r_profile_enable(64); h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, p, dout, dst, dws, nullptr); CK(hipDeviceSynchronize()); float ms[64]; unsigned c = 0; h2r_profile_read(H2R_KERNEL_TRACE, ms, 64, &c); float tot = 0; for (unsigned q = 0; q < c && q < 64; ++q) tot += ms[q]; if (rep) sum += tot; }
        std::printf("plain hipMalloc %d             record kernel %.4f ms avg = %.3f of the peak\n", k, sum / 3, (double)batch * pl.num_mul_mods * 64338.0 / (sum / 3) / 8e9); }
    return 0;
}
