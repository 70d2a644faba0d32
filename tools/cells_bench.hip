// Developer bench of cells_kernel (h2r_cells.hpp): ablations and residency sweeps at BASELINE config 2's size without rebuilding
// the library.  Operands come from a real h2r_pow_mod_fixed_exp_batch call (no records), the row table and the constants' table are
// rebuilt here the way h2r_ctx_create builds them.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ihalo2_rsa_amd/csrc tools/cells_bench.hip -o /tmp/cells_bench -Lhalo2_rsa_amd/lib -lh2r -Wl,-rpath,$PWD/halo2_rsa_amd/lib
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "h2r.h"
#include "h2r_layout.hpp"
#include "h2r_cells.hpp"
using namespace h2r;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static u64 rng_state = 0x9e3779b97f4a7c15ull;
static u64 rnd() { u64 z = (rng_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

__global__ void diff_kernel(const uint4 *x, const uint4 *y, u64 n, unsigned long long *cnt, unsigned long long *first) {
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const uint4 a = x[i], b = y[i];
        if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) { atomicAdd(cnt, 1ull); atomicMin(first, (unsigned long long)i); }
    }
}

template <int LW, int ABL>
float run(const CellsArgs &ca, u32 lds, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cells_kernel<LW, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((cells_kernel<LW, ABL>), dim3((unsigned)ca.n_items), dim3(64), lds, 0, ca);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((cells_kernel<LW, ABL>), dim3((unsigned)ca.n_items), dim3(64), lds, 0, ca);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const u32 w = argc > 1 ? (u32)std::atoi(argv[1]) : 64, bits = argc > 2 ? (u32)std::atoi(argv[2]) : 2048;
    const u64 batch = argc > 3 ? (u64)std::atoll(argv[3]) : 1024;
    const u32 L = bits / w;
    h2r_params pr; std::memset(&pr, 0, sizeof pr);
    pr.limb_width = w; pr.bits_len = bits; pr.field = H2R_FIELD_BN254_FR; pr.device = 0;
    h2r_ctx *ctx = nullptr;
    if (h2r_ctx_create(&pr, &ctx)) { std::printf("ctx_create failed\n"); return 1; }
    const u8 e_le[3] = {1, 0, 1};
    h2r_pow_layout pl; h2r_pow_fixed_layout(ctx, e_le, 3, &pl);
    const u32 T = pl.num_mul_mods;
    const u64 lb = w / 8;
    std::vector<u8> hx(batch * L * lb), hn(batch * L * lb);
    for (u64 e = 0; e < batch; ++e) {
        for (u32 k = 0; k < L * lb / 8; ++k) { u64 a = rnd(), b = rnd(); std::memcpy(&hn[(e * L * lb) + 8 * k], &a, 8); std::memcpy(&hx[(e * L * lb) + 8 * k], &b, 8); }
        hn[e * L * lb] |= 1; hn[(e + 1) * L * lb - 1] |= 0x80; hx[(e + 1) * L * lb - 1] &= 0x7f;
    }
    void *dx, *dn, *dout, *dws; u8 *dst;
    const u64 wsb = h2r_workspace_bytes(ctx, batch, T);
    CK(hipMalloc(&dx, hx.size())); CK(hipMalloc(&dn, hn.size())); CK(hipMalloc(&dout, hx.size())); CK(hipMalloc(&dws, wsb)); CK(hipMalloc(reinterpret_cast<void **>(&dst), batch));
    CK(hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), hn.size(), hipMemcpyHostToDevice));
    CK(hipMemset(dst, 0, batch));
    int32_t rc = h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, nullptr, dout, dst, dws, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<u8> hst(batch); CK(hipMemcpy(hst.data(), dst, batch, hipMemcpyDeviceToHost));
    u64 nbad = 0; for (u8 s : hst) nbad += s != 0;
    std::printf("pow rc=%d, %llu elements with a status\n", rc, (unsigned long long)nbad);
    h2r_layout lo; h2r_trace_layout(ctx, &lo);
    const u32 rows = h2r_advice_rows(ctx), nrc = (lo.carry_nsub + 3) / 4;
    u64 kt[CELLS_TAB_WORDS] = {0};
    {   // the accumulated_extra chain (chip.rs:869-875)
        const U256 wm = compute_mul_word_max(w, L);
        U256 acc;
        for (u32 i = 0; i < 3; ++i) {
            acc = acc + wm;
            const U256 q = acc.shr(w), nq = q.shl(w);
            u64 *e = &kt[10 * i];
            e[0] = acc.v[0]; e[1] = acc.v[1]; e[2] = acc.v[2]; e[3] = q.v[0]; e[4] = q.v[1]; e[5] = acc.low(w);
            e[6] = nq.v[0]; e[7] = nq.v[1]; e[8] = nq.v[2]; e[9] = (acc - nq).low(w);
            acc = q;
        }
    }
    { const U256 wm = compute_mul_word_max(w, L); kt[CELLS_KT_WM] = wm.v[0]; kt[CELLS_KT_WM + 1] = wm.v[1]; kt[CELLS_KT_WM + 2] = wm.v[2]; }
    { u64 p[4]; FieldConsts fc; field_modulus(pr.field, p); field_consts_init(p, &fc); for (int k = 0; k < 4; ++k) kt[CELLS_KT_P + k] = p[k]; std::memcpy(&kt[CELLS_KT_FC], &fc, sizeof fc); }
    {
        const CellsLds lp = cells_lds_plan(w, L);
        u32 *fs = reinterpret_cast<u32 *>(&kt[CELLS_KT_FSRC]);
        for (u32 k = 0; k < ADVICE_COL_ROWS * 3; ++k) {
            fs[k] = cells_pack_fast_src(lp, w, cells_fast_src(k / 3, k % 3, false));
            fs[CELLS_SRC_WORDS + k] = cells_pack_fast_src(lp, w, cells_fast_src(k / 3, k % 3, true));
        }
    }
    u64 *dkt; u8 *dimg;
    const u64 out_stride = (2ull + (u64)T * rows) * 160, img = batch * out_stride;
    CK(hipMalloc(reinterpret_cast<void **>(&dkt), sizeof kt)); CK(hipMalloc(reinterpret_cast<void **>(&dimg), img));
    CK(hipMemcpy(dkt, kt, sizeof kt, hipMemcpyHostToDevice));
    CellsArgs ca; std::memset(&ca, 0, sizeof ca);
    const u8 *ws = reinterpret_cast<const u8 *>(round_up(reinterpret_cast<u64>(dws), 256));
    ca.ktab = dkt; ca.per_col_magic = (u32)(((1ull << 32) + (ADVICE_COL_ROWS + nrc) - 1) / (ADVICE_COL_ROWS + nrc)); ca.opA = ws; ca.opB = ws + L * lb; ca.opQ = ws + 2 * L * lb; ca.opR = ws + 3 * L * lb; ca.op_stride = ca.qr_stride = 4ull * L;
    ca.n = dn; ca.n_stride = L; ca.status = dst; ca.T = T; ca.n_items = batch * T; ca.out = dimg; ca.out_stride = out_stride;
    ca.rows = rows; ca.pre_rows = 2; ca.L = L; ca.carry_sub_bits = lo.carry_sub_bits; ca.carry_nsub = lo.carry_nsub;
    const u32 base = cells_lds_bytes(w, L);
    const double gb = (double)img / 1e9;
    std::printf("w=%u L=%u batch=%llu T=%u rows=%u image %.2f GB, lds/wave %u\n", w, L, (unsigned long long)batch, T, rows, gb, base);
    auto line = [&](const char *nm, float ms) { std::printf("  %-34s %.3f ms  %.2f TB/s\n", nm, ms, gb / ms); std::fflush(stdout); };
    const int R = 5;
    {   // the fast paths against the general path (which tools/cells_check.py pins against advice_kernel)
        u8 *dref; CK(hipMalloc(reinterpret_cast<void **>(&dref), img));
        CK(hipMemset(dimg, 0xee, img)); CK(hipMemset(dref, 0xee, img));
        CellsArgs cr = ca; cr.out = dref;
        if (w == 64) { run<64, 0>(ca, base, 1); run<64, 64 | 128>(cr, base, 1); } else { run<32, 0>(ca, base, 1); run<32, 64 | 128>(cr, base, 1); }
        unsigned long long *dc; CK(hipMalloc(reinterpret_cast<void **>(&dc), 16)); CK(hipMemset(dc, 0, 8)); CK(hipMemset(dc + 1, 0xff, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(dimg), reinterpret_cast<const uint4 *>(dref), img / 16, dc, dc + 1);
        unsigned long long hc[2]; CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
        if (hc[0]) {
            const u64 piece = hc[1], e = piece * 16 / out_stride, rrow = (piece * 16 % out_stride) / 160;
            std::printf("FAST vs GENERAL: %llu differing pieces; first at element %llu row %llu (record row %lld) piece %llu\n", hc[0], (unsigned long long)e,
                        (unsigned long long)rrow, (long long)((rrow >= 2 ? rrow - 2 : 0) % rows), (unsigned long long)(piece * 16 % 160 / 16));
        } else std::printf("FAST vs GENERAL: identical (%.2f GB)\n", gb);
        CK(hipFree(dref));
    }
    {   // per-chunk cycle stamps of one wave (item in the middle of the grid), next to the full grid and alone
        u64 *ddbg; CK(hipMalloc(reinterpret_cast<void **>(&ddbg), 3000 * 8));
        for (int alone = 0; alone < 2; ++alone) {
            CK(hipMemset(ddbg, 0, 3000 * 8));
            CellsArgs cd = ca; cd.dbg = ddbg; cd.dbg_item = alone ? 0 : (u32)(ca.n_items / 2);
            if (alone) cd.n_items = 1;
            if (w == 64) run<64, 256>(cd, base, 1); else run<32, 256>(cd, base, 1);
            std::vector<u64> h(3000); CK(hipMemcpy(h.data(), ddbg, 3000 * 8, hipMemcpyDeviceToHost));
            u64 sb[3] = {0, 0, 0}, sr[3] = {0, 0, 0}, n[3] = {0, 0, 0};
            const u32 nchunks = (rows + 63) / 64 + 1;
            for (u32 c = 0; c < nchunks && c < 1000; ++c) { const u64 pth = h[3 * c] < 3 ? h[3 * c] : 0; sb[pth] += h[3 * c + 1]; sr[pth] += h[3 * c + 2]; ++n[pth]; }
            std::printf("  cycles per chunk (%s): general %llu build + %llu readout (%llu chunks); mul rows %llu + %llu (%llu); column rows %llu + %llu (%llu)\n",
                        alone ? "one wave alone" : "full grid", n[0] ? sb[0] / n[0] : 0, n[0] ? sr[0] / n[0] : 0, n[0], n[1] ? sb[1] / n[1] : 0, n[1] ? sr[1] / n[1] : 0, n[1],
                        n[2] ? sb[2] / n[2] : 0, n[2] ? sr[2] / n[2] : 0, n[2]);
        }
        CK(hipFree(ddbg));
    }
    if (argc > 4 && !std::strcmp(argv[4], "vmm")) {   // images stitched from physical chunks of one size (HIP virtual-memory API), in creation order and shuffled
        const u32 l4 = (160u * 1024 / 4 - 512) & ~15u, lds = l4 > base ? l4 : base;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        std::printf("  allocation granularity %zu\n", gran);
        for (size_t chunk_mb : {2, 16, 64, 256, 1024, 4096}) {
            for (int shuffled = 0; shuffled < 2; ++shuffled) {
                for (int rep = 0; rep < 3; ++rep) {
                    const size_t chunk = ((chunk_mb << 20) + gran - 1) / gran * gran, n = (img + chunk - 1) / chunk;
                    void *va = nullptr;
                    if (hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0) != hipSuccess) { std::printf("reserve failed\n"); return 1; }
                    std::vector<hipMemGenericAllocationHandle_t> hs(n);
                    bool ok = true;
                    for (size_t i = 0; i < n && ok; ++i) ok = hipMemCreate(&hs[i], chunk, &prop, 0) == hipSuccess;
                    if (!ok) { std::printf("  chunk %zu MB: out of memory\n", chunk_mb); (void)hipGetLastError(); break; }
                    std::vector<size_t> order(n);
                    for (size_t i = 0; i < n; ++i) order[i] = i;
                    if (shuffled) for (size_t i = n - 1; i > 0; --i) std::swap(order[i], order[rnd() % (i + 1)]);
                    for (size_t i = 0; i < n; ++i) CK(hipMemMap((char *)va + i * chunk, chunk, 0, hs[order[i]], 0));
                    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
                    CellsArgs c2 = ca; c2.out = static_cast<u8 *>(va);
                    const float ms = w == 64 ? run<64, 0>(c2, lds, 3) : run<32, 0>(c2, lds, 3);
                    std::printf("  chunks of %4zu MB%s (%zu): full %.3f ms %.2f TB/s\n", chunk_mb, shuffled ? ", shuffled" : "          ", n, ms, gb / ms); std::fflush(stdout);
                    if (rep == 2 || chunk_mb >= 1024) {   // (keep two images of every kind allocated so that later ones land elsewhere; the big-chunk ones are freed)
                        CK(hipDeviceSynchronize()); CK(hipMemUnmap(va, n * chunk));
                        for (size_t i = 0; i < n; ++i) CK(hipMemRelease(hs[i]));
                        CK(hipMemAddressFree(va, n * chunk));
                    }
                }
            }
        }
        return 0;
    }
    if (argc > 4 && !std::strcmp(argv[4], "vmm2")) {   // eight images as allocated, then eight stitched from 256 MB / 1 GB physical chunks (all eight held): is the stitched kind consistently one class?
        const u32 l4 = (160u * 1024 / 4 - 512) & ~15u, lds = l4 > base ? l4 : base;
        auto rate = [&](u8 *p) { CellsArgs c2 = ca; c2.out = p; const float ms = w == 64 ? run<64, 0>(c2, lds, 3) : run<32, 0>(c2, lds, 3); return gb / ms; };
        {
            std::vector<u8 *> keep;
            std::printf("  as allocated (hipMalloc):");
            for (int i = 0; i < 8; ++i) { u8 *p = nullptr; if (hipMalloc(reinterpret_cast<void **>(&p), img) != hipSuccess) { (void)hipGetLastError(); break; } keep.push_back(p); std::printf(" %.2f", rate(p)); std::fflush(stdout); }
            std::printf(" TB/s\n");
            for (u8 *p : keep) CK(hipFree(p));
        }
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        for (size_t chunk_mb : {256, 1024, 128, 512}) {
            const size_t chunk = ((chunk_mb << 20) + gran - 1) / gran * gran, n = (img + chunk - 1) / chunk;
            struct Img { void *va; std::vector<hipMemGenericAllocationHandle_t> hs; };
            std::vector<Img> imgs;
            std::printf("  stitched from %4zu MB chunks (%zu per image):", chunk_mb, n);
            for (int k = 0; k < 8; ++k) {
                Img im; im.va = nullptr; im.hs.resize(n);
                if (hipMemAddressReserve(&im.va, n * chunk, 0, nullptr, 0) != hipSuccess) { std::printf(" reserve failed"); break; }
                bool ok = true; size_t made = 0;
                for (; made < n && ok; ++made) ok = hipMemCreate(&im.hs[made], chunk, &prop, 0) == hipSuccess;
                if (!ok) { (void)hipGetLastError(); for (size_t i = 0; i + 1 < made; ++i) (void)hipMemRelease(im.hs[i]); (void)hipMemAddressFree(im.va, n * chunk); std::printf(" (out of memory)"); break; }
                for (size_t i = 0; i < n; ++i) CK(hipMemMap((char *)im.va + i * chunk, chunk, 0, im.hs[i], 0));
                hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                CK(hipMemSetAccess(im.va, n * chunk, &acc, 1));
                std::printf(" %.2f", rate(static_cast<u8 *>(im.va))); std::fflush(stdout);
                imgs.push_back(std::move(im));
            }
            std::printf(" TB/s\n");
            CK(hipDeviceSynchronize());
            for (Img &im : imgs) { CK(hipMemUnmap(im.va, n * chunk)); for (auto h : im.hs) CK(hipMemRelease(h)); CK(hipMemAddressFree(im.va, n * chunk)); }
        }
        return 0;
    }
    if (argc > 4 && !std::strcmp(argv[4], "fragrec")) {
        // The RECORD kernel's regions (one call's op-trace: batch x elem_stride bytes) as allocated, and allocated from the holes of a
        // fragmented free memory (chunks of argv[5] MB, a random half released): does conditioning the allocator give fast regions?
        const u64 region = batch * pl.elem_stride + 4096;
        auto rec_ms = [&](void *p) {
            void *t = reinterpret_cast<void *>(round_up(reinterpret_cast<u64>(p), 256));
            for (int i = 0; i < 2; ++i) (void)h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, t, dout, dst, dws, nullptr);
            CK(hipDeviceSynchronize());
            (void)h2r_profile_enable(16);
            for (int i = 0; i < 4; ++i) (void)h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, t, dout, dst, dws, nullptr);
            CK(hipDeviceSynchronize());
            float ms[16]; u32 n = 0; (void)h2r_profile_read(H2R_KERNEL_TRACE, ms, 16, &n);
            (void)h2r_profile_enable(0);
            float sum = 0; for (u32 i = 0; i < n && i < 16; ++i) sum += ms[i];
            return n ? sum / (n < 16 ? n : 16) : 0.f;
        };
        const int NR = argc > 6 ? std::atoi(argv[6]) : 8;
        std::printf("  region %.2f GB; record kernel alone, ms per launch\n", (double)region / 1e9);
        for (int round = 0; round < 2; ++round) {
            {
                std::vector<void *> keep;
                std::printf("  as allocated:      ");
                for (int i = 0; i < NR; ++i) { void *p = nullptr; CK(hipMalloc(&p, region)); keep.push_back(p); std::printf(" %.4f", rec_ms(p)); std::fflush(stdout); }
                std::printf("\n");
                for (void *p : keep) CK(hipFree(p));
            }
            const size_t chunk = (size_t)(argc > 5 ? std::atoi(argv[5]) : 16) << 20;
            std::vector<void *> ballast;
            const auto t0 = std::chrono::steady_clock::now();
            while (true) {
                size_t f2 = 0, tot = 0; CK(hipMemGetInfo(&f2, &tot));
                if (f2 < (4ull << 30)) break;
                void *q = nullptr; if (hipMalloc(&q, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
                ballast.push_back(q);
            }
            u64 x = 0x9e3779b97f4a7c15ull + 1315423911ull * (u64)round; size_t freed = 0;
            for (size_t i = 0; i < ballast.size(); ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; if (x & 1) { CK(hipFree(ballast[i])); ballast[i] = nullptr; ++freed; } }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<void *> keep;
            std::printf("  from random holes: ");
            for (int i = 0; i < NR; ++i) { void *p = nullptr; if (hipMalloc(&p, region) != hipSuccess) { (void)hipGetLastError(); break; } keep.push_back(p); std::printf(" %.4f", rec_ms(p)); std::fflush(stdout); }
            std::printf("   (%zu chunks of %zu MB, %zu released, %.1f s)\n", ballast.size(), chunk >> 20, freed, secs);
            for (void *q : ballast) if (q) CK(hipFree(q));
            std::printf("  ballast released:  ");
            for (void *p : keep) std::printf(" %.4f", rec_ms(p));
            std::printf("\n");
            for (void *p : keep) CK(hipFree(p));
        }
        return 0;
    }
    if (argc > 4 && !std::strcmp(argv[4], "churn")) {
        // Regions of the record kernel as allocated in a fresh process, then after the whole free memory has been taken as chunks of argv[5] MB
        // and released again (in allocation order / argv[6] = 1: in a pseudo-random order): does a churned allocator hand out fast regions?
        const u64 region = batch * pl.elem_stride + 4096;
        auto rec_ms = [&](void *p) {
            void *t = reinterpret_cast<void *>(round_up(reinterpret_cast<u64>(p), 256));
            for (int i = 0; i < 2; ++i) (void)h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, t, dout, dst, dws, nullptr);
            CK(hipDeviceSynchronize());
            (void)h2r_profile_enable(16);
            for (int i = 0; i < 4; ++i) (void)h2r_pow_mod_fixed_exp_batch(ctx, dx, dn, e_le, 3, batch, 0, t, dout, dst, dws, nullptr);
            CK(hipDeviceSynchronize());
            float ms[16]; u32 n = 0; (void)h2r_profile_read(H2R_KERNEL_TRACE, ms, 16, &n);
            (void)h2r_profile_enable(0);
            float sum = 0; for (u32 i = 0; i < n && i < 16; ++i) sum += ms[i];
            return n ? sum / (n < 16 ? n : 16) : 0.f;
        };
        auto look = [&](const char *what, int nr) {
            std::vector<void *> keep;
            std::printf("  %-28s", what);
            for (int i = 0; i < nr; ++i) { void *p = nullptr; if (hipMalloc(&p, region) != hipSuccess) { (void)hipGetLastError(); break; } keep.push_back(p); std::printf(" %.3f", rec_ms(p)); std::fflush(stdout); }
            std::printf("\n");
            for (void *p : keep) CK(hipFree(p));
        };
        const bool shuffle = argc > 6 && std::atoi(argv[6]) != 0;
        look("fresh process:", 12);
        look("again:", 12);
        for (int round = 0; round < 2; ++round) {
            const size_t chunk = (size_t)(argc > 5 ? std::atoi(argv[5]) : 2) << 20;
            std::vector<void *> ballast;
            const auto t0 = std::chrono::steady_clock::now();
            while (true) {
                size_t f2 = 0, tot = 0; CK(hipMemGetInfo(&f2, &tot));
                if (f2 < (2ull << 30)) break;
                void *q = nullptr; if (hipMalloc(&q, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
                ballast.push_back(q);
            }
            if (shuffle) for (size_t i = ballast.size() - 1; i > 0; --i) std::swap(ballast[i], ballast[rnd() % (i + 1)]);
            for (void *q : ballast) CK(hipFree(q));
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("  churn: %zu chunks of %zu MB taken and released%s, %.1f s\n", ballast.size(), chunk >> 20, shuffle ? " in a random order" : "", secs);
            look("after the churn:", 16);
            look("again:", 16);
        }
        return 0;
    }
    if (argc > 4 && !std::strcmp(argv[4], "fragment")) {
        // Does memory that the driver has to ASSEMBLE from scattered free blocks land in the fast class?  (1) six images as allocated;
        // (2) the free memory taken as chunks of argv[5] MB, every other chunk released, six images allocated from the holes.
        const u32 l4 = (160u * 1024 / 4 - 512) & ~15u, lds = l4 > base ? l4 : base;
        auto rate = [&](u8 *p) { CellsArgs c2 = ca; c2.out = p; const float ms = w == 64 ? run<64, 0>(c2, lds, 3) : run<32, 0>(c2, lds, 3); return gb / ms; };
        const int NI = argc > 6 ? std::atoi(argv[6]) : 6;
        {
            std::vector<u8 *> keep;
            std::printf("  as allocated:");
            for (int i = 0; i < NI; ++i) { u8 *p = nullptr; if (hipMalloc(reinterpret_cast<void **>(&p), img) != hipSuccess) { (void)hipGetLastError(); break; } keep.push_back(p); std::printf(" %.2f", rate(p)); std::fflush(stdout); }
            std::printf(" TB/s\n");
            for (u8 *p : keep) CK(hipFree(p));
        }
        for (int rep = 0; rep < 2; ++rep) {
            const size_t chunk = (size_t)(argc > 5 ? std::atoi(argv[5]) : 16) << 20;
            size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
            std::vector<void *> ballast;
            const auto t0 = std::chrono::steady_clock::now();
            while (true) {
                size_t f2 = 0; CK(hipMemGetInfo(&f2, &tot));
                if (f2 < (4ull << 30)) break;
                void *q = nullptr; if (hipMalloc(&q, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
                ballast.push_back(q);
            }
            size_t freed = 0;
            if (rep == 0 && !std::getenv("FRAG_RANDOM_ONLY")) { for (size_t i = 0; i < ballast.size(); i += 2) { CK(hipFree(ballast[i])); ballast[i] = nullptr; ++freed; } }
            else {   // pseudo-random half
                u64 x = 0x9e3779b97f4a7c15ull + 77u * (u64)rep;
                for (size_t i = 0; i < ballast.size(); ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; if (x & 1) { CK(hipFree(ballast[i])); ballast[i] = nullptr; ++freed; } }
            }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("  %zu chunks of %zu MB taken, %zu released (%s) in %.2f s; images from the holes:", ballast.size(), chunk >> 20, freed, (rep || std::getenv("FRAG_RANDOM_ONLY")) ? "a random half" : "every other one", secs);
            std::vector<u8 *> keep;
            for (int i = 0; i < NI; ++i) { u8 *p = nullptr; if (hipMalloc(reinterpret_cast<void **>(&p), img) != hipSuccess) { (void)hipGetLastError(); break; } keep.push_back(p); std::printf(" %.2f", rate(p)); std::fflush(stdout); }
            std::printf(" TB/s\n");
            for (void *q : ballast) if (q) CK(hipFree(q));
            std::printf("    the same images after the ballast is released:");
            for (u8 *p : keep) std::printf(" %.2f", rate(p));
            std::printf(" TB/s\n");
            for (u8 *p : keep) CK(hipFree(p));
        }
        return 0;
    }
    if (argc > 4 && !std::strcmp(argv[4], "placement")) {   // the same launch into images allocated one after the other (all kept)
        const u32 l4 = (160u * 1024 / 4 - 512) & ~15u;
        std::vector<u8 *> keep;
        for (int i = 0; i < 12; ++i) {
            u8 *p = nullptr;
            if (hipMalloc(reinterpret_cast<void **>(&p), img) != hipSuccess) { (void)hipGetLastError(); break; }
            keep.push_back(p);
            CellsArgs c2 = ca; c2.out = p;
            const float ms = w == 64 ? run<64, 0>(c2, l4 > base ? l4 : base, 3) : run<32, 0>(c2, l4 > base ? l4 : base, 3);
            const float ms1 = w == 64 ? run<64, 1>(c2, l4 > base ? l4 : base, 3) : run<32, 1>(c2, l4 > base ? l4 : base, 3);
            std::printf("  image %2d at %p: full %.3f ms %.2f TB/s   stores alone %.3f ms %.2f TB/s\n", i, (void *)p, ms, gb / ms, ms1, gb / ms1); std::fflush(stdout);
        }
        if (w == 64) {
            std::printf("  stores alone, TB/s:  xcd nt 4w | xcd plain 4w | identity nt 4w | identity plain 4w | xcd nt 2w | xcd nt 6w | xcd nt 8w || full xcd nt 4w | full identity nt 4w\n");
            const u32 l2 = (160u * 1024 / 2 - 512) & ~15u, l8 = (160u * 1024 / 8 - 512) & ~15u;
            for (size_t i = 0; i < keep.size(); ++i) {
                CellsArgs c2 = ca; c2.out = keep[i];
                const float a0 = run<64, 1>(c2, l4, 3), a1 = run<64, 1 | 4>(c2, l4, 3), a2 = run<64, 1 | 512>(c2, l4, 3), a3 = run<64, 1 | 4 | 512>(c2, l4, 3),
                            a4 = run<64, 1>(c2, l2, 3), a5 = run<64, 1>(c2, base, 3), a6 = run<64, 1>(c2, l8 > base ? l8 : base, 3), f0 = run<64, 0>(c2, l4, 3), f1 = run<64, 512>(c2, l4, 3);
                std::printf("  image %2zu: %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f || %.2f | %.2f   spread (full, nt, 4w):", i, gb / a0, gb / a1, gb / a2, gb / a3, gb / a4, gb / a5, gb / a6, gb / f0, gb / f1);
                for (u32 sp : {8u, 16u, 64u, 256u, 1024u, 2048u, 4864u}) { c2.spread = sp; std::printf(" %u: %.2f", sp, gb / run<64, 1024>(c2, l4, 3)); }
                std::printf("\n"); std::fflush(stdout);
            }
        }
        return 0;
    }
    if (w == 64) {
        line("full", run<64, 0>(ca, base, R));
        line("no build (stores of the stage)", run<64, 1>(ca, base, R));
        line("no stores (build only)", run<64, 2>(ca, base, R));
        line("plain stores", run<64, 4>(ca, base, R));
        line("no build, plain stores", run<64, 4 | 1>(ca, base, R));
        line("no line alignment", run<64, 128>(ca, base, R));
        line("no line alignment, plain stores", run<64, 128 | 4>(ca, base, R));
        line("no fast paths", run<64, 64>(ca, base, R));
        line("build only, no is_equal_muled rows", run<64, 2 | 8>(ca, base, R));
        line("build only, no mul, no is_eq rows", run<64, 2 | 8 | 16>(ca, base, R));
        line("build only, no mul rows", run<64, 2 | 16>(ca, base, R));
        for (u32 per_cu : {5u, 4u, 3u, 2u}) {
            const u32 lds = (160u * 1024 / per_cu - 512) & ~15u;
            if (lds < base) continue;
            char nm[64]; std::snprintf(nm, sizeof nm, "full, %u waves per CU", per_cu); line(nm, run<64, 0>(ca, lds, R));
            std::snprintf(nm, sizeof nm, "no build, %u waves per CU", per_cu); line(nm, run<64, 1>(ca, lds, R));
        }
        line("full", run<64, 0>(ca, base, R));
    } else {
        line("full", run<32, 0>(ca, base, R));
        line("no build (stores of the stage)", run<32, 1>(ca, base, R));
        line("no stores (build only)", run<32, 2>(ca, base, R));
        for (u32 per_cu : {4u, 3u, 2u, 1u}) {
            const u32 lds = (160u * 1024 / per_cu - 512) & ~15u;
            if (lds < base) continue;
            char nm[64]; std::snprintf(nm, sizeof nm, "full, %u waves per CU", per_cu); line(nm, run<32, 0>(ca, lds, R));
        }
    }
    return 0;
}
