// Developer bench of cells_kernel's REPRESENTATIONS (h2r_cells.hpp: planar columns, Montgomery cells): per-path chunk cycles and
// ablations at BASELINE config 2's size without rebuilding the library (derived from tools/cells_bench.hip).  Operands come from a real h2r_pow_mod_fixed_exp_batch call (no records), the row table and the constants' table are
// rebuilt here the way h2r_ctx_create builds them.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ihalo2_rsa_amd/csrc tools/cells_repr_bench.hip -o /tmp/cells_repr_bench -Lhalo2_rsa_amd/lib -lh2r -Wl,-rpath,$PWD/halo2_rsa_amd/lib
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

template <int LW, int ABL, bool MONT = false>
float run(const CellsArgs &ca, u32 lds, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cells_kernel<LW, ABL, MONT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((cells_kernel<LW, ABL, MONT>), dim3((unsigned)ca.n_items), dim3(64), lds, 0, ca);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((cells_kernel<LW, ABL, MONT>), dim3((unsigned)ca.n_items), dim3(64), lds, 0, ca);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const u32 w = argc > 1 ? (u32)std::atoi(argv[1]) : 64, bits = argc > 2 ? (u32)std::atoi(argv[2]) : 2048;
    const u64 batch = argc > 3 ? (u64)std::atoll(argv[3]) : 1024;
    const bool mont = argc > 4 && std::atoi(argv[4]) & 1, planar = argc > 4 && std::atoi(argv[4]) & 2;
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
        const CellsLds lp = cells_lds_plan(w, L, mont);
        u32 *fs = reinterpret_cast<u32 *>(&kt[CELLS_KT_FSRC]);
        for (u32 k = 0; k < ADVICE_COL_ROWS * 3; ++k) {
            fs[k] = cells_pack_fast_src(lp, w, cells_fast_src(k / 3, k % 3, false), mont);
            fs[CELLS_SRC_WORDS + k] = cells_pack_fast_src(lp, w, cells_fast_src(k / 3, k % 3, true), mont);
        }
    }
    u64 *dkt; u8 *dimg;
    const u64 out_stride = (2ull + (u64)T * rows) * 160, img = batch * out_stride;
    CK(hipMalloc(reinterpret_cast<void **>(&dkt), sizeof kt)); CK(hipMalloc(reinterpret_cast<void **>(&dimg), img));
    CK(hipMemcpy(dkt, kt, sizeof kt, hipMemcpyHostToDevice));
    CellsArgs ca; std::memset(&ca, 0, sizeof ca);
    const u8 *ws = reinterpret_cast<const u8 *>(round_up(reinterpret_cast<u64>(dws), 256));
    ca.ktab = dkt; ca.per_col_magic = (u32)(((1ull << 32) + (ADVICE_COL_ROWS + nrc) - 1) / (ADVICE_COL_ROWS + nrc)); ca.opA = ws; ca.opB = ws + L * lb; ca.opQ = ws + 2 * L * lb; ca.opR = ws + 3 * L * lb; ca.op_stride = ca.qr_stride = 4ull * L;
    ca.n = dn; ca.n_stride = L; ca.status = dst; ca.T = T; ca.n_items = batch * T; ca.dst.base = dimg; ca.dst.elem_stride = out_stride; ca.dst.mont = mont; ca.dst.row_pitch = planar ? 32 : 160; ca.dst.col_pitch = planar ? out_stride / 5 : 32;
    { u64 p[4]; field_modulus(pr.field, p); MontK hmk; montk_init(p, &hmk); MontK *dmk; CK(hipMalloc(reinterpret_cast<void **>(&dmk), sizeof hmk)); CK(hipMemcpy(dmk, &hmk, sizeof hmk, hipMemcpyHostToDevice)); ca.mk = dmk; }
    ca.rows = rows; ca.pre_rows = 2; ca.L = L; ca.carry_sub_bits = lo.carry_sub_bits; ca.carry_nsub = lo.carry_nsub;
    const u32 base = cells_lds_bytes(w, L, mont);
    const double gb = (double)img / 1e9;
    std::printf("w=%u L=%u batch=%llu T=%u rows=%u image %.2f GB, lds/wave %u\n", w, L, (unsigned long long)batch, T, rows, gb, base);
    auto line = [&](const char *nm, float ms) { std::printf("  %-34s %.3f ms  %.2f TB/s\n", nm, ms, gb / ms); std::fflush(stdout); };
    const int R = 5;

    const u32 wpc = argc > 6 ? (u32)std::atoi(argv[6]) : 4;   // waves per CU (by the LDS request)
    const u32 l4 = argc > 7 ? (u32)std::atoi(argv[7]) : (mont && argc <= 6 ? 0u : (160u * 1024 / wpc - 512) & ~15u), lds = l4 > base ? l4 : base;   // argv[7]: the raw LDS request; Montgomery: whatever fits
    std::printf("representation: %s%s, lds request %u\n", mont ? "montgomery " : "canonical ", planar ? "planar" : "row-major", lds);
#define RUN(ABL_) (w == 64 ? (mont ? run<64, ABL_, true>(ca, lds, R) : run<64, ABL_, false>(ca, lds, R)) : (mont ? run<32, ABL_, true>(ca, lds, R) : run<32, ABL_, false>(ca, lds, R)))
    {   // per-chunk cycle stamps of one wave (item in the middle of the grid), next to the full grid and alone
        u64 *ddbg; CK(hipMalloc(reinterpret_cast<void **>(&ddbg), 3000 * 8));
        for (int alone = 0; alone < 2; ++alone) {
            CK(hipMemset(ddbg, 0, 3000 * 8));
            CellsArgs cd = ca; cd.dbg = ddbg; cd.dbg_item = alone ? 0 : (u32)(ca.n_items / 2);
            if (alone) cd.n_items = 1;
            if (w == 64) { if (mont) run<64, 256, true>(cd, lds, 1); else run<64, 256, false>(cd, lds, 1); } else { if (mont) run<32, 256, true>(cd, lds, 1); else run<32, 256, false>(cd, lds, 1); }
            std::vector<u64> h(3000); CK(hipMemcpy(h.data(), ddbg, 3000 * 8, hipMemcpyDeviceToHost));
            u64 sb[3] = {0, 0, 0}, sr[3] = {0, 0, 0}, n[3] = {0, 0, 0};
            const u32 nchunks = (rows + 63) / 64 + 1;
            for (u32 c = 0; c < nchunks && c < 1000; ++c) { const u64 pth = h[3 * c] < 3 ? h[3 * c] : 0; sb[pth] += h[3 * c + 1]; sr[pth] += h[3 * c + 2]; ++n[pth]; }
            std::printf("  cycles per chunk (%s): general %llu build + %llu readout (%llu chunks); mul rows %llu + %llu (%llu); column rows %llu + %llu (%llu); item total %llu\n",
                        alone ? "one wave alone" : "full grid", n[0] ? sb[0] / n[0] : 0, n[0] ? sr[0] / n[0] : 0, n[0], n[1] ? sb[1] / n[1] : 0, n[1] ? sr[1] / n[1] : 0, n[1],
                        n[2] ? sb[2] / n[2] : 0, n[2] ? sr[2] / n[2] : 0, n[2], sb[0] + sb[1] + sb[2] + sr[0] + sr[1] + sr[2]);
            if (alone) { std::printf("  general chunks:"); for (u32 c = 0; c < nchunks && c < 1000; ++c) if (h[3 * c] == 0 && (h[3 * c + 1] | h[3 * c + 2])) std::printf(" [%u] %llu", c, h[3 * c + 1]); std::printf("\n"); }
        }
        CK(hipFree(ddbg));
    }
    if (mont && argc > 5) {   // developer: what the mul rows' chunk is made of (cycle stamps of one wave alone)
        u64 *ddbg; CK(hipMalloc(reinterpret_cast<void **>(&ddbg), 3000 * 8));
        auto stamp = [&](const char *nm, auto abl) {
            constexpr int A = decltype(abl)::value;
            CK(hipMemset(ddbg, 0, 3000 * 8));
            CellsArgs cd = ca; cd.dbg = ddbg; cd.dbg_item = 0; cd.n_items = 1;
            if (w == 64) run<64, 256 | A, true>(cd, lds, 1); else run<32, 256 | A, true>(cd, lds, 1);
            std::vector<u64> h(3000); CK(hipMemcpy(h.data(), ddbg, 3000 * 8, hipMemcpyDeviceToHost));
            u64 sb = 0, n = 0; const u32 nchunks = (rows + 63) / 64 + 1;
            for (u32 c = 0; c < nchunks && c < 1000; ++c) if (h[3 * c] == 1) { sb += h[3 * c + 1]; ++n; }
            std::printf("  mul-row chunk build, %-34s %llu cycles\n", nm, (unsigned long long)(n ? sb / n : 0));
        };
        stamp("as shipped", std::integral_constant<int, 0>{});
        stamp("no conversion", std::integral_constant<int, 2048>{});
        CK(hipFree(ddbg));
    }
    line("full", RUN(0));
    line("no stores (build only)", RUN(2));
    line("no build (stores of the stage)", RUN(1));
    line("build only, no is_equal_muled rows", RUN(2 | 8));
    line("build only, no mul rows", RUN(2 | 16));
    line("full", RUN(0));
    if (mont) {   // [r6] ceilings of any faster conversion (wrong cells, timing only): the mul rows' conversions free / EVERY conversion free
        line("mul-row conversions free", RUN(2048));
        line("mul-row conversions free, build only", RUN(2 | 2048));
        line("every conversion free", RUN(16384));
        line("every conversion free, build only", RUN(2 | 16384));
        line("full", RUN(0));
    }
    if (argc > 8) {   // the SAME image buffer under several LDS requests (= waves per CU): 6, 5, 4, 3 per CU and the shipped one again
        for (u32 req : {26880u, 32000u, 40448u, 53760u, lds}) {
            if (req < base) continue;
            const u32 keep = lds;
            const float ms = w == 64 ? (mont ? run<64, 0, true>(ca, req, R) : run<64, 0, false>(ca, req, R)) : (mont ? run<32, 0, true>(ca, req, R) : run<32, 0, false>(ca, req, R));
            char nm[64]; std::snprintf(nm, sizeof nm, "full, lds request %u", req); line(nm, ms); (void)keep;
        }
    }
    return 0;
}
