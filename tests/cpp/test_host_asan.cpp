// AddressSanitizer / UBSan build of the HOST side (SURVEY section 5's sanitizer build): the C oracle compiled instrumented, and
// every host-only export of libh2r.so driven through exact-size heap buffers (the library's writes into caller memory are
// memcpy / memset calls, which the sanitizer's interceptors bound-check even though libh2r.so itself is not instrumented).
// No device work: contexts are host-only (device = -1).  TEST CODE: links the oracle (checker).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <vector>

#include "h2r.h"
#include "../../oracle/h2r_oracle.h"

#define REQUIRE(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static uint64_t rng_state = 0x68327273ull;
static uint64_t next64() { uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

template <typename T> static std::unique_ptr<T[]> heap(size_t n) { return std::unique_ptr<T[]>(new T[n]()); }

static int one_shape(uint32_t w, uint32_t L, uint32_t field) {
    h2r_params prm{w, w * L, field, -1};
    h2r_ctx *ctx = nullptr;
    REQUIRE(h2r_ctx_create(&prm, &ctx) == H2R_OK);
    h2r_layout lo;
    REQUIRE(h2r_trace_layout(ctx, &lo) == H2R_OK && lo.num_limbs == L);
    h2ro_params op;
    REQUIRE(h2ro_params_init(&op, w, L) == 0 && op.mul_mod_stream_bytes == lo.stream_bytes);
    // oracle: a random mul_mod and a short pow, streams into exact-size heap buffers
    const size_t lb = w / 8;
    auto a = heap<uint8_t>(L * lb), b = heap<uint8_t>(L * lb), n = heap<uint8_t>(L * lb), r = heap<uint8_t>(L * lb);
    for (size_t i = 0; i < L * lb; ++i) { a[i] = (uint8_t)next64(); b[i] = (uint8_t)next64(); n[i] = (uint8_t)next64(); }
    n[0] |= 1; n[L * lb - 1] |= 0x80; a[L * lb - 1] &= 0x7f; b[L * lb - 1] &= 0x7f;   // a, b < n
    auto st = heap<uint8_t>(op.mul_mod_stream_bytes);
    REQUIRE(h2ro_mul_mod(&op, a.get(), b.get(), n.get(), st.get(), r.get()) == 0);
    const uint8_t e_le[1] = {0x0b};
    const uint64_t psb = h2ro_pow_fixed_stream_bytes(&op, e_le, 1);
    auto pst = heap<uint8_t>(psb); auto pout = heap<uint8_t>(L * lb), ref = heap<uint8_t>(L * lb);
    REQUIRE(h2ro_pow_mod_fixed_exp(&op, a.get(), n.get(), e_le, 1, pst.get(), pout.get()) == 0);
    REQUIRE(h2ro_big_pow_mod(&op, a.get(), e_le, 1, n.get(), ref.get()) == 0 && std::memcmp(pout.get(), ref.get(), L * lb) == 0);
    auto ist = heap<uint8_t>(h2ro_in_field_stream_bytes(&op)); int lt = -1;
    REQUIRE(h2ro_assert_in_field(&op, a.get(), n.get(), ist.get(), &lt) == 0 && lt == 1);
    // libh2r host-only exports: layouts, flatten of a record (zero planes: only the walk's bounds matter), streams
    h2r_pow_layout pl, vpl;
    REQUIRE(h2r_pow_fixed_layout(ctx, e_le, 1, &pl) == H2R_OK && pl.stream_bytes == psb);
    REQUIRE(h2r_pow_var_layout(ctx, 2, 5, &vpl) == H2R_OK);
    auto rec = heap<uint8_t>(lo.record_stride);
    for (uint32_t flags = 0; flags < 2; ++flags) {
        const uint64_t sb = h2r_stream_bytes(ctx, flags);
        auto out = heap<uint8_t>(sb);
        REQUIRE(h2r_trace_flatten_ex(ctx, rec.get(), flags, out.get()) == H2R_OK);
        auto elem = heap<uint8_t>(pl.elem_stride), pout2 = heap<uint8_t>(h2r_pow_stream_bytes(ctx, &pl, flags));
        REQUIRE(h2r_pow_trace_flatten_ex(ctx, &pl, elem.get(), flags, pout2.get()) == H2R_OK);
        auto velem = heap<uint8_t>(vpl.elem_stride), vout = heap<uint8_t>(h2r_pow_stream_bytes(ctx, &vpl, flags));
        REQUIRE(h2r_pow_trace_flatten_ex(ctx, &vpl, velem.get(), flags, vout.get()) == H2R_OK);
    }
    { auto out = heap<uint8_t>(h2r_mul_stream_bytes(ctx)); REQUIRE(h2r_mul_trace_flatten(ctx, rec.get(), out.get()) == H2R_OK); }
    { auto out = heap<uint8_t>(h2r_is_equal_muled_stream_bytes(ctx)); REQUIRE(h2r_is_equal_muled_flatten(ctx, rec.get(), out.get()) == H2R_OK); }
    for (uint32_t opk = 0; opk < H2R_OP_COUNT; ++opk) {
        uint64_t es = 0, sb = 0; uint32_t vl = 0;
        REQUIRE(h2r_fresh_op_layout(ctx, opk, &es, &sb, &vl) == H2R_OK);
        auto e = heap<uint8_t>(es), o = heap<uint8_t>(sb ? sb : 1);
        REQUIRE(h2r_fresh_op_flatten(ctx, opk, e.get(), o.get()) == H2R_OK);
    }
    if (w == 64 && L >= 9) {
        h2r_verify_layout vl;
        const uint8_t e3[3] = {1, 0, 1};
        REQUIRE(h2r_verify_layout_fixed(ctx, e3, 3, &vl) == H2R_OK);
        auto e = heap<uint8_t>(vl.elem_stride), o = heap<uint8_t>(vl.stream_bytes);
        REQUIRE(h2r_verify_trace_flatten(ctx, &vl, e.get(), o.get()) == H2R_OK);
        // the Var arm's layout and its flatten; the row programs of the verify element and of the hashed-message limbs (host-side walks)
        h2r_verify_layout vv;
        REQUIRE(h2r_verify_layout_var(ctx, 4, 5, &vv) == H2R_OK && vv.pow.num_mul_mods == 40);
        auto ve = heap<uint8_t>(vv.elem_stride), vo = heap<uint8_t>(vv.stream_bytes);
        REQUIRE(h2r_verify_trace_flatten(ctx, &vv, ve.get(), vo.get()) == H2R_OK);
        uint64_t sec[4];
        const uint64_t vr = h2r_verify_advice_rows(ctx, &vl, sec);
        REQUIRE(vr == sec[0] + sec[1] + sec[2] + sec[3]);
        auto vk = heap<uint8_t>(vr);
        REQUIRE(h2r_verify_row_kinds(ctx, &vl, vk.get()) == H2R_OK);
        // the Var element: to_bits rows of the four 5-bit limbs, acc = 1, per bit mul_mod / select / square_mod (big_integer/chip.rs:674-694)
        uint64_t vsec[4];
        const uint64_t vvr = h2r_verify_advice_rows(ctx, &vv, vsec);
        REQUIRE(vvr == vsec[0] + vsec[1] + vsec[2] + vsec[3] && vsec[2] == h2r_pow_advice_rows(ctx, &vv.pow));
        REQUIRE(vsec[2] == 4 * (5 + 2 + 1) + 2 + 20ull * (2ull * h2r_advice_rows(ctx) + L));
        auto vvk = heap<uint8_t>(vvr);
        REQUIRE(h2r_verify_row_kinds(ctx, &vv, vvk.get()) == H2R_OK);
        auto pk = heap<uint8_t>(vsec[2]);
        REQUIRE(h2r_pow_row_kinds(ctx, &vv.pow, pk.get()) == H2R_OK);
        for (uint64_t i = 0; i < 4 * 8; ++i) { h2r_fixed_row fr; REQUIRE(h2r_advice_fixed_row(ctx, nullptr, pk[i], &fr) == H2R_OK); }
        REQUIRE(pk[5] == H2R_ROW_BITS_COMPOSE && pk[6] == H2R_ROW_BITS_COMPOSE_LAST + 4 * 1 + 0 && pk[7] == H2R_ROW_ASSERT_EQ);
        const uint32_t hr = h2r_hashed_msg_advice_rows(ctx);
        REQUIRE(hr == 68);
        auto hk = heap<uint8_t>(hr);
        REQUIRE(h2r_hashed_msg_row_kinds(ctx, hk.get()) == H2R_OK);
        for (uint32_t i = 0; i < hr; ++i) { h2r_fixed_row fr; REQUIRE(h2r_advice_fixed_row(ctx, nullptr, hk[i], &fr) == H2R_OK); }
    }
    for (uint32_t opk = 0; opk < H2R_OP_COUNT; ++opk) {   // row programs of the Fresh-integer family
        const uint32_t fr_rows = h2r_fresh_op_advice_rows(ctx, opk, 0);
        REQUIRE(fr_rows > 0);
        auto fk = heap<uint8_t>(fr_rows);
        REQUIRE(h2r_fresh_op_row_kinds(ctx, opk, 0, fk.get()) == H2R_OK);
    }
    {   // the oracle's SHA-256 / hashed-message restatement through exact-size buffers (55 / 56 / 64-byte padding edges)
        for (size_t len : {0u, 1u, 55u, 56u, 63u, 64u, 65u, 119u, 120u, 128u, 200u}) {
            auto m = heap<uint8_t>(len ? len : 1), d = heap<uint8_t>(32), hs = heap<uint8_t>(h2ro_hashed_msg_stream_bytes());
            for (size_t i = 0; i < len; ++i) m[i] = (uint8_t)next64();
            uint64_t h4[4];
            h2ro_sha256(m.get(), len, d.get());
            h2ro_hashed_msg(d.get(), h4, hs.get());
            REQUIRE(h4[0] == (((uint64_t)d[24] << 56) | ((uint64_t)d[25] << 48) | ((uint64_t)d[26] << 40) | ((uint64_t)d[27] << 32) |
                              ((uint64_t)d[28] << 24) | ((uint64_t)d[29] << 16) | ((uint64_t)d[30] << 8) | d[31]));
        }
    }
    // lookup argument + advice image: host side
    h2r_lookup_config cfg;
    REQUIRE(h2r_lookup_config_default(ctx, w == 64, &cfg) == H2R_OK);
    auto tcol = heap<uint64_t>(4 * cfg.n_rows), vcol = heap<uint64_t>(4 * cfg.n_rows);
    REQUIRE(h2r_lookup_table_image(ctx, &cfg, tcol.get(), vcol.get()) == H2R_OK);
    const uint32_t rows = h2r_advice_rows(ctx);
    auto kinds = heap<uint8_t>(rows);
    REQUIRE(h2r_advice_row_kinds(ctx, kinds.get()) == H2R_OK);
    for (uint32_t i = 0; i < rows; ++i) { h2r_fixed_row fr; REQUIRE(h2r_advice_fixed_row(ctx, &cfg, kinds[i], &fr) == H2R_OK); }
    {   // the layout table is public data: NULL outputs and hand-filled non-permutations are refused, not dereferenced / indexed
        auto lay = heap<h2r_advice_layout>(1);
        REQUIRE(h2r_advice_layout_default(lay.get()) == H2R_OK);
        h2r_fixed_row fr;
        REQUIRE(h2r_advice_fixed_row_ex(ctx, &cfg, lay.get(), kinds[0], &fr) == H2R_OK);
        REQUIRE(h2r_advice_fixed_row_ex(ctx, &cfg, lay.get(), kinds[0], nullptr) == H2R_E_NULL);
        REQUIRE(h2r_advice_fixed_row_ex(nullptr, &cfg, lay.get(), kinds[0], &fr) == H2R_E_NULL);
        lay.get()->column_of[kinds[0]][1] = 9;
        REQUIRE(h2r_advice_fixed_row_ex(ctx, &cfg, lay.get(), kinds[0], &fr) == H2R_E_SHAPE);
        lay.get()->column_of[kinds[0]][1] = 0;   // a repeated column: not a permutation
        REQUIRE(h2r_advice_fixed_row_ex(ctx, &cfg, lay.get(), kinds[0], &fr) == H2R_E_SHAPE);
        // the representation of a ctx: struct_size guards the ABI, unknown flags and a stride without the columns flag are refused
        REQUIRE(h2r_abi_version() == H2R_VERSION);
        h2r_params hp{lo.limb_width, lo.limb_width * lo.num_limbs, H2R_FIELD_BN254_FR, -1};
        h2r_ctx *c2 = nullptr;
        h2r_advice_repr rp{(uint32_t)sizeof(h2r_advice_repr) - 4, 0, 0};
        REQUIRE(h2r_ctx_create_ex(&hp, &rp, &c2) == H2R_E_UNSUPPORTED && !c2);
        rp = h2r_advice_repr{(uint32_t)sizeof(h2r_advice_repr), H2R_ADVICE_MONTGOMERY, 4096};
        REQUIRE(h2r_ctx_create_ex(&hp, &rp, &c2) == H2R_E_SHAPE && !c2);
        rp = h2r_advice_repr{(uint32_t)sizeof(h2r_advice_repr), H2R_ADVICE_COLUMNS | H2R_ADVICE_MONTGOMERY, 1u << 20};
        REQUIRE(h2r_ctx_create_ex(&hp, &rp, &c2) == H2R_OK && c2);
        h2r_advice_repr got{};
        REQUIRE(h2r_ctx_advice_repr(c2, &got) == H2R_OK && got.flags == rp.flags && got.col_stride == rp.col_stride);
        uint64_t a5[4] = {5, 0, 0, 0}, m5[4], back[4];
        REQUIRE(h2r_field_eval(c2, 6, a5, nullptr, m5) == H2R_OK && h2r_field_eval(c2, 7, m5, nullptr, back) == H2R_OK && back[0] == 5 && !back[1]);
        REQUIRE(h2r_field_eval(c2, 9, a5, nullptr, back) == H2R_OK && !std::memcmp(back, m5, 32));
        REQUIRE(h2r_advice_fixed_row(c2, &cfg, kinds[0], &fr) == H2R_OK);
        REQUIRE(h2r_advice_check(c2, &cfg, nullptr, kinds.get(), rows, tcol.get(), rows * 160, 1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0,
                                 nullptr, nullptr, nullptr) == H2R_E_NULL);
        h2r_ctx_destroy(c2);
    }
    uint64_t x[4] = {5, 0, 0, 0}, y[4] = {7, 0, 0, 0}, z[4], zi[4], one[4];
    REQUIRE(h2r_field_eval(ctx, 2, x, y, z) == H2R_OK && z[0] == 35);
    REQUIRE(h2r_field_eval(ctx, 3, z, nullptr, zi) == H2R_OK && h2r_field_eval(ctx, 2, z, zi, one) == H2R_OK && one[0] == 1 && !one[1] && !one[2] && !one[3]);
    uint64_t sizes[64]; uint32_t ns = 0, paced = 0;
    REQUIRE(h2r_pipeline_call_plan(ctx, 8192, 0, sizes, 64, &ns, &paced) == H2R_OK && ns >= 1);
    uint64_t s_lo = 0, s_hi = 0;
    REQUIRE(h2r_dist_shard_range(65536, 7, 8, &s_lo, &s_hi) == H2R_OK && s_hi == 65536);
    REQUIRE(h2r_workspace_bytes(ctx, 1024, 19) > 0);
    h2r_ctx_destroy(ctx);
    return 0;
}

// ---- no C++ exception crosses the C ABI: allocation failures injected through a replaced operator new ----
// (libh2r.so's own `new` / std::vector / std::map allocations resolve to this definition: the executable comes first in symbol lookup)
static bool g_fail_new = false;
static long g_fail_after = -1;   // >= 0: that many allocations still succeed, then every one fails
static unsigned long g_new_calls = 0;
static bool fail_now() {
    ++g_new_calls;
    if (g_fail_after == 0) return true;
    if (g_fail_after > 0) --g_fail_after;
    return g_fail_new;
}
void *operator new(std::size_t n) {
    if (fail_now()) throw std::bad_alloc();
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new[](std::size_t n) { return operator new(n); }
void *operator new(std::size_t n, const std::nothrow_t &) noexcept { return fail_now() ? nullptr : std::malloc(n ? n : 1); }
void *operator new[](std::size_t n, const std::nothrow_t &) noexcept { return fail_now() ? nullptr : std::malloc(n ? n : 1); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

static int allocation_failures_stay_inside() {
    h2r_params pr;
    std::memset(&pr, 0, sizeof pr);
    pr.limb_width = 64; pr.bits_len = 2048; pr.field = H2R_FIELD_BN254_FR; pr.device = -1;
    h2r_ctx *ctx = nullptr;
    // h2r_ctx_create builds the constant record in a std::vector: with every allocation failing it must return a status, not unwind
    g_fail_new = true;
    int32_t rc = h2r_ctx_create(&pr, &ctx);
    g_fail_new = false;
    REQUIRE((rc == H2R_E_NOMEM) && ctx == nullptr);
    for (long ok_allocs = 1; ok_allocs <= 3; ++ok_allocs) {   // the ctx itself is allocated, a later std::vector is not: the exception path
        g_fail_after = ok_allocs;
        rc = h2r_ctx_create(&pr, &ctx);
        g_fail_after = -1;
        REQUIRE((rc == H2R_E_NOMEM || rc == H2R_OK) && (rc == H2R_OK) == (ctx != nullptr));
        if (ctx) { h2r_ctx_destroy(ctx); ctx = nullptr; }
    }
    REQUIRE(std::strcmp(h2r_status_str(H2R_E_NOMEM), "unknown") != 0 && std::strcmp(h2r_status_str(H2R_E_INTERNAL), "unknown") != 0);
    REQUIRE(h2r_ctx_create(&pr, &ctx) == H2R_OK && ctx);
    // the row programs of the Fresh-integer family are built on first use (std::vector / std::map / std::function inside the export)
    const unsigned long before = g_new_calls;
    g_fail_new = true;
    const uint32_t r0 = h2r_fresh_op_advice_rows(ctx, 10 /* is_in_field */, 0);   // a size query: 0 on failure
    h2r_verify_layout vl;
    const uint8_t e3[3] = {1, 0, 1};
    const int32_t lrc = h2r_verify_layout_fixed(ctx, e3, 3, &vl);   // (allocates nothing)
    const uint64_t vr = h2r_verify_advice_rows(ctx, &vl, nullptr);
    uint8_t kinds[8];
    const int32_t krc = h2r_fresh_op_row_kinds(ctx, 10, 0, kinds);
    g_fail_new = false;
    REQUIRE(g_new_calls > before);              // the library's allocations do come through here
    REQUIRE(r0 == 0 && lrc == H2R_OK && vr == 0 && krc == H2R_E_NOMEM);
    // and the same calls succeed afterwards (nothing was left half-built)
    const uint32_t r1 = h2r_fresh_op_advice_rows(ctx, 10, 0);
    REQUIRE(r1 > 1000 && h2r_verify_advice_rows(ctx, &vl, nullptr) > r1);
    h2r_ctx_destroy(ctx);
    return 0;
}

int main() {
    if (allocation_failures_stay_inside()) return 1;
    const uint32_t shapes[][3] = {{64, 32, H2R_FIELD_BN254_FR}, {32, 128, H2R_FIELD_PASTA_FP}, {64, 4, H2R_FIELD_BN254_FQ}, {64, 48, H2R_FIELD_PASTA_FQ}, {32, 8, H2R_FIELD_BN254_FR}};
    for (auto &s : shapes) if (one_shape(s[0], s[1], s[2])) return 1;
    std::printf("ASAN_HOST_OK %zu shapes\n", sizeof shapes / sizeof shapes[0]);
    return 0;
}
