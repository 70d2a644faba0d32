// C++ host-mirror test: reads like the reference's RSA tests (src/chip.rs:683-816) but runs the
// batch path on the GPU through include/h2r_chips.hpp and checks every byte against the CPU oracle.
//   test_rsa_signature_circuit1 / circuit2 : valid pkcs1v15 signatures  -> is_valid == 1
//   test_bad_rsa_signature_circuit2        : one digit off              -> is_valid == 0
// TEST CODE: links the oracle (checker).  Build: see tests/test_cpp_host.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "h2r_chips.hpp"
#include "../../oracle/h2r_oracle.h"

using namespace h2r_host;

#define REQUIRE(cond)                                                                  \
    do {                                                                               \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

struct Kat { std::string name; int is_valid; std::vector<uint64_t> n, sig, hashed; };

static std::vector<Kat> load(const char *path) {
    std::vector<Kat> out; std::ifstream f(path); std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line); Kat k; is >> k.name >> k.is_valid;
        auto rd = [&](std::vector<uint64_t> &v, int n) { for (int i = 0; i < n; ++i) { std::string h; is >> h; v.push_back(std::stoull(h, nullptr, 16)); } };
        rd(k.n, 32); rd(k.sig, 32); rd(k.hashed, 4);
        out.push_back(k);
    }
    return out;
}

int main(int argc, char **argv) {
    REQUIRE(argc == 2);
    std::vector<Kat> kats = load(argv[1]);
    REQUIRE(kats.size() == 3);
    const size_t B = kats.size();
    // configure(): RSAChip::compute_range_lens(BITS_LEN / LIMB_WIDTH)  (src/chip.rs:639-642)
    auto lens = RSAChip::compute_range_lens(2048 / RSAChip::LIMB_WIDTH);
    REQUIRE((lens.first == std::vector<uint32_t>{8, 1, 8, 4}) && (lens.second == std::vector<uint32_t>{0, 0, 6}));

    RSAChip rsa_chip(2048, 5);                       // RSAChip::new(config, BITS_LEN, EXP_LIMB_BITS)
    const BigIntChip &bigint_chip = rsa_chip.bigint_chip();
    std::vector<uint64_t> n_limbs, sign_limbs, hashed_limbs;
    for (auto &k : kats) { n_limbs.insert(n_limbs.end(), k.n.begin(), k.n.end()); sign_limbs.insert(sign_limbs.end(), k.sig.begin(), k.sig.end());
                           hashed_limbs.insert(hashed_limbs.end(), k.hashed.begin(), k.hashed.end()); }
    RSAPublicKey public_key{UnassignedInteger::from(n_limbs, B, 32), RSAPubE::fix(65537)};       // e_fix = 65537
    AssignedRSAPublicKey pk = rsa_chip.assign_public_key(public_key);
    AssignedRSASignature sign = rsa_chip.assign_signature(RSASignature{UnassignedInteger::from(sign_limbs, B, 32)});
    AssignedInteger hashed_msg_assigned = bigint_chip.assign_integer(UnassignedInteger::from(hashed_limbs, B, 4));
    VerifyResult res = rsa_chip.verify_pkcs1v15_signature(pk, hashed_msg_assigned, sign);

    h2ro_params op;
    REQUIRE(h2ro_params_init(&op, 64, 32) == 0);
    const uint8_t e_le[3] = {0x01, 0x00, 0x01};
    for (size_t i = 0; i < B; ++i) {
        REQUIRE(res.status[i] == H2R_OK);
        REQUIRE(res.is_valid[i] == kats[i].is_valid);                     // main_gate.assert_one(is_valid) / the Bad twin
        // oracle: in-field + pow + EM streams
        std::vector<uint8_t> s_if(h2ro_in_field_stream_bytes(&op)), s_pow(h2ro_pow_fixed_stream_bytes(&op, e_le, 3)), s_em(h2ro_pkcs1v15_stream_bytes(&op));
        int lt = -1, ok = -1; std::vector<uint64_t> powed(32);
        REQUIRE(h2ro_assert_in_field(&op, kats[i].sig.data(), kats[i].n.data(), s_if.data(), &lt) == 0 && lt == 1);
        REQUIRE(h2ro_pow_mod_fixed_exp(&op, kats[i].sig.data(), kats[i].n.data(), e_le, 3, s_pow.data(), powed.data()) == 0);
        REQUIRE(h2ro_pkcs1v15_em_check(&op, powed.data(), kats[i].hashed.data(), s_em.data(), &ok) == 0 && ok == kats[i].is_valid);
        std::vector<uint8_t> want(s_if); want.insert(want.end(), s_pow.begin(), s_pow.end()); want.insert(want.end(), s_em.begin(), s_em.end());
        std::vector<uint8_t> got = rsa_chip.flatten(res, i);
        REQUIRE(got == want);
        std::vector<uint64_t> gp = res.powed.limbs();
        REQUIRE(std::vector<uint64_t>(gp.begin() + 32 * i, gp.begin() + 32 * (i + 1)) == powed);
    }
    // RSASignatureVerifier::verify_pkcs1v15_signature (src/lib.rs:183-246) from the message BYTES: the three KATs sign b"hello world"
    // (src/chip.rs:713 is its SHA-256); a different message for the first signature must fail
    {
        RSASignatureVerifier verifier(rsa_chip, 128 + 64);
        const std::string hw = "hello world";
        std::vector<std::vector<uint8_t>> msgs(B, std::vector<uint8_t>(hw.begin(), hw.end()));
        SignatureVerifyResult sv = verifier.verify_pkcs1v15_signature(pk, msgs, sign);
        std::vector<uint64_t> hl(4 * B);
        sv.hashed_msg.download(hl.data(), hl.size() * 8);
        for (size_t i = 0; i < B; ++i) {
            REQUIRE(sv.verify.status[i] == H2R_OK && sv.verify.is_valid[i] == kats[i].is_valid);
            uint8_t d[32]; uint64_t h4[4]; std::vector<uint8_t> st(h2ro_hashed_msg_stream_bytes()), got(st.size());
            h2ro_sha256(reinterpret_cast<const uint8_t *>(hw.data()), hw.size(), d);
            h2ro_hashed_msg(d, h4, st.data());
            REQUIRE(std::equal(d, d + 32, sv.hashed_bytes.begin() + 32 * i));
            REQUIRE(std::equal(h4, h4 + 4, hl.begin() + 4 * i) && std::equal(h4, h4 + 4, kats[i].hashed.begin()));
            sv.hashed_msg_trace.download(got.data(), got.size(), i * H2R_HASHED_MSG_STREAM_BYTES);
            REQUIRE(got == st);
            REQUIRE(rsa_chip.flatten(sv.verify, i) == rsa_chip.flatten(res, i));
        }
        msgs[0].push_back('!'); msgs[1].clear();
        SignatureVerifyResult sv2 = verifier.verify_pkcs1v15_signature(pk, msgs, sign);
        REQUIRE(sv2.verify.is_valid[0] == 0 && sv2.verify.is_valid[1] == 0 && sv2.verify.is_valid[2] == 0);
        bool refused = false;
        msgs[2].assign(193, 0);
        try { verifier.verify_pkcs1v15_signature(pk, msgs, sign); } catch (const Error &e) { refused = e.code == H2R_E_SHAPE; }
        REQUIRE(refused);
    }
    // RSAPubE::Var arm of verify_pkcs1v15_signature (src/chip.rs:108-110): e = 65537 = 1 + 2 * 32^3 as four 5-bit limbs
    // (EXP_LIMB_BITS = 5): same verdicts and powed limbs as the Fix arm
    {
        std::vector<uint64_t> e_limbs;
        for (size_t i = 0; i < B; ++i) { const uint64_t el[4] = {1, 0, 0, 2}; e_limbs.insert(e_limbs.end(), el, el + 4); }
        RSAPublicKey pk_var{UnassignedInteger::from(n_limbs, B, 32), RSAPubE{RSAPubE::Var{UnassignedInteger::from(e_limbs, B, 4)}}};
        AssignedRSAPublicKey pkv = rsa_chip.assign_public_key(pk_var);
        VerifyResult rv = rsa_chip.verify_pkcs1v15_signature(pkv, hashed_msg_assigned, sign);
        for (size_t i = 0; i < B; ++i) REQUIRE(rv.status[i] == H2R_OK && rv.is_valid[i] == kats[i].is_valid);
        REQUIRE(rv.powed.limbs() == res.powed.limbs());
        REQUIRE(rv.layout.pow.num_mul_mods == 2 * 20);
    }
    // BigIntInstructions::mul_mod identity (n - 1) * (n - 1) mod n = 1   (big_integer/chip.rs:3204)
    {
        std::vector<uint64_t> n(kats[0].n), a(n); a[0] -= 1;   // n is odd -> no borrow
        AssignedInteger an = bigint_chip.assign_integer(UnassignedInteger::from(n, 1, 32));
        AssignedInteger aa = bigint_chip.assign_integer(UnassignedInteger::from(a, 1, 32));
        BatchResult r = bigint_chip.mul_mod(aa, aa, an);
        REQUIRE(r.status[0] == H2R_OK);
        std::vector<uint64_t> one(32, 0); one[0] = 1;
        REQUIRE(r.value.limbs() == one);
        std::vector<uint8_t> st(op.mul_mod_stream_bytes); std::vector<uint64_t> rr(32);
        REQUIRE(h2ro_mul_mod(&op, a.data(), a.data(), n.data(), st.data(), rr.data()) == 0);
        REQUIRE(r.trace.flatten(0) == st);
        // the same mul_mod as cells (what main_gate.mul_add / range_chip.assign / the is_equal_muled ops assign, chip.rs:408, :590, :598, :851-893):
        // converted from the record and written directly from the operands -- the same bytes; the copy map holds value for value
        const uint64_t rows = bigint_chip.advice_rows(r);
        REQUIRE(rows == 3974 && bigint_chip.advice_row_kinds(r).size() == rows);
        DeviceBuffer img = bigint_chip.emit_advice(r, aa, aa, an), img_d = bigint_chip.emit_advice(r, aa, aa, an, true);
        std::vector<uint8_t> ia(rows * H2R_ADVICE_ROW_BYTES), ib(ia.size());
        img.download(ia.data(), ia.size()); img_d.download(ib.data(), ib.size());
        REQUIRE(ia == ib);
        size_t inside = 0, outside = 0;
        for (const h2r_copy &c : bigint_chip.advice_copy_map()) {
            const uint8_t *cell = ia.data() + (size_t)c.row * 160 + c.col * 32;
            if (c.src_row < 0xFFFFFF00u) { REQUIRE(c.src_row < rows && !std::memcmp(cell, ia.data() + (size_t)c.src_row * 160 + c.src_col * 32, 32)); ++inside; }
            else {   // limb src_col of operand a / b / n
                const uint64_t want = c.src_row == H2R_COPY_SRC_N ? n[c.src_col] : a[c.src_col];
                uint64_t got[4]; std::memcpy(got, cell, 32);
                REQUIRE(got[0] == want && got[1] == 0 && got[2] == 0 && got[3] == 0);
                ++outside;
            }
        }
        REQUIRE(inside > 0 && outside == 3u * 32 * 32);   // x_j, y_{i-j} of both muls: 2 L^2 from a, b (here a = b) and L^2 from n
        // the placement as data: mul_add's (a, b) <-> (c, d) through the descriptor -- the image's MUL_ADD rows trade their column pairs, every
        // other row is untouched, the selectors follow; a table that splits the product over the gate's pairs is refused
        const std::vector<uint8_t> kinds = bigint_chip.advice_row_kinds(r);
        h2r_advice_layout lay = bigint_chip.advice_layout({(uint8_t)H2R_ROW_MUL_ADD}, {{2, 3, 0, 1, 4}});
        bigint_chip.apply_layout(lay, kinds, img_d, rows * H2R_ADVICE_ROW_BYTES, 1);
        img_d.download(ib.data(), ib.size());
        size_t moved = 0;
        for (uint64_t row = 0; row < rows; ++row) {
            const uint8_t *x = ia.data() + row * 160, *y = ib.data() + row * 160;
            if (kinds[row] == H2R_ROW_MUL_ADD) { REQUIRE(!std::memcmp(y + 64, x, 64) && !std::memcmp(y, x + 64, 64) && !std::memcmp(y + 128, x + 128, 32)); ++moved; }
            else REQUIRE(!std::memcmp(x, y, 160));
        }
        REQUIRE(moved == 2u * 32 * 32);
        const h2r_fixed_row f0 = bigint_chip.advice_fixed_row(H2R_ROW_MUL_ADD, bigint_chip.advice_layout()), f1 = bigint_chip.advice_fixed_row(H2R_ROW_MUL_ADD, lay);
        REQUIRE(f0.s_mul_ab[0] == 1 && f0.s_mul_cd[0] == 0 && f1.s_mul_ab[0] == 0 && f1.s_mul_cd[0] == 1);
        bool refused = false;
        try { bigint_chip.advice_layout({(uint8_t)H2R_ROW_MUL_ADD}, {{0, 2, 1, 3, 4}}); } catch (const Error &e) { refused = e.code != H2R_OK; }
        REQUIRE(refused);
    }
    // one verify_pkcs1v15_signature element and one modpow_public_key element as cells (src/chip.rs:128-199, :99-114)
    {
        uint64_t sec[4], sec2[2];
        const uint64_t rows = rsa_chip.advice_rows(res, sec);
        REQUIRE(sec[0] == 1 && sec[1] == 1532 && sec[2] == 75508 && sec[3] == 178 && rows == 77219);
        DeviceBuffer img = rsa_chip.emit_advice(res, pk, hashed_msg_assigned, sign), img_d = rsa_chip.emit_advice(res, pk, hashed_msg_assigned, sign, true);
        std::vector<uint8_t> ia(B * rows * H2R_ADVICE_ROW_BYTES), ib(ia.size());
        img.download(ia.data(), ia.size()); img_d.download(ib.data(), ib.size());
        REQUIRE(ia == ib);
        for (size_t i = 0; i < B; ++i) REQUIRE(ia[i * rows * 160] == 1);   // is_eq = assign_constant(1), :137
        // assert_in_field(sig, n) on its own (src/chip.rs:106) = the second section of the element
        FreshResult inf = bigint_chip.assert_in_field(sign.c, pk.n);
        REQUIRE(inf.advice_rows(true) == 1532);
        DeviceBuffer fimg = inf.emit_advice(sign.c.data(), pk.n.data(), nullptr, 0, true);
        std::vector<uint8_t> fa(B * 1532 * H2R_ADVICE_ROW_BYTES);
        fimg.download(fa.data(), fa.size());
        for (size_t i = 0; i < B; ++i) REQUIRE(!std::memcmp(fa.data() + i * 1532 * 160, ia.data() + (i * rows + 1) * 160, 1532 * 160));
        ModpowResult mp = rsa_chip.modpow_public_key(sign.c, pk);
        REQUIRE(rsa_chip.advice_rows(mp, sec2) == 1532 + 75508 && sec2[0] == 1532);
        DeviceBuffer m1 = rsa_chip.emit_advice(mp, sign.c, pk, true), m2 = rsa_chip.emit_advice(mp, sign.c, pk, false);
        std::vector<uint8_t> ma(B * (1532 + 75508) * H2R_ADVICE_ROW_BYTES), mb(ma.size());
        m1.download(ma.data(), ma.size()); m2.download(mb.data(), mb.size());
        REQUIRE(ma == mb);
        // the pow rows of both elements are the same rows (same x = sig, same n, same e)
        for (size_t i = 0; i < B; ++i)
            REQUIRE(!std::memcmp(ma.data() + (i * (1532 + 75508) + 1532) * 160, ia.data() + (i * rows + 1 + 1532) * 160, 75508ull * 160));
        // ... and the pipelined form of it: three calls over two buffer sets and two side streams, every image = the plain one
        {
            Pipeline apipe(rsa_chip, 2, 2);
            const std::vector<uint8_t> e65537 = {0x01, 0x00, 0x01};
            Pipeline::Buffers ab[2] = {apipe.make_buffers(B, e65537), apipe.make_buffers(B, e65537)};
            uint64_t es = 0;
            REQUIRE(h2r_fresh_op_layout(bigint_chip.ctx(), H2R_OP_IS_IN_FIELD, &es, nullptr, nullptr) == H2R_OK);
            const uint64_t stride = (1532 + 75508) * (uint64_t)H2R_ADVICE_ROW_BYTES;
            DeviceBuffer inf2[2] = {DeviceBuffer(B * es), DeviceBuffer(B * es)}, adv[2] = {DeviceBuffer(B * stride), DeviceBuffer(B * stride)};
            for (int k = 0; k < 3; ++k) apipe.modpow_public_key_advice(sign.c, pk, ab[k & 1], inf2[k & 1], adv[k & 1], stride);
            apipe.join();
            REQUIRE(hipDeviceSynchronize() == hipSuccess);
            for (int s2 = 0; s2 < 2; ++s2) { adv[s2].download(mb.data(), mb.size()); REQUIRE(ma == mb); }
        }
        // ... and the WHOLE verify element without records, pipelined (h2r_pipeline_verify_pkcs1v15_advice): every image = the record-based one
        {
            Pipeline vpipe(rsa_chip, 2, 2);
            const std::vector<uint8_t> e65537 = {0x01, 0x00, 0x01};
            Pipeline::Buffers vb[2] = {vpipe.make_buffers(B, e65537), vpipe.make_buffers(B, e65537)};
            const h2r_verify_layout cl = Pipeline::compact_layout(rsa_chip, pk);
            REQUIRE(cl.off_in_field == 0 && cl.elem_stride < 16384);
            const uint64_t vstride = rows * (uint64_t)H2R_ADVICE_ROW_BYTES;
            DeviceBuffer wit[2] = {DeviceBuffer(B * cl.elem_stride), DeviceBuffer(B * cl.elem_stride)};
            DeviceBuffer vadv[2] = {DeviceBuffer(B * vstride), DeviceBuffer(B * vstride)};
            for (int k = 0; k < 3; ++k) vpipe.verify_pkcs1v15_signature_advice(pk, hashed_msg_assigned, sign, vb[k & 1], wit[k & 1], vadv[k & 1], vstride);
            vpipe.join();
            REQUIRE(hipDeviceSynchronize() == hipSuccess);
            std::vector<uint8_t> va(ia.size()), valid(B);
            for (int s2 = 0; s2 < 2; ++s2) {
                vadv[s2].download(va.data(), va.size());
                REQUIRE(va == ia);
                vb[s2].is_valid.download(valid.data(), B);
                for (size_t i = 0; i < B; ++i) REQUIRE(valid[i] == kats[i].is_valid);
            }
        }
        // ... and the RSAPubE::Var arm without records (h2r_pipeline_modpow_public_key_var_advice): = the image of the call with records
        {
            std::vector<uint64_t> ev(B);
            for (size_t i = 0; i < B; ++i) ev[i] = 17 + 3 * i;
            RSAPublicKey pkv_u{UnassignedInteger::from(n_limbs, B, 32), RSAPubE{RSAPubE::Var{UnassignedInteger::from(ev, B, 1)}}};
            AssignedRSAPublicKey pkv = rsa_chip.assign_public_key(pkv_u);
            ModpowResult mv = rsa_chip.modpow_public_key(sign.c, pkv);
            uint64_t secv[2];
            const uint64_t vrows = rsa_chip.advice_rows(mv, secv);
            DeviceBuffer want = rsa_chip.emit_advice(mv, sign.c, pkv);
            std::vector<uint8_t> wa(B * vrows * H2R_ADVICE_ROW_BYTES), ga(wa.size());
            want.download(wa.data(), wa.size());
            Pipeline vp(rsa_chip, 2, 2);
            const h2r_pow_layout cl = Pipeline::compact_pow_layout(rsa_chip, 1);
            REQUIRE(cl.off_records == UINT64_MAX && cl.num_mul_mods == mv.pow.pow_layout.num_mul_mods);
            Pipeline::Buffers vb2[2] = {vp.make_buffers(B, {0x01, 0x00, 0x01}), vp.make_buffers(B, {0x01, 0x00, 0x01})};
            uint64_t es2 = 0;
            REQUIRE(h2r_fresh_op_layout(bigint_chip.ctx(), H2R_OP_IS_IN_FIELD, &es2, nullptr, nullptr) == H2R_OK);
            const uint64_t vstride = vrows * (uint64_t)H2R_ADVICE_ROW_BYTES;
            DeviceBuffer inf3[2] = {DeviceBuffer(B * es2), DeviceBuffer(B * es2)}, wit3[2] = {DeviceBuffer(B * cl.elem_stride), DeviceBuffer(B * cl.elem_stride)};
            DeviceBuffer adv3[2] = {DeviceBuffer(B * vstride), DeviceBuffer(B * vstride)};
            for (int k = 0; k < 3; ++k) vp.modpow_public_key_var_advice(sign.c, pkv, vb2[k & 1], inf3[k & 1], wit3[k & 1], adv3[k & 1], vstride);
            vp.join();
            REQUIRE(hipDeviceSynchronize() == hipSuccess);
            for (int s2 = 0; s2 < 2; ++s2) { adv3[s2].download(ga.data(), ga.size()); REQUIRE(ga == wa); }
        }
        BatchResult pw = bigint_chip.pow_mod_fixed_exp(sign.c, {0x01, 0x00, 0x01}, pk.n);
        REQUIRE(bigint_chip.advice_rows(pw) == 75508);
        DeviceBuffer p1 = bigint_chip.emit_advice(pw, pk.n), p2 = bigint_chip.emit_advice(pw, pk.n, true);
        std::vector<uint8_t> pa(B * 75508ull * 160), pb(pa.size());
        p1.download(pa.data(), pa.size()); p2.download(pb.data(), pb.size());
        REQUIRE(pa == pb);
        for (size_t i = 0; i < B; ++i) REQUIRE(!std::memcmp(pa.data() + i * 75508ull * 160, ia.data() + (i * rows + 1 + 1532) * 160, 75508ull * 160));
    }
    // pipelined verifier: three back-to-back batches over two buffer sets give the same witnesses as the batch call
    {
        Pipeline pipe(rsa_chip, 2, 1);
        const std::vector<uint8_t> e65537 = {0x01, 0x00, 0x01};
        Pipeline::Buffers bufs[2] = {pipe.make_buffers(B, e65537), pipe.make_buffers(B, e65537)};
        for (int k = 0; k < 3; ++k) pipe.verify_pkcs1v15_signature(pk, hashed_msg_assigned, sign, bufs[k & 1]);
        pipe.join();
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        for (int s = 0; s < 2; ++s) {
            std::vector<uint8_t> valid(B);
            bufs[s].is_valid.download(valid.data(), B);
            for (size_t i = 0; i < B; ++i) {
                REQUIRE(valid[i] == kats[i].is_valid);
                REQUIRE(pipe.flatten(bufs[s], i) == rsa_chip.flatten(res, i));
            }
        }
    }
    // pipelined RSASignatureVerifier from message bytes: same verdicts and element bytes as the batch call on the digests
    {
        Pipeline pipe(rsa_chip, 2, 1);
        const std::vector<uint8_t> e65537 = {0x01, 0x00, 0x01};
        Pipeline::Buffers bufs[2] = {pipe.make_buffers(B, e65537), pipe.make_buffers(B, e65537)};
        const std::string hw = "hello world";
        std::vector<uint8_t> bytes; std::vector<uint64_t> off(B + 1, 0);
        for (size_t i = 0; i < B; ++i) { bytes.insert(bytes.end(), hw.begin(), hw.end()); off[i + 1] = bytes.size(); }
        DeviceBuffer dmsg(bytes.size()), doff(off.size() * 8), digest(B * 32), hashed(B * 32);
        dmsg.upload(bytes.data(), bytes.size()); doff.upload(off.data(), off.size() * 8);
        for (int k = 0; k < 3; ++k) pipe.signature_verifier(pk, dmsg, doff, sign, bufs[k & 1], digest, hashed);
        pipe.join();
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        std::vector<uint64_t> hl(4 * B);
        hashed.download(hl.data(), hl.size() * 8);
        for (int s = 0; s < 2; ++s) {
            std::vector<uint8_t> valid(B);
            bufs[s].is_valid.download(valid.data(), B);
            for (size_t i = 0; i < B; ++i) {
                REQUIRE(valid[i] == kats[i].is_valid);
                REQUIRE(std::equal(kats[i].hashed.begin(), kats[i].hashed.end(), hl.begin() + 4 * i));
                REQUIRE(pipe.flatten(bufs[s], i) == rsa_chip.flatten(res, i));
            }
        }
    }
    // pipelined modpow_public_key: the powed limbs equal the verifier's
    {
        Pipeline pipe(rsa_chip, 2, 1);
        const std::vector<uint8_t> e65537 = {0x01, 0x00, 0x01};
        Pipeline::Buffers bufs[2] = {pipe.make_buffers(B, e65537), pipe.make_buffers(B, e65537)};
        for (int k = 0; k < 2; ++k) pipe.modpow_public_key(sign.c, pk, bufs[k]);
        pipe.join();
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        std::vector<uint64_t> got(B * 32), want = res.powed.limbs();
        bufs[1].powed.download(got.data(), got.size() * 8);
        REQUIRE(got == want);
    }
    // placement-aware trace arena: the kept regions come fastest first and are ordinary device memory
    {
        const std::vector<uint8_t> e65537 = {0x01, 0x00, 0x01};
        TraceArena arena(rsa_chip, B, e65537, /*regions*/ 2, /*candidates*/ 3);
        REQUIRE(arena.region(0) != nullptr && arena.region(1) != nullptr && arena.region(2) == nullptr);
        REQUIRE(arena.region_ms(0) > 0.0 && arena.region_ms(0) <= arena.region_ms(1));
        h2r_verify_layout vl{};
        REQUIRE(h2r_verify_layout_fixed(bigint_chip.ctx(), e65537.data(), e65537.size(), &vl) == H2R_OK);
        REQUIRE(arena.region_bytes() == B * vl.elem_stride);
        REQUIRE(hipMemset(arena.region(0), 0xff, arena.region_bytes()) == hipSuccess);
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
    }
    // RSAInstructions::modpow_public_key (src/chip.rs:99-114): assert_in_field witness, then the pow witness
    {
        ModpowResult mp = rsa_chip.modpow_public_key(sign.c, pk);
        for (size_t i = 0; i < B; ++i) {
            REQUIRE(mp.pow.status[i] == H2R_OK);
            std::vector<uint8_t> s_if(h2ro_in_field_stream_bytes(&op)), s_pow(h2ro_pow_fixed_stream_bytes(&op, e_le, 3));
            int lt = -1; std::vector<uint64_t> powed(32);
            REQUIRE(h2ro_assert_in_field(&op, kats[i].sig.data(), kats[i].n.data(), s_if.data(), &lt) == 0 && lt == 1);
            REQUIRE(h2ro_pow_mod_fixed_exp(&op, kats[i].sig.data(), kats[i].n.data(), e_le, 3, s_pow.data(), powed.data()) == 0);
            std::vector<uint8_t> want(s_if); want.insert(want.end(), s_pow.begin(), s_pow.end());
            REQUIRE(mp.flatten(i) == want);
        }
    }
    // The rest of BigIntInstructions through the mirror, in the style of the reference's tests (operands against assigned
    // constants, big_integer/chip.rs:1470-1660, 2797-2870): add / sub / comparisons / assert_*, mul vs a Muled constant,
    // refresh(mul(a, b)) = a * b
    {
        std::vector<uint64_t> n(kats[0].n), small(32, 0); small[0] = 7;
        AssignedInteger an = bigint_chip.assign_integer(UnassignedInteger::from(n, 1, 32));
        AssignedInteger seven = bigint_chip.assign_constant_fresh({7});
        AssignedInteger five = bigint_chip.assign_constant_fresh({5});
        REQUIRE(seven.limbs() == small);
        FreshResult s = bigint_chip.add(seven, five);
        REQUIRE(s.status[0] == H2R_OK && s.limbs()[0] == 12 && s.value_limbs == 33);
        std::vector<uint8_t> st(h2ro_fresh_op_stream_bytes(&op, H2RO_OP_ADD)); std::vector<uint64_t> vout(40); uint32_t nv = 0; int fl = -1;
        std::vector<uint64_t> five_l(32, 0); five_l[0] = 5;
        REQUIRE(h2ro_fresh_op(&op, H2RO_OP_ADD, small.data(), five_l.data(), nullptr, st.data(), vout.data(), &nv, &fl) == 0);
        REQUIRE(s.flatten(0) == st);
        REQUIRE(bigint_chip.is_less_than(five, seven).flag[0] == 1 && bigint_chip.is_less_than(seven, five).flag[0] == 0);
        REQUIRE(bigint_chip.assert_less_than(five, seven).status[0] == H2R_OK);
        REQUIRE(bigint_chip.assert_less_than(seven, five).status[0] == H2R_E_ASSERTION);
        REQUIRE(bigint_chip.assert_in_field(seven, an).status[0] == H2R_OK);
        REQUIRE(bigint_chip.assert_equal_fresh(seven, five).status[0] == H2R_E_ASSERTION);
        REQUIRE(bigint_chip.assert_zero(bigint_chip.assign_constant_fresh({})).status[0] == H2R_OK);
        REQUIRE(bigint_chip.sub_mod(five, seven, an).status[0] == H2R_OK);
        std::vector<uint64_t> mx = bigint_chip.max_value(32).limbs();
        REQUIRE(mx == std::vector<uint64_t>(32, ~0ull));
        MuledInteger prod = bigint_chip.mul(seven, five);
        REQUIRE(bigint_chip.assert_equal_muled(prod, bigint_chip.assign_constant_muled({35}, 32, 32))[0] == H2R_OK);
        REQUIRE(bigint_chip.assert_equal_muled(prod, bigint_chip.assign_constant_muled({36}, 32, 32))[0] == H2R_E_ASSERTION);
        auto fr = bigint_chip.refresh(bigint_chip.square(an));
        REQUIRE(fr.second[0] == H2R_OK && fr.first.num_limbs() == 64);
    }
    // BigIntChip::new asserts bits_len % limb_width == 0 (big_integer/chip.rs:1175) -> exception
    bool threw = false;
    try { BigIntChip bad(64, 2048 + 8); } catch (const Error &e) { threw = e.code == H2R_E_SHAPE; }
    REQUIRE(threw);
    // operands of different lengths: mul(d0 = 3, d1 = 32), refresh with RefreshAux::new(64, 3, 32), is_equal_muled(3, 32)
    {
        AssignedInteger three = bigint_chip.assign_integer(UnassignedInteger::from({7, 1, 2}, 1, 3));
        AssignedInteger nn = bigint_chip.assign_integer(UnassignedInteger::from(kats[0].n, 1, 32));
        MuledInteger p1 = bigint_chip.mul_general(three, nn), p2 = bigint_chip.mul_general(three, nn);
        REQUIRE(bigint_chip.is_equal_muled_general(p1, p2, 3, 32)[0] == 1);
        auto fr = bigint_chip.refresh_general(p1, 3, 32);
        REQUIRE(fr.second[0] == H2R_OK && fr.first.num_limbs() == 35);
        // (7 + 2^64 + 2 * 2^128) * n, limb 0 = 7 * n[0] mod 2^64
        REQUIRE(fr.first.limbs()[0] == 7 * kats[0].n[0]);
    }
    // halo2's lookup argument for the range checks of these three circuits: table, multiplicities, permuted columns
    {
        h2r_lookup_config cfg;
        REQUIRE(h2r_lookup_config_default(bigint_chip.ctx(), 1, &cfg) == H2R_OK && cfg.n_rows == 339);   // bit lengths 1, 4, 6, 8
        std::vector<uint64_t> tag_col(4 * cfg.n_rows), val_col(4 * cfg.n_rows);
        REQUIRE(h2r_lookup_table_image(bigint_chip.ctx(), &cfg, tag_col.data(), val_col.data()) == H2R_OK);
        REQUIRE(tag_col[0] == 0 && val_col[0] == 0 && tag_col[4 * 338] == 4 && val_col[4 * 338] == 255);
        const uint32_t usable = (1u << 17) - 6;
        DeviceBuffer hist(B * 5 * cfg.n_rows * 4), theta(B * 32), a_perm(B * 5ull * usable * 32), s_perm(B * 5ull * usable * 32), lst(B),
            lws(h2r_lookup_workspace_bytes(&cfg, B));
        hip_check(hipMemset(hist.get(), 0, hist.size()), "hipMemset");
        REQUIRE(h2r_lookup_hist_records(bigint_chip.ctx(), &cfg, res.trace.get(), res.layout.pow.off_records, res.layout.elem_stride, B,
                                        res.layout.pow.num_mul_mods, nullptr, static_cast<uint32_t *>(hist.get()), nullptr) == H2R_OK);
        std::vector<uint64_t> th(B * 4, 0);
        for (size_t i = 0; i < B; ++i) { th[4 * i] = 0x1234567 + i; th[4 * i + 2] = 99; }
        theta.upload(th.data(), th.size() * 8);
        REQUIRE(h2r_lookup_permuted_columns(bigint_chip.ctx(), &cfg, static_cast<const uint32_t *>(hist.get()), static_cast<const uint64_t *>(theta.get()), B,
                                            usable, 31, a_perm.get(), s_perm.get(), 5ull * usable * 32, static_cast<uint8_t *>(lst.get()),
                                            lws.get(), nullptr) == H2R_OK);
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        // the lookup argument's rule on circuit 1 (BAD: no pow trace -> all-zero columns), and on circuit 0, argument composition_a
        std::vector<uint64_t> A(4ull * usable), S(4ull * usable);
        a_perm.download(A.data(), A.size() * 8); s_perm.download(S.data(), S.size() * 8);
        size_t heads = 0;
        for (uint32_t r = 0; r < usable; ++r) {
            const bool eq_s = std::equal(A.begin() + 4 * r, A.begin() + 4 * r + 4, S.begin() + 4 * r);
            const bool eq_p = r && std::equal(A.begin() + 4 * r, A.begin() + 4 * r + 4, A.begin() + 4 * (r - 1));
            REQUIRE(eq_s || eq_p);
            heads += !eq_p;
        }
        REQUIRE(heads > 200 && heads <= 339);   // the zero run + every looked-up 8-bit value
    }
    // The multi-GPU exports over RCCL (h2r_dist_*, SURVEY section 2 component C1), as a Rust host would call them: a ONE-rank
    // communicator on this box's single GPU -- id, init, shard ranges, parameter broadcast, result all-gather, MAX, barrier.
    {
        uint8_t id[H2R_DIST_ID_BYTES];
        REQUIRE(h2r_dist_unique_id(id) == H2R_OK);
        h2r_dist *d = nullptr;
        REQUIRE(h2r_dist_init(bigint_chip.ctx(), id, 1, 1, &d) == H2R_E_SHAPE);   // rank >= world
        REQUIRE(h2r_dist_init(bigint_chip.ctx(), id, 0, 1, &d) == H2R_OK && d);
        REQUIRE(h2r_dist_rank(d) == 0 && h2r_dist_world(d) == 1);
        uint64_t lo = 0, hi = 0;
        REQUIRE(h2r_dist_shard_range(65536, 3, 8, &lo, &hi) == H2R_OK && lo == 3 * 8192 && hi == 4 * 8192);   // BASELINE config 3
        REQUIRE(h2r_dist_shard_range(10, 2, 3, &lo, &hi) == H2R_OK && lo == 7 && hi == 10);
        std::vector<uint64_t> cfg = {65537, 1024, 4}, back(3, 0);
        DeviceBuffer cb(sizeof(uint64_t) * 3), all(res.powed.limbs().size() * 8), st_all(B);
        cb.upload(cfg.data(), sizeof(uint64_t) * 3);
        DeviceBuffer st_dev(B);
        st_dev.upload(res.status.data(), B);
        REQUIRE(h2r_dist_bcast(d, cb.get(), sizeof(uint64_t) * 3, 0, nullptr) == H2R_OK);
        REQUIRE(h2r_dist_gather_results(d, res.powed.data(), static_cast<const uint8_t *>(st_dev.get()), B, all.get(), static_cast<uint8_t *>(st_all.get()), nullptr) == H2R_OK);
        DeviceBuffer tm(sizeof(double));
        double t = 1.25;
        tm.upload(&t, sizeof t);
        REQUIRE(h2r_dist_allreduce_max_f64(d, static_cast<double *>(tm.get()), 1, nullptr) == H2R_OK);
        REQUIRE(h2r_dist_allreduce_max_f64(d, nullptr, 0, nullptr) == H2R_OK);    // barrier
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        cb.download(back.data(), sizeof(uint64_t) * 3);
        REQUIRE(back == cfg);
        std::vector<uint64_t> gathered(res.powed.limbs().size());
        all.download(gathered.data(), gathered.size() * 8);
        REQUIRE(gathered == res.powed.limbs());
        std::vector<uint8_t> gst(B);
        st_all.download(gst.data(), B);
        REQUIRE(gst == res.status);
        tm.download(&t, sizeof t);
        REQUIRE(t == 1.25);
        h2r_dist_destroy(d);
    }
    std::printf("CPP_HOST_MIRROR_OK %zu signatures\n", B);
    return 0;
}
