// h2r_chips.hpp -- header-only C++17 host mirror of the reference's chip API over the C ABI (h2r.h).
//
// The reference is a Rust crate; its API for the accelerated path is
//   BigIntChip / BigIntInstructions<F>   reference src/big_integer/chip.rs:42-51, instructions.rs:7-260
//   RSAChip / RSAInstructions<F>         reference src/chip.rs:38-255, src/instructions.rs:8-39
//   RSAPublicKey / RSAPubE / RSASignature  reference src/lib.rs:25-140
// This header keeps those names, argument order and error behaviour (a C++ exception where the
// reference panics or returns plonk::Error at construction; a per-element status byte where it would
// panic on a value) in BATCH form: every integer is a batch of integers, one per independent circuit,
// resident in HBM.  All arithmetic happens in libh2r.so; there is no CPU fallback here.
//
// Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude ... -lh2r -lamdhip64
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <array>
#include <variant>
#include <vector>

#include "h2r.h"

namespace h2r_host {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string &where)
        : std::runtime_error(where + ": " + h2r_status_str(c) + (c == H2R_E_HIP ? std::string(" (") + h2r_last_hip_error() + ")" : "")), code(c) {}
};
inline void check(int32_t rc, const char *where) { if (rc != H2R_OK) throw Error(rc, where); }
inline void hip_check(hipError_t e, const char *where) { if (e != hipSuccess) throw std::runtime_error(std::string(where) + ": " + hipGetErrorString(e)); }

// RAII device allocation
class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) : n_(bytes) { if (bytes) hip_check(hipMalloc(&p_, bytes), "hipMalloc"); }
    DeviceBuffer(DeviceBuffer &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void *get() const { return p_; }
    size_t size() const { return n_; }
    void upload(const void *src, size_t bytes) { hip_check(hipMemcpy(p_, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
    void download(void *dst, size_t bytes, size_t offset = 0) const { hip_check(hipMemcpy(dst, static_cast<const uint8_t *>(p_) + offset, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
  private:
    void release() { if (p_) (void)hipFree(p_); p_ = nullptr; }
    void *p_ = nullptr; size_t n_ = 0;
};

// reference src/big_integer/mod.rs:270-302 -- limb values about to be assigned (host side, 64-bit limbs)
struct UnassignedInteger {
    std::vector<uint64_t> limbs;  // [batch][num_limbs], limb 0 least significant (decompose_big)
    size_t batch = 0, num_limbs = 0;
    static UnassignedInteger from(std::vector<uint64_t> l, size_t batch, size_t num_limbs) {
        if (l.size() != batch * num_limbs) throw std::invalid_argument("UnassignedInteger: shape");
        return UnassignedInteger{std::move(l), batch, num_limbs};
    }
};

// reference src/big_integer/mod.rs:306-382 (range type Fresh): a batch of limb vectors in HBM
class AssignedInteger {
  public:
    AssignedInteger(DeviceBuffer d, size_t batch, size_t num_limbs) : dev_(std::move(d)), batch_(batch), num_limbs_(num_limbs) {}
    size_t num_limbs() const { return num_limbs_; }
    size_t batch() const { return batch_; }
    const void *data() const { return dev_.get(); }
    std::vector<uint64_t> limbs() const { std::vector<uint64_t> h(batch_ * num_limbs_); dev_.download(h.data(), h.size() * 8); return h; }
  private:
    DeviceBuffer dev_; size_t batch_, num_limbs_;
};

class BigIntChip;

// witness records of one batch call + what is needed to walk them in the reference's order
class Trace {
  public:
    Trace(const BigIntChip *chip, DeviceBuffer buf, size_t batch, bool is_pow, h2r_pow_layout pl, uint64_t elem_stride, uint64_t stream_bytes)
        : chip_(chip), buf_(std::move(buf)), batch_(batch), is_pow_(is_pow), pl_(pl), elem_stride_(elem_stride), stream_bytes_(stream_bytes) {}
    uint64_t stream_bytes() const { return stream_bytes_; }
    const void *data() const { return buf_.get(); }
    std::vector<uint8_t> flatten(size_t elem) const;  // the element's op-trace in assignment order
  private:
    const BigIntChip *chip_; DeviceBuffer buf_; size_t batch_; bool is_pow_; h2r_pow_layout pl_; uint64_t elem_stride_, stream_bytes_;
};

struct BatchResult {
    AssignedInteger value;        // a*b mod n  /  a^e mod n
    Trace trace;
    std::vector<uint8_t> status;  // H2R_* per element
    // kept for the advice image (BigIntChip::emit_advice): the per-element status on the device, a pow call's workspace (it holds every
    // mul_mod's operands) and layout
    DeviceBuffer status_dev, workspace;
    h2r_pow_layout pow_layout{};
    bool is_pow = false;
};

// result of a Fresh-integer op (add / sub / add_mod / sub_mod / comparisons): value where the op has one, the
// predicate / overflow bit, the per-element status and the op's witness (flat stream via flatten())
struct FreshResult {
    DeviceBuffer value; size_t value_limbs = 0;
    std::vector<uint8_t> flag, status;
    DeviceBuffer trace; uint64_t elem_stride = 0, stream_bytes = 0; uint32_t op = 0; const h2r_ctx *ctx = nullptr; size_t batch = 0;
    std::vector<uint64_t> limbs() const { std::vector<uint64_t> h(batch * value_limbs); if (!h.empty()) value.download(h.data(), h.size() * 8); return h; }
    std::vector<uint8_t> flatten(size_t elem) const {
        std::vector<uint8_t> host(elem_stride), out(stream_bytes);
        trace.download(host.data(), host.size(), elem * elem_stride);
        check(h2r_fresh_op_flatten(ctx, op, host.data(), out.data()), "h2r_fresh_op_flatten");
        return out;
    }
    // the op as cells (rows of the main gate's five advice columns); a, b, n: the call's operands (b / n null where the op has none),
    // shared_flags: H2R_F_SHARED_MODULUS as in the call; assert_form: the assert_* twin (main_gate.assert_one at the end)
    uint64_t advice_rows(bool assert_form = false) const { return h2r_fresh_op_advice_rows(ctx, op, assert_form ? H2R_ADVICE_ASSERT_ONE : 0u); }
    DeviceBuffer emit_advice(const void *a, const void *b, const void *n, uint32_t shared_flags = 0, bool assert_form = false) const {
        const uint64_t stride = advice_rows(assert_form) * H2R_ADVICE_ROW_BYTES;
        DeviceBuffer out(batch * stride);
        check(h2r_fresh_op_emit_advice(ctx, op, shared_flags | (assert_form ? H2R_ADVICE_ASSERT_ONE : 0u), a, b, n, trace.get(), 0, elem_stride, batch,
                                       nullptr, out.get(), stride, nullptr), "h2r_fresh_op_emit_advice");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return out;
    }
};

// reference src/big_integer/mod.rs:216-232, 306-382 (range type Muled): the un-carried product columns, 2L columns of
// 4 x u64 per element (column 2L-1 is zero), plus the accumulator trace of the mul that produced them (if any)
struct MuledInteger {
    DeviceBuffer cols; size_t batch = 0, num_limbs = 0;   // num_limbs = L of the operands
    DeviceBuffer trace;                                   // one record per element (AB planes written), may be empty
};

// reference src/big_integer/chip.rs:42-51, 1161-1249
class BigIntChip {
  public:
    static constexpr unsigned NUM_LOOKUP_LIMBS = 8;  // big_integer/chip.rs:1163
    // BigIntChip::new(config, limb_width, bits_len), big_integer/chip.rs:1174-1185
    BigIntChip(uint32_t limb_width, uint32_t bits_len, uint32_t field = H2R_FIELD_BN254_FR, int device = 0)
        : limb_width_(limb_width), num_limbs_(limb_width ? bits_len / limb_width : 0) {
        h2r_params p{limb_width, bits_len, field, device};
        check(h2r_ctx_create(&p, &ctx_), "BigIntChip::new");
        check(h2r_trace_layout(ctx_, &layout_), "h2r_trace_layout");
    }
    ~BigIntChip() { h2r_ctx_destroy(ctx_); }
    BigIntChip(const BigIntChip &) = delete;
    BigIntChip &operator=(const BigIntChip &) = delete;

    // big_integer/chip.rs:1220-1249
    static std::pair<std::vector<uint32_t>, std::vector<uint32_t>> compute_range_lens(uint32_t limb_width, uint32_t num_limbs) {
        std::vector<uint32_t> c(3), o(3);
        check(h2r_compute_range_lens(limb_width, num_limbs, c.data(), o.data()), "compute_range_lens");
        return {c, o};
    }
    uint32_t num_limbs() const { return num_limbs_; }
    const h2r_ctx *ctx() const { return ctx_; }
    const h2r_layout &layout() const { return layout_; }

    // big_integer/chip.rs:62-82
    AssignedInteger assign_integer(const UnassignedInteger &integer) const {
        DeviceBuffer d(integer.limbs.size() * 8);
        d.upload(integer.limbs.data(), integer.limbs.size() * 8);
        return AssignedInteger(std::move(d), integer.batch, integer.num_limbs);
    }
    // big_integer/chip.rs:1252-1281 (assign_constant): `limbs` = the constant's little-endian 64-bit limbs (at most
    // num_limbs of them: the reference asserts that the value fits, :1266); the same constant for every batch element
    AssignedInteger assign_constant(const std::vector<uint64_t> &limbs, size_t num_limbs, size_t batch = 1) const {
        if (limbs.size() > num_limbs) throw Error(H2R_E_SHAPE, "assign_constant");
        std::vector<uint64_t> h(batch * num_limbs, 0);
        for (size_t e = 0; e < batch; ++e) std::copy(limbs.begin(), limbs.end(), h.begin() + e * num_limbs);
        return assign_integer(UnassignedInteger::from(std::move(h), batch, num_limbs));
    }
    // instructions.rs:16 / big_integer/chip.rs:95-101
    AssignedInteger assign_constant_fresh(const std::vector<uint64_t> &limbs, size_t batch = 1) const { return assign_constant(limbs, num_limbs_, batch); }
    // instructions.rs:23 / big_integer/chip.rs:119-127: n_l + n_r - 1 limbs of limb_width bits, as Muled columns
    MuledInteger assign_constant_muled(const std::vector<uint64_t> &limbs, size_t num_limbs_l, size_t num_limbs_r, size_t batch = 1) const {
        if (num_limbs_l != num_limbs_ || num_limbs_r != num_limbs_ || limbs.size() > 2 * num_limbs_ - 1) throw Error(H2R_E_SHAPE, "assign_constant_muled");
        std::vector<uint64_t> h(batch * 2 * num_limbs_ * 4, 0);
        for (size_t e = 0; e < batch; ++e)
            for (size_t i = 0; i < limbs.size(); ++i) h[(e * 2 * num_limbs_ + i) * 4] = limbs[i];
        MuledInteger m; m.cols = DeviceBuffer(h.size() * 8); m.cols.upload(h.data(), h.size() * 8); m.batch = batch; m.num_limbs = num_limbs_;
        return m;
    }
    // instructions.rs:32 / big_integer/chip.rs:138-154: every limb = 2^limb_width - 1
    AssignedInteger max_value(size_t num_limbs, size_t batch = 1) const { return assign_constant(std::vector<uint64_t>(num_limbs, ~0ull), num_limbs, batch); }

    // ---- the Fresh-integer family (instructions.rs:47-60, 78-95, 132-145, 157-195) ----------------------------------------
    FreshResult add(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_ADD, a, &b, nullptr); }                    // chip.rs:245-297
    FreshResult sub(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_SUB, a, &b, nullptr); }                    // chip.rs:310-373
    FreshResult add_mod(const AssignedInteger &a, const AssignedInteger &b, const AssignedInteger &n) const { return fresh(H2R_OP_ADD_MOD, a, &b, &n); }   // :452-481
    FreshResult sub_mod(const AssignedInteger &a, const AssignedInteger &b, const AssignedInteger &n) const { return fresh(H2R_OP_SUB_MOD, a, &b, &n); }   // :495-528
    FreshResult is_zero(const AssignedInteger &a) const { return fresh(H2R_OP_IS_ZERO, a, nullptr, nullptr); }                                 // chip.rs:754-767
    FreshResult is_equal_fresh(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_IS_EQUAL_FRESH, a, &b, nullptr); }   // :780-805
    FreshResult is_less_than(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_IS_LESS_THAN, a, &b, nullptr); }       // :908-919
    FreshResult is_less_than_or_equal(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_IS_LESS_THAN_OR_EQUAL, a, &b, nullptr); }
    FreshResult is_greater_than(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_IS_GREATER_THAN, a, &b, nullptr); }
    FreshResult is_greater_than_or_equal(const AssignedInteger &a, const AssignedInteger &b) const { return fresh(H2R_OP_IS_GREATER_THAN_OR_EQUAL, a, &b, nullptr); }
    FreshResult is_in_field(const AssignedInteger &a, const AssignedInteger &n) const { return fresh(H2R_OP_IS_IN_FIELD, a, &n, nullptr); }         // :998-1006
    // assert_* (instructions.rs:197-254, chip.rs:1020-1158): the predicate, then main_gate.assert_one on its bit; an
    // element whose assertion does not hold gets status H2R_E_ASSERTION (the reference's circuit is unsatisfiable)
    FreshResult assert_zero(const AssignedInteger &a) const { return asserted(is_zero(a)); }
    FreshResult assert_equal_fresh(const AssignedInteger &a, const AssignedInteger &b) const { return asserted(is_equal_fresh(a, b)); }
    FreshResult assert_less_than(const AssignedInteger &a, const AssignedInteger &b) const { return asserted(is_less_than(a, b)); }
    FreshResult assert_less_than_or_equal(const AssignedInteger &a, const AssignedInteger &b) const { return asserted(is_less_than_or_equal(a, b)); }
    FreshResult assert_greater_than(const AssignedInteger &a, const AssignedInteger &b) const { return asserted(is_greater_than(a, b)); }
    FreshResult assert_greater_than_or_equal(const AssignedInteger &a, const AssignedInteger &b) const { return asserted(is_greater_than_or_equal(a, b)); }
    FreshResult assert_in_field(const AssignedInteger &a, const AssignedInteger &n) const { return asserted(is_in_field(a, n)); }

    // ---- Muled integers (instructions.rs:39-45, 63-76, 147-155, 212-220) --------------------------------------------------
    // big_integer/chip.rs:386-419 (square = mul(a, a), :431-437)
    MuledInteger mul(const AssignedInteger &a, const AssignedInteger &b) const {
        const size_t batch = a.batch();
        MuledInteger m; m.cols = DeviceBuffer(batch * 2 * num_limbs_ * 4 * 8); m.trace = DeviceBuffer(batch * layout_.record_stride);
        m.batch = batch; m.num_limbs = num_limbs_;
        hip_check(hipMemset(m.cols.get(), 0, m.cols.size()), "hipMemset");
        check(h2r_mul_batch(ctx_, a.data(), b.data(), batch, m.trace.get(), static_cast<uint64_t *>(m.cols.get()), nullptr), "mul");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return m;
    }
    MuledInteger square(const AssignedInteger &a) const { return mul(a, a); }
    // big_integer/chip.rs:822-895: the eq bit per element (and the step witness in `trace_out`, one record per element)
    std::vector<uint8_t> is_equal_muled(const MuledInteger &a, const MuledInteger &b, DeviceBuffer *trace_out = nullptr) const {
        const size_t batch = a.batch;
        DeviceBuffer trace(batch * layout_.record_stride), eq(batch);
        check(h2r_is_equal_muled_batch(ctx_, static_cast<const uint64_t *>(a.cols.get()), static_cast<const uint64_t *>(b.cols.get()), batch,
                                       trace.get(), static_cast<uint8_t *>(eq.get()), nullptr), "is_equal_muled");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<uint8_t> bits(batch); eq.download(bits.data(), batch);
        if (trace_out) *trace_out = std::move(trace);
        return bits;
    }
    // big_integer/chip.rs:1053-1063: status H2R_E_ASSERTION where the bit is 0
    std::vector<uint8_t> assert_equal_muled(const MuledInteger &a, const MuledInteger &b) const {
        std::vector<uint8_t> st = is_equal_muled(a, b);
        for (auto &v : st) v = v ? (uint8_t)H2R_OK : (uint8_t)H2R_E_ASSERTION;
        return st;
    }
    // big_integer/chip.rs:168-233 with RefreshAux::new(limb_width, L, L): 2L Fresh limbs per element
    std::pair<AssignedInteger, std::vector<uint8_t>> refresh(const MuledInteger &a, DeviceBuffer *stream_out = nullptr) const {
        const size_t batch = a.batch;
        const uint64_t stride = (h2r_refresh_stream_bytes(ctx_) + 255) / 256 * 256;
        DeviceBuffer trace(batch * stride), fresh(batch * 2 * num_limbs_ * 8), st(batch);
        check(h2r_refresh_batch(ctx_, static_cast<const uint64_t *>(a.cols.get()), batch, trace.get(), fresh.get(), static_cast<uint8_t *>(st.get()), nullptr), "refresh");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<uint8_t> status(batch); st.download(status.data(), batch);
        if (stream_out) *stream_out = std::move(trace);
        return {AssignedInteger(std::move(fresh), batch, 2 * num_limbs_), std::move(status)};
    }

    // ---- any operand shape (the reference's mul takes d0 != d1, chip.rs:395-397; refresh any RefreshAux::new(w, n_l, n_r),
    //      mod.rs:428; is_equal_muled n_l != n_r, chip.rs:822-842), operands of at most num_limbs limbs ----
    MuledInteger mul_general(const AssignedInteger &a, const AssignedInteger &b) const {
        const size_t batch = a.batch();
        MuledInteger m; m.cols = DeviceBuffer(batch * 2 * num_limbs_ * 4 * 8); m.trace = DeviceBuffer(batch * layout_.record_stride);
        m.batch = batch; m.num_limbs = num_limbs_;
        hip_check(hipMemset(m.cols.get(), 0, m.cols.size()), "hipMemset");
        check(h2r_mul_batch_ex(ctx_, a.data(), (uint32_t)a.num_limbs(), b.data(), (uint32_t)b.num_limbs(), batch, m.trace.get(),
                               static_cast<uint64_t *>(m.cols.get()), nullptr), "mul_general");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return m;
    }
    // refresh with RefreshAux::new(limb_width, n_l, n_r): (Fresh limbs, status); stream_out receives the flat streams
    std::pair<AssignedInteger, std::vector<uint8_t>> refresh_general(const MuledInteger &a, uint32_t n_l, uint32_t n_r, DeviceBuffer *stream_out = nullptr) const {
        uint32_t nf = 0; uint64_t sb = 0, stride = 0;
        check(h2r_refresh_layout(ctx_, n_l, n_r, &nf, &sb, &stride), "h2r_refresh_layout");
        const size_t batch = a.batch;
        DeviceBuffer trace(batch * stride), fresh(batch * nf * 8), st(batch);
        check(h2r_refresh_batch_ex(ctx_, static_cast<const uint64_t *>(a.cols.get()), 2 * num_limbs_, n_l, n_r, batch, trace.get(), fresh.get(),
                                   static_cast<uint8_t *>(st.get()), nullptr), "refresh_general");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<uint8_t> status(batch); st.download(status.data(), batch);
        if (stream_out) *stream_out = std::move(trace);
        return {AssignedInteger(std::move(fresh), batch, nf), std::move(status)};
    }
    std::vector<uint8_t> is_equal_muled_general(const MuledInteger &a, const MuledInteger &b, uint32_t n_l, uint32_t n_r) const {
        const uint64_t sb = h2r_is_equal_muled_stream_bytes_ex(ctx_, n_l, n_r, 0), stride = (sb + 15) / 16 * 16;
        if (!sb) throw Error(H2R_E_SHAPE, "is_equal_muled_general");
        const size_t batch = a.batch;
        DeviceBuffer out(batch * stride), eq(batch);
        check(h2r_is_equal_muled_batch_ex(ctx_, static_cast<const uint64_t *>(a.cols.get()), static_cast<const uint64_t *>(b.cols.get()), 2 * num_limbs_,
                                          n_l, n_r, batch, 0, out.get(), stride, static_cast<uint8_t *>(eq.get()), nullptr), "is_equal_muled_general");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<uint8_t> bits(batch); eq.download(bits.data(), batch);
        return bits;
    }

    // big_integer/chip.rs:542-629
    BatchResult mul_mod(const AssignedInteger &a, const AssignedInteger &b, const AssignedInteger &n) const {
        if (a.num_limbs() != n.num_limbs() || a.num_limbs() != num_limbs_) throw Error(H2R_E_SHAPE, "mul_mod");  // :555
        const size_t batch = a.batch();
        DeviceBuffer trace(batch * layout_.record_stride), r(batch * num_limbs_ * 8), st(batch);
        check(h2r_mul_mod_batch(ctx_, a.data(), b.data(), n.data(), batch, flags(n, batch), trace.get(), r.get(),
                                static_cast<uint8_t *>(st.get()), nullptr, nullptr), "mul_mod");
        return finish(std::move(trace), std::move(r), std::move(st), DeviceBuffer(), batch, false, h2r_pow_layout{}, layout_.record_stride, layout_.stream_bytes);
    }
    // big_integer/chip.rs:642-649
    BatchResult square_mod(const AssignedInteger &a, const AssignedInteger &n) const { return mul_mod(a, a, n); }
    // big_integer/chip.rs:710-742; e = BigUint::to_bytes_le()
    BatchResult pow_mod_fixed_exp(const AssignedInteger &a, const std::vector<uint8_t> &e_le, const AssignedInteger &n) const {
        return pow_fixed(a, e_le, n, nullptr);
    }
    // big_integer/chip.rs:664-696
    BatchResult pow_mod(const AssignedInteger &a, const AssignedInteger &e, const AssignedInteger &n, uint32_t exp_limb_bits) const {
        return pow_var(a, e, n, exp_limb_bits, nullptr);
    }

    // ---- the witness as cells: rows of the main gate's five advice columns, 160 bytes per row (h2r.h "advice image"; what the reference
    // assigns cell by cell: main_gate.mul_add chip.rs:408, range_chip.assign :590, :598, :880-885, the is_equal_muled ops :851-893) ----
    // rows of one element of `r` and their kinds (H2R_ROW_*)
    uint64_t advice_rows(const BatchResult &r) const { return r.is_pow ? h2r_pow_advice_rows(ctx_, &r.pow_layout) : h2r_advice_rows(ctx_); }
    std::vector<uint8_t> advice_row_kinds(const BatchResult &r) const {
        std::vector<uint8_t> k(advice_rows(r));
        check(r.is_pow ? h2r_pow_row_kinds(ctx_, &r.pow_layout, k.data()) : h2r_advice_row_kinds(ctx_, k.data()), "advice row kinds");
        return k;
    }
    // the image of a mul_mod result (a, b, n: the call's operands); direct = H2R_ADVICE_DIRECT: written from the operands and the record's
    // q, r limbs by cells_kernel instead of converted from the record planes -- the same bytes
    DeviceBuffer emit_advice(const BatchResult &r, const AssignedInteger &a, const AssignedInteger &b, const AssignedInteger &n, bool direct = false) const {
        if (r.is_pow) throw Error(H2R_E_SHAPE, "emit_advice(a, b, n) is for mul_mod results");
        const size_t batch = a.batch();
        const uint64_t stride = advice_rows(r) * H2R_ADVICE_ROW_BYTES;
        DeviceBuffer out(batch * stride);
        check(h2r_mul_mod_emit_advice(ctx_, a.data(), b.data(), n.data(), flags(n, batch) | (direct ? H2R_ADVICE_DIRECT : 0u), r.trace.data(), batch,
                                      static_cast<const uint8_t *>(r.status_dev.get()), out.get(), stride, nullptr), "h2r_mul_mod_emit_advice");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return out;
    }
    // the image of a pow_mod_fixed_exp / pow_mod result: [to_bits rows (Var)] [acc = 1: CONST1, CONST0] [every mul_mod's rows (+ select rows)]
    DeviceBuffer emit_advice(const BatchResult &r, const AssignedInteger &n, bool direct = false) const {
        if (!r.is_pow) throw Error(H2R_E_SHAPE, "emit_advice(n) is for pow results");
        const size_t batch = r.status.size();
        const uint64_t stride = advice_rows(r) * H2R_ADVICE_ROW_BYTES;
        DeviceBuffer out(batch * stride);
        check(h2r_pow_trace_emit_advice(ctx_, &r.pow_layout, n.data(), flags(n, batch) | (direct ? H2R_ADVICE_DIRECT : 0u), r.trace.data(), 0,
                                        r.workspace.get(), batch, static_cast<const uint8_t *>(r.status_dev.get()), out.get(), stride, nullptr),
              "h2r_pow_trace_emit_advice");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return out;
    }
    // ---- the placement of an op's cells in the five columns as DATA (h2r.h "LAYOUT": third-party maingate shapes, restated) ----
    // a layout in which the given row kinds have their five documented cells in the given physical columns (identity elsewhere);
    // throws H2R_E_SHAPE for a table that would change what a row means
    h2r_advice_layout advice_layout(const std::vector<uint8_t> &kinds = {}, const std::vector<std::array<uint8_t, 5>> &column_of = {}) const {
        h2r_advice_layout lay;
        if (kinds.empty()) { check(h2r_advice_layout_default(&lay), "h2r_advice_layout_default"); return lay; }
        if (kinds.size() != column_of.size()) throw Error(H2R_E_SHAPE, "advice_layout");
        check(h2r_advice_layout_custom(ctx_, kinds.data(), reinterpret_cast<const uint8_t (*)[5]>(column_of.data()), (uint32_t)kinds.size(), &lay),
              "h2r_advice_layout_custom");
        return lay;
    }
    // permute an emitted image (batch elements of `rows` rows whose kinds are `kinds`, element stride `stride`) to the layout, in place
    void apply_layout(const h2r_advice_layout &lay, const std::vector<uint8_t> &kinds, DeviceBuffer &image, uint64_t stride, size_t batch) const {
        DeviceBuffer kd(kinds.size());
        kd.upload(kinds.data(), kinds.size());
        check(h2r_advice_apply_layout(ctx_, &lay, static_cast<const uint8_t *>(kd.get()), kinds.size(), image.get(), stride, batch, nullptr, nullptr),
              "h2r_advice_apply_layout");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    // the fixed (selector) row of a kind under a layout
    h2r_fixed_row advice_fixed_row(uint32_t kind, const h2r_advice_layout &lay) const {
        h2r_fixed_row f;
        check(h2r_advice_fixed_row_ex(ctx_, nullptr, &lay, kind, &f), "h2r_advice_fixed_row_ex");
        return f;
    }
    // the equality (copy) constraints of one mul_mod record's rows: (row, col) <- (src_row, src_col) | limb src_col of operand a / b / n
    std::vector<h2r_copy> advice_copy_map() const {
        const uint32_t n = h2r_advice_copy_map(ctx_, nullptr, 0);
        std::vector<h2r_copy> v(n);
        if (n && h2r_advice_copy_map(ctx_, v.data(), n) != n) throw Error(H2R_E_INTERNAL, "h2r_advice_copy_map");
        return v;
    }

  private:
    static uint32_t flags(const AssignedInteger &n, size_t batch) { return (n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u; }
    FreshResult fresh(uint32_t op, const AssignedInteger &a, const AssignedInteger *b, const AssignedInteger *n) const {
        FreshResult r; uint32_t vl = 0;
        check(h2r_fresh_op_layout(ctx_, op, &r.elem_stride, &r.stream_bytes, &vl), "h2r_fresh_op_layout");
        const size_t batch = a.batch();
        r.op = op; r.ctx = ctx_; r.batch = batch; r.value_limbs = vl;
        r.trace = DeviceBuffer(batch * r.elem_stride); r.value = DeviceBuffer(batch * vl * 8);
        DeviceBuffer fl(batch), st(batch);
        hip_check(hipMemset(r.trace.get(), 0, batch * r.elem_stride), "hipMemset");
        uint32_t fl_shared = n ? flags(*n, batch) : 0u;
        if (b && b->batch() != batch) {   // one `b` for the whole batch (is_in_field's modulus, a shared comparand): only the ops without an `n`
            if (b->batch() == 1 && !n) fl_shared |= H2R_F_SHARED_MODULUS;
            else check(H2R_E_SHAPE, "fresh op: operand batches differ");
        }
        check(h2r_fresh_op_batch(ctx_, op, a.data(), b ? b->data() : nullptr, n ? n->data() : nullptr, batch, fl_shared,
                                 r.trace.get(), vl ? r.value.get() : nullptr, static_cast<uint8_t *>(fl.get()), static_cast<uint8_t *>(st.get()), nullptr),
              "h2r_fresh_op_batch");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        r.flag.resize(batch); r.status.resize(batch);
        fl.download(r.flag.data(), batch); st.download(r.status.data(), batch);
        return r;
    }
    static FreshResult asserted(FreshResult r) {
        for (size_t i = 0; i < r.status.size(); ++i) if (r.status[i] == H2R_OK && !r.flag[i]) r.status[i] = H2R_E_ASSERTION;
        return r;
    }
    // in_field != nullptr: RSAChip::modpow_public_key (assert_in_field witness into *in_field, status H2R_E_NOT_IN_FIELD)
    BatchResult pow_fixed(const AssignedInteger &a, const std::vector<uint8_t> &e_le, const AssignedInteger &n, DeviceBuffer *in_field) const {
        h2r_pow_layout pl;
        check(h2r_pow_fixed_layout(ctx_, e_le.data(), e_le.size(), &pl), "h2r_pow_fixed_layout");
        const size_t batch = a.batch();
        DeviceBuffer trace(batch * pl.elem_stride), out(batch * num_limbs_ * 8), st(batch), ws(h2r_workspace_bytes(ctx_, batch, pl.num_mul_mods));
        if (in_field)
            check(h2r_modpow_public_key_batch(ctx_, a.data(), n.data(), e_le.data(), e_le.size(), batch, flags(n, batch), trace.get(),
                                              in_field->get(), out.get(), static_cast<uint8_t *>(st.get()), ws.get(), nullptr), "modpow_public_key");
        else
            check(h2r_pow_mod_fixed_exp_batch(ctx_, a.data(), n.data(), e_le.data(), e_le.size(), batch, flags(n, batch), trace.get(),
                                              out.get(), static_cast<uint8_t *>(st.get()), ws.get(), nullptr), "pow_mod_fixed_exp");
        return finish(std::move(trace), std::move(out), std::move(st), std::move(ws), batch, true, pl, pl.elem_stride, pl.stream_bytes);
    }
    BatchResult pow_var(const AssignedInteger &a, const AssignedInteger &e, const AssignedInteger &n, uint32_t exp_limb_bits, DeviceBuffer *in_field) const {
        h2r_pow_layout pl;
        check(h2r_pow_var_layout(ctx_, (uint32_t)e.num_limbs(), exp_limb_bits, &pl), "h2r_pow_var_layout");
        const size_t batch = a.batch();
        DeviceBuffer trace(batch * pl.elem_stride), out(batch * num_limbs_ * 8), st(batch), ws(h2r_workspace_bytes(ctx_, batch, pl.num_mul_mods));
        if (in_field)
            check(h2r_modpow_public_key_var_batch(ctx_, a.data(), e.data(), (uint32_t)e.num_limbs(), exp_limb_bits, n.data(), batch,
                                                  flags(n, batch), trace.get(), in_field->get(), out.get(), static_cast<uint8_t *>(st.get()),
                                                  ws.get(), nullptr), "modpow_public_key");
        else
            check(h2r_pow_mod_batch(ctx_, a.data(), e.data(), (uint32_t)e.num_limbs(), exp_limb_bits, n.data(), batch, flags(n, batch),
                                    trace.get(), out.get(), static_cast<uint8_t *>(st.get()), ws.get(), nullptr), "pow_mod");
        return finish(std::move(trace), std::move(out), std::move(st), std::move(ws), batch, true, pl, pl.elem_stride, pl.stream_bytes);
    }
    BatchResult finish(DeviceBuffer trace, DeviceBuffer value, DeviceBuffer st, DeviceBuffer ws, size_t batch, bool is_pow, h2r_pow_layout pl,
                       uint64_t elem_stride, uint64_t stream_bytes) const {
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        std::vector<uint8_t> status(batch);
        st.download(status.data(), batch);
        return BatchResult{AssignedInteger(std::move(value), batch, num_limbs_), Trace(this, std::move(trace), batch, is_pow, pl, elem_stride, stream_bytes),
                           std::move(status), std::move(st), std::move(ws), pl, is_pow};
    }
    uint32_t limb_width_, num_limbs_;
    h2r_ctx *ctx_ = nullptr;
    h2r_layout layout_{};
    friend class Trace;
    friend class RSAChip;
    friend class RSASignatureVerifier;
};

inline std::vector<uint8_t> Trace::flatten(size_t elem) const {
    std::vector<uint8_t> host(elem_stride_), out(stream_bytes_);
    buf_.download(host.data(), elem_stride_, elem * elem_stride_);
    if (is_pow_) check(h2r_pow_trace_flatten(chip_->ctx(), &pl_, host.data(), out.data()), "h2r_pow_trace_flatten");
    else check(h2r_trace_flatten(chip_->ctx(), host.data(), out.data()), "h2r_trace_flatten");
    return out;
}

// reference src/lib.rs:25-140
struct RSAPubE {
    struct Fix { std::vector<uint8_t> e_le; };        // RSAPubE::Fix(BigUint)
    struct Var { UnassignedInteger e; };               // RSAPubE::Var(UnassignedInteger)
    std::variant<Fix, Var> v;
    static RSAPubE fix(uint64_t e) { std::vector<uint8_t> b; do { b.push_back((uint8_t)e); e >>= 8; } while (e); return RSAPubE{Fix{b}}; }
};
struct RSAPublicKey { UnassignedInteger n; RSAPubE e; };
struct RSASignature { UnassignedInteger c; };
struct AssignedRSAPublicKey { AssignedInteger n; std::variant<RSAPubE::Fix, AssignedInteger> e; };
struct AssignedRSASignature { AssignedInteger c; };

// RSAInstructions::modpow_public_key: the assert_in_field witness (src/chip.rs:106) followed by the pow path's
struct ModpowResult {
    BatchResult pow;               // value = x^e mod n, trace = the pow path, status (H2R_E_NOT_IN_FIELD where x >= n)
    DeviceBuffer in_field;         // batch elements, stride in_field_stride, flat stream of in_field_stream_bytes each
    uint64_t in_field_stride = 0, in_field_stream_bytes = 0;
    const h2r_ctx *ctx = nullptr;
    // the element's whole witness in the reference's assignment order: in-field stream, then the pow stream
    std::vector<uint8_t> flatten(size_t elem) const {
        std::vector<uint8_t> host(in_field_stride), out(in_field_stream_bytes);
        in_field.download(host.data(), host.size(), elem * in_field_stride);
        check(h2r_fresh_op_flatten(ctx, H2R_OP_IS_IN_FIELD, host.data(), out.data()), "h2r_fresh_op_flatten");
        const std::vector<uint8_t> p = pow.trace.flatten(elem);
        out.insert(out.end(), p.begin(), p.end());
        return out;
    }
};

struct VerifyResult {
    std::vector<uint8_t> is_valid;   // one byte per signature
    std::vector<uint8_t> status;
    AssignedInteger powed;
    DeviceBuffer trace; h2r_verify_layout layout;
    DeviceBuffer status_dev, workspace;   // kept for RSAChip::emit_advice (the workspace holds every mul_mod's operands)
};

// reference src/chip.rs:38-255
class RSAChip {
  public:
    static constexpr uint32_t LIMB_WIDTH = 64;  // src/chip.rs:203
    // RSAChip::new(config, bits_len, exp_limb_bits), src/chip.rs:214-221
    RSAChip(uint32_t bits_len, uint32_t exp_limb_bits, uint32_t field = H2R_FIELD_BN254_FR, int device = 0)
        : bits_len_(bits_len), exp_limb_bits_(exp_limb_bits), bigint_(LIMB_WIDTH, bits_len, field, device) {}
    const BigIntChip &bigint_chip() const { return bigint_; }  // src/chip.rs:224-230
    uint32_t exp_limb_bits() const { return exp_limb_bits_; }
    // src/chip.rs:249-254
    static std::pair<std::vector<uint32_t>, std::vector<uint32_t>> compute_range_lens(uint32_t num_limbs) {
        std::vector<uint32_t> c(4), o(3);
        check(h2r_rsa_compute_range_lens(num_limbs, c.data(), o.data()), "RSAChip::compute_range_lens");
        return {c, o};
    }
    // src/chip.rs:58-70
    AssignedRSAPublicKey assign_public_key(const RSAPublicKey &pk) const {
        AssignedInteger n = bigint_.assign_integer(pk.n);
        if (auto *f = std::get_if<RSAPubE::Fix>(&pk.e.v)) return AssignedRSAPublicKey{std::move(n), *f};
        return AssignedRSAPublicKey{std::move(n), bigint_.assign_integer(std::get<RSAPubE::Var>(pk.e.v).e)};
    }
    // src/chip.rs:80-88
    AssignedRSASignature assign_signature(const RSASignature &s) const { return AssignedRSASignature{bigint_.assign_integer(s.c)}; }
    // src/chip.rs:99-114: assert_in_field(x, n) (:106; witness + status H2R_E_NOT_IN_FIELD), then Fix -> pow_mod_fixed_exp,
    // Var -> pow_mod with the chip's exp_limb_bits
    ModpowResult modpow_public_key(const AssignedInteger &x, const AssignedRSAPublicKey &pk) const {
        uint64_t es = 0, sb = 0;
        check(h2r_fresh_op_layout(bigint_.ctx(), H2R_OP_IS_IN_FIELD, &es, &sb, nullptr), "h2r_fresh_op_layout");
        DeviceBuffer inf(x.batch() * es);
        BatchResult r = std::holds_alternative<RSAPubE::Fix>(pk.e)
                            ? bigint_.pow_fixed(x, std::get<RSAPubE::Fix>(pk.e).e_le, pk.n, &inf)
                            : bigint_.pow_var(x, std::get<AssignedInteger>(pk.e), pk.n, exp_limb_bits_, &inf);
        return ModpowResult{std::move(r), std::move(inf), es, sb, bigint_.ctx()};
    }
    // src/chip.rs:128-199 (hashed_msg = 4 little-endian 64-bit limbs of the SHA-256 digest per signature, :141-144)
    VerifyResult verify_pkcs1v15_signature(const AssignedRSAPublicKey &pk, const AssignedInteger &hashed_msg,
                                           const AssignedRSASignature &sig) const {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        h2r_verify_layout vl;
        const size_t batch = sig.c.batch();
        if (f) check(h2r_verify_layout_fixed(bigint_.ctx(), f->e_le.data(), f->e_le.size(), &vl), "h2r_verify_layout_fixed");
        else check(h2r_verify_layout_var(bigint_.ctx(), (uint32_t)std::get<AssignedInteger>(pk.e).num_limbs(), exp_limb_bits_, &vl), "h2r_verify_layout_var");
        DeviceBuffer trace(batch * vl.elem_stride), powed(batch * bigint_.num_limbs() * 8), valid(batch), st(batch),
            ws(h2r_workspace_bytes(bigint_.ctx(), batch, vl.pow.num_mul_mods));
        if (f)
            check(h2r_verify_pkcs1v15_batch(bigint_.ctx(), sig.c.data(), pk.n.data(), f->e_le.data(), f->e_le.size(),
                                            static_cast<const uint64_t *>(hashed_msg.data()), batch, BigIntChip::flags(pk.n, batch), trace.get(),
                                            powed.get(), static_cast<uint8_t *>(valid.get()), static_cast<uint8_t *>(st.get()), ws.get(), nullptr),
                  "verify_pkcs1v15_signature");
        else {   // RSAPubE::Var (src/chip.rs:108-110)
            const AssignedInteger &e = std::get<AssignedInteger>(pk.e);
            check(h2r_verify_pkcs1v15_var_batch(bigint_.ctx(), sig.c.data(), pk.n.data(), e.data(), (uint32_t)e.num_limbs(), exp_limb_bits_,
                                                static_cast<const uint64_t *>(hashed_msg.data()), batch, BigIntChip::flags(pk.n, batch), trace.get(),
                                                powed.get(), static_cast<uint8_t *>(valid.get()), static_cast<uint8_t *>(st.get()), ws.get(), nullptr),
                  "verify_pkcs1v15_signature (Var)");
        }
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        VerifyResult r{std::vector<uint8_t>(batch), std::vector<uint8_t>(batch), AssignedInteger(std::move(powed), batch, bigint_.num_limbs()),
                       std::move(trace), vl, DeviceBuffer(), std::move(ws)};
        valid.download(r.is_valid.data(), batch);
        st.download(r.status.data(), batch);
        r.status_dev = std::move(st);
        return r;
    }
    // ---- the witness as cells (h2r.h "advice image") ----
    // one whole verify_pkcs1v15_signature element (src/chip.rs:128-199): [is_eq = 1] [assert_in_field rows] [pow rows] [encoded-message check];
    // pk, hashed_msg, sig: what the call was given.  direct: the pow rows written from the operands (H2R_ADVICE_DIRECT), the same bytes.
    uint64_t advice_rows(const VerifyResult &r, uint64_t section_rows[4] = nullptr) const { return h2r_verify_advice_rows(bigint_.ctx(), &r.layout, section_rows); }
    DeviceBuffer emit_advice(const VerifyResult &r, const AssignedRSAPublicKey &pk, const AssignedInteger &hashed_msg, const AssignedRSASignature &sig,
                             bool direct = false) const {
        const size_t batch = sig.c.batch();
        const uint64_t stride = advice_rows(r) * H2R_ADVICE_ROW_BYTES;
        DeviceBuffer out(batch * stride);
        check(h2r_verify_emit_advice(bigint_.ctx(), &r.layout, sig.c.data(), pk.n.data(), static_cast<const uint64_t *>(hashed_msg.data()), r.powed.data(),
                                     BigIntChip::flags(pk.n, batch) | (direct ? H2R_ADVICE_DIRECT : 0u), r.trace.get(), r.workspace.get(), batch,
                                     static_cast<const uint8_t *>(r.status_dev.get()), out.get(), stride, nullptr), "h2r_verify_emit_advice");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return out;
    }
    // one modpow_public_key element (src/chip.rs:99-114): [assert_in_field rows] [pow rows]; with_records = false: as if the call had
    // written no records (the pow rows come from the operands alone)
    uint64_t advice_rows(const ModpowResult &r, uint64_t section_rows[2] = nullptr) const { return h2r_modpow_public_key_advice_rows(bigint_.ctx(), &r.pow.pow_layout, section_rows); }
    DeviceBuffer emit_advice(const ModpowResult &r, const AssignedInteger &x, const AssignedRSAPublicKey &pk, bool with_records = true) const {
        const size_t batch = x.batch();
        const uint64_t stride = advice_rows(r) * H2R_ADVICE_ROW_BYTES;
        DeviceBuffer out(batch * stride);
        check(h2r_modpow_public_key_emit_advice(bigint_.ctx(), &r.pow.pow_layout, x.data(), pk.n.data(), BigIntChip::flags(pk.n, batch), r.in_field.get(),
                                                with_records ? r.pow.trace.data() : nullptr, r.pow.workspace.get(), batch,
                                                static_cast<const uint8_t *>(r.pow.status_dev.get()), out.get(), stride, nullptr),
              "h2r_modpow_public_key_emit_advice");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        return out;
    }
    std::vector<uint8_t> flatten(const VerifyResult &r, size_t elem) const {
        std::vector<uint8_t> host(r.layout.elem_stride), out(r.layout.stream_bytes);
        r.trace.download(host.data(), host.size(), elem * r.layout.elem_stride);
        check(h2r_verify_trace_flatten(bigint_.ctx(), &r.layout, host.data(), out.data()), "h2r_verify_trace_flatten");
        return out;
    }

  private:
    uint32_t bits_len_, exp_limb_bits_;
    BigIntChip bigint_;
};

// reference src/lib.rs:149-246: RSASignatureVerifier { rsa_chip, sha256_chip }.  The SHA-256 chip's own circuit is third-party and
// outside the accelerated path; its digest values, the reversed-byte limb composition of the verifier's region (:210-239) and the
// RSAChip verification run on the device (h2r_signature_verifier_batch).
struct SignatureVerifyResult {
    VerifyResult verify;                 // is_valid (:243), status, powed, trace
    std::vector<uint8_t> hashed_bytes;   // 32 per signature, digest order: the second return value of the reference (:243-245)
    DeviceBuffer hashed_msg;             // 4 limbs per signature (the operand of RSAChip::verify_pkcs1v15_signature)
    DeviceBuffer hashed_msg_trace;       // H2R_HASHED_MSG_STREAM_BYTES per signature
};
class RSASignatureVerifier {
  public:
    // sha256_max_byte_size: the capacity the reference's Sha256Config is created with (src/lib.rs:321); 0 = unbounded
    explicit RSASignatureVerifier(const RSAChip &rsa_chip, size_t sha256_max_byte_size = 0) : rsa_(rsa_chip), max_(sha256_max_byte_size) {}
    // msgs: one message per signature (ragged; empty allowed)
    SignatureVerifyResult verify_pkcs1v15_signature(const AssignedRSAPublicKey &pk, const std::vector<std::vector<uint8_t>> &msgs,
                                                    const AssignedRSASignature &sig) const {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "RSASignatureVerifier::verify_pkcs1v15_signature (batch path takes RSAPubE::Fix)");
        const BigIntChip &bi = rsa_.bigint_chip();
        const size_t batch = sig.c.batch();
        if (msgs.size() != batch) throw Error(H2R_E_SHAPE, "one message per signature");
        std::vector<uint64_t> off(batch + 1, 0);
        std::vector<uint8_t> bytes;
        for (size_t i = 0; i < batch; ++i) {
            if (max_ && msgs[i].size() > max_) throw Error(H2R_E_SHAPE, "message longer than the SHA-256 chip's max_byte_size");
            bytes.insert(bytes.end(), msgs[i].begin(), msgs[i].end());
            off[i + 1] = bytes.size();
        }
        DeviceBuffer dmsg(bytes.size() + 16), doff(off.size() * 8);
        if (!bytes.empty()) dmsg.upload(bytes.data(), bytes.size());
        doff.upload(off.data(), off.size() * 8);
        h2r_verify_layout vl;
        check(h2r_verify_layout_fixed(bi.ctx(), f->e_le.data(), f->e_le.size(), &vl), "h2r_verify_layout_fixed");
        DeviceBuffer trace(batch * vl.elem_stride), powed(batch * bi.num_limbs() * 8), valid(batch), st(batch), digest(batch * 32),
            hashed(batch * 32), hm(batch * H2R_HASHED_MSG_STREAM_BYTES);
        check(h2r_signature_verifier_batch(bi.ctx(), static_cast<const uint8_t *>(dmsg.get()), static_cast<const uint64_t *>(doff.get()), 0,
                                           sig.c.data(), pk.n.data(), f->e_le.data(), f->e_le.size(), batch, BigIntChip::flags(pk.n, batch),
                                           trace.get(), hm.get(), H2R_HASHED_MSG_STREAM_BYTES, static_cast<uint8_t *>(digest.get()),
                                           static_cast<uint64_t *>(hashed.get()), powed.get(), static_cast<uint8_t *>(valid.get()),
                                           static_cast<uint8_t *>(st.get()), nullptr, nullptr), "h2r_signature_verifier_batch");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        SignatureVerifyResult r{VerifyResult{std::vector<uint8_t>(batch), std::vector<uint8_t>(batch),
                                             AssignedInteger(std::move(powed), batch, bi.num_limbs()), std::move(trace), vl},
                                std::vector<uint8_t>(batch * 32), std::move(hashed), std::move(hm)};
        valid.download(r.verify.is_valid.data(), batch);
        st.download(r.verify.status.data(), batch);
        digest.download(r.hashed_bytes.data(), batch * 32);
        return r;
    }

  private:
    const RSAChip &rsa_;
    size_t max_;
};

// h2r_pipeline_*: consecutive verifier batches overlap the off-circuit chain of batch k+1 with the record emission
// of batch k.  The caller rotates through `depth` buffer sets; join() orders `stream` after every record kernel.
// h2r_arena: the fastest `regions` of `candidates` mapped-and-measured trace regions for `batch` elements of the verifier
// layout of public exponent e_le (where a trace buffer lies physically decides how fast the record kernel writes it).
class TraceArena {
  public:
    TraceArena(const RSAChip &chip, size_t batch, const std::vector<uint8_t> &e_le, uint32_t regions = 2, uint32_t candidates = 16,
               hipStream_t stream = nullptr) {
        h2r_verify_layout vl{};
        check(h2r_verify_layout_fixed(chip.bigint_chip().ctx(), e_le.data(), e_le.size(), &vl), "h2r_verify_layout_fixed");
        check(h2r_arena_create(chip.bigint_chip().ctx(), vl.elem_stride, vl.pow.off_records, vl.pow.num_mul_mods, batch, regions, candidates,
                               stream, &a_), "h2r_arena_create");
    }
    ~TraceArena() { h2r_arena_destroy(a_); }
    TraceArena(const TraceArena &) = delete;
    TraceArena &operator=(const TraceArena &) = delete;
    void *region(uint32_t i) const { return h2r_arena_region(a_, i); }        // fastest first
    double region_ms(uint32_t i) const { return h2r_arena_region_ms(a_, i); }
    uint64_t region_bytes() const { return h2r_arena_region_bytes(a_); }
  private:
    h2r_arena *a_ = nullptr;
};

class Pipeline {
  public:
    struct Buffers {   // one buffer set (sized for `batch` signatures and public exponent `e_le`)
        DeviceBuffer trace, workspace, powed, is_valid, status;
        h2r_verify_layout layout{};
    };
    explicit Pipeline(const RSAChip &chip, uint32_t depth = 2, uint32_t side_streams = 1) : chip_(chip) {
        check(h2r_pipeline_create_ex(chip.bigint_chip().ctx(), depth, side_streams, &p_), "h2r_pipeline_create_ex");
    }
    ~Pipeline() { h2r_pipeline_destroy(p_); }
    Pipeline(const Pipeline &) = delete;
    Pipeline &operator=(const Pipeline &) = delete;
    Buffers make_buffers(size_t batch, const std::vector<uint8_t> &e_le) const {
        Buffers b;
        const BigIntChip &bc = chip_.bigint_chip();
        check(h2r_verify_layout_fixed(bc.ctx(), e_le.data(), e_le.size(), &b.layout), "h2r_verify_layout_fixed");
        const uint64_t ws = h2r_workspace_bytes(bc.ctx(), batch, b.layout.pow.num_mul_mods);
        b.trace = DeviceBuffer(batch * b.layout.elem_stride); b.workspace = DeviceBuffer(ws);
        b.powed = DeviceBuffer(batch * bc.num_limbs() * 8); b.is_valid = DeviceBuffer(batch); b.status = DeviceBuffer(batch);
        return b;
    }
    // RSAInstructions::verify_pkcs1v15_signature (src/chip.rs:128-199), asynchronous on `stream`
    void verify_pkcs1v15_signature(const AssignedRSAPublicKey &pk, const AssignedInteger &hashed_msg, const AssignedRSASignature &sig,
                                   Buffers &b, hipStream_t stream = nullptr) {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "Pipeline::verify_pkcs1v15_signature (takes RSAPubE::Fix)");
        const size_t batch = sig.c.batch();
        check(h2r_pipeline_verify_pkcs1v15(p_, sig.c.data(), pk.n.data(), f->e_le.data(), f->e_le.size(),
                                           static_cast<const uint64_t *>(hashed_msg.data()), batch,
                                           (pk.n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u, b.trace.get(), b.powed.get(),
                                           static_cast<uint8_t *>(b.is_valid.get()), static_cast<uint8_t *>(b.status.get()),
                                           b.workspace.get(), stream), "h2r_pipeline_verify_pkcs1v15");
    }
    // RSASignatureVerifier::verify_pkcs1v15_signature from message BYTES (src/lib.rs:183-246), asynchronous on `stream`:
    // msgs / msg_off are DEVICE buffers (message i = msgs[msg_off[i], msg_off[i + 1])), digest / hashed receive 32 bytes / 4 limbs
    // per signature.  On the one-launch-step shapes the SHA-256 step and the verifier's witness ride inside the step launch.
    void signature_verifier(const AssignedRSAPublicKey &pk, const DeviceBuffer &msgs, const DeviceBuffer &msg_off, const AssignedRSASignature &sig,
                            Buffers &b, DeviceBuffer &digest, DeviceBuffer &hashed, hipStream_t stream = nullptr) {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "Pipeline::signature_verifier (takes RSAPubE::Fix)");
        const size_t batch = sig.c.batch();
        check(h2r_pipeline_signature_verifier(p_, static_cast<const uint8_t *>(msgs.get()), static_cast<const uint64_t *>(msg_off.get()), 0,
                                              sig.c.data(), pk.n.data(), f->e_le.data(), f->e_le.size(), batch,
                                              (pk.n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u, b.trace.get(), nullptr, 0,
                                              static_cast<uint8_t *>(digest.get()), static_cast<uint64_t *>(hashed.get()), b.powed.get(),
                                              static_cast<uint8_t *>(b.is_valid.get()), static_cast<uint8_t *>(b.status.get()),
                                              b.workspace.get(), stream), "h2r_pipeline_signature_verifier");
    }
    // RSAInstructions::modpow_public_key (src/chip.rs:99-114, RSAPubE::Fix), asynchronous on `stream`; uses the
    // trace / workspace / powed / status members of the buffer set (its trace region is large enough for the pow trace)
    // (the in-field witness goes to the element's in-field region of the verify layout: same format, stride = elem_stride)
    void modpow_public_key(const AssignedInteger &x, const AssignedRSAPublicKey &pk, Buffers &b, hipStream_t stream = nullptr,
                           DeviceBuffer *in_field = nullptr) {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "Pipeline::modpow_public_key (takes RSAPubE::Fix)");
        const size_t batch = x.batch();
        check(h2r_pipeline_modpow_public_key(p_, x.data(), pk.n.data(), f->e_le.data(), f->e_le.size(), batch,
                                             (pk.n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u, b.trace.get(),
                                             in_field ? in_field->get() : nullptr, b.powed.get(),
                                             static_cast<uint8_t *>(b.status.get()), b.workspace.get(), stream), "h2r_pipeline_modpow_public_key");
    }
    // The same call WITHOUT records whose product is the element's advice image ([assert_in_field rows] [pow rows], 160 bytes per row:
    // RSAChip::advice_rows(ModpowResult) rows per element): chains, in-field witness and in-field rows on `stream`, the pow rows
    // (cells_kernel) on the pipeline's side streams next to the following call's chains.  `advice` is complete after depth - 1 further
    // calls or join().  in_field: batch * h2r_fresh_op_layout(H2R_OP_IS_IN_FIELD) stride bytes.
    void modpow_public_key_advice(const AssignedInteger &x, const AssignedRSAPublicKey &pk, Buffers &b, DeviceBuffer &in_field, DeviceBuffer &advice,
                                  uint64_t advice_stride, hipStream_t stream = nullptr) {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "Pipeline::modpow_public_key_advice (takes RSAPubE::Fix)");
        const size_t batch = x.batch();
        check(h2r_pipeline_modpow_public_key_advice(p_, x.data(), pk.n.data(), f->e_le.data(), f->e_le.size(), batch,
                                                    (pk.n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u, in_field.get(), b.powed.get(),
                                                    static_cast<uint8_t *>(b.status.get()), b.workspace.get(), advice.get(), advice_stride, stream),
              "h2r_pipeline_modpow_public_key_advice");
    }
    // The RSAPubE::Var arm of modpow_public_key_advice (src/chip.rs:108-110): per-element exponents, no records; `witness` keeps the exponent
    // bits, the selected operands and the result: batch * compact_pow_layout(...).elem_stride bytes.
    static h2r_pow_layout compact_pow_layout(const RSAChip &chip, uint32_t e_num_limbs) {
        h2r_pow_layout full, pl;
        check(h2r_pow_var_layout(chip.bigint_chip().ctx(), e_num_limbs, chip.exp_limb_bits(), &full), "h2r_pow_var_layout");
        check(h2r_pow_layout_compact(chip.bigint_chip().ctx(), &full, &pl), "h2r_pow_layout_compact");
        return pl;
    }
    void modpow_public_key_var_advice(const AssignedInteger &x, const AssignedRSAPublicKey &pk, Buffers &b, DeviceBuffer &in_field, DeviceBuffer &witness,
                                      DeviceBuffer &advice, uint64_t advice_stride, hipStream_t stream = nullptr) {
        auto *v = std::get_if<AssignedInteger>(&pk.e);
        if (!v) throw Error(H2R_E_UNSUPPORTED, "Pipeline::modpow_public_key_var_advice (takes RSAPubE::Var)");
        const size_t batch = x.batch();
        check(h2r_pipeline_modpow_public_key_var_advice(p_, x.data(), v->data(), (uint32_t)v->num_limbs(), chip_.exp_limb_bits(), pk.n.data(), batch,
                                                        (pk.n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u, in_field.get(), witness.get(),
                                                        b.powed.get(), static_cast<uint8_t *>(b.status.get()), b.workspace.get(), advice.get(),
                                                        advice_stride, stream), "h2r_pipeline_modpow_public_key_var_advice");
    }
    // The WHOLE RSAInstructions::verify_pkcs1v15_signature element (src/chip.rs:128-199) as advice rows without records
    // ([is_eq seed][assert_in_field][pow rows][encoded-message check]: h2r_verify_advice_rows rows per element): chains, witness, is_valid and
    // the short row programs on `stream`, the pow rows on the pipeline's side streams.  witness: batch * compact_layout(pk).elem_stride bytes.
    static h2r_verify_layout compact_layout(const RSAChip &chip, const AssignedRSAPublicKey &pk) {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "Pipeline::compact_layout (takes RSAPubE::Fix)");
        h2r_verify_layout full, vl;
        check(h2r_verify_layout_fixed(chip.bigint_chip().ctx(), f->e_le.data(), f->e_le.size(), &full), "h2r_verify_layout_fixed");
        check(h2r_verify_layout_compact(chip.bigint_chip().ctx(), &full, &vl), "h2r_verify_layout_compact");
        return vl;
    }
    void verify_pkcs1v15_signature_advice(const AssignedRSAPublicKey &pk, const AssignedInteger &hashed_msg, const AssignedRSASignature &sig, Buffers &b,
                                DeviceBuffer &witness, DeviceBuffer &advice, uint64_t advice_stride, hipStream_t stream = nullptr) {
        auto *f = std::get_if<RSAPubE::Fix>(&pk.e);
        if (!f) throw Error(H2R_E_UNSUPPORTED, "Pipeline::verify_pkcs1v15_signature_advice (takes RSAPubE::Fix)");
        const size_t batch = sig.c.batch();
        check(h2r_pipeline_verify_pkcs1v15_advice(p_, sig.c.data(), pk.n.data(), f->e_le.data(), f->e_le.size(),
                                                  static_cast<const uint64_t *>(hashed_msg.data()), batch,
                                                  (pk.n.batch() == 1 && batch != 1) ? H2R_F_SHARED_MODULUS : 0u, witness.get(), b.powed.get(),
                                                  static_cast<uint8_t *>(b.is_valid.get()), static_cast<uint8_t *>(b.status.get()), b.workspace.get(),
                                                  advice.get(), advice_stride, stream), "h2r_pipeline_verify_pkcs1v15_advice");
    }
    void join(hipStream_t stream = nullptr) { check(h2r_pipeline_join(p_, stream), "h2r_pipeline_join"); }
    // an element's whole verify witness in the reference's order (after join() + synchronisation)
    std::vector<uint8_t> flatten(const Buffers &b, size_t elem) const {
        std::vector<uint8_t> host(b.layout.elem_stride), out(b.layout.stream_bytes);
        b.trace.download(host.data(), host.size(), elem * b.layout.elem_stride);
        check(h2r_verify_trace_flatten(chip_.bigint_chip().ctx(), &b.layout, host.data(), out.data()), "h2r_verify_trace_flatten");
        return out;
    }

  private:
    const RSAChip &chip_;
    h2r_pipeline *p_ = nullptr;
};

}  // namespace h2r_host
