/*
 * h2r.h -- C ABI of the MI355X-native halo2-rsa witness engine (libh2r.so).
 *
 * Drop-in boundary for ONE path of SoraSuegami/halo2-rsa: the witness ("assign_*") values of
 *   RSAChip::modpow_public_key            reference src/chip.rs:99-114
 *     -> BigIntChip::pow_mod_fixed_exp    reference src/big_integer/chip.rs:710-742
 *     -> BigIntChip::pow_mod              reference src/big_integer/chip.rs:664-696
 *     -> BigIntChip::mul_mod / square_mod reference src/big_integer/chip.rs:542-629 / 642-649
 *        (mul :386-419, is_equal_muled :822-895, div_mod_main_gate :1323-1349)
 *   plus the RangeChip sub-limb decomposition of every range-checked value on that path
 *   (call sites big_integer/chip.rs:74, 590, 598, 880-885) and its lookup multiplicities.
 *
 * The reference has no FFI; it is a Rust crate whose API is the two traits
 * BigIntInstructions<F> (src/big_integer/instructions.rs:7-260) and RSAInstructions<F>
 * (src/instructions.rs:8-39).  Each export below is the batch form of one trait method: a Rust
 * `impl BigIntInstructions<F> for GpuBigIntChip<F>` calls the export, then walks the returned
 * trace with h2r_trace_flatten()/the h2r_layout offsets and assigns the values to cells
 * (INTEGRATION.md shows the `extern "C"` block).
 *
 * Conventions
 *  - Integers are little-endian limb vectors, limb 0 least significant (big_integer/mod.rs:348-359),
 *    `num_limbs` limbs of `limb_width` bits: uint64_t limbs for limb_width 64, uint32_t for 32.
 *    Batches are element-major: element e occupies limbs [e*num_limbs, (e+1)*num_limbs).
 *  - Every data pointer is a DEVICE (HBM) pointer valid on the ctx's device.  `stream` is a
 *    hipStream_t (NULL = the null stream).  Calls only enqueue work; they never synchronise the
 *    device.  Calls on distinct streams may run concurrently; a ctx is immutable after creation.
 *  - The library owns nothing but the ctx.  `workspace` is caller-provided scratch of at least
 *    h2r_workspace_bytes(); pass NULL to let the call use stream-ordered hipMallocAsync/hipFreeAsync.
 *  - Every call returns an int32_t status and never throws or aborts.  Where the reference panics
 *    on a per-value condition (division by a zero modulus big_integer/chip.rs:566; quotient not
 *    fitting num_limbs limbs :583-584) the element's byte in `status` is set instead and that
 *    element's trace is unspecified; other elements are unaffected.
 */
#ifndef H2R_H
#define H2R_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libh2r.so is built with -fvisibility=hidden: exactly the prototypes of this header are exported (tests/test_cabi_host.py
 * asserts it), so that a Rust cdylib / C++ host linking the library never meets an un-prefixed internal symbol. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define H2R_VERSION 4

/* ---- status codes (function return values and per-element status bytes) ---------------------- */
enum {
    H2R_OK = 0,
    H2R_E_SHAPE = 1,           /* reference asserts big_integer/chip.rs:555, 1175, 1266 */
    H2R_E_ZERO_MODULUS = 2,    /* reference divides by zero, big_integer/chip.rs:566 */
    H2R_E_NOT_REDUCED = 3,     /* a*b/n does not fit num_limbs limbs, big_integer/chip.rs:583-584 */
    H2R_E_FIELD_TOO_SMALL = 4, /* reference assert big_integer/chip.rs:1178 */
    H2R_E_HIP = 5,             /* a HIP runtime call failed (h2r_last_hip_error) */
    H2R_E_UNSUPPORTED = 6,     /* (limb_width, num_limbs) has no compiled kernel */
    H2R_E_NULL = 7,            /* required pointer is NULL */
    H2R_E_NOT_IN_FIELD = 8,    /* x >= n: assert_in_field fails, src/chip.rs:106 */
    H2R_E_ASSERTION = 9,       /* an assert_* constraint does not hold (main_gate.assert_one / assert_zero on the
                                  predicate bit, big_integer/chip.rs:1020-1158): set by the host mirrors */
    H2R_E_NOMEM = 10,          /* a host allocation failed inside the library (no C++ exception crosses the boundary) */
    H2R_E_INTERNAL = 11        /* any other C++ exception, stopped at the boundary */
};

/* ---- fields (only F::NUM_BITS and the encoding of negative a_b depend on it) ------------------ */
enum {
    H2R_FIELD_BN254_FR = 0, /* halo2curves bn256::Fr (examples/rsa_example.rs:148, benches/bench.rs:35) */
    H2R_FIELD_BN254_FQ = 1, /* bn256::Fq   (tests: big_integer/chip.rs:1461) */
    H2R_FIELD_PASTA_FP = 2, /* pasta::Fp   (big_integer/chip.rs:1462) */
    H2R_FIELD_PASTA_FQ = 3  /* pasta::Fq   (big_integer/chip.rs:1463) */
};

/* flags for the batch calls */
#define H2R_F_SHARED_MODULUS 1u /* `n` holds ONE modulus used by every element */

typedef struct h2r_ctx h2r_ctx;
typedef void *h2r_stream_t; /* hipStream_t */

/* BigIntChip::new(config, limb_width, bits_len) -- big_integer/chip.rs:1174-1185 */
typedef struct h2r_params {
    uint32_t limb_width; /* 64 (RSAChip::LIMB_WIDTH, src/chip.rs:203) or 32 */
    uint32_t bits_len;   /* bits_len % limb_width == 0; num_limbs = bits_len / limb_width */
    uint32_t field;      /* H2R_FIELD_* */
    int32_t device;      /* HIP device ordinal; < 0 = host-only ctx (layouts + flatten, no device work) */
} h2r_params;

/* ---- trace layout ------------------------------------------------------------------------------
 * One mul_mod produces one RECORD: a struct of planes.  Plane p starts at byte plane_off[p] of the
 * record, has plane_count[p] entries of plane_elem[p] bytes (0 = plane absent for this config).
 * Values wider than 16 bytes are split into a 16-byte LO plane and an 8-byte HI plane.
 * C = 2*num_limbs-1 is the number of un-carried product columns.  Index maps:
 *   H2R_PL_Q, _R           [k]            quotient / remainder limb k          (chip.rs:570-582)
 *   H2R_PL_Q_SUB, _R_SUB   [k][t]         sub-limb t of limb k, one byte each  (chip.rs:590, 598)
 *   H2R_PL_AB_*, _QN_*     [j][i % L]     accumulator of column i=j+k right after adding
 *                                         a[j]*b[k] (resp. q[j]*n[k]); the reference's order is
 *                                         i ascending, then j ascending        (chip.rs:400-412).
 *                                         Addressing (strides in h2r_layout), with
 *                                         g = j / acc_steps_per_group, s = j % acc_steps_per_group:
 *                                           LO(j, i%L) at plane_off[P_LO] + g*acc_lo_group_bytes
 *                                                         + s*acc_lo_row_bytes + (i%L)*16
 *                                           HI(j, i%L) at plane_off[P_HI] + (j/2)*acc_hi_group_bytes
 *                                                         + (i%L)*16 + (j%2)*8      (limb_width 64 only)
 *                                         The four planes may be separate regions (planar) or
 *                                         interleaved row by row ([ab | qn] LO rows of a group of
 *                                         steps, then one shared [ab | qn] HI row) so that a step
 *                                         writes one contiguous run; the strides say which.
 *   H2R_PL_EQB_*           [i], i < L     qn[i] + r[i]                         (chip.rs:617)
 *   per-column planes      [i], i < C     is_equal_muled step i                (chip.rs:857-893)
 *   H2R_PL_CARRY_DUP/_SUB  [i], i < C-1   the range-assigned carry and its sub-limbs (:879-885);
 *                                         one byte per sub-limb in a 16-byte slot per column
 * Per-column planes reserve 2L entries; entries past the plane's count are written as zeros.
 *   H2R_PL_FLAGS           [i][4]         cs_acc_eq, eq_bit, range_eq|final_carry_eq, eq_bit
 */
enum {
    H2R_PL_Q = 0, H2R_PL_R, H2R_PL_Q_SUB, H2R_PL_R_SUB,
    H2R_PL_AB_LO, H2R_PL_AB_HI, H2R_PL_QN_LO, H2R_PL_QN_HI,
    H2R_PL_EQB_LO, H2R_PL_EQB_HI,
    H2R_PL_AMB_LO, H2R_PL_AMB_HI,   /* a_b = a[i]-b[i], two's complement (chip.rs:859) */
    H2R_PL_SUM_LO, H2R_PL_SUM_HI,   /* a_b + carry[i] + word_max (chip.rs:860-861) */
    H2R_PL_CARRY,                   /* carry[i+1] (div_mod q, chip.rs:864) */
    H2R_PL_CMOD,                    /* c = sum mod 2^w (div_mod r) */
    H2R_PL_NQ1_LO, H2R_PL_NQ1_HI,   /* 2^w * carry[i+1] (chip.rs:1345) */
    H2R_PL_AMNQ1,                   /* sum - nq (chip.rs:1346) */
    H2R_PL_ACCX_LO, H2R_PL_ACCX_HI, /* accumulated_extra + word_max (chip.rs:869-870) */
    H2R_PL_QACC, H2R_PL_MODACC,     /* div_mod of it (chip.rs:871) */
    H2R_PL_NQ2_LO, H2R_PL_NQ2_HI, H2R_PL_AMNQ2,
    H2R_PL_FLAGS, H2R_PL_CARRY_DUP, H2R_PL_CARRY_SUB,
    H2R_PL_COUNT
};

typedef struct h2r_layout {
    uint32_t limb_width, num_limbs, num_cols;
    uint32_t limb_bytes, wide_bytes, carry_bytes; /* flat-stream widths: LIMB, WIDE, CARRY */
    uint32_t limb_sub_bits, limb_nsub;            /* RangeChip::assign(v, limb_width/8, limb_width) */
    uint32_t carry_bits, carry_sub_bits, carry_nsub, carry_sub_stride;
    uint32_t word_max_bits;
    uint32_t reserved0;
    uint64_t record_stride;  /* bytes from one mul_mod record to the next */
    uint64_t stream_bytes;   /* flat-stream (algorithmic) bytes of one mul_mod */
    uint64_t plane_off[H2R_PL_COUNT];
    uint32_t plane_elem[H2R_PL_COUNT];
    uint32_t plane_count[H2R_PL_COUNT];
    /* interleaving of the AB/QN accumulator planes (see above) */
    uint32_t acc_steps_per_group; /* LO rows per group */
    uint32_t acc_lo_row_bytes;    /* bytes between the LO rows of one group */
    uint64_t acc_lo_group_bytes;  /* bytes from one group's LO rows to the next group's */
    uint64_t acc_hi_group_bytes;  /* bytes from one HI row (two steps) to the next */
} h2r_layout;

/* Layout of one element's pow trace: `num_mul_mods` records back to back, then extras.
 *   fixed exponent (pow_mod_fixed_exp): record order = the reference's call order: for each
 *     exponent bit (LSB first) square_mod, then mul_mod if the bit is set (chip.rs:731-740).
 *   variable exponent (pow_mod): per exponent bit mul_mod(acc, squared) then square_mod
 *     (chip.rs:684-694); e_bits[] (one byte per bit) and selected[bit][limb] are the extra planes. */
typedef struct h2r_pow_layout {
    uint32_t num_mul_mods;
    uint32_t num_exp_bits;
    uint64_t elem_stride;     /* bytes from one element's trace to the next */
    uint64_t off_records;     /* record t at off_records + t*record_stride */
    uint64_t off_result;      /* num_limbs limbs: x^e mod n */
    uint64_t off_e_bits;      /* variable exponent only, else UINT64_MAX */
    uint64_t off_selected;    /* variable exponent only: [bit][limb] */
    uint64_t selected_stride; /* bytes between consecutive bits' selected[] */
    uint64_t stream_bytes;    /* flat-stream bytes of one element */
    uint32_t exp_limb_bits;   /* variable exponent only (else 0): bits main_gate.to_bits takes from every exponent limb */
    uint32_t e_num_limbs;     /* variable exponent only (else 0): limbs of the exponent; num_exp_bits = e_num_limbs * exp_limb_bits */
} h2r_pow_layout;

/* ---- context ---------------------------------------------------------------------------------- */

/* BigIntChip::new (big_integer/chip.rs:1174-1185).  Returns H2R_E_SHAPE where the reference
 * asserts bits_len % limb_width == 0 (:1175), H2R_E_FIELD_TOO_SMALL for (:1178), and
 * H2R_E_UNSUPPORTED when no kernel is compiled for the shape. */
int32_t h2r_ctx_create(const h2r_params *params, h2r_ctx **out);
void h2r_ctx_destroy(h2r_ctx *ctx);

/* ---- representation of the field elements that cross the boundary (a property of the CONSUMER, fixed per ctx) -------------
 * The reference hands every value to halo2 as `Value<F>` (big_integer/chip.rs:408, 590, 598; benches/bench.rs:35, 321-329:
 * F = bn256::Fr).  [3P: halo2curves / pasta_curves, not in the reference tree] such an F lives in memory as four 64-bit words in
 * MONTGOMERY form (x * R mod p, R = 2^256), and halo2 keeps ONE CONTIGUOUS VECTOR PER ADVICE COLUMN.  The default image of this
 * library is row-major (five 32-byte cells per 160-byte row) with canonical little-endian integers: what a host-side shim that
 * calls `F::from_repr` per cell wants.  A consumer that takes the witness as columns of F -- a device prover, or a Rust shim that
 * memcpys into `Vec<F>` -- creates its ctx with
 *   H2R_ADVICE_COLUMNS     every image is planar: cell (element e, row r, column c) at
 *                            advice_out + e * out_stride + c * col_stride + r * 32
 *                          col_stride = bytes between an element's column vectors (2^k * 32 for halo2's 2^k-row columns; multiple of
 *                          16); 0 = packed: the five columns of a call's image back to back (col_stride = rows of that image * 32).
 *                          A col_stride that is a multiple of 128 bytes (any 2^k-row column) keeps every column on the 128-byte line
 *                          grid: cells_kernel then writes whole lines (0.80 of the HBM peak); packed columns of a row count that is
 *                          not a multiple of 4 start mid-line and write partial lines (0.57, profiles/r05_cells_representations.txt).
 *                          Both [element][column][row] (out_stride >= 4 col_stride + rows * 32) and [column][element][row]
 *                          (col_stride >= batch * out_stride) arrangements are accepted; a region that starts at row r0 of the
 *                          caller's columns is addressed by passing advice_out + r0 * 32.
 *   H2R_ADVICE_MONTGOMERY  every FIELD ELEMENT that crosses the boundary is x * R mod p: the cells of every image, the selectors of
 *                          h2r_advice_fixed_row[_ex], the table of h2r_lookup_table_image, theta and the A' / S' columns of
 *                          h2r_lookup_permuted_columns.  (Limbs, records, streams and statuses are integers, not field elements:
 *                          unchanged.)
 * Every *_emit_advice export, h2r_pipeline_modpow_public_key_advice and h2r_advice_apply_layout follow the ctx's representation;
 * their `out_stride` stays the ELEMENT stride.  h2r_ctx_create == h2r_ctx_create_ex(params, NULL): row-major, canonical.
 * struct_size must be sizeof(h2r_advice_repr) (H2R_E_UNSUPPORTED otherwise: a caller built against another header);
 * h2r_abi_version() returns the H2R_VERSION the library was built from -- a binder checks it against its own header before it
 * passes any struct (h2r_pow_layout, h2r_verify_layout, h2r_lookup_config, h2r_advice_layout carry no size field). */
#define H2R_ADVICE_COLUMNS 0x400u
#define H2R_ADVICE_MONTGOMERY 0x800u
typedef struct h2r_advice_repr {
    uint32_t struct_size; /* sizeof(h2r_advice_repr) */
    uint32_t flags;       /* H2R_ADVICE_COLUMNS | H2R_ADVICE_MONTGOMERY */
    uint64_t col_stride;  /* H2R_ADVICE_COLUMNS only; 0 = packed */
} h2r_advice_repr;
int32_t h2r_ctx_create_ex(const h2r_params *params, const h2r_advice_repr *repr, h2r_ctx **out);
int32_t h2r_ctx_advice_repr(const h2r_ctx *ctx, h2r_advice_repr *out);
uint32_t h2r_abi_version(void);
/* The build ID of the library: 64 hex digits, the SHA-256 of every source file it was compiled from (halo2_rsa_amd/csrc/ and
 * include/, as halo2_rsa_amd/_build.py hashes them).  `python -m halo2_rsa_amd._build` reuses a shipped libh2r.so exactly when
 * this equals the hash of the tree (file times play no part); a binder can log it next to h2r_abi_version(). */
const char *h2r_build_id(void);

/* BigIntChip::compute_range_lens (big_integer/chip.rs:1220-1249).  Host-only, no ctx needed. */
int32_t h2r_compute_range_lens(uint32_t limb_width, uint32_t num_limbs,
                               uint32_t composition_bit_lens[3], uint32_t overflow_bit_lens[3]);
/* RSAChip::compute_range_lens (src/chip.rs:249-254): appends 32/8 to the composition lens. */
int32_t h2r_rsa_compute_range_lens(uint32_t num_limbs, uint32_t composition_bit_lens[4],
                                   uint32_t overflow_bit_lens[3]);

int32_t h2r_trace_layout(const h2r_ctx *ctx, h2r_layout *out);
int32_t h2r_pow_fixed_layout(const h2r_ctx *ctx, const uint8_t *e_le_bytes, size_t e_len,
                             h2r_pow_layout *out);
int32_t h2r_pow_var_layout(const h2r_ctx *ctx, uint32_t e_num_limbs, uint32_t exp_limb_bits,
                           h2r_pow_layout *out);

/* Scratch needed by one batch call that runs `num_mul_mods` mul_mods per element. */
uint64_t h2r_workspace_bytes(const h2r_ctx *ctx, uint64_t batch, uint32_t num_mul_mods);

/* ---- the hot path ------------------------------------------------------------------------------ */

/* BigIntInstructions::mul_mod (big_integer/chip.rs:542-629).  trace: batch records
 * (record_stride apart).  r_out (nullable): batch*num_limbs limbs of a*b mod n. */
int32_t h2r_mul_mod_batch(const h2r_ctx *ctx, const void *a, const void *b, const void *n,
                          uint64_t batch, uint32_t flags, void *trace, void *r_out,
                          uint8_t *status, void *workspace, h2r_stream_t stream);

/* BigIntInstructions::square_mod (big_integer/chip.rs:642-649) = mul_mod(a, a, n). */
int32_t h2r_square_mod_batch(const h2r_ctx *ctx, const void *a, const void *n, uint64_t batch,
                             uint32_t flags, void *trace, void *r_out, uint8_t *status,
                             void *workspace, h2r_stream_t stream);

/* BigIntInstructions::pow_mod_fixed_exp (big_integer/chip.rs:710-742).  `e` is a HOST buffer:
 * e.to_bytes_le() (chip.rs:719-720), the same exponent for every element (RSAPubE::Fix).
 * trace: batch elements laid out per h2r_pow_fixed_layout().  out (nullable): x^e mod n.
 * Stream-ordered: everything the call queues is ordered within `stream`.  A large call with a trace (more than ~1.5k
 * RSA-1536/2048 elements, ~1.5k RSA-3072/4096 ones) is walked as sub-batches so that a sub-batch's off-circuit chains run
 * next to the previous sub-batch's record writes: for the shapes with a one-launch step (see h2r_pipeline_*) as step
 * launches on `stream` itself, for the others with the record kernels on a side stream owned by the ctx, joined back onto
 * `stream` before the call returns (8,192 RSA-2048 elements: 3.7 -> 4.8 M assigns/s); same results, same buffers, same
 * ordering guarantee.  A LONG exponent (512 bits or more) on a small batch (at most two elements per CU; here and in
 * h2r_pow_mod_batch, the modpow_public_key exports and the pipelined forms) is walked as up to 16 segments of its bits the
 * same way: the records of a segment are written next to the chains of the next one (256 RSA-2048 elements with a 2,048-bit
 * exponent, one call: 17.4 -> 30.3 k assigns/s); the running (squared, acc) pair crosses launches in the workspace (included
 * in h2r_workspace_bytes).  The trace of an element whose status is not H2R_OK is unspecified, as ever. */
int32_t h2r_pow_mod_fixed_exp_batch(const h2r_ctx *ctx, const void *x, const void *n,
                                    const uint8_t *e_le_bytes, size_t e_len, uint64_t batch,
                                    uint32_t flags, void *trace, void *out, uint8_t *status,
                                    void *workspace, h2r_stream_t stream);

/* BigIntInstructions::pow_mod (big_integer/chip.rs:664-696): per-element variable exponent given
 * as e_num_limbs limbs (same limb type as x), each decomposed into exp_limb_bits bits
 * (main_gate.to_bits, chip.rs:677).  An element with an e limb >= 2^exp_limb_bits gets status
 * H2R_E_SHAPE (to_bits cannot be satisfied: the reference's circuit fails). */
int32_t h2r_pow_mod_batch(const h2r_ctx *ctx, const void *x, const void *e_limbs,
                          uint32_t e_num_limbs, uint32_t exp_limb_bits, const void *n,
                          uint64_t batch, uint32_t flags, void *trace, void *out, uint8_t *status,
                          void *workspace, h2r_stream_t stream);

/* RSAInstructions::modpow_public_key (src/chip.rs:99-114) = bigint_chip.assert_in_field(x, n) (:106) followed by
 * pow_mod_fixed_exp (RSAPubE::Fix, :111) or pow_mod (RSAPubE::Var with the chip's exp_limb_bits, :108-110).
 *   in_field_trace (nullable): batch elements holding the assert_in_field witness -- the is_in_field Fresh op
 *     (big_integer/chip.rs:1150-1158 -> 998-1006 -> 908-919) -- laid out and flattened exactly like
 *     h2r_fresh_op_batch(H2R_OP_IS_IN_FIELD, x, n): element stride from h2r_fresh_op_layout, flat stream from
 *     h2r_fresh_op_flatten.  The reference's assignment order is: the in-field stream, then the pow stream
 *     (h2r_pow_trace_flatten of `trace`).  NULL: only the status is produced.
 *   status: H2R_E_NOT_IN_FIELD where x >= n (the in-field witness of such an element is still written; its pow
 *     trace and `out` are not).  `trace` is laid out as for h2r_pow_mod_fixed_exp_batch / h2r_pow_mod_batch. */
int32_t h2r_modpow_public_key_batch(const h2r_ctx *ctx, const void *x, const void *n,
                                    const uint8_t *e_le_bytes, size_t e_len, uint64_t batch,
                                    uint32_t flags, void *trace, void *in_field_trace, void *out,
                                    uint8_t *status, void *workspace, h2r_stream_t stream);
int32_t h2r_modpow_public_key_var_batch(const h2r_ctx *ctx, const void *x, const void *e_limbs,
                                        uint32_t e_num_limbs, uint32_t exp_limb_bits, const void *n,
                                        uint64_t batch, uint32_t flags, void *trace, void *in_field_trace,
                                        void *out, uint8_t *status, void *workspace, h2r_stream_t stream);

/* ---- pipelined form (opt-in): overlap batch k+1's off-circuit chain with batch k's witness emission
 * h2r_pipeline_modpow_public_key() is h2r_modpow_public_key_batch except that the call's TRACE (records and in-field
 * witness) is complete, in `stream` order, only once the NEXT pipelined call has returned or after h2r_pipeline_join();
 * until then it must not be read.  Consecutive calls must use distinct trace / in_field_trace / out / status / workspace
 * buffers (workspace is mandatory here).  The INPUTS x, n (and sig, hashed) are read in `stream` order inside the call that
 * is given them, for every shape: a producer may refill its staging buffers in stream order as soon as the call has
 * returned (the records are written from the workspace, and a call's assert_in_field witness is written by the call's own
 * launches).  Not thread-safe: one pipeline per producer thread.  The chain's results (`out`, `status`) are stream-ordered
 * on `stream` as usual.  h2r_pipeline_join() must be called before a stream that pipelined calls were issued on is
 * destroyed (h2r_pipeline_destroy flushes records still owed on the stream of the last call).
 * How the overlap is obtained depends on the shape:
 *  - RSA-2048, RSA-1024, RSA-3072, RSA-4096 (64-bit limbs: 32 / 16 / 48 / 64 limbs) and 128 x 32-bit limbs (BASELINE config 4), more
 *    than 512 elements per call: ONE launch per call on `stream` (step_kernel; a call above 4,096 elements is
 *    walked as equal parts of at most 4,096, one launch each) whose
 *    workgroups run this call's chains and write this call's in-field witness and the PREVIOUS call's records; the
 *    last call's records go out alone at the join.  Everything is on the caller's stream, no side stream is involved.
 *    EXCEPT RSA-2048 (32 x 64-bit limbs) calls of up to 2,048 elements and RSA-1024 (16 x 64-bit limbs) calls of 1,280 and more on a pipeline
 *    created with side_streams = 2 and depth >= 3: those take the two-queue
 *    form below with the record kernels alternating between the two side streams, so that call k + 1's record kernel starts while
 *    call k's tail workgroups drain -- RSA-2048 5.40-5.47 M assigns/s against 5.2-5.3 M as one-launch steps at 1,024 per call, RSA-1024
 *    15.8-16.7 M against 14.5-15.9 M at 1,536-2,048 per call (the other step shapes are chain-bound enough to lose that way and keep
 *    the step whatever the pipeline's streams).  The overlap needs the
 *    caller's stream and the two side streams on three different HARDWARE queues: HIP hands a process's streams GPU_MAX_HW_QUEUES
 *    queues (default 4) in creation order, so a host that also runs RCCL or many streams of its own should export
 *    GPU_MAX_HW_QUEUES=8 before the runtime initialises (4.8 M against 5.6 M assigns/s per GPU under torchrun without it:
 *    profiles/r04_two_queue.txt), or create the pipeline with one side stream (the one-launch step: 5.5 M).
 *  - every other shape and size: the record-writing kernel runs on a side HIP stream the pipeline owns (created at the
 *    lowest stream priority so that it gets a hardware queue of its own), behind the call's chain kernel, next to the
 *    following call's chain kernel; the in-field witness kernel runs on `stream` right behind the chain kernel.
 *
 * h2r_pipeline_create_ex: `depth` (2..4) = buffer sets the caller rotates through -- call k may reuse
 * the buffers of call k - depth, and `stream` is ordered after call k - depth + 1's records when call k
 * returns; `side_streams` (1 or 2) = streams the record kernels alternate between (with 2, call k+1's
 * record kernel may start while call k's still drains).  h2r_pipeline_create == (depth 2, 1 stream). */
typedef struct h2r_pipeline h2r_pipeline;
int32_t h2r_pipeline_create(const h2r_ctx *ctx, h2r_pipeline **out);
int32_t h2r_pipeline_create_ex(const h2r_ctx *ctx, uint32_t depth, uint32_t side_streams,
                               h2r_pipeline **out);
void h2r_pipeline_destroy(h2r_pipeline *p);
int32_t h2r_pipeline_modpow_public_key(h2r_pipeline *p, const void *x, const void *n,
                                       const uint8_t *e_le_bytes, size_t e_len, uint64_t batch,
                                       uint32_t flags, void *trace, void *in_field_trace, void *out,
                                       uint8_t *status, void *workspace, h2r_stream_t stream);
/* the RSAPubE::Var arm (per-element exponents, as h2r_modpow_public_key_var_batch), pipelined the same way */
int32_t h2r_pipeline_modpow_public_key_var(h2r_pipeline *p, const void *x, const void *e_limbs, uint32_t e_num_limbs,
                                           uint32_t exp_limb_bits, const void *n, uint64_t batch, uint32_t flags, void *trace,
                                           void *in_field_trace, void *out, uint8_t *status, void *workspace, h2r_stream_t stream);
int32_t h2r_pipeline_join(h2r_pipeline *p, h2r_stream_t stream);
/* Which form a pipelined modpow_public_key call of `batch` elements on `stream` takes.  The two-queue form (RSA-2048, calls of up to
 * 2,048; RSA-1024, calls of 1,280 and more; depth >= 3, two side streams) needs the caller's stream and the two side streams on three different HARDWARE queues; the
 * library cannot read HIP's stream -> queue assignment, so the pipeline MEASURES it the first time it meets a caller stream (three
 * 150 us one-wave spinners, one per stream, after synchronising the three streams: ~0.5 ms once per (pipeline, stream); never inside a
 * stream capture, which takes the step) and falls back to the one-launch step when two of them share a queue -- no environment
 * variable is needed for correctness or for the 5.4-5.5 M assigns/s floor; GPU_MAX_HW_QUEUES=8 merely makes the faster form available
 * to a process that has many streams.  The verdict is cached per (pipeline, stream handle); h2r_pipeline_info MEASURES AGAIN every time
 * it is called (it synchronises the three streams: not for a hot path) and refreshes the cache -- call it after creating or destroying
 * streams (HIP may re-assign queues; a destroyed stream's handle can come back for a new stream).  The first pipelined call on a new caller
 * stream therefore synchronises that stream once (~0.5 ms).  Three queues = the spinners' own device-clock stamps show all three running
 * at one instant AND first start to last end is <= 0.157 ms (0.154 measured on three queues, 0.160 when two streams share one; the host-side
 * wall time, which carries launch jitter, only has to be <= 0.195 ms). */
enum { H2R_PIPE_ONE_LAUNCH_STEP = 0, H2R_PIPE_TWO_QUEUE = 1, H2R_PIPE_SIDE_STREAM = 2 };
typedef struct h2r_pipeline_info_t {
    uint32_t struct_size;   /* in: sizeof(h2r_pipeline_info_t) */
    uint32_t depth, side_streams;
    uint32_t record_form;   /* H2R_PIPE_* */
    uint32_t three_queues;  /* 1: the three streams overlap; 0: two of them share a hardware queue; 2: not measured (the shape / size has no two-queue form) */
    float probe_ms;         /* host wall time of the three 150 us spinners, best of three rounds (0.178-0.186 on three queues; 0.199-0.203 when two streams shared one; 0.3-0.45 with one queue) */
    float probe_span_ms;    /* the same round on the device's clock, first start to last end (what decides).  A caller compiled against the struct without this field
                               (struct_size 24) is still served */
} h2r_pipeline_info_t;
int32_t h2r_pipeline_info(h2r_pipeline *p, h2r_stream_t stream, uint64_t batch, h2r_pipeline_info_t *out);

/* ---- multi-GPU: one process per GPU, signatures sharded, RCCL over xGMI behind the C ABI ---------------------------------
 * Signatures are independent: every rank owns a contiguous shard of the batch (h2r_dist_shard_range), runs the
 * exports above on it and keeps its traces resident on its own GPU -- there is NO data-path collective.  What a prover
 * service exchanges is small: the call's parameters (exponent, sizes; a shared modulus) from rank 0, the per-signature
 * results (num_limbs limbs + one status byte) to every rank, and a barrier / MAX for timing.  These exports are that, over
 * RCCL directly (librccl.so.1 is dlopen'ed on first use -- the copy already loaded in the process, e.g. torch's, if any --
 * so the library has no link-time dependency on it); a Rust host needs no torch / MPI for the collectives, only a way to hand
 * rank 0's 128-byte id to the other ranks (its launcher, a file, a socket).  All buffers are DEVICE pointers on the ctx's
 * device; calls enqueue on `stream`.  H2R_E_HIP with h2r_last_hip_error() carrying RCCL's message on failure. */
#define H2R_DIST_ID_BYTES 128u
typedef struct h2r_dist h2r_dist;
int32_t h2r_dist_unique_id(uint8_t id_out[H2R_DIST_ID_BYTES]);                     /* rank 0: ncclGetUniqueId */
int32_t h2r_dist_init(const h2r_ctx *ctx, const uint8_t id[H2R_DIST_ID_BYTES], uint32_t rank, uint32_t world, h2r_dist **out);
void h2r_dist_destroy(h2r_dist *d);
int32_t h2r_dist_version(void);   /* ncclGetVersion of the RCCL these exports run on (major * 10000 + minor * 100 + patch); 0: not loadable */
uint32_t h2r_dist_rank(const h2r_dist *d);
uint32_t h2r_dist_world(const h2r_dist *d);
/* contiguous shards: rank g gets [g * total / world, (g + 1) * total / world), the remainder spread over the low ranks */
int32_t h2r_dist_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi);
int32_t h2r_dist_bcast(h2r_dist *d, void *buf, uint64_t bytes, uint32_t root, h2r_stream_t stream);            /* ncclBroadcast */
/* all-gather of equally sized shards: results_all[rank] = results_shard (shard_elems * num_limbs limbs), status likewise
 * (status_shard / status_all nullable).  256 B + 1 B per RSA-2048 signature: 16.8 MB for BASELINE config 3. */
int32_t h2r_dist_gather_results(h2r_dist *d, const void *results_shard, const uint8_t *status_shard, uint64_t shard_elems,
                                void *results_all, uint8_t *status_all, h2r_stream_t stream);
/* in-place MAX over the ranks of `count` doubles (timing); with count = 0 it is a barrier on `stream` */
int32_t h2r_dist_allreduce_max_f64(h2r_dist *d, double *values, uint64_t count, h2r_stream_t stream);

/* ---- placement-aware trace arena ---------------------------------------------------------------
 * How fast the record kernel writes a trace buffer depends on where the buffer lies physically: for a 1.25 GB region
 * 5.65 .. 6.8 TB/s, stable for the life of the allocation.  The arena maps `candidates` regions of
 * batch * elem_stride bytes (HIP virtual-memory API), times the record kernel on each in the geometry given
 * (records_per_elem records per element from first_record_off, elem_stride apart: e.g. h2r_pow_layout's
 * num_mul_mods / off_records / elem_stride), keeps the `regions` fastest -- h2r_arena_region(a, 0) is the fastest --
 * and gives the others back (rejected candidates keep their memory during the look so that the next candidate lands
 * elsewhere -- at most 64 GB of them and never more than half of the memory free on the device when the call starts).
 * While the kept regions are not of one class (regions of up to 12 GB: the slowest kept more than 4 % behind the fastest) up to
 * 3 x `candidates` further candidates are tried one at a time.
 * While the kept regions are not 7 % faster than the median of everything measured (regions of up to 12 GB) up to four more rounds
 * of `candidates` are tried, each after giving everything back and behind a placeholder allocation of another size (48, 96, ...
 * GB, at most half of the free memory): fresh, unchurned device memory hands out the slow class only.
 * Synchronises `stream`.  The regions are ordinary device memory for every other purpose. */
typedef struct h2r_arena h2r_arena;
int32_t h2r_arena_create(const h2r_ctx *ctx, uint64_t elem_stride, uint64_t first_record_off, uint32_t records_per_elem,
                         uint64_t batch, uint32_t regions, uint32_t candidates, h2r_stream_t stream, h2r_arena **out);
/* The same with a bound on the memory the look may hold: max_look_bytes != 0 caps the rejected candidates kept mapped during the look
 * (besides the `regions` best so far) and skips the placeholder rounds -- for a process that shares the device's memory with others
 * (e.g. max_look_bytes = a tenth of hipMemGetInfo's free bytes).  0 = the policy above. */
int32_t h2r_arena_create_ex(const h2r_ctx *ctx, uint64_t elem_stride, uint64_t first_record_off, uint32_t records_per_elem,
                            uint64_t batch, uint32_t regions, uint32_t candidates, uint64_t max_look_bytes, h2r_stream_t stream,
                            h2r_arena **out);
/* The same look for ANY large buffer the kernels stream into -- advice images, the lookup argument's A' / S' columns: `regions`
 * regions of region_bytes each, the fastest of `candidates` allocations timed with a streaming fill in the product kernels' store
 * pattern (16 bytes per lane, non-temporal); max_look_bytes as for h2r_arena_create_ex.  A prover allocates such buffers once:
 * this is "time a few, keep the fastest" as an export.  The accessors below apply.
 * [measured] The fill's ranking carries over to the cells kernel's images; it does NOT predict h2r_lookup_permuted_columns, whose rate
 * belongs to the PAIR of column buffers it writes (0.65 of the HBM peak on the pair this look kept, 0.77 on the pair fastest under the
 * call itself, same box): for that call time candidate pairs with the call (bench.py --lookup does).  What the measurements show: allocations
 * fall into two placement classes (runs of consecutive allocations), and TWO CONCURRENT store streams -- A' and S', the images of two cells
 * kernels in flight -- are fast exactly when their buffers are of different classes (1.64-1.70 against 2.0-2.3 ms for the lookup call): rank
 * the candidates of one output next to a fixed buffer for the other, under the call that will write them. */
int32_t h2r_image_arena_create(const h2r_ctx *ctx, uint64_t region_bytes, uint32_t regions, uint32_t candidates, uint64_t max_look_bytes,
                               h2r_stream_t stream, h2r_arena **out);
void *h2r_arena_region(const h2r_arena *a, uint32_t i);
uint64_t h2r_arena_region_bytes(const h2r_arena *a);
double h2r_arena_region_ms(const h2r_arena *a, uint32_t i);          /* measured record-kernel time of kept region i */
uint32_t h2r_arena_measurements(const h2r_arena *a, double *ms_out, uint32_t cap);   /* all candidates, allocation order */
void h2r_arena_destroy(h2r_arena *a);

/* How a pipelined (or a large plain) fixed-exponent call of `batch` elements is walked on this ctx: the sizes of the
 * sub-batches that get their own chain and record kernel (one entry = the call is one launch pair), and whether the
 * chain kernels are paced by the record kernels.  pipeline_busy: a record kernel of the previous call is still in
 * flight (the library asks the runtime; here the caller says).  Host-only, works on a context without a device.
 * sizes_out (nullable): up to `cap` entries; *n_out: the number of sub-batches.  (Not covered by this host-only query: an RSA-1024 call
 * above 4,096 that takes the two-queue form -- h2r_pipeline_info says whether it does -- is walked as uniform sub-batches of 2,048, unpaced.) */
int32_t h2r_pipeline_call_plan(const h2r_ctx *ctx, uint64_t batch, uint32_t pipeline_busy, uint64_t *sizes_out,
                               uint32_t cap, uint32_t *n_out, uint32_t *paced_out);

/* How a call of `batch` elements with a LONG exponent is walked on this ctx (see h2r_pow_mod_fixed_exp_batch): the segments of
 * the exponent's bits, each with its own chain kernel and record kernel.  e_le_bytes / e_len: the fixed exponent
 * (RSAPubE::Fix; pow_mod_fixed_exp, big_integer/chip.rs:710-742: one squaring per bit and one multiply per set bit);
 * var_exp_bits != 0: a variable exponent of that many bits instead (pow_mod, chip.rs:664-696: two mul_mods per bit).
 * *n_out: the number of segments (1 = the call is not cut); bit_bounds_out / mul_mod_bounds_out (nullable, up to `cap` entries
 * each): the n + 1 boundaries in exponent bits and in mul_mods per element.  Host-only, works on a context without a device. */
int32_t h2r_exp_segment_plan(const h2r_ctx *ctx, uint64_t batch, const uint8_t *e_le_bytes, size_t e_len, uint32_t var_exp_bits,
                             uint32_t *bit_bounds_out, uint32_t *mul_mod_bounds_out, uint32_t cap, uint32_t *n_out);

/* ---- RSAInstructions::verify_pkcs1v15_signature after the SHA step (src/chip.rs:128-199) ------
 * For every element: the assert_in_field(sig, n) witness (src/chip.rs:106 ->
 * big_integer/chip.rs:1150-1158, 908-919, 310-373, 245-297, 1286-1318, 780-805), the
 * pow_mod_fixed_exp trace, the encoded-message check against `hashed` (4 little-endian 64-bit limbs
 * of the SHA-256 digest per element, src/chip.rs:141-144) and is_valid (one byte per element).
 * limb_width must be 64 (RSAChip::LIMB_WIDTH).  Element layout: the pow trace at offset 0, then the
 * in-field region and the encoded-message region; the two latter hold their flat streams directly,
 * section by section, each section starting on a 16-byte boundary (h2r_verify_trace_flatten packs
 * them).  status: H2R_E_NOT_IN_FIELD elements get is_valid = 0 and no pow / EM trace. */
typedef struct h2r_verify_layout {
    h2r_pow_layout pow;
    uint64_t off_in_field, in_field_stream_bytes;
    uint64_t off_em, em_stream_bytes;
    uint64_t elem_stride;
    uint64_t stream_bytes; /* in-field + pow + EM flat streams, in the reference's order */
} h2r_verify_layout;
int32_t h2r_verify_layout_fixed(const h2r_ctx *ctx, const uint8_t *e_le_bytes, size_t e_len,
                                h2r_verify_layout *out);
int32_t h2r_verify_pkcs1v15_batch(const h2r_ctx *ctx, const void *sig, const void *n,
                                  const uint8_t *e_le_bytes, size_t e_len, const uint64_t *hashed,
                                  uint64_t batch, uint32_t flags, void *trace, void *powed_out,
                                  uint8_t *is_valid_out, uint8_t *status, void *workspace,
                                  h2r_stream_t stream);
/* the RSAPubE::Var arm (src/chip.rs:108-110): e_limbs = e_num_limbs limbs per element, as h2r_pow_mod_batch takes them; the chip's
 * exp_limb_bits.  An exponent limb wider than exp_limb_bits gets H2R_E_SHAPE (to_bits cannot be satisfied). */
int32_t h2r_verify_layout_var(const h2r_ctx *ctx, uint32_t e_num_limbs, uint32_t exp_limb_bits, h2r_verify_layout *out);
int32_t h2r_verify_pkcs1v15_var_batch(const h2r_ctx *ctx, const void *sig, const void *n, const void *e_limbs,
                                      uint32_t e_num_limbs, uint32_t exp_limb_bits, const uint64_t *hashed, uint64_t batch,
                                      uint32_t flags, void *trace, void *powed_out, uint8_t *is_valid_out, uint8_t *status,
                                      void *workspace, h2r_stream_t stream);
int32_t h2r_verify_trace_flatten(const h2r_ctx *ctx, const h2r_verify_layout *vl,
                                 const void *elem_host, void *stream_out);
/* Pipelined form of h2r_verify_pkcs1v15_batch (see h2r_pipeline_create).  Where the call is issued as one-launch steps and a call is
 * in flight, the chain role of the step launch writes the element's in-field / encoded-message witness and is_valid itself (no kernel
 * of its own); otherwise that kernel runs on `stream` right behind the chain kernel.  Either way powed_out / is_valid_out / status are
 * stream-ordered on `stream`, the trace follows the pipeline's join rule. */
int32_t h2r_pipeline_verify_pkcs1v15(h2r_pipeline *p, const void *sig, const void *n,
                                     const uint8_t *e_le_bytes, size_t e_len, const uint64_t *hashed,
                                     uint64_t batch, uint32_t flags, void *trace, void *powed_out,
                                     uint8_t *is_valid_out, uint8_t *status, void *workspace,
                                     h2r_stream_t stream);

int32_t h2r_pipeline_verify_pkcs1v15_var(h2r_pipeline *p, const void *sig, const void *n, const void *e_limbs,
                                         uint32_t e_num_limbs, uint32_t exp_limb_bits, const uint64_t *hashed,
                                         uint64_t batch, uint32_t flags, void *trace, void *powed_out,
                                         uint8_t *is_valid_out, uint8_t *status, void *workspace, h2r_stream_t stream);

/* ---- the CALLER of the path: RSASignatureVerifier::verify_pkcs1v15_signature (src/lib.rs:183-246) ----------
 * The reference hashes the signed message bytes with its SHA-256 chip (:205-209), reverses the 32 digest bytes (:210-213),
 * composes them eight at a time into four 64-bit limbs -- limb_val = 0; limb_val = mul_add(2^(8j), hashed_bytes[8i + j], limb_val)
 * (:225-239) -- and hands that integer to RSAChip::verify_pkcs1v15_signature (:240-241); it returns (is_valid, hashed_bytes in
 * digest order) (:243-245).  Here, per element, on the device:
 *   h2r_sha256_hashed_msg_batch   SHA-256 (FIPS 180-4; the values the sha2 crate / the SHA chip's digest cells hold) of message e =
 *                                 msgs[msg_off[e], msg_off[e+1]) (msg_off: batch + 1 byte offsets in device memory; NULL: every
 *                                 message has fixed_len bytes, message e at e * fixed_len; empty messages are fine).
 *                                 digest_out (nullable): 32 bytes per element, digest order.  hashed_out (nullable): the 4 limbs.
 *                                 hm_trace (nullable): the step's flat stream, H2R_HASHED_MSG_STREAM_BYTES per element every
 *                                 hm_stride bytes (0 = packed): the 32 reversed byte cells (1 byte each), then the 32 limb_val
 *                                 cells the mul_add chain assigns, 8 bytes little-endian each.  The constants are not streamed.
 *                                 The SHA chip's own cells (its message schedule / compression witness: third-party Table16
 *                                 circuit) are NOT produced.  Output pointers and hm_stride must be multiples of 16.
 *   h2r_signature_verifier_batch  that, then h2r_verify_pkcs1v15_batch on hashed_out (required), in stream order: the whole
 *                                 RSASignatureVerifier call from message and signature limbs to is_valid.
 *   h2r_hashed_msg_*advice*       the limb composition as advice rows (4 x (CONST0 + 8 x (CONST_COEFF8 + j, MUL_ADD)) = 68 rows),
 *                                 placed in front of the h2r_verify_emit_advice rows in the reference's region (:220-241). */
#define H2R_SHA256_DIGEST_BYTES 32u
#define H2R_HASHED_MSG_STREAM_BYTES 288u
int32_t h2r_sha256_hashed_msg_batch(const h2r_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, uint64_t fixed_len,
                                    uint64_t batch, uint8_t *digest_out, uint64_t *hashed_out, void *hm_trace,
                                    uint64_t hm_stride, h2r_stream_t stream);
int32_t h2r_signature_verifier_batch(const h2r_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, uint64_t fixed_len,
                                     const void *sig, const void *n, const uint8_t *e_le_bytes, size_t e_len, uint64_t batch,
                                     uint32_t flags, void *trace, void *hm_trace, uint64_t hm_stride, uint8_t *digest_out,
                                     uint64_t *hashed_out, void *powed_out, uint8_t *is_valid_out, uint8_t *status,
                                     void *workspace, h2r_stream_t stream);
/* Pipelined form (see h2r_pipeline_create): the SHA-256 / hashed-message step of THIS call's messages rides on the call's step launch
 * as one more role -- next to the records of the previous call, at no cost to the step -- where the shape has one-launch steps and a
 * call is in flight (the chain role of the same launch then consumes its limbs inside the launch: message count + acquire at agent
 * scope); otherwise it is a kernel of its own on `stream` in front of the call.  The messages are read in stream order
 * inside the call, like every other input; digest_out / hashed_out / hm_trace / powed_out / is_valid_out / status are
 * stream-ordered on `stream`, the trace follows the pipeline's join rule. */
int32_t h2r_pipeline_signature_verifier(h2r_pipeline *p, const uint8_t *msgs, const uint64_t *msg_off, uint64_t fixed_len,
                                        const void *sig, const void *n, const uint8_t *e_le_bytes, size_t e_len, uint64_t batch,
                                        uint32_t flags, void *trace, void *hm_trace, uint64_t hm_stride, uint8_t *digest_out,
                                        uint64_t *hashed_out, void *powed_out, uint8_t *is_valid_out, uint8_t *status,
                                        void *workspace, h2r_stream_t stream);
uint32_t h2r_hashed_msg_advice_rows(const h2r_ctx *ctx);
int32_t h2r_hashed_msg_row_kinds(const h2r_ctx *ctx, uint8_t *kinds_out);
int32_t h2r_hashed_msg_emit_advice(const h2r_ctx *ctx, const void *hm_trace, uint64_t hm_stride, uint64_t batch,
                                   const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream);

/* ---- the Fresh-integer family of BigIntInstructions ---------------------------
 * add (big_integer/chip.rs:245-297), sub (:310-373; flag = is_overflowed, value = |a-b|), add_mod (:452-481),
 * sub_mod (:495-528), is_zero (:754-767), is_equal_fresh (:780-805), is_less_than (:908-919),
 * is_less_than_or_equal (:932-941), is_greater_than (:954-963), is_greater_than_or_equal (:976-985),
 * is_in_field (:998-1006), all on num_limbs-limb operands.  The element trace holds the op's flat stream
 * section by section (16-byte aligned sections; h2r_fresh_op_flatten packs them).  value_out
 * (nullable): value_limbs limbs per element (num_limbs+1 for add/sub, num_limbs for add_mod/sub_mod);
 * flag_out (nullable): the predicate / overflow bit.  As in the reference, sub's overflow bit is 1 iff
 * a <= b, so add_mod returns a+b un-reduced when a+b == n and sub_mod(a, a, n) returns n.
 * status: H2R_E_NOT_IN_FIELD where sub_mod's assert_zero(is_overflowed2) (:510) fails, H2R_E_NOT_REDUCED
 * where the high limbs of an add_mod/sub_mod result are non-zero (:475-478, :522-525).
 * flags: H2R_F_SHARED_MODULUS -- add_mod / sub_mod: `n` is ONE integer used by every element; the ops without an `n`
 * (comparisons, is_in_field(a, b = modulus)): `b` is ONE integer used by every element.  Without the flag `b` (and `n`)
 * hold `batch` integers. */
enum { H2R_OP_ADD = 0, H2R_OP_SUB, H2R_OP_ADD_MOD, H2R_OP_SUB_MOD, H2R_OP_IS_ZERO, H2R_OP_IS_EQUAL_FRESH,
       H2R_OP_IS_LESS_THAN, H2R_OP_IS_LESS_THAN_OR_EQUAL, H2R_OP_IS_GREATER_THAN,
       H2R_OP_IS_GREATER_THAN_OR_EQUAL, H2R_OP_IS_IN_FIELD, H2R_OP_COUNT };
int32_t h2r_fresh_op_layout(const h2r_ctx *ctx, uint32_t op, uint64_t *elem_stride, uint64_t *stream_bytes,
                            uint32_t *value_limbs);
int32_t h2r_fresh_op_batch(const h2r_ctx *ctx, uint32_t op, const void *a, const void *b, const void *n,
                           uint64_t batch, uint32_t flags, void *trace, void *value_out, uint8_t *flag_out,
                           uint8_t *status, h2r_stream_t stream);
int32_t h2r_fresh_op_flatten(const h2r_ctx *ctx, uint32_t op, const void *elem_host, void *stream_out);

/* ---- Muled integers: BigIntInstructions::mul / square, is_equal_muled, refresh ----
 * A Muled integer (the un-carried product columns, big_integer/mod.rs:216-232) is passed as 2*num_limbs
 * columns of 4 x uint64_t (256-bit little-endian) per element; column 2*num_limbs-1 is zero.
 *  h2r_mul_batch            BigIntChip::mul (chip.rs:386-419; square = mul(a, a)): trace = one record per
 *                           element with only the AB planes written; stream = its L*L accumulators.
 *  h2r_is_equal_muled_batch BigIntChip::is_equal_muled (chip.rs:822-895, num_limbs_l = num_limbs_r =
 *                           num_limbs): trace = one record per element with only the per-column planes
 *                           written; eq_out[elem] = the returned bit (assert_equal_muled = bit must be 1).
 *  h2r_refresh_batch        BigIntChip::refresh (chip.rs:168-233) with RefreshAux::new(limb_width, L, L)
 *                           (mod.rs:428-482): 2*num_limbs Fresh limbs; the element trace (stride
 *                           h2r_refresh_stream_bytes rounded up to 256) IS the flat stream. */
uint64_t h2r_mul_stream_bytes(const h2r_ctx *ctx);
uint64_t h2r_is_equal_muled_stream_bytes(const h2r_ctx *ctx);
uint64_t h2r_refresh_stream_bytes(const h2r_ctx *ctx);
int32_t h2r_mul_batch(const h2r_ctx *ctx, const void *a, const void *b, uint64_t batch, void *trace,
                      uint64_t *muled_out, h2r_stream_t stream);
int32_t h2r_mul_trace_flatten(const h2r_ctx *ctx, const void *record_host, void *stream_out);
int32_t h2r_is_equal_muled_batch(const h2r_ctx *ctx, const uint64_t *muled_a, const uint64_t *muled_b,
                                 uint64_t batch, void *trace, uint8_t *eq_out, h2r_stream_t stream);
int32_t h2r_is_equal_muled_flatten(const h2r_ctx *ctx, const void *record_host, void *stream_out);
int32_t h2r_refresh_batch(const h2r_ctx *ctx, const uint64_t *muled, uint64_t batch, void *trace,
                          void *fresh_out, uint8_t *status, h2r_stream_t stream);

/* General operand shapes (the reference's mul takes d0 != d1 limbs, big_integer/chip.rs:395-397; refresh any
 * RefreshAux::new(limb_width, n_l, n_r), mod.rs:428 + chip.rs:178-181; is_equal_muled n_l != n_r, chip.rs:822-835), for
 * operands of at most the ctx's num_limbs limbs (the stream's WIDE / CARRY widths are the ctx's).  Muled integers are passed
 * as muled_stride_cols columns of 4 x uint64_t per element, of which the first n_l + n_r - 1 are read.
 *  h2r_mul_batch_ex            a: [batch][d0] limbs, b: [batch][d1] limbs; trace = one (num_limbs-shaped) record per element,
 *                              muled_out [batch][2 * num_limbs][4] with d0 + d1 - 1 columns of the product;
 *                              h2r_mul_trace_flatten_ex walks the record in the reference's (d0, d1) order.
 *  h2r_refresh_batch_ex        any (n_l, n_r): h2r_refresh_layout gives the number of Fresh limbs (RefreshAux), the stream size and
 *                              the element stride of `trace` (which IS the flat stream); status H2R_E_NOT_REDUCED where a limb
 *                              does not fit increased_limbs_vec (assert_zero, chip.rs:213).  h2r_refresh_batch == (L, L).
 *  h2r_is_equal_muled_batch_ex word_max = compute_mul_word_max(limb_width, min(n_l, n_r)) (chip.rs:838), carry range check of
 *                              bits(2 word_max) - limb_width bits; stream_out: element e's flat stream at + e * out_stride
 *                              (a multiple of 16); flags: H2R_STREAM_FIELD_AB as for the emitters; eq_out nullable. */
uint64_t h2r_mul_stream_bytes_ex(const h2r_ctx *ctx, uint32_t d0, uint32_t d1);
int32_t h2r_mul_batch_ex(const h2r_ctx *ctx, const void *a, uint32_t d0, const void *b, uint32_t d1, uint64_t batch, void *trace,
                         uint64_t *muled_out, h2r_stream_t stream);
int32_t h2r_mul_trace_flatten_ex(const h2r_ctx *ctx, const void *record_host, uint32_t d0, uint32_t d1, void *stream_out);
int32_t h2r_refresh_layout(const h2r_ctx *ctx, uint32_t num_limbs_l, uint32_t num_limbs_r, uint32_t *num_limbs_fresh,
                           uint64_t *stream_bytes, uint64_t *elem_stride);
int32_t h2r_refresh_batch_ex(const h2r_ctx *ctx, const uint64_t *muled, uint64_t muled_stride_cols, uint32_t num_limbs_l,
                             uint32_t num_limbs_r, uint64_t batch, void *trace, void *fresh_out, uint8_t *status, h2r_stream_t stream);
uint64_t h2r_is_equal_muled_stream_bytes_ex(const h2r_ctx *ctx, uint32_t num_limbs_l, uint32_t num_limbs_r, uint32_t flags);
int32_t h2r_is_equal_muled_batch_ex(const h2r_ctx *ctx, const uint64_t *muled_a, const uint64_t *muled_b, uint64_t muled_stride_cols,
                                    uint32_t num_limbs_l, uint32_t num_limbs_r, uint64_t batch, uint32_t flags, void *stream_out,
                                    uint64_t out_stride, uint8_t *eq_out, h2r_stream_t stream);

/* ---- the lookup range-check batch --------------------------------------------------------------
 * RangeChip::assign(value, sublimb_bits, bit_len) decomposition of `count` values of `value_bytes`
 * bytes each (8 or 16) into ceil(bit_len/sublimb_bits) one-byte sub-limbs (stride sub_stride),
 * plus (hist nullable) the multiplicity of every (tag, value) lookup-table row:
 * hist[0 .. 2^sublimb_bits) for the composition table, then 2^(bit_len % sublimb_bits) entries
 * for the overflow table if any (uint32 counters, ADDED to). */
int32_t h2r_range_decompose_batch(const h2r_ctx *ctx, const void *values, uint32_t value_bytes,
                                  uint64_t count, uint32_t bit_len, uint32_t sublimb_bits,
                                  uint8_t *sublimbs_out, uint32_t sub_stride, uint32_t *hist,
                                  h2r_stream_t stream);

/* Multiplicities of every lookup-table row touched by the range checks of `num_records` mul_mod
 * records (q, r limbs and carries): hist_out[elem][h2r_hist_len()] uint32, where `records_per_elem`
 * consecutive records belong to one element (= one circuit).  Row order: the composition table of
 * limb_sub_bits, then of carry_sub_bits if different, then the carry overflow table if any. */
uint32_t h2r_hist_len(const h2r_ctx *ctx);
int32_t h2r_trace_lookup_hist(const h2r_ctx *ctx, const void *trace, uint64_t first_record_off,
                              uint64_t elem_stride, uint64_t num_elems, uint32_t records_per_elem,
                              uint32_t *hist_out, h2r_stream_t stream);

/* Grouped arrangement of each element's lookup inputs (stable counting sort by table row; the
 * theta-independent core of the halo2 lookup permutation, which itself is third-party and unpinned).
 * Cell ids follow the flat stream: record t contributes q sub-limbs, r sub-limbs, carry sub-limbs
 * (h2r_lookups_per_record() cells).  perm_out[elem][k] = id of the cell at sorted position k;
 * rows_out (nullable) [elem][k] = its table row in h2r_trace_lookup_hist order. */
uint32_t h2r_lookups_per_record(const h2r_ctx *ctx);
int32_t h2r_trace_lookup_permutation(const h2r_ctx *ctx, const void *trace, uint64_t first_record_off,
                                     uint64_t elem_stride, uint64_t num_elems, uint32_t records_per_elem,
                                     uint32_t *perm_out, uint16_t *rows_out, h2r_stream_t stream);
/* The same with the multiplicities as a by-product: hist_out (nullable) [elem][h2r_hist_len()] receives exactly what
 * h2r_trace_lookup_hist writes (the permutation's counting pass produces them), saving that launch. */
int32_t h2r_trace_lookup_permutation_hist(const h2r_ctx *ctx, const void *trace, uint64_t first_record_off,
                                          uint64_t elem_stride, uint64_t num_elems, uint32_t records_per_elem,
                                          uint32_t *perm_out, uint16_t *rows_out, uint32_t *hist_out,
                                          h2r_stream_t stream);

/* ---- halo2's lookup argument for the range checks: table image, per-argument multiplicities, permuted columns A' / S' ----
 * What `range_chip.load_table` + `create_proof` do with the sub-limb cells (reference benches/bench.rs:141-142, 321-329; the
 * RangeChip call sites big_integer/chip.rs:74, 590, 598, 880-885).  THIRD-PARTY behaviour (maingate / halo2wrong rev 63bde545,
 * halo2's plonk::lookup::prover -- neither is in the reference tree), restated in DESIGN.md section 2c; parity is pinned against
 * a Python restatement of the same algorithm run on the ORACLE's cells (tests/advice_ref.py), not against the upstream code.
 *  - RangeChip::configure gives every distinct nonzero bit length of composition_bit_lens + overflow_bit_lens a TAG and
 *    load_table writes ONE (tag, value) table: row 0 = (0, 0), then, bit lengths ascending, (tag_b, 0 .. 2^b - 1).
 *    h2r_lookup_config_default: tags 1, 2, ... in ascending bit-length order for BigIntChip::compute_range_lens
 *    (big_integer/chip.rs:1220-1249) [+ RSAChip's 32/8, src/chip.rs:252 when rsa_chip != 0]; h2r_lookup_config_custom: the
 *    bit_len -> tag map of whatever maingate revision the caller links (a shim reads it from its RangeConfig).
 *  - FIVE lookup arguments read the main gate's advice columns: composition_a..d = columns a..d under the fixed column
 *    tag_composition, overflow_a = column a under tag_overflow; a row with the lookup off contributes (0, 0).
 *    RangeChip::assign(v, s, bit_len) = main_gate.decompose: rows of four s-bit terms in columns a..d (the LAST row reversed so
 *    that the last term -- the overflow sub-limb, if bit_len % s != 0 -- is in column a; missing terms are zero cells).
 *    h2r_lookup_hist_*: multiplicity of every table row per element (= circuit) and argument, hist[elem][5][n_rows] uint32,
 *    ADDED to (accumulate the mul_mod records, then the range-assigned inputs / any other range-checked values of the circuit).
 *  - h2r_lookup_permuted_columns: for every element and argument, halo2's permute_expression_pair on the usable rows: inputs and
 *    table compressed with the element's challenge (tag * theta + value), A' = the inputs sorted by the field's Ord (order of the
 *    canonical integers), S'[i] = A'[i] on the first row of every run of A', the leftover table values elsewhere (ascending,
 *    handed out from the LAST repeated row backwards, as the upstream Vec::pop does); the table column is the table's rows
 *    followed by its default row (0, 0).  The blinding tail (random) is the caller's.  Output: 32-byte little-endian elements
 *    (canonical; of a H2R_ADVICE_MONTGOMERY ctx: Montgomery form, and theta is then given in Montgomery form too -- the sort
 *    order is that of the canonical integers either way); element e, argument k at + e * out_elem_stride + k * usable_rows * 32
 *    in a_perm_out and in s_perm_out (planar already: one contiguous vector per argument).
 *    At most 65,535 circuits per call.  theta: [num_elems][4] uint64 on the device, canonical (< p) little-endian (every proof has its own challenge).  status
 *    (nullable, [num_elems]): H2R_E_SHAPE where the lookup inputs or the table do not fit usable_rows or theta is not canonical
 *    (that circuit's columns are left untouched).  arg_mask: bit k = produce argument k. */
#define H2R_LOOKUP_ARGS 5u
#define H2R_LOOKUP_MAX_LENS 8u
enum { H2R_LOOKUP_COMPOSITION_A = 0, H2R_LOOKUP_COMPOSITION_B, H2R_LOOKUP_COMPOSITION_C, H2R_LOOKUP_COMPOSITION_D, H2R_LOOKUP_OVERFLOW_A };
typedef struct h2r_lookup_config {
    uint32_t n_lens;                          /* distinct nonzero bit lengths */
    uint32_t bit_len[H2R_LOOKUP_MAX_LENS];    /* ascending */
    uint32_t tag[H2R_LOOKUP_MAX_LENS];        /* RangeConfig::bit_len_tag */
    uint32_t row_off[H2R_LOOKUP_MAX_LENS];    /* table row of (tag, 0) */
    uint32_t n_rows;                          /* 1 + sum of 2^bit_len (at most 1,024) */
} h2r_lookup_config;
int32_t h2r_lookup_config_default(const h2r_ctx *ctx, uint32_t rsa_chip, h2r_lookup_config *out);
int32_t h2r_lookup_config_custom(const uint32_t *bit_lens, const uint32_t *tags, uint32_t n, h2r_lookup_config *out);
/* the table `load_table` writes: tag_col / value_col receive n_rows canonical field elements (4 x uint64, HOST buffers) */
int32_t h2r_lookup_table_image(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint64_t *tag_col, uint64_t *value_col);
int32_t h2r_lookup_hist_records(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *trace, uint64_t first_record_off,
                                uint64_t elem_stride, uint64_t num_elems, uint32_t records_per_elem, const uint8_t *status,
                                uint32_t *hist, h2r_stream_t stream);
/* RangeChip::assign(value, sublimb_bits, bit_len) of values_per_elem values (value_bytes = 4, 8 or 16) per element, e.g. the
 * limbs of assign_integer(x), assign_integer(n) (big_integer/chip.rs:71-76: sublimb_bits = limb_width / 8, bit_len = limb_width) */
int32_t h2r_lookup_hist_values(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *values, uint32_t value_bytes,
                               uint64_t values_per_elem, uint64_t num_elems, uint32_t bit_len, uint32_t sublimb_bits,
                               uint32_t *hist, h2r_stream_t stream);
/* the same for values that are not packed: element e's value k at values + e * elem_stride + k * value_stride (bytes; multiples of 4
 * for 4-byte values, of 8 otherwise); elements with a nonzero status byte (nullable) are skipped */
int32_t h2r_lookup_hist_values_strided(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const void *values, uint32_t value_bytes,
                                       uint64_t values_per_elem, uint64_t num_elems, uint64_t elem_stride, uint64_t value_stride,
                                       uint32_t bit_len, uint32_t sublimb_bits, const uint8_t *status, uint32_t *hist,
                                       h2r_stream_t stream);
/* every lookup the witness of one verify_pkcs1v15_signature element holds (trace of h2r_verify_pkcs1v15_batch / h2r_signature_verifier_batch):
 * the range assigns inside assert_in_field (= h2r_lookup_hist_fresh_op), the q / r limbs and carries of every mul_mod record
 * (= h2r_lookup_hist_records) and the two RangeChip::assign(half, 4, 32) of the encoded-message check (src/chip.rs:170-171; cfg
 * must hold RSAChip's 4-bit length: h2r_lookup_config_default(ctx, 1, ..)).  In-field rows are counted for every element (that
 * witness is written whatever the status), records and EM rows only where status is 0.  What the caller adds: assign_integer of
 * the signature and of n (h2r_lookup_hist_values). */
int32_t h2r_lookup_hist_verify(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const h2r_verify_layout *vl, const void *trace,
                               uint64_t num_elems, const uint8_t *status, uint32_t *hist, h2r_stream_t stream);
/* the range assigns INSIDE a Fresh-op witness (h2r_fresh_op_batch's trace; op = H2R_OP_IS_IN_FIELD for the in_field_trace of
 * h2r_modpow_public_key_batch, or for the in-field region of a verify element with first_off = h2r_verify_layout.off_in_field):
 * add's c / carry per limb (big_integer/chip.rs:279-282) and sub_unchecked's difference limbs (:1307-1308), each a
 * RangeChip::assign(limb, limb_width / 8, limb_width) */
int32_t h2r_lookup_hist_fresh_op(const h2r_ctx *ctx, const h2r_lookup_config *cfg, uint32_t op, const void *trace, uint64_t first_off,
                                 uint64_t elem_stride, uint64_t num_elems, uint32_t *hist, h2r_stream_t stream);
/* The multiplicities of an advice IMAGE (any *_emit_advice / pipelined advice export's output, in the ctx's representation, under `layout`;
 * kinds_dev: the rows' kinds on the device as for h2r_advice_check): every row whose kind enables the composition lookup adds its cells
 * a..d to arguments 0..3, every row with the overflow lookup its cell a to argument 4 -- what h2r_lookup_hist_verify / _records / _fresh_op
 * count from a trace, for a witness that has no records (h2r_pipeline_*_advice).  A cell that is not a row of the table is not counted
 * (h2r_advice_check reports it).  Elements with a nonzero status byte (nullable) are skipped.  What the caller still adds: the
 * assign_integer range assigns of the circuit's inputs (h2r_lookup_hist_values), which are not rows of these images.
 * 0.32 ms per 1,024 RSA-2048 modpow_public_key elements (12.6 GB image; 0.58 ms in planar Montgomery form). */
struct h2r_advice_layout;
int32_t h2r_lookup_hist_advice(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const struct h2r_advice_layout *layout, const uint8_t *kinds_dev,
                               uint64_t rows, const void *image, uint64_t image_stride, uint64_t batch, const uint8_t *status, uint32_t *hist,
                               h2r_stream_t stream);
uint64_t h2r_lookup_workspace_bytes(const h2r_lookup_config *cfg, uint64_t num_elems);
int32_t h2r_lookup_permuted_columns(const h2r_ctx *ctx, const h2r_lookup_config *cfg, const uint32_t *hist, const uint64_t *theta,
                                    uint64_t num_elems, uint32_t usable_rows, uint32_t arg_mask, void *a_perm_out,
                                    void *s_perm_out, uint64_t out_elem_stride, uint8_t *status, void *workspace,
                                    h2r_stream_t stream);
/* Arithmetic of the ctx's field on canonical elements (host): op 0 = a + b, 1 = a - b, 2 = a * b, 3 = a^-1 (b ignored; a != 0;
 * binary extended Euclid), 4 = a^(p-2) (Fermat: the cross-check of 3), 5 = a^-1 as the kernels compute main_gate.is_zero's witness
 * (classical Euclid on (p, s) when a = +-s with s < 2^64 -- the only differences this path produces --, op 3 otherwise),
 * 6 = a * R mod p (R = 2^256: into the Montgomery form of H2R_ADVICE_MONTGOMERY, by the short product the kernels use for a cell),
 * 7 = a * R^-1 mod p (a Montgomery-form element back to its canonical integer), 8 = a * R mod p by the generic R^2 product,
 * 9 = a * R mod p in radix 2^30 (carry-free multiply-add columns: what the cells kernel runs).
 * The same code the kernels run (lookup compression, main_gate.is_zero's inverse witness, the cells' representation). */
int32_t h2r_field_eval(const h2r_ctx *ctx, uint32_t op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);

/* ---- host-side helpers (no device work) --------------------------------------------------------
 * h2r_trace_flatten: walk ONE record (host copy, record_stride bytes) in the reference's assignment
 * order and write its flat op-trace stream (layout.stream_bytes bytes; widths in h2r_layout).
 * This is the order in which a layouter shim assigns the values to cells. */
int32_t h2r_trace_flatten(const h2r_ctx *ctx, const void *record_host, void *stream_out);
int32_t h2r_pow_trace_flatten(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *elem_host,
                              void *stream_out);

/* ---- flat stream in HBM: the device-side flatten ----------------------------------------------------
 * h2r_trace_flatten walks ONE host copy; these exports produce the same bytes for every record of a batch on the
 * device, so that a layouter shim / advice-column builder consumes the witness in the reference's assignment order at
 * HBM speed.  Element e's stream is written at stream_out + e * out_stride + out_off (any byte alignment).
 *   flags  H2R_STREAM_FIELD_AB: a_b = a[i] - b[i] (big_integer/chip.rs:859, a FIELD subtraction) is emitted as the
 *          canonical element of the ctx's field, 32 bytes little-endian (x >= 0 -> x, x < 0 -> p - |x|), instead of the
 *          WIDE two's complement integer; every other value is < 2^135 << p, so its integer IS its canonical element.
 *          The *_ex host walks take the same flags (flags = 0: identical to the plain forms).
 *   h2r_trace_emit_stream      mul_mod batch: record i at trace + i * record_stride.
 *   h2r_pow_trace_emit_stream  pow traces laid out per `pl`; elem_stride = 0 means pl->elem_stride (pass the enclosing
 *                              element's stride for a pow trace embedded in a verify element).  Byte-equal to
 *                              h2r_pow_trace_flatten_ex of every element. */
#define H2R_STREAM_FIELD_AB 1u
uint64_t h2r_stream_bytes(const h2r_ctx *ctx, uint32_t flags);                         /* one mul_mod record */
uint64_t h2r_pow_stream_bytes(const h2r_ctx *ctx, const h2r_pow_layout *pl, uint32_t flags);   /* one pow element */
int32_t h2r_trace_flatten_ex(const h2r_ctx *ctx, const void *record_host, uint32_t flags, void *stream_out);
int32_t h2r_pow_trace_flatten_ex(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *elem_host, uint32_t flags,
                                 void *stream_out);
int32_t h2r_trace_emit_stream(const h2r_ctx *ctx, const void *trace, uint64_t num_records, uint32_t flags,
                              void *stream_out, uint64_t out_stride, uint64_t out_off, h2r_stream_t stream);
int32_t h2r_pow_trace_emit_stream(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *trace, uint64_t elem_stride,
                                  uint64_t batch, uint32_t flags, void *stream_out, uint64_t out_stride,
                                  uint64_t out_off, h2r_stream_t stream);

/* ---- advice-column image: the witness as rows of the main gate's five advice columns ---------------------------
 * What a halo2 prover consumes (reference benches/bench.rs:141-142, 321-329 via RangeChip::assign / main_gate ops,
 * big_integer/chip.rs:74, 408, 590, 598, 880-885) is not a value stream but advice columns of field elements.  These
 * exports write, for every mul_mod record of a batch, the rows of a 5-column image in HBM: row = 5 cells (columns a..e)
 * of 32 bytes, each the canonical little-endian element of the ctx's field; one row per main-gate op and
 * ceil(sub-limbs / 4) rows per RangeChip::assign, in the reference's op order.  Row shapes: DESIGN.md section 2b.  The
 * VALUES are the ones the flat stream pins; the third-party PLACEMENT (maingate / halo2wrong are not in the reference
 * tree) is restated from SURVEY Appendix A and is unpinned.
 *   h2r_advice_rows(ctx)            rows of one mul_mod (3,974 for RSA-2048 as 32 x 64-bit limbs): EVERY cell the ops assign --
 *                                   the flat stream's values, the assign_constant cells, the assign_bit(1) seeds, is_zero's inverses
 *   element e's image starts at advice_out + e * out_stride; record t of the element at + (pre + t * rows) * 160 bytes
 *   h2r_pow_trace_emit_advice: a pow element = [acc = assign_constant(1): CONST1 [1], CONST0 [0]] then its records back to back
 *   (pow_mod_fixed_exp, big_integer/chip.rs:729-740).  A VARIABLE-exponent element (pow_mod, :674-694) is
 *       [main_gate.to_bits(limb, exp_limb_bits) of every exponent limb: exp_limb_bits x BIT [b, b, b], ceil(exp_limb_bits / 4) x
 *        BITS_COMPOSE [four bits, what remains to be composed] (the last row reversed and zero-padded, as RangeChip's rows), ASSERT_EQ [result, limb]]
 *       [CONST1, CONST0]  per exponent bit: [mul_mod(acc, squared) rows] [num_limbs x SELECT [e_bit, muled_j, e_bit, acc_j, selected_j] :688-691]
 *       [square_mod rows];  its trace must be given (the e_bits / selected planes live there), also with H2R_ADVICE_DIRECT.
 *   h2r_pow_advice_rows / h2r_pow_row_kinds: rows of one such element and their kinds.
 *   h2r_mul_mod_emit_advice         records of h2r_mul_mod_batch with the same a, b, n, flags
 *   h2r_pow_trace_emit_advice       the records of a pow / modpow / verify trace; `workspace` is the workspace that call
 *                                   was given (it holds every mul_mod's operands), elem_stride = 0 means pl->elem_stride. */
#define H2R_ADVICE_ROW_BYTES 160u
/* flags bit of h2r_mul_mod_emit_advice / h2r_pow_trace_emit_advice / h2r_verify_emit_advice: write the mul_mod rows DIRECTLY from
 * the operands (a, b, q, r, n) -- one wave per mul_mod recomputes every cell the reference assigns (main_gate.mul_add
 * big_integer/chip.rs:408, range_chip.assign :590, :598, :880-885, the is_equal_muled ops :851-893) and nothing of the record planes
 * is read (mul_mod form: only its q, r limbs).  Byte-identical to the image of the stored records for every valid witness, and the
 * store-bound form: the record-reading kernel is bound by its scattered loads.  With this flag h2r_pow_trace_emit_advice accepts
 * trace = NULL (a pow call that wrote no records: trace = NULL, caller workspace -- the workspace holds every mul_mod's operands). */
#define H2R_ADVICE_DIRECT 0x200u
/* Row kinds (one per main-gate op shape) and their FIXED columns.  The gate every row satisfies:
 *   sa*a + sb*b + sc*c + sd*d + se*e + s_mul_ab*a*b + s_mul_cd*c*d + se_next*e(next row) + s_const = 0
 * RANGE_LIMB + j / RANGE_CARRY + j = row j of RangeChip::assign of a limb / of a carry: four sub-limb terms in a..d (the LAST
 * row reversed and zero-padded), e = what remains to be composed; tag_composition / tag_overflow = the lookup tags enabled on
 * the row (0 = off).  h2r_advice_row_kinds: the kind of each of the h2r_advice_rows() rows of a mul_mod (host, input-independent);
 * h2r_advice_fixed_row: the selectors of a kind as canonical field elements (cfg nullable: no tags). */
enum { H2R_ROW_NOP = 0, H2R_ROW_CONST0, H2R_ROW_CONST1, H2R_ROW_CONST_B /* assign_constant(0 / 1 / 2^w) */, H2R_ROW_BIT /* assign_bit [v,v,v] */,
       H2R_ROW_VALUE /* assign_value [v] */, H2R_ROW_MUL_ADD /* [a,b,c,a*b+c] */, H2R_ROW_ADD, H2R_ROW_SUB /* [a,b,a+-b] */,
       H2R_ROW_ADD_WM /* add_with_constant(word_max) */, H2R_ROW_ADDC_WM /* add_constant(word_max): [a, a+W] */, H2R_ROW_MUL /* mul, and */,
       H2R_ROW_ASSERT_EQ /* [a,b] */, H2R_ROW_ISZERO_INV /* is_zero: [a, 1/a or 1, r], a*a' + r - 1 = 0 */, H2R_ROW_ISZERO_RA /* [r, a], r*a = 0 */,
       H2R_ROW_SELECT /* select(a,b,cond): [cond,a,cond,b,res], cond*a - cond*b + b - res = 0 */, H2R_ROW_NOT /* [c, 1-c] */,
       H2R_ROW_ASSERT_ONE /* [a], a - 1 = 0 */, H2R_ROW_CONST_BM1 /* assign_constant(2^w - 1) */, H2R_ROW_ASSERT_ZERO /* [a] */,
       H2R_ROW_CONST_EM /* + j, j < 6: the encoded-message constants prefix_64_1, prefix_64_2, 2^32, prefix_32, ff_32, last_em */,
       H2R_ROW_RANGE_U32 = 48 /* + row of RangeChip::assign(value, 4, 32): eight 4-bit sub-limbs (src/chip.rs:170-171) */,
       H2R_ROW_CONST_COEFF8 = 56 /* + j, j < 8: assign_constant(2^(8j)), the byte coefficients of a hashed-message limb (src/lib.rs:228-229) */,
       H2R_ROW_BITS_COMPOSE = 64 /* + row: main_gate.to_bits' composition of four bits 4 row .. 4 row + 3 (coefficients 2^bit), e = what remains */,
       H2R_ROW_BITS_COMPOSE_LAST = 80 /* + 4 row + (terms - 1): its last row, `terms` bits reversed (the highest in column a), zero-padded */,
       H2R_ROW_RANGE_LIMB = 32, H2R_ROW_RANGE_CARRY = 40 };
typedef struct h2r_fixed_row {
    uint64_t sa[4], sb[4], sc[4], sd[4], se[4], s_mul_ab[4], s_mul_cd[4], se_next[4], s_const[4];
    uint32_t tag_composition, tag_overflow;
} h2r_fixed_row;
struct h2r_lookup_config;
int32_t h2r_advice_row_kinds(const h2r_ctx *ctx, uint8_t *kinds_out);
int32_t h2r_advice_fixed_row(const h2r_ctx *ctx, const struct h2r_lookup_config *cfg, uint32_t kind, h2r_fixed_row *out);
/* rows of one element of h2r_pow_trace_emit_advice: for a fixed exponent two constant rows first -- CONST1 [1], CONST0 [0], the
 * limbs of pow_mod_fixed_exp's acc = assign_constant(1, num_limbs) (big_integer/chip.rs:729 -> :1272-1276) -- then the records */
uint64_t h2r_pow_advice_rows(const h2r_ctx *ctx, const h2r_pow_layout *pl);
int32_t h2r_pow_row_kinds(const h2r_ctx *ctx, const h2r_pow_layout *pl, uint8_t *kinds_out);
/* The Fresh-integer family as advice rows: every cell of BigIntChip::add / sub / add_mod / sub_mod / is_zero / is_equal_fresh /
 * the comparisons / is_in_field (big_integer/chip.rs:245-373, 452-528, 754-805, 908-1006; helpers max_value :138-154,
 * sub_unchecked :1286-1318), in the reference's op order, from the witness h2r_fresh_op_batch wrote (or the in_field_trace of
 * h2r_modpow_public_key_batch / the in-field region of a verify element: op = H2R_OP_IS_IN_FIELD, first_off =
 * h2r_verify_layout.off_in_field, elem_stride = the verify element's).  elem_stride = 0: h2r_fresh_op_layout's.
 * flags: H2R_F_SHARED_MODULUS as for h2r_fresh_op_batch (a, b, n are the operands that call was given);
 *        H2R_ADVICE_ASSERT_ONE  the op's bit is then given to main_gate.assert_one -- assert_in_field (:1150-1158), assert_equal_fresh,
 *        assert_less_than ... (instructions.rs:197-254): one more row; H2R_E_UNSUPPORTED for add / sub / add_mod / sub_mod.
 * Rows (kinds from h2r_fresh_op_row_kinds; fixed columns from h2r_advice_fixed_row): 1,532 for assert_in_field of RSA-2048
 * (32 x 64-bit limbs).  Elements with a nonzero status byte (nullable) are skipped. */
#define H2R_ADVICE_ASSERT_ONE 0x100u
uint32_t h2r_fresh_op_advice_rows(const h2r_ctx *ctx, uint32_t op, uint32_t flags);
int32_t h2r_fresh_op_row_kinds(const h2r_ctx *ctx, uint32_t op, uint32_t flags, uint8_t *kinds_out);
/* One whole RSAChip::verify_pkcs1v15_signature element (src/chip.rs:128-199, after the SHA step) as advice rows, in the
 * reference's op order: [is_eq = assign_constant(1) :137] [assert_in_field(sig, n) :106] [pow_mod_fixed_exp :111 -- the rows of
 * h2r_pow_trace_emit_advice] [the encoded-message check :138-198: is_equal / and per limb, the constants (H2R_ROW_CONST_EM + j),
 * the two RangeChip::assign(half, 4, 32) of limb 6 (H2R_ROW_RANGE_U32 + row), their mul_add recomposition and assert_equal].
 * section_rows (nullable): the four sections' row counts (1, 1,532, 75,508, 178 for RSA-2048 with e = 65537).
 * sig, n, hashed, flags, trace, workspace: what h2r_verify_pkcs1v15_batch / h2r_pipeline_verify_pkcs1v15 were given (a caller
 * workspace is required: it holds every mul_mod's operands); powed: their powed_out.  Elements with a nonzero status are skipped.
 * Both exponent arms (src/chip.rs:108-111): a Var element (h2r_verify_layout_var) has the pow_mod rows of h2r_pow_trace_emit_advice
 * (to_bits, select) in its third section.
 * Stream semantics: the call is ordered on `stream` like every export (its inputs are read, and the image is complete, in `stream`
 * order).  Inside, the three short row programs run on a side stream the ctx owns NEXT to the pow rows' kernel -- forked from and
 * joined back into `stream` with events, so nothing is visible to the caller but the shorter call (legal inside a stream capture;
 * concurrent callers on one ctx are serialised for the enqueue by a mutex of the ctx).  h2r_modpow_public_key_emit_advice does the
 * same with its assert_in_field rows. */
uint64_t h2r_verify_advice_rows(const h2r_ctx *ctx, const h2r_verify_layout *vl, uint64_t section_rows[4]);
int32_t h2r_verify_row_kinds(const h2r_ctx *ctx, const h2r_verify_layout *vl, uint8_t *kinds_out);
int32_t h2r_verify_emit_advice(const h2r_ctx *ctx, const h2r_verify_layout *vl, const void *sig, const void *n,
                               const uint64_t *hashed, const void *powed, uint32_t flags, const void *trace,
                               const void *workspace, uint64_t batch, const uint8_t *status, void *advice_out,
                               uint64_t out_stride, h2r_stream_t stream);
int32_t h2r_fresh_op_emit_advice(const h2r_ctx *ctx, uint32_t op, uint32_t flags, const void *a, const void *b, const void *n,
                                 const void *trace, uint64_t first_off, uint64_t elem_stride, uint64_t batch,
                                 const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream);
/* ---- what a layouter backend needs besides the cells: the copy constraints, and the layout as data --------------------------
 * COPY MAP.  maingate gives every op a fresh row and ties the row's INPUT cells to the cells where their values were first assigned
 * with equality constraints (`AssignedValue::from(a.limb(j))` into main_gate.mul_add big_integer/chip.rs:406-408, `carry` across the
 * steps of is_equal_muled :861, ...).  The image repeats the values; h2r_advice_copy_map lists, for ONE mul_mod record (rows of
 * h2r_advice_rows), every input cell (row, col) with its origin: (src_row, src_col) of the same record, or -- src_row =
 * H2R_COPY_SRC_A / _B / _N -- limb src_col of the mul_mod's operand a / b / n, assigned outside the record.  Returns the number of
 * pairs (out may be NULL / cap 0 to ask).  h2r_pow_operand_sources says where those operands come from in a fixed-exponent pow
 * element (pow_mod_fixed_exp, :729-740), per record t: a_src[t], b_src[t] = t' >= 0: the r limbs of record t' (limb j = column e of
 * its row 2 (num_limbs + j)); H2R_SRC_X: the assigned input x; H2R_SRC_ONE: acc = assign_constant(1) (limb 0 = CONST1 row, the others
 * the CONST0 row in front of the records).  n always comes from the assigned modulus.
 * LAYOUT.  The placement of an op's cells in the five columns is third-party (restated, unpinned).  h2r_advice_layout
 * holds per row kind the PHYSICAL column of each of the restated shape's five cells (default: identity).  A maintainer who re-pins a
 * row shape against upstream maingate edits data: h2r_advice_layout_custom validates a table (a permutation per kind; a * b / c * d
 * products stay on one of the gate's column pairs; decompose rows keep their running value in e and their first term in a),
 * h2r_advice_apply_layout permutes an emitted image in place (kinds_dev: the rows' kinds on the device, e.g. h2r_pow_row_kinds
 * uploaded once), h2r_advice_fixed_row_ex gives the selectors of a kind under the layout. */
typedef struct h2r_copy { uint32_t row, col, src_row, src_col; } h2r_copy;
#define H2R_COPY_SRC_A 0xFFFFFF01u
#define H2R_COPY_SRC_B 0xFFFFFF02u
#define H2R_COPY_SRC_N 0xFFFFFF03u
#define H2R_SRC_X (-1)
#define H2R_SRC_ONE (-2)
uint32_t h2r_advice_copy_map(const h2r_ctx *ctx, h2r_copy *out, uint32_t cap);
int32_t h2r_pow_operand_sources(const h2r_ctx *ctx, const h2r_pow_layout *pl, const uint8_t *e_le_bytes, size_t e_len, int32_t *a_src, int32_t *b_src);
#define H2R_ADVICE_LAYOUT_VERSION 1u
typedef struct h2r_advice_layout { uint32_t version; uint8_t column_of[256][5]; } h2r_advice_layout;
int32_t h2r_advice_layout_default(h2r_advice_layout *out);
int32_t h2r_advice_layout_custom(const h2r_ctx *ctx, const uint8_t *kinds, const uint8_t (*column_of)[5], uint32_t n_kinds, h2r_advice_layout *out);
int32_t h2r_advice_fixed_row_ex(const h2r_ctx *ctx, const struct h2r_lookup_config *cfg, const h2r_advice_layout *layout, uint32_t kind, h2r_fixed_row *out);
int32_t h2r_advice_apply_layout(const h2r_ctx *ctx, const h2r_advice_layout *layout, const uint8_t *kinds_dev, uint64_t rows, void *image,
                                uint64_t out_stride, uint64_t batch, const uint8_t *status, h2r_stream_t stream);

/* ---- audit of an advice image in HBM: the device-side MockProver --------------------------------------------------------------
 * Every reference test ends in `MockProver::run(k, &circuit, ..).verify()` (src/chip.rs:338-345, 667; big_integer/chip.rs:1454-1458;
 * examples/rsa_example.rs:207-212).  h2r_advice_check checks, where the image lies and independently of the kernels that wrote it,
 * what MockProver checks of these rows: every row against the main-gate equation with the fixed row of its kind (under `layout`;
 * NULL = the default), every cell of a lookup-enabled row against the (tag, value) table of `cfg` (composition: cells a..d below
 * 2^bit_len(tag); overflow: cell a), every pair of `copies_dev` (n_copies h2r_copy entries ON THE DEVICE, rows counted from the
 * element image's first row, logical columns; src_row = H2R_COPY_SRC_A / _B / _N: limb src_col of the element's src_a / src_b / src_n
 * operand, num_limbs limbs per element, src_n shared with H2R_F_SHARED_MODULUS in flags) for equality, and that every cell is a
 * canonical representative.  The image is read in the ctx's representation.  kinds_dev: the rows' kinds on the device (h2r_*_row_kinds
 * uploaded once); bad_out[elem] (zeroed by the call) = violated checks, first_bad_out[elem] = (row << 8) | code of one of them
 * (1 gate, 2 lookup, 3 copy, 4 a kind without a fixed row, 5 a cell >= p); elements with a nonzero status byte are skipped.
 * h2r_pow_copy_map: the pairs of one fixed-exponent pow element (every record's h2r_advice_copy_map pairs + its operand limbs tied to
 * the cells pow_mod_fixed_exp takes them from: h2r_pow_operand_sources; H2R_COPY_SRC_A = the assigned base x, _N = the modulus), rows
 * counted from row_offset (the pow section's first row in a larger element image); returns the number of pairs (out NULL / cap 0 to ask). */
int32_t h2r_advice_check(const h2r_ctx *ctx, const struct h2r_lookup_config *cfg, const h2r_advice_layout *layout, const uint8_t *kinds_dev,
                         uint64_t rows, const void *image, uint64_t out_stride, uint64_t batch, const uint8_t *status,
                         const h2r_copy *copies_dev, uint64_t n_copies, const void *src_a, const void *src_b, const void *src_n,
                         uint32_t flags, uint32_t *bad_out, uint64_t *first_bad_out, h2r_stream_t stream);
uint64_t h2r_pow_copy_map(const h2r_ctx *ctx, const h2r_pow_layout *pl, const uint8_t *e_le_bytes, size_t e_len, uint64_t row_offset,
                          h2r_copy *out, uint64_t cap);

/* One RSAChip::modpow_public_key element (src/chip.rs:99-114) as advice rows, in the reference's op order: [assert_in_field(x, n) :106:
 * the rows of h2r_fresh_op_emit_advice(H2R_OP_IS_IN_FIELD, H2R_ADVICE_ASSERT_ONE)] [pow_mod_fixed_exp / pow_mod :108-111: the rows of
 * h2r_pow_trace_emit_advice].  x, n, flags, in_field_trace, trace, workspace: what h2r_modpow_public_key_batch was given (a caller
 * workspace is required).  trace = NULL: that call wrote no records -- the pow rows are written directly from the operands
 * (H2R_ADVICE_DIRECT, also selectable in flags with a trace): the prover-consumable witness without the record planes in between.
 * section_rows (nullable): {1,532, 75,508} for RSA-2048, e = 65537. */
uint64_t h2r_modpow_public_key_advice_rows(const h2r_ctx *ctx, const h2r_pow_layout *pl, uint64_t section_rows[2]);
int32_t h2r_modpow_public_key_emit_advice(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *x, const void *n, uint32_t flags,
                                          const void *in_field_trace, const void *trace, const void *workspace, uint64_t batch,
                                          const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream);
/* Pipelined form of the two calls above for a fixed exponent (see h2r_pipeline_create): h2r_modpow_public_key_batch WITHOUT records --
 * the chains, `out`, `status`, the assert_in_field witness -- and the element's in-field rows on `stream`; its pow rows (cells_kernel,
 * H2R_ADVICE_DIRECT) on a side stream of the pipeline, next to the chains of the following call.  The IMAGE follows the pipeline's join
 * rule (complete, in `stream` order, once `depth` - 1 further pipelined calls have returned or after h2r_pipeline_join); `out` and `status`
 * are stream-ordered as usual; x and n are read inside the call (the moduli are copied into the workspace).  Consecutive calls rotate
 * through `depth` sets of in_field_trace / out / status / workspace / advice_out.  This is what `bench.py --advice` measures
 * (RSA-2048, 1,024 per call: 0.5 M elements/s of 12.3 MB each, the cells kernel at 0.79-0.82 of the HBM peak). */
int32_t h2r_pipeline_modpow_public_key_advice(h2r_pipeline *p, const void *x, const void *n, const uint8_t *e_le_bytes, size_t e_len,
                                              uint64_t batch, uint32_t flags, void *in_field_trace, void *out, uint8_t *status,
                                              void *workspace, void *advice_out, uint64_t out_stride, h2r_stream_t stream);
/* The WHOLE RSAChip::verify_pkcs1v15_signature element (src/chip.rs:128-199) as advice rows without records, pipelined: the image of
 * h2r_verify_emit_advice -- [is_eq seed][assert_in_field][pow rows][encoded-message check] -- for a fixed exponent, produced like
 * h2r_pipeline_modpow_public_key_advice produces its two sections: the chains, powed_out, the in-field / encoded-message witness,
 * is_valid_out, status and the three short row programs on `stream`; the pow rows (cells_kernel) on a side stream of the pipeline next
 * to the chains of the following call.  The image follows the pipeline's join rule; everything else is stream-ordered.
 * witness (16-byte aligned): batch * h2r_verify_layout_compact(...).elem_stride bytes -- the element's in-field and EM witness, the only part of a verify
 * element's trace the rows need (9,984 B instead of 1.26 MB per RSA-2048 element).  h2r_verify_layout_compact turns a verify layout into
 * that form (off_in_field = 0, off_em, elem_stride; pow.off_records = UINT64_MAX -- the record exports (flatten, check, hist, emit_stream and the record-read image) answer H2R_E_SHAPE; the rest of
 * `pow` unchanged for a Fix layout, the witness-only pow layout inside the element for a Var one); with it h2r_verify_emit_advice (flags | H2R_ADVICE_DIRECT,
 * trace = the witness), h2r_verify_advice_rows and h2r_verify_row_kinds work as with the full layout.
 * Consecutive calls rotate through `depth` sets of witness / powed_out / is_valid_out / status / workspace / advice_out.
 * `bench.py --advice --verify` measures this form (77,219 rows = 12.35 MB per RSA-2048 element). */
int32_t h2r_verify_layout_compact(const h2r_ctx *ctx, const h2r_verify_layout *full, h2r_verify_layout *out);
int32_t h2r_pipeline_verify_pkcs1v15_advice(h2r_pipeline *p, const void *sig, const void *n, const uint8_t *e_le_bytes, size_t e_len,
                                            const uint64_t *hashed, uint64_t batch, uint32_t flags, void *witness, void *powed_out,
                                            uint8_t *is_valid_out, uint8_t *status, void *workspace, void *advice_out,
                                            uint64_t out_stride, h2r_stream_t stream);
/* the RSAPubE::Var arm: h2r_verify_layout_compact of a Var layout also keeps the pow_mod's exponent bits, selected operands and result
 * (its `pow` is then the witness-only form, off_records = UINT64_MAX, offsets inside the element's witness) */
int32_t h2r_pipeline_verify_pkcs1v15_var_advice(h2r_pipeline *p, const void *sig, const void *n, const void *e_limbs, uint32_t e_num_limbs,
                                                uint32_t exp_limb_bits, const uint64_t *hashed, uint64_t batch, uint32_t flags,
                                                void *witness, void *powed_out, uint8_t *is_valid_out, uint8_t *status, void *workspace,
                                                void *advice_out, uint64_t out_stride, h2r_stream_t stream);
/* The RSAPubE::Var arm (src/chip.rs:108-110; pow_mod, big_integer/chip.rs:664-696) of h2r_pipeline_modpow_public_key_advice: per-element
 * exponents, no records.  A Var element's to_bits / select rows read its exponent bits and selected operands: they are kept -- with the
 * result -- in `witness`, batch * h2r_pow_layout_compact(h2r_pow_var_layout(...)).elem_stride bytes (16-byte aligned), the witness-only
 * form of a pow layout (off_records = UINT64_MAX: no record planes; for a Fix layout just the result).  A compact layout is accepted by
 * h2r_pow_advice_rows, h2r_pow_row_kinds, h2r_modpow_public_key_advice_rows and -- with H2R_ADVICE_DIRECT and the witness as `trace` --
 * h2r_pow_trace_emit_advice / h2r_modpow_public_key_emit_advice; the record exports (flatten, check, hist, emit_stream) need the full one.
 * Rows: [assert_in_field][to_bits rows of every exponent limb][CONST1, CONST0][per bit: two mul_mods' rows + the select rows]. */
int32_t h2r_pow_layout_compact(const h2r_ctx *ctx, const h2r_pow_layout *full, h2r_pow_layout *out);
int32_t h2r_pipeline_modpow_public_key_var_advice(h2r_pipeline *p, const void *x, const void *e_limbs, uint32_t e_num_limbs,
                                                  uint32_t exp_limb_bits, const void *n, uint64_t batch, uint32_t flags,
                                                  void *in_field_trace, void *witness, void *out, uint8_t *status, void *workspace,
                                                  void *advice_out, uint64_t out_stride, h2r_stream_t stream);
uint32_t h2r_advice_rows(const h2r_ctx *ctx);
int32_t h2r_mul_mod_emit_advice(const h2r_ctx *ctx, const void *a, const void *b, const void *n, uint32_t flags,
                                const void *trace, uint64_t batch, const uint8_t *status, void *advice_out,
                                uint64_t out_stride, h2r_stream_t stream);
int32_t h2r_pow_trace_emit_advice(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *n, uint32_t flags,
                                  const void *trace, uint64_t elem_stride, const void *workspace, uint64_t batch,
                                  const uint8_t *status, void *advice_out, uint64_t out_stride, h2r_stream_t stream);

/* ---- in-place audit of a trace (test / diagnosis instrument; not needed by a consumer) ---------------------
 * Checks, where the records lie in HBM, that every record is a valid witness of ITS mul_mod (sub-limb recomposition,
 * every accumulator = its predecessor + one limb product, eq_b, every is_equal_muled step against the stored previous
 * carry, flags, range-assigned carries, r < n, final eq_bit = 1: SURVEY Appendix C 1-4, reference
 * big_integer/chip.rs:400-412, 562-599, 617, 857-893) and -- for pow traces -- that the operands of every mul_mod are
 * the values the reference's control flow feeds it and the result limbs are the final acc (Appendix C 5-6,
 * chip.rs:682-694, 729-740).  Every relation is checked pointwise from stored values, independently of the kernels
 * that produced them.  bad_out[elem] (zeroed by the call) = number of violated relations; first_bad_out (nullable)
 * = (mul_mod index << 8) | relation code of one of them.  Elements with a nonzero status byte are skipped.
 *   h2r_mul_mod_trace_check  records of h2r_mul_mod_batch / h2r_square_mod_batch with the same a, b, n, flags.
 *   h2r_pow_trace_check      traces of the pow / modpow / verify exports: `workspace` is the workspace that call was
 *                            given (it still holds the operands of every mul_mod), elem_stride = 0 means
 *                            pl->elem_stride; e_le_bytes = NULL for a variable-exponent trace. */
int32_t h2r_mul_mod_trace_check(const h2r_ctx *ctx, const void *a, const void *b, const void *n, uint32_t flags,
                                const void *trace, uint64_t batch, const uint8_t *status, uint32_t *bad_out,
                                uint32_t *first_bad_out, h2r_stream_t stream);
int32_t h2r_pow_trace_check(const h2r_ctx *ctx, const h2r_pow_layout *pl, const void *x, const void *n,
                            const uint8_t *e_le_bytes, size_t e_len, uint32_t flags, const void *trace,
                            uint64_t elem_stride, const void *workspace, uint64_t batch, const uint8_t *status,
                            uint32_t *bad_out, uint32_t *first_bad_out, h2r_stream_t stream);

/* ---- per-kernel timing (HIP events recorded on the launch stream around each kernel) -----------
 * h2r_profile_enable(capacity) arms process-wide recording of up to `capacity` launches (0 disarms
 * and frees the events).  h2r_profile_read() synchronises the recorded events of one kernel class
 * and returns their durations in milliseconds, in launch order. */
enum { H2R_KERNEL_CHAIN = 0, H2R_KERNEL_TRACE = 1, H2R_KERNEL_HIST = 2, H2R_KERNEL_AUX = 3, H2R_KERNEL_EMIT = 4,
       H2R_KERNEL_STEP = 5 /* a pipeline step as one launch: records of call k + chains of call k+1 */,
       H2R_KERNEL_LOOKUP = 6 /* lookup_fill_kernel: the permuted columns */, H2R_KERNEL_SHA256 = 7 /* sha256_kernel */,
       H2R_KERNEL_CELLS = 8 /* cells_kernel: the advice image written directly from the operands */,
       H2R_KERNEL_COUNT = 9 };
int32_t h2r_profile_enable(uint32_t capacity);
int32_t h2r_profile_read(uint32_t kernel, float *ms_out, uint32_t max_count, uint32_t *count);

const char *h2r_status_str(int32_t status);
const char *h2r_last_hip_error(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* H2R_H */
