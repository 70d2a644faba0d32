/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C) of the halo2-rsa hot path; see
 * h2r_oracle.c for the reference file:line each function follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (halo2_rsa_amd/, libh2r.so) never links, loads or calls it.
 *
 * Output format: the flat op-trace stream documented in oracle/pyref.py and DESIGN.md --
 * every witness value in the reference's assignment order, little-endian, fixed width.
 */
#ifndef H2R_ORACLE_H
#define H2R_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { H2RO_OK = 0, H2RO_E_SHAPE = 1, H2RO_E_ZERO_MODULUS = 2, H2RO_E_NOT_REDUCED = 3,
       H2RO_E_NOT_IN_FIELD = 8 };

typedef struct h2ro_params {
    uint32_t w, L;            /* limb_width, num_limbs */
    uint32_t LB, WB, CB;      /* stream widths in bytes: LIMB, WIDE, CARRY */
    uint32_t word_max_bits, carry_bits;
    uint32_t limb_sub_bits, limb_nsub, carry_sub_bits, carry_nsub;
    uint64_t word_max[4];     /* compute_mul_word_max(w, L), 256-bit little-endian */
    uint64_t mul_mod_stream_bytes;
} h2ro_params;

/* BigIntChip::new parameter derivation; limbs are uint64_t for w == 64 and uint32_t for w == 32. */
int h2ro_params_init(h2ro_params *p, uint32_t limb_width, uint32_t num_limbs);
void h2ro_compute_range_lens(uint32_t limb_width, uint32_t num_limbs, uint32_t comp[3], uint32_t over[3]);

/* Each returns H2RO_*; `stream` may be NULL (values only).  Sizes via the *_stream_bytes calls. */
int h2ro_mul_columns(const h2ro_params *p, const void *a, const void *b, uint8_t *stream, uint64_t *cols_out /* (2L-1)*4 u64 */);
int h2ro_mul_mod(const h2ro_params *p, const void *a, const void *b, const void *n, uint8_t *stream, void *r_out);
uint64_t h2ro_pow_fixed_stream_bytes(const h2ro_params *p, const uint8_t *e_le, size_t e_len);
int h2ro_pow_mod_fixed_exp(const h2ro_params *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len, uint8_t *stream, void *out);
uint64_t h2ro_pow_var_stream_bytes(const h2ro_params *p, uint32_t e_num_limbs, uint32_t exp_limb_bits);
int h2ro_pow_mod(const h2ro_params *p, const void *x, const void *e_limbs, uint32_t e_num_limbs, uint32_t exp_limb_bits, const void *n, uint8_t *stream, void *out);
/* big_integer/utils.rs:2-17 */
int h2ro_big_pow_mod(const h2ro_params *p, const void *a, const uint8_t *e_le, size_t e_len, const void *n, void *out);

/* SURVEY 8(f) next #1 / #2 */
uint64_t h2ro_in_field_stream_bytes(const h2ro_params *p);
int h2ro_assert_in_field(const h2ro_params *p, const void *a, const void *n, uint8_t *stream, int *is_less);
uint64_t h2ro_pkcs1v15_stream_bytes(const h2ro_params *p);
int h2ro_pkcs1v15_em_check(const h2ro_params *p, const void *powed, const uint64_t hashed[4], uint8_t *stream, int *is_valid);

/* Fresh-integer family (add, sub, add_mod, sub_mod, comparisons) */
enum { H2RO_OP_ADD = 0, H2RO_OP_SUB, H2RO_OP_ADD_MOD, H2RO_OP_SUB_MOD, H2RO_OP_IS_ZERO, H2RO_OP_IS_EQUAL_FRESH,
       H2RO_OP_IS_LESS_THAN, H2RO_OP_IS_LESS_THAN_OR_EQUAL, H2RO_OP_IS_GREATER_THAN, H2RO_OP_IS_GREATER_THAN_OR_EQUAL,
       H2RO_OP_IS_IN_FIELD, H2RO_OP_COUNT };
uint64_t h2ro_fresh_op_stream_bytes(const h2ro_params *p, int op);
int h2ro_fresh_op(const h2ro_params *p, int op, const void *a, const void *b, const void *n, uint8_t *stream,
                  void *value_out, uint32_t *nvalue, int *flag_out);

uint64_t h2ro_refresh_stream_bytes(const h2ro_params *p);
int h2ro_refresh(const h2ro_params *p, const uint64_t *muled, uint8_t *stream, void *fresh_out);
uint64_t h2ro_is_equal_muled_stream_bytes(const h2ro_params *p);
int h2ro_is_equal_muled(const h2ro_params *p, const uint64_t *a, const uint64_t *b, uint8_t *stream, int *eq_bit);

/* The caller of the path, RSASignatureVerifier::verify_pkcs1v15_signature (src/lib.rs:183-246): FIPS 180-4 SHA-256 of the
 * message (:205-209) and the hashed-message limbs composed from the reversed digest bytes (:210-239). */
void h2ro_sha256(const uint8_t *msg, uint64_t len, uint8_t digest[32]);
uint64_t h2ro_hashed_msg_stream_bytes(void);
void h2ro_hashed_msg(const uint8_t digest[32], uint64_t hashed[4], uint8_t *stream);

/* Batch drivers used by the CPU-baseline timing leg: element-major inputs, `nthreads` pthreads,
 * one element per task.  stream (nullable) holds batch*stream_bytes bytes. */
int h2ro_pow_mod_fixed_exp_batch(const h2ro_params *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len,
                                 uint64_t batch, uint8_t *stream, void *out, uint8_t *status, int nthreads);

/* cpu_baseline timing: persistent threads, per-thread reusable stream buffers, `passes` passes over the batch. */
int h2ro_pow_mod_fixed_exp_timed(const h2ro_params *p, const void *x, const void *n, const uint8_t *e_le, size_t e_len,
                                 uint64_t batch, uint64_t passes, int nthreads, double *seconds, uint64_t *failed);

#ifdef __cplusplus
}
#endif
#endif
