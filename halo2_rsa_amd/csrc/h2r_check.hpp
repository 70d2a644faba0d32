// gfx950 kernels of libh2r: the audit of an ADVICE IMAGE where it lies in HBM -- what halo2's MockProver::verify checks of these rows
// (every reference test ends in `MockProver::run(k, &circuit, ..).verify()`: src/chip.rs:338-345, 667; big_integer/chip.rs:1454-1458;
// examples/rsa_example.rs:207-212), independently of the kernels that wrote the image:
//   * GATE      every row satisfies the main-gate equation with the fixed row of its kind
//                   sa a + sb b + sc c + sd d + se e + s_mul_ab a b + s_mul_cd c d + se_next e(next row) + s_const = 0   (mod p)
//   * LOOKUP    on a row whose kind enables the composition lookup, cells a..d are rows of the (tag, value) table, i.e. below
//               2^bit_len(tag_composition); with the overflow lookup, cell a is below 2^bit_len(tag_overflow)
//   * COPY      every pair of the copy map holds equal cells (or a cell equal to the operand limb it is a copy of)
//   * every cell is a canonical representative (< p)
// The image is read in the ctx's representation (row-major / planar, canonical / Montgomery); the gate is evaluated in the Montgomery
// domain with plain field arithmetic (fe_mont_mul) -- nothing of the producing kernels' short conversions is reused.  One thread per
// row (then per copy pair); a test / diagnosis instrument, bound by its field multiplications, not a product path.
#pragma once

#include "h2r_kernels.hpp"

namespace h2r {

struct CheckKind {                 // one row kind under the caller's layout and lookup configuration
    Fe s[9];                       // Montgomery form: sa, sb, sc, sd, se, s_mul_ab, s_mul_cd, se_next, s_const
    u32 comp_bits, ov_bits;        // bit lengths of the lookups enabled on the row (0 = off)
    u32 valid, nz;                 // a kind h2r_advice_fixed_row knows; bit k of nz = s[k] is nonzero
};
enum : u32 { ACHK_GATE = 1, ACHK_LOOKUP = 2, ACHK_COPY = 3, ACHK_KIND = 4, ACHK_RANGE = 5 };

struct AdviceCheckArgs {
    AdviceDst img;                 // (read only)
    const u8 *kinds; u64 rows, batch; const u8 *status;
    const CheckKind *tab;
    FieldConsts f;
    u32 *bad; unsigned long long *first;
    const h2r_copy *copies; u64 n_copies;
    const void *ext[3]; u64 ext_stride[3]; u32 limb_bytes;   // operand limbs behind H2R_COPY_SRC_A / _B / _N (limbs between elements; 0 = shared)
    u8 perm[256][5];               // the layout: physical column of a kind's logical cell
};

__device__ __forceinline__ Fe achk_cell(const AdviceDst &d, const u8 *img, u64 row, u32 col) {
    const uint4 *p = reinterpret_cast<const uint4 *>(img + row * d.row_pitch + (u64)col * d.col_pitch);
    const uint4 lo = p[0], hi = p[1];
    Fe r;
    r.v[0] = ((u64)lo.y << 32) | lo.x; r.v[1] = ((u64)lo.w << 32) | lo.z; r.v[2] = ((u64)hi.y << 32) | hi.x; r.v[3] = ((u64)hi.w << 32) | hi.z;
    return r;
}
__device__ __forceinline__ void achk_fail(const AdviceCheckArgs &a, u64 elem, u64 row, u32 code) {
    atomicAdd(a.bad + elem, 1u);
    atomicCAS(a.first + elem, 0ull, (unsigned long long)((row << 8) | code));
}

__global__ __launch_bounds__(256) void advice_check_rows_kernel(AdviceCheckArgs a) {
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 elem = gid / a.rows;
    if (elem >= a.batch || (a.status && a.status[elem])) return;
    const u64 r = gid - elem * a.rows;
    const u8 *img = a.img.base + elem * a.img.elem_stride;
    const CheckKind &ck = a.tab[a.kinds[r]];
    if (!ck.valid) { achk_fail(a, elem, r, ACHK_KIND); return; }
    Fe c[6];
#pragma unroll
    for (u32 k = 0; k < 5; ++k) c[k] = achk_cell(a.img, img, r, k);
    const bool has_next = r + 1 < a.rows;
    c[5] = has_next ? achk_cell(a.img, img, r + 1, 4) : fe_zero();
    bool canon = true;
#pragma unroll
    for (u32 k = 0; k < 5; ++k) canon = canon && !ge_p(c[k].v, a.f.p);
    if (!canon) { achk_fail(a, elem, r, ACHK_RANGE); return; }
    if ((ck.nz & (1u << 7)) && !has_next) { achk_fail(a, elem, r, ACHK_GATE); return; }   // a row that refers to a next row the image does not have
    // ---- the lookups: on the canonical integers ----
    if (ck.comp_bits | ck.ov_bits) {
        bool ok = true;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const Fe v = a.img.mont ? fe_from_mont(c[k], a.f) : c[k];
            const bool small = (v.v[1] | v.v[2] | v.v[3]) == 0;
            if (ck.comp_bits) ok = ok && small && (ck.comp_bits >= 64 || (v.v[0] >> ck.comp_bits) == 0);
            if (k == 0 && ck.ov_bits) ok = ok && small && (ck.ov_bits >= 64 || (v.v[0] >> ck.ov_bits) == 0);
        }
        if (!ok) achk_fail(a, elem, r, ACHK_LOOKUP);
    }
    // ---- the gate, in the Montgomery domain ----
    Fe m[6];
#pragma unroll
    for (u32 k = 0; k < 6; ++k) m[k] = a.img.mont ? c[k] : fe_to_mont(c[k], a.f);
    Fe sum = ck.s[8];
#pragma unroll
    for (u32 k = 0; k < 5; ++k) if (ck.nz & (1u << k)) sum = fe_add(sum, fe_mont_mul(ck.s[k], m[k], a.f), a.f.p);
    if (ck.nz & (1u << 5)) sum = fe_add(sum, fe_mont_mul(ck.s[5], fe_mont_mul(m[0], m[1], a.f), a.f), a.f.p);
    if (ck.nz & (1u << 6)) sum = fe_add(sum, fe_mont_mul(ck.s[6], fe_mont_mul(m[2], m[3], a.f), a.f), a.f.p);
    if (ck.nz & (1u << 7)) sum = fe_add(sum, fe_mont_mul(ck.s[7], m[5], a.f), a.f.p);
    if (!fe_is_zero(sum)) achk_fail(a, elem, r, ACHK_GATE);
}

__global__ __launch_bounds__(256) void advice_check_copies_kernel(AdviceCheckArgs a) {
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 elem = gid / a.n_copies;
    if (elem >= a.batch || (a.status && a.status[elem])) return;
    const h2r_copy cp = a.copies[gid - elem * a.n_copies];
    const u8 *img = a.img.base + elem * a.img.elem_stride;
    if (cp.row >= a.rows || cp.col > 4) { achk_fail(a, elem, cp.row, ACHK_COPY); return; }
    const Fe x = achk_cell(a.img, img, cp.row, a.perm[a.kinds[cp.row]][cp.col]);
    bool ok;
    if (cp.src_row >= 0xFFFFFF00u) {   // a limb of an operand assigned outside the image
        const u32 which = (cp.src_row & 0xffu) - 1u;
        if (which > 2 || !a.ext[which]) { achk_fail(a, elem, cp.row, ACHK_COPY); return; }
        const u64 idx = elem * a.ext_stride[which] + cp.src_col;
        const u64 limb = a.limb_bytes == 8 ? reinterpret_cast<const u64 *>(a.ext[which])[idx] : reinterpret_cast<const u32 *>(a.ext[which])[idx];
        const Fe v = a.img.mont ? fe_from_mont(x, a.f) : x;
        ok = v.v[0] == limb && (v.v[1] | v.v[2] | v.v[3]) == 0;
    } else {
        if (cp.src_row >= a.rows || cp.src_col > 4) { achk_fail(a, elem, cp.row, ACHK_COPY); return; }
        ok = fe_eq(x, achk_cell(a.img, img, cp.src_row, a.perm[a.kinds[cp.src_row]][cp.src_col]));
    }
    if (!ok) achk_fail(a, elem, cp.row, ACHK_COPY);
}

// ---- the lookup multiplicities of an advice image: what h2r_lookup_permuted_columns needs of a witness that has no records ----
// For every row whose kind enables the composition lookup, cells a..d each add one to their (tag, value) table row in arguments 0..3;
// with the overflow lookup, cell a adds one in argument 4 (h2r.h: hist[elem][5][n_rows]).  A workgroup walks HIST_ROWS_PER_WG rows of
// one element with a histogram of its own in LDS (the hot counters of a circuit are a few hundred) and adds what it gathered at the end.
struct AdviceHistArgs {
    AdviceDst img;
    const u8 *kinds; u64 rows, batch; const u8 *status;
    const CheckKind *tab;
    FieldConsts f;
    u32 *hist; u32 n_rows, n_lens;
    u32 bit_len[8], row_off[8];
};
constexpr u32 HIST_ROWS_PER_WG = 16384;
__global__ __launch_bounds__(256) void advice_hist_kernel(AdviceHistArgs a) {
    extern __shared__ u32 hist_lds[];                        // [5][n_rows]
    const u32 chunks = (u32)((a.rows + HIST_ROWS_PER_WG - 1) / HIST_ROWS_PER_WG);
    const u64 elem = blockIdx.x / chunks;
    const u64 r_lo = (u64)(blockIdx.x - elem * chunks) * HIST_ROWS_PER_WG, r_hi = r_lo + HIST_ROWS_PER_WG < a.rows ? r_lo + HIST_ROWS_PER_WG : a.rows;
    if (a.status && a.status[elem]) return;
    for (u32 k = threadIdx.x; k < 5 * a.n_rows; k += 256) hist_lds[k] = 0;
    __shared__ unsigned short kbits[256];                               // per kind: composition bits | overflow bits << 8 (0: no lookup on the row) -- the scan reads nothing else per row
    { const CheckKind &ck = a.tab[threadIdx.x]; kbits[threadIdx.x] = ck.valid ? (unsigned short)(ck.comp_bits | (ck.ov_bits << 8)) : (unsigned short)0; }
    __syncthreads();
    const u8 *img = a.img.base + elem * a.img.elem_stride;
    auto table_row = [&](u32 bits, const Fe &cell, u32 &row) -> bool {
        const Fe v = a.img.mont ? fe_from_mont(cell, a.f) : cell;
        if ((v.v[1] | v.v[2] | v.v[3]) != 0 || (bits < 64 && (v.v[0] >> bits) != 0)) return false;   // not a table row: h2r_advice_check's finding, not counted
        for (u32 i = 0; i < a.n_lens; ++i) if (a.bit_len[i] == bits) { row = a.row_off[i] + (u32)v.v[0]; return true; }
        return false;
    };
    // Lookup rows are a few per cent of an image: each pass lists those of 1,024 rows densely, then one thread takes one (row, cell) of the
    // list -- every lane of the waves that convert and count has a cell (a scan that did the work where it found a row ran 4x longer in
    // Montgomery form: one active lane in a wave pays the whole conversion)
    __shared__ u32 l_row[1024];
    __shared__ u32 l_cnt;
    for (u64 base = r_lo; base < r_hi; base += 1024) {
        if (threadIdx.x == 0) l_cnt = 0;
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
            const u64 r = base + q * 256 + threadIdx.x;
            if (r < r_hi) {
                const u32 kb = kbits[a.kinds[r]];
                if (kb) l_row[atomicAdd(&l_cnt, 1u)] = (u32)(r - base) | (kb << 10);
            }
        }
        __syncthreads();
        const u32 n = l_cnt;
        for (u32 t = threadIdx.x; t < 4 * n; t += 256) {
            const u32 e = l_row[t >> 2], k = t & 3u, comp_bits = (e >> 10) & 0xffu, ov_bits = e >> 18;
            if (!comp_bits && k) continue;
            const Fe cell = achk_cell(a.img, img, base + (e & 1023u), k);
            u32 row;
            if (comp_bits && table_row(comp_bits, cell, row)) atomicAdd(&hist_lds[k * a.n_rows + row], 1u);
            if (k == 0 && ov_bits && table_row(ov_bits, cell, row)) atomicAdd(&hist_lds[4 * a.n_rows + row], 1u);
        }
        __syncthreads();
    }
    __syncthreads();
    u32 *out = a.hist + elem * 5ull * a.n_rows;
    for (u32 k = threadIdx.x; k < 5 * a.n_rows; k += 256) if (hist_lds[k]) atomicAdd(out + k, hist_lds[k]);
}

}  // namespace h2r
