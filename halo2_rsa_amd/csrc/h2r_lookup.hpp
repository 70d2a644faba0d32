// halo2's lookup argument for the range checks of the path -- the prover-side "lookup permutation" (SURVEY 8a row a10,
// 8f next #3).  Third-party algorithm (halo2 plonk::lookup::prover, not in the reference tree; triggered by
// range_chip.load_table + create_proof, reference benches/bench.rs:141-142, 321-329), restated in DESIGN.md section 2c:
//   RangeChip registers FIVE lookup arguments over one (tag, value) table: composition_a..d read the main gate's advice
//   columns a..d under the fixed column tag_composition, overflow_a reads column a under tag_overflow.  Per argument and
//   circuit the prover compresses inputs and table with the challenge theta (tag * theta + value), sorts the inputs into A'
//   and arranges the table into S' so that on every row A'[i] == S'[i] or A'[i] == A'[i-1].
// Both columns are a pure function of (per-argument multiplicity of every table row, theta, usable rows):
//   lookup_hist_*   multiplicities per (element = circuit, argument, table row) from the records' sub-limb planes or from
//                   arrays of range-assigned values; LDS atomics, one workgroup per element.
//   lookup_setup    per (element, argument): compress the <= 1,024 table rows with the element's theta, rank-sort them by
//                   the field's Ord, merge equal values, prefix-sum the run lengths of A' and of the leftover table values.
//   lookup_fill     writes A' and S' (usable_rows x 32 bytes each): every position finds its run by binary search in the
//                   LDS copy of the prefix sums; lanes exchange halves so that every store instruction covers 1 KB.
//                   Bound: HBM writes, 64 bytes per row and argument.
#pragma once

#include "h2r_field.hpp"
#include "h2r_kernels.hpp"

namespace h2r {

constexpr int LOOKUP_ARGS = 5;
constexpr int LOOKUP_MAX_ROWS = 1024;   // table rows: 1 + sum 2^bit_len (RSA-2048: 339)
constexpr int LOOKUP_MAX_LENS = 8;

// where a RangeChip::assign(value, s, bit_len) puts its sub-limbs (main_gate.decompose: rows of four terms in columns
// a..d; the LAST row is reversed so that the last term -- the overflow sub-limb when there is one -- sits in column a,
// and padded with zero terms, which are looked up like any other cell of the row)
struct RangeShape {
    u32 nsub, sub_bits, ov_bits;   // sub-limbs (overflow one included), their width, width of the overflow sub-limb (0 = none)
    u32 row_comp, row_ov;          // table row of (tag(sub_bits), 0) / (tag(ov_bits), 0)
};
__host__ __device__ inline u32 range_last_row(const RangeShape &s) { return (s.nsub - 1) / 4; }

// adds the lookups of one range assign with sub-limb bytes `sub(t)` to the per-argument LDS histogram h[arg * n_rows + row]
template <typename SubFn>
__device__ __forceinline__ void range_count(u32 *h, u32 n_rows, const RangeShape &s, SubFn sub) {
    const u32 last = range_last_row(s), last_len = s.nsub - 4 * last;
    for (u32 t = 0; t < s.nsub; ++t) {
        const u32 v = sub(t);
        const u32 rr = t / 4;
        const u32 arg = rr < last ? (t & 3u) : (s.nsub - 1 - t);
        atomicAdd(&h[arg * n_rows + s.row_comp + v], 1u);
        if (s.ov_bits && t == s.nsub - 1) atomicAdd(&h[4 * n_rows + s.row_ov + v], 1u);
    }
    for (u32 k = last_len; k < 4; ++k) atomicAdd(&h[k * n_rows + s.row_comp], 1u);   // zero terms of the last row
}

struct LookupHistArgs {
    const u8 *trace; u64 first_record_off, elem_stride, record_stride; u64 num_elems; u32 records_per_elem;
    u64 off_q_sub, off_r_sub, off_carry_sub; u32 L, C, carry_sub_stride;
    RangeShape limb, carry;
    u32 n_rows;
    u32 *hist;          // [elem][5][n_rows], ADDED to
    const u8 *status;   // nullable: elements with a nonzero status are skipped
};
__global__ __launch_bounds__(256) void lookup_hist_records_kernel(LookupHistArgs a) {
    extern __shared__ u32 lh[];
    const u32 tid = threadIdx.x, n = LOOKUP_ARGS * a.n_rows;
    const u64 elem = blockIdx.x;
    if (a.status && a.status[elem]) return;
    for (u32 k = tid; k < n; k += 256) lh[k] = 0;
    __syncthreads();
    const u8 *base = a.trace + elem * a.elem_stride + a.first_record_off;
    const u32 per_rec = 2 * a.L + (a.C - 1);
    for (u64 idx = tid; idx < (u64)a.records_per_elem * per_rec; idx += 256) {
        const u32 t = (u32)(idx / per_rec), k = (u32)(idx - (u64)t * per_rec);
        const u8 *rec = base + (u64)t * a.record_stride;
        if (k < 2 * a.L) {   // q limbs then r limbs: eight sub-limb bytes each
            const u64 sb = *reinterpret_cast<const u64 *>(rec + (k < a.L ? a.off_q_sub + (u64)k * 8 : a.off_r_sub + (u64)(k - a.L) * 8));
            range_count(lh, a.n_rows, a.limb, [&](u32 i) { return (u32)((sb >> (8 * i)) & 0xff); });
        } else {             // range-assigned carry of column k - 2L
            const ulonglong2 sb = *reinterpret_cast<const ulonglong2 *>(rec + a.off_carry_sub + (u64)(k - 2 * a.L) * a.carry_sub_stride);
            range_count(lh, a.n_rows, a.carry, [&](u32 i) { return (u32)(((i < 8 ? sb.x : sb.y) >> (8 * (i & 7))) & 0xff); });
        }
    }
    __syncthreads();
    u32 *out = a.hist + elem * n;
    for (u32 k = tid; k < n; k += 256) if (lh[k]) out[k] += lh[k];
}

struct LookupValuesArgs {
    const u8 *values; u32 value_bytes; u64 values_per_elem, num_elems;
    u64 elem_stride, value_stride;      // bytes between the elements' first values / between the values of an element
    const u8 *status;                   // nullable: elements with a nonzero status are skipped
    RangeShape shape; u32 n_rows; u32 *hist;
};
__global__ __launch_bounds__(256) void lookup_hist_values_kernel(LookupValuesArgs a) {
    extern __shared__ u32 lh[];
    const u32 tid = threadIdx.x, n = LOOKUP_ARGS * a.n_rows;
    const u64 elem = blockIdx.x;
    if (a.status && a.status[elem]) return;
    for (u32 k = tid; k < n; k += 256) lh[k] = 0;
    __syncthreads();
    const u32 m = (1u << a.shape.sub_bits) - 1;
    for (u64 idx = tid; idx < a.values_per_elem; idx += 256) {
        const u8 *p = a.values + elem * a.elem_stride + idx * a.value_stride;
        u128 v = a.value_bytes == 4 ? (u128) * reinterpret_cast<const u32 *>(p) : (u128) * reinterpret_cast<const u64 *>(p);
        if (a.value_bytes == 16) v |= (u128)(*reinterpret_cast<const u64 *>(p + 8)) << 64;
        range_count(lh, a.n_rows, a.shape, [&](u32 i) { return (u32)(v >> (i * a.shape.sub_bits)) & m; });
    }
    __syncthreads();
    u32 *out = a.hist + elem * n;
    for (u32 k = tid; k < n; k += 256) if (lh[k]) out[k] += lh[k];
}

// The range assigns inside a Fresh-op witness (assert_in_field's add / sub_unchecked steps: every one a
// RangeChip::assign(limb, w/8, w), big_integer/chip.rs:279-282, 1307-1308): runs of entries {limb, 8 sub-limb bytes} at a stride
struct LookupFreshArgs {
    const u8 *trace; u64 first_off, elem_stride, num_elems;
    u32 n_runs; u32 run_off[16], run_n[16], run_stride[16]; u32 sub_off;   // sub-limb bytes at entry + sub_off
    RangeShape limb; u32 n_rows; u32 *hist; const u8 *status;
};
__global__ __launch_bounds__(64) void lookup_hist_fresh_kernel(LookupFreshArgs a) {
    extern __shared__ u32 lh[];
    const u32 tid = threadIdx.x, n = LOOKUP_ARGS * a.n_rows;
    const u64 elem = blockIdx.x;
    if (a.status && a.status[elem]) return;   // (a failed element's witness is not written: nothing to count)
    for (u32 k = tid; k < n; k += 64) lh[k] = 0;
    __syncthreads();
    const u8 *base = a.trace + elem * a.elem_stride + a.first_off;
    for (u32 r = 0; r < a.n_runs; ++r)
        for (u32 i = tid; i < a.run_n[r]; i += 64) {
            const u8 *p = base + a.run_off[r] + (u64)i * a.run_stride[r] + a.sub_off;
            const u64 sb = (u64) * reinterpret_cast<const u32 *>(p) | ((u64) * reinterpret_cast<const u32 *>(p + 4) << 32);   // 4-byte aligned for 32-bit limbs
            range_count(lh, a.n_rows, a.limb, [&](u32 t) { return (u32)((sb >> (8 * t)) & 0xff); });
        }
    __syncthreads();
    u32 *out = a.hist + elem * n;
    for (u32 k = tid; k < n; k += 64) if (lh[k]) out[k] += lh[k];
}

// ---- tables of one (element, argument) in the workspace ---------------------------------------------------------------
// [0, 32 G)  val      sorted distinct compressed values (Fe)
// then u32 a_start[G + 1] (prefix of the run lengths of A'), a_rank[G] (non-empty runs before g), l_start[G + 1] (prefix of the
// leftover counts), G itself; G <= n_rows.  Slot size is fixed by n_rows.
__host__ __device__ inline u64 lookup_slot_bytes(u32 n_rows) { return (u64)n_rows * 32 + ((u64)(3 * n_rows + 2 + 2) * 4 + 15) / 16 * 16; }

struct LookupSetupArgs {
    const u32 *hist;         // [elem][5][n_rows]
    const u64 *theta;        // [elem][4] canonical (mont: x * R mod p, like every field element of a Montgomery ctx)
    u32 mont;                // H2R_ADVICE_MONTGOMERY: theta comes in, the columns' values go out, in Montgomery form
    u64 num_elems; u32 usable_rows, n_rows, n_lens, arg_mask;
    u32 tag[LOOKUP_MAX_LENS], row_off[LOOKUP_MAX_LENS], bit_len[LOOKUP_MAX_LENS];
    FieldConsts f;
    u8 *ws;                  // [elem][5] slots
    u8 *status;              // [elem]: H2R_E_SHAPE when the inputs do not fit the usable rows (nullable)
};

// inclusive scan of x[0 .. n) (n <= 1,024) by 256 threads: four elements per thread + a Hillis-Steele scan of the partials
__device__ __forceinline__ void block_scan_1024(u32 *x, u32 n, u32 *part /* 256 */) {
    const u32 tid = threadIdx.x;
    const u32 per = (n + 255) / 256, lo = tid * per;
    u32 s = 0;
    for (u32 k = lo; k < lo + per && k < n; ++k) { s += x[k]; x[k] = s; }
    part[tid] = s;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
        const u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const u32 add = tid ? part[tid - 1] : 0;
    for (u32 k = lo; k < lo + per && k < n; ++k) x[k] += add;
    __syncthreads();
}

// LDS of the set-up kernel for a table of n rows (sized by the table, not by LOOKUP_MAX_ROWS: 19 KB instead of 54.5 KB for RSA-2048's 339 rows --
// next to a cells kernel that holds most of a CU's LDS a 54.5 KB workgroup waited for room, 0.9 ms per 1,024 circuits in the records-free flow)
__host__ __device__ inline u32 lookup_setup_lds_bytes(u32 n) { return n * 32u + LOOKUP_MAX_LENS * 32u + 5u * n * 4u + 256u * 4u + 16u; }
__global__ __launch_bounds__(256) void lookup_setup_kernel(LookupSetupArgs a) {
    extern __shared__ uint4 ls_dyn[];
    const u32 tid = threadIdx.x, n = a.n_rows;
    Fe *T = reinterpret_cast<Fe *>(ls_dyn);                 // compressed table rows, then the sorted distinct values
    Fe *tagth = T + n;
    u32 *order = reinterpret_cast<u32 *>(tagth + LOOKUP_MAX_LENS);   // row at sorted position i
    u32 *gid = order + n;                                    // group of sorted position i (inclusive scan of the run heads, minus one)
    u32 *gm = gid + n, *gs = gm + n, *gr = gs + n;           // per group: inputs, table entries, non-empty flag
    u32 *part = gr + n;                                      // [256]
    u32 &total_in = part[256];
    const u32 arg = blockIdx.x % LOOKUP_ARGS;
    const u64 elem = blockIdx.x / LOOKUP_ARGS;
    if (!((a.arg_mask >> arg) & 1u)) return;
    const u32 *h = a.hist + (elem * LOOKUP_ARGS + arg) * n;
    Fe theta;
    for (int k = 0; k < 4; ++k) theta.v[k] = a.theta[elem * 4 + k];
    const bool theta_ok = !ge_p(theta.v, a.f.p);
    if (a.mont && theta_ok) theta = fe_from_mont(theta, a.f);
    if (tid < a.n_lens) tagth[tid] = fe_mul_small(theta, a.tag[tid], a.f.p);
    if (tid == 0) total_in = 0;
    __syncthreads();
    // 1. compress: T[0] = 0 (the (0, 0) row), T[row_off[i] + v] = tag_i * theta + v
    u32 my_in = 0;
    for (u32 r = tid; r < n; r += 256) {
        Fe t = fe_zero();
        if (r) {
            u32 i = 0;
            while (i + 1 < a.n_lens && r >= a.row_off[i + 1]) ++i;
            t = fe_add(tagth[i], fe_small(r - a.row_off[i]), a.f.p);
        }
        T[r] = t;
        my_in += h[r];
    }
    atomicAdd(&total_in, my_in);
    __syncthreads();
    const u32 usable = a.usable_rows;
    const bool fits = total_in <= usable && n <= usable && theta_ok;   // (a challenge that is not a canonical element is refused too)
    if (!fits) {   // more lookup inputs (or table rows) than usable rows: no such circuit.  G = 0 tells the fill kernel to leave the columns alone
        if (tid == 0) {
            if (a.status) a.status[elem] = (u8)H2R_E_SHAPE;
            u32 *gc = reinterpret_cast<u32 *>(a.ws + (elem * LOOKUP_ARGS + arg) * lookup_slot_bytes(n) + (u64)n * 32) + (3 * n + 2);
            gc[0] = 0; gc[1] = 0;
        }
        return;
    }
    // 2. rank by the field's Ord (ties by row index), 3. scatter.  The rows of a bit length are CONSECUTIVE field elements
    //    tag theta + 0, + 1, ... (mod p), so the number of them below a value is a closed form -- n_lens field subtractions per row
    //    instead of n comparisons: the set-up was a fifth of the whole call when it compared every pair of rows
    for (u32 r = tid; r < n; r += 256) {
        const Fe me = T[r];
        u32 rank = r ? 1u : 0u;                               // row 0 = (0, 0): below every other row (or equal with the lower index)
        for (u32 j = 0; j < a.n_lens; ++j) {
            const Fe B = tagth[j];
            const u32 S = 1u << a.bit_len[j];
            auto small = [](const Fe &x, u32 cap) -> u32 { return ((x.v[1] | x.v[2] | x.v[3]) == 0 && x.v[0] < cap) ? (u32)x.v[0] : cap; };
            const Fe d = fe_sub(me, B, a.f.p);               // the offset of `me` behind B on the circle of residues
            const bool ge = !fe_lt(me, B);
            Fe pm; for (int k = 0; k < 4; ++k) pm.v[k] = a.f.p[k];
            const Fe wj = fe_is_zero(B) ? pm : fe_sub(fe_zero(), B, a.f.p);   // p - B: rows before the residues wrap
            const u32 W = small(wj, S);                       // (W = S: the group does not wrap)
            u32 less = ge ? small(d, W) : 0u;                 // rows B + u < me with u < W
            if (W < S) less += small(me, S - W);              // the wrapped rows 0 .. S - W - 1
            const u32 eq_u = small(d, S);                     // the group's row equal to `me`, if any: ties go to the lower row index
            if (eq_u < S && a.row_off[j] + eq_u < r) ++less;
            rank += less;
        }
        order[rank] = r;
    }
    __syncthreads();
    // 4. run heads -> group ids
    for (u32 i = tid; i < n; i += 256) gid[i] = (i == 0 || !fe_eq(T[order[i]], T[order[i - 1]])) ? 1u : 0u;
    for (u32 i = tid; i < n; i += 256) { gm[i] = 0; gs[i] = 0; }
    __syncthreads();
    block_scan_1024(gid, n, part);
    const u32 G = gid[n - 1];
    // 5. per group: inputs M_g and table entries S_g.  Row 0 = (0, 0) also stands for every row where the lookup is off
    //    (inputs) and for the default rows behind the table (table column).
    for (u32 i = tid; i < n; i += 256) {
        const u32 r = order[i], g = gid[i] - 1;
        const u32 m = h[r] + (r == 0 ? usable - total_in : 0u), s = 1u + (r == 0 ? usable - n : 0u);
        if (m) atomicAdd(&gm[g], m);
        atomicAdd(&gs[g], s);
    }
    __syncthreads();
    u8 *slot = a.ws + (elem * LOOKUP_ARGS + arg) * lookup_slot_bytes(n);
    Fe *val = reinterpret_cast<Fe *>(slot);
    u32 *a_start = reinterpret_cast<u32 *>(slot + (u64)n * 32), *a_rank = a_start + n + 1, *l_start = a_rank + n, *gcount = l_start + n + 1;
    // sorted distinct values: the head of every run
    for (u32 i = tid; i < n; i += 256) if (i == 0 || gid[i] != gid[i - 1]) val[gid[i] - 1] = a.mont ? fe_to_mont(T[order[i]], a.f) : T[order[i]];
    // 6. prefix sums over the groups
    for (u32 g = tid; g < G; g += 256) { gr[g] = gm[g] ? 1u : 0u; gs[g] -= gr[g]; }   // leftover = table entries - [value occurs in A]
    __syncthreads();
    block_scan_1024(gm, G, part);
    block_scan_1024(gr, G, part);
    block_scan_1024(gs, G, part);
    for (u32 g = tid; g < G; g += 256) { a_start[g + 1] = gm[g]; a_rank[g] = g ? gr[g - 1] : 0u; l_start[g + 1] = gs[g]; }
    if (tid == 0) { a_start[0] = 0; l_start[0] = 0; gcount[0] = G; gcount[1] = gr[G - 1]; }
}

struct LookupFillArgs {
    const u8 *ws; u64 num_elems; u32 usable_rows, n_rows, arg_mask, rows_per_block, round_robin;
    const u8 *status;
    u8 *a_perm, *s_perm; u64 out_elem_stride;   // element e, argument k at + e * out_elem_stride + k * usable_rows * 32
};
// last index g in [0, G] with start[g] <= x  (start is non-decreasing, start[0] = 0)
__device__ __forceinline__ u32 upper_group(const u32 *start, u32 G, u32 x) {
    u32 lo = 0, hi = G;   // invariant: start[lo] <= x; answer in [lo, hi]
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (start[mid] <= x) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ __launch_bounds__(256) void lookup_fill_kernel(LookupFillArgs a) {
    extern __shared__ uint4 lf_dyn[];
    const u32 tid = threadIdx.x, lane = tid & 63, n = a.n_rows;
    const u32 chunk = blockIdx.x, arg = blockIdx.y;
    const u64 elem = blockIdx.z;
    if (!((a.arg_mask >> arg) & 1u)) return;
    if (a.status && a.status[elem]) return;
    const u64 slot_bytes = lookup_slot_bytes(n);
    const uint4 *src = reinterpret_cast<const uint4 *>(a.ws + (elem * LOOKUP_ARGS + arg) * slot_bytes);
    for (u32 k = tid; k < slot_bytes / 16; k += 256) lf_dyn[k] = src[k];
    __syncthreads();
    const u8 *slot = reinterpret_cast<const u8 *>(lf_dyn);
    const Fe *val = reinterpret_cast<const Fe *>(slot);
    const u32 *a_start = reinterpret_cast<const u32 *>(slot + (u64)n * 32), *a_rank = a_start + n + 1, *l_start = a_rank + n, *gcount = l_start + n + 1;
    const u32 G = gcount[0], n_heads = gcount[1];
    if (G == 0) return;   // the set-up refused this circuit (status H2R_E_SHAPE)
    const u32 usable = a.usable_rows, n_rep = usable - n_heads;   // repeated rows = leftover table entries
    u8 *ap = a.a_perm + elem * a.out_elem_stride + (u64)arg * usable * 32;
    u8 *sp = a.s_perm + elem * a.out_elem_stride + (u64)arg * usable * 32;
    // [r6] A workgroup takes ONE contiguous run of rows_per_block rows of the column (256 shipped: 8 KB of A' and of S', one 64-row pass per wave).
    // With round 5's 8,192 rows per workgroup the resident workgroups wrote a comb of 8 KB pieces 256 KB apart, which is what the placement
    // classes punish; with 256 they write one dense window and the same buffers give 2.02-2.16 -> 1.76-1.90 ms (one class) / 1.63 -> 1.56-1.59 ms
    // (two classes) per 10.7 GB call (tools/store_pattern_probe.hip, tools/lookup_geometry_probe.py, profiles/r06_lookup_geometry.txt).  The slot
    // every workgroup stages comes from the L2.  round_robin (developer sweeps only): the column's workgroups take its 256-row blocks in turn
    // instead -- measured equal or worse at every size, off in the product.
    const bool rr = a.round_robin != 0;
    const u32 p0 = rr ? chunk * 256 : chunk * a.rows_per_block;
    const u32 p1 = rr ? usable : (p0 + a.rows_per_block < usable ? p0 + a.rows_per_block : usable);
    const u32 base_step = rr ? 256 * gridDim.x : 256;
    // a wave works on 64 consecutive rows: every lane computes one row, then the lanes exchange halves so that each of the
    // four store instructions writes 64 consecutive 16-byte units (rows base .. base+31, then base+32 .. base+63)
    for (u32 base = p0 + (tid & ~63u); base < p1; base += base_step) {
        const u32 pos = base + lane;
        Fe av = fe_zero(), sv = fe_zero();
        if (pos < p1) {
            const u32 g = upper_group(a_start, G, pos);      // start[g] <= pos < start[g + 1] (runs of length 0 share their successor's start)
            av = val[g];
            if (pos == a_start[g]) sv = av;
            else {
                const u32 j = pos - (a_rank[g] + 1);          // index among the repeated rows
                const u32 li = n_rep - 1 - j;                  // leftover values ascending are handed out from the LAST repeated row
                sv = val[upper_group(l_start, G, li)];
            }
        }
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int srcl = 32 * hlf + (int)(lane >> 1);
            const bool hi = lane & 1;
            const u64 a0 = __shfl(av.v[0], srcl), a1 = __shfl(av.v[1], srcl), a2 = __shfl(av.v[2], srcl), a3 = __shfl(av.v[3], srcl);
            const u64 s0 = __shfl(sv.v[0], srcl), s1 = __shfl(sv.v[1], srcl), s2 = __shfl(sv.v[2], srcl), s3 = __shfl(sv.v[3], srcl);
            const u32 row = base + 32 * hlf + (lane >> 1);
            if (row < p1) {
                st16(ap + (u64)row * 32 + (hi ? 16 : 0), hi ? a2 : a0, hi ? a3 : a1);
                st16(sp + (u64)row * 32 + (hi ? 16 : 0), hi ? s2 : s0, hi ? s3 : s1);
            }
        }
    }
}

}  // namespace h2r
