//! The reference-side binding of libh2r: `GpuBigIntChip` / `GpuRSAChip`, drop-in for the hot path of
//! `BigIntInstructions` (`mul`, `mul_mod`, `square_mod`, `pow_mod`, `pow_mod_fixed_exp`: reference
//! src/big_integer/instructions.rs:7-260) and `RSAInstructions::modpow_public_key` (src/instructions.rs:8-39).
//!
//! What it does: the WITNESS VALUES come from the GPU -- one `h2r_pow_mod_fixed_exp_batch` / `h2r_pow_mod_batch` call, then
//! `h2r_pow_trace_emit_stream`, which lays every value the reference assigns out in the order its code issues the calls -- and this
//! file replays that control flow (big_integer/chip.rs:386-419 mul, 542-629 mul_mod, 642-649 square_mod, 664-696 pow_mod, 710-742
//! pow_mod_fixed_exp, 822-895 is_equal_muled, 1053-1063 assert_equal_muled, 1323-1349 div_mod_main_gate; src/chip.rs:99-114
//! modpow_public_key), handing each value to the main gate / range chip as an UNASSIGNED term of the same gate the reference uses.
//! No BigUint arithmetic and no field multiplication happens on the host.
//!
//! Status of this file: Rust cannot be compiled in the image this library was built in (no cargo, no rustc), so
//! the file is written against the crates the reference pins (Cargo.toml:12-16: halo2wrong / maingate rev 63bde545, num-bigint 0.4)
//! and its compiled TWIN is tests/cpp/test_chip_replay.cpp: the same functions, statement for statement, against a mock RegionCtx
//! that checks every gate relation, counts every call (19 mul_mods, 38,912 mul_add, 2,394 range assigns per RSA-2048 e = 65537
//! element) and requires the stream to be consumed to the last byte, for the Fix and the Var arm, on the GPU.  tests/test_chip_replay.py
//! keeps the two files in step (same methods, same reference line anchors, every `ffi::` name declared in include/h2r.h).
//! [3P] marks what is restated from recollection of maingate's API (`MainGate::apply`, `Term`, `CombinationOptionCommon`,
//! `RangeChip::assign`'s decompose): a maintainer who links the real crate adjusts those few call sites, not the replay.
//!
//! A witness-only backend (a device prover that takes columns instead of `assign_advice` calls) skips this file altogether: a ctx
//! created with H2R_ADVICE_COLUMNS | H2R_ADVICE_MONTGOMERY makes `h2r_modpow_public_key_emit_advice` write the advice columns as
//! vectors of F in its in-memory form, and `h2r_advice_check` is the MockProver for that image.
use crate::big_integer::{AssignedInteger, AssignedLimb, BigIntChip, BigIntConfig, Fresh, Muled};
use crate::{AssignedRSAPublicKey, AssignedRSAPubE};
use halo2wrong::halo2::{arithmetic::FieldExt, circuit::Value, plonk::Error};
use maingate::{
    big_to_fe, AssignedValue, CombinationOptionCommon, MainGate, MainGateInstructions, RangeChip, RangeInstructions, RegionCtx, Term,
};
use num_bigint::BigUint;
use std::os::raw::c_void;

use crate::gpu::ffi; // the `extern "C"` block of INTEGRATION.md section 1 (bindgen of include/h2r.h)

/// One element's flat witness stream (`h2r_pow_trace_emit_stream` with H2R_STREAM_FIELD_AB), copied to the host.
/// Widths come from `h2r_layout`: LIMB `limb_bytes`, WIDE `wide_bytes`, CARRY `carry_bytes`; a_b is a 32-byte canonical element.
pub struct WitnessStream<'a> {
    bytes: &'a [u8],
    pos: usize,
    lo: &'a ffi::h2r_layout,
}

impl<'a> WitnessStream<'a> {
    fn take<F: FieldExt>(&mut self, n: usize) -> Value<F> {
        let mut repr = [0u8; 32];
        repr[..n].copy_from_slice(&self.bytes[self.pos..self.pos + n]);
        self.pos += n;
        // every value of the path is below 2^135 << p, and a_b arrives as its canonical element: from_repr never fails
        Value::known(F::from_repr(repr.into()).unwrap())
    }
    fn limb<F: FieldExt>(&mut self) -> Value<F> { self.take(self.lo.limb_bytes as usize) }
    fn wide<F: FieldExt>(&mut self) -> Value<F> { self.take(self.lo.wide_bytes as usize) }
    fn carry<F: FieldExt>(&mut self) -> Value<F> { self.take(self.lo.carry_bytes as usize) }
    fn field<F: FieldExt>(&mut self) -> Value<F> { self.take(32) }
    fn flag<F: FieldExt>(&mut self) -> Value<F> { self.take(1) }
    fn bytes(&mut self, n: usize) -> &'a [u8] { let r = &self.bytes[self.pos..self.pos + n]; self.pos += n; r }
    pub fn finished(&self) -> bool { self.pos == self.bytes.len() }
}

pub struct GpuBigIntChip<F: FieldExt> {
    cpu: BigIntChip<F>,          // configuration, main gate, range chip: the reference's own (big_integer/chip.rs:17-51)
    ctx: *mut ffi::h2r_ctx,      // h2r_ctx_create(limb_width, bits_len, field)
    lo: ffi::h2r_layout,         // h2r_trace_layout
}

impl<F: FieldExt> GpuBigIntChip<F> {
    fn main_gate(&self) -> MainGate<F> { self.cpu.main_gate() }
    fn range_chip(&self) -> RangeChip<F> { self.cpu.range_chip() }

    // ---- main-gate ops whose OUTPUT is given instead of computed.  [3P] maingate builds every op from `apply`; these are the
    // same term lists its own mul_add / add / sub / ... use, with the result as an unassigned term.
    fn mul_add_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, b: &AssignedValue<F>, acc: &AssignedValue<F>, out: Value<F>)
        -> Result<AssignedValue<F>, Error> {
        // a * b + acc - out = 0
        Ok(self.main_gate().apply(ctx, [Term::assigned_to_mul(a), Term::assigned_to_mul(b), Term::assigned_to_add(acc), Term::unassigned_to_sub(out)],
                                  F::zero(), CombinationOptionCommon::OneLinerMul.into())?.swap_remove(3))
    }
    fn add_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, b: &AssignedValue<F>, k: F, out: Value<F>) -> Result<AssignedValue<F>, Error> {
        // a + b + k - out = 0   (add, add_with_constant)
        Ok(self.main_gate().apply(ctx, [Term::assigned_to_add(a), Term::assigned_to_add(b), Term::unassigned_to_sub(out)], k,
                                  CombinationOptionCommon::OneLinerAdd.into())?.swap_remove(2))
    }
    fn sub_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, b: &AssignedValue<F>, out: Value<F>) -> Result<AssignedValue<F>, Error> {
        // a - b - out = 0
        Ok(self.main_gate().apply(ctx, [Term::assigned_to_add(a), Term::assigned_to_sub(b), Term::unassigned_to_sub(out)], F::zero(),
                                  CombinationOptionCommon::OneLinerAdd.into())?.swap_remove(2))
    }
    fn add_constant_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, k: F, out: Value<F>) -> Result<AssignedValue<F>, Error> {
        Ok(self.main_gate().apply(ctx, [Term::assigned_to_add(a), Term::unassigned_to_sub(out)], k, CombinationOptionCommon::OneLinerAdd.into())?.swap_remove(1))
    }
    fn mul_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, b: &AssignedValue<F>, out: Value<F>) -> Result<AssignedValue<F>, Error> {
        Ok(self.main_gate().apply(ctx, [Term::assigned_to_mul(a), Term::assigned_to_mul(b), Term::unassigned_to_sub(out)], F::zero(),
                                  CombinationOptionCommon::OneLinerMul.into())?.swap_remove(2))
    }
    // is_equal / and keep maingate's own gadgets (their internal inverse witness is maingate's business); the stream's flag byte is
    // the value the gadget will arrive at -- consumed so that the stream stays in step, asserted in debug builds
    fn is_equal_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, b: &AssignedValue<F>, flag: Value<F>) -> Result<AssignedValue<F>, Error> {
        let r = self.main_gate().is_equal(ctx, a, b)?;
        debug_assert!(r.value().zip(flag.as_ref()).map(|(x, y)| x == y).assert_if_known(|ok| *ok));
        Ok(r)
    }
    fn and_w(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, b: &AssignedValue<F>, flag: Value<F>) -> Result<AssignedValue<F>, Error> {
        let r = self.main_gate().and(ctx, a, b)?;
        debug_assert!(r.value().zip(flag.as_ref()).map(|(x, y)| x == y).assert_if_known(|ok| *ok));
        Ok(r)
    }
    // RangeChip::assign decomposes the value itself ([3P] maingate's `decompose`): byte / 5-bit splits of a 64-bit word are cheap on the
    // host, so the stream's sub-limb bytes are only skipped (they are what the lookup argument is built from: h2r_lookup_hist_*)
    fn range_assign_w(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, value: Value<F>, limb_bit_len: usize, bit_len: usize, nsub: usize)
        -> Result<AssignedValue<F>, Error> {
        let _sub_limbs = s.bytes(nsub);
        self.range_chip().assign(ctx, value, limb_bit_len, bit_len)
    }

    /// big_integer/chip.rs:386-419
    pub fn mul(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, a: &AssignedInteger<F, Fresh>, b: &AssignedInteger<F, Fresh>)
        -> Result<AssignedInteger<F, Muled>, Error> {
        let (d0, d1) = (a.num_limbs(), b.num_limbs());
        let d = d0 + d1 - 1;
        let main_gate = self.main_gate();
        let mut c_vals = Vec::new();
        for i in 0..d {
            let mut acc = main_gate.assign_constant(ctx, big_to_fe(BigUint::default()))?;
            let mut j = if d1 >= i + 1 { 0 } else { i + 1 - d1 };
            while j < d0 && j <= i {
                let k = i - j;
                let (a_limb, b_limb) = (AssignedValue::from(a.limb(j)), AssignedValue::from(b.limb(k)));
                acc = self.mul_add_w(ctx, &a_limb, &b_limb, &acc, s.wide())?;          // :408
                j += 1;
            }
            c_vals.push(acc);
        }
        Ok(AssignedInteger::new(&c_vals.into_iter().map(AssignedLimb::<_, Muled>::from).collect::<Vec<_>>()))
    }

    /// big_integer/chip.rs:1323-1349
    fn div_mod_main_gate(&self, ctx: &mut RegionCtx<'_, F>, a: &AssignedValue<F>, n: &AssignedValue<F>, q_v: Value<F>, r_v: Value<F>,
                         nq_v: Value<F>, a_sub_nq_v: Value<F>) -> Result<(AssignedValue<F>, AssignedValue<F>), Error> {
        let main_gate = self.main_gate();
        let (q, a_mod_n) = (main_gate.assign_value(ctx, q_v)?, main_gate.assign_value(ctx, r_v)?);
        let nq = self.mul_w(ctx, n, &q, nq_v)?;
        let a_sub_nq = self.sub_w(ctx, a, &nq, a_sub_nq_v)?;
        main_gate.assert_equal(ctx, &a_mod_n, &a_sub_nq)?;
        Ok((q, a_mod_n))
    }

    /// big_integer/chip.rs:822-895
    pub fn is_equal_muled(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, a: &AssignedInteger<F, Muled>, b: &AssignedInteger<F, Muled>,
                          num_limbs_l: usize, num_limbs_r: usize) -> Result<AssignedValue<F>, Error> {
        let min_n = num_limbs_l.min(num_limbs_r);
        let word_max = BigIntChip::<F>::compute_mul_word_max(self.cpu.limb_width(), min_n);
        let limb_width = self.cpu.limb_width();
        let num_limbs = num_limbs_l + num_limbs_r - 1;
        let carry_bits = BigIntChip::<F>::bits_size(&(&word_max * 2u32)) - limb_width;
        let main_gate = self.main_gate();
        let limb_max = main_gate.assign_constant(ctx, F::from_u128(1 << limb_width))?;
        let mut accumulated_extra = main_gate.assign_constant(ctx, F::zero())?;
        let mut carry = Vec::with_capacity(num_limbs);
        let mut cs = Vec::with_capacity(num_limbs);
        carry.push(main_gate.assign_constant(ctx, F::zero())?);
        let mut eq_bit = main_gate.assign_bit(ctx, Value::known(F::one()))?;
        for i in 0..num_limbs {
            let a_b = self.sub_w(ctx, &a.limb(i), &b.limb(i), s.field())?;                                   // :859
            let sum = self.add_w(ctx, &a_b, &carry[i], big_to_fe(word_max.clone()), s.wide())?;              // :860-861
            let (q1, r1, nq1, amnq1) = (s.carry(), s.limb(), s.wide(), s.limb());
            let (new_carry, c) = self.div_mod_main_gate(ctx, &sum, &limb_max, q1, r1, nq1, amnq1)?;          // :864
            carry.push(new_carry);
            cs.push(c);
            accumulated_extra = self.add_constant_w(ctx, &accumulated_extra, big_to_fe(word_max.clone()), s.wide())?;   // :869-870
            let (q2, r2, nq2, amnq2) = (s.carry(), s.limb(), s.wide(), s.limb());
            let (q_acc, mod_acc) = self.div_mod_main_gate(ctx, &accumulated_extra, &limb_max, q2, r2, nq2, amnq2)?;     // :871
            let cs_acc_eq = self.is_equal_w(ctx, &cs[i], &mod_acc, s.flag())?;                               // :873
            eq_bit = self.and_w(ctx, &eq_bit, &cs_acc_eq, s.flag())?;
            accumulated_extra = q_acc;
            if i < num_limbs - 1 {
                let carry_value = s.carry();                                                                 // = carry[i + 1]
                let range_assigned = self.range_assign_w(ctx, s, carry_value, BigIntChip::<F>::sublimb_bit_len(carry_bits), carry_bits,
                                                         self.lo.carry_nsub as usize)?;                      // :879-885
                let range_eq = self.is_equal_w(ctx, &carry[i + 1], &range_assigned, s.flag())?;
                eq_bit = self.and_w(ctx, &eq_bit, &range_eq, s.flag())?;
            } else {
                let final_carry_eq = self.is_equal_w(ctx, &carry[i + 1], &accumulated_extra, s.flag())?;     // :890
                eq_bit = self.and_w(ctx, &eq_bit, &final_carry_eq, s.flag())?;
            }
        }
        Ok(eq_bit)
    }

    /// big_integer/chip.rs:542-629
    pub fn mul_mod(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, a: &AssignedInteger<F, Fresh>, b: &AssignedInteger<F, Fresh>,
                   n: &AssignedInteger<F, Fresh>) -> Result<AssignedInteger<F, Fresh>, Error> {
        let limb_width = self.cpu.limb_width();
        let (n1, n2) = (a.num_limbs(), b.num_limbs());
        assert_eq!(n1, n.num_limbs());                                                                       // :555
        let sub_bits = BigIntChip::<F>::sublimb_bit_len(limb_width);
        let nsub = self.lo.limb_nsub as usize;
        let mut quotient_limbs = Vec::with_capacity(n2);
        for _ in 0..n2 { let q = s.limb(); quotient_limbs.push(AssignedLimb::<F, Fresh>::from(self.range_assign_w(ctx, s, q, sub_bits, limb_width, nsub)?)); }   // :588-591
        let mut prod_limbs = Vec::with_capacity(n1);
        for _ in 0..n1 { let p = s.limb(); prod_limbs.push(AssignedLimb::<F, Fresh>::from(self.range_assign_w(ctx, s, p, sub_bits, limb_width, nsub)?)); }       // :596-599
        let (quotient_int, prod_int) = (AssignedInteger::new(&quotient_limbs), AssignedInteger::new(&prod_limbs));
        let ab = self.mul(ctx, s, a, b)?;                                                                    // :608
        let qn = self.mul(ctx, s, &quotient_int, n)?;                                                        // :609
        let n_sum = n1 + n2;
        let (mut eq_a_limbs, mut eq_b_limbs) = (Vec::with_capacity(n_sum - 1), Vec::with_capacity(n_sum - 1));
        for i in 0..(n_sum - 1) {
            eq_a_limbs.push(AssignedLimb::<F, Muled>::from(ab.limb(i)));
            if i < n1 {
                let sum = self.add_w(ctx, &qn.limb(i), &prod_int.limb(i), F::zero(), s.wide())?;             // :617
                eq_b_limbs.push(AssignedLimb::<F, Muled>::from(sum));
            } else {
                eq_b_limbs.push(AssignedLimb::<F, Muled>::from(qn.limb(i)));
            }
        }
        let (eq_a, eq_b) = (AssignedInteger::new(&eq_a_limbs), AssignedInteger::new(&eq_b_limbs));
        let eq_bit = self.is_equal_muled(ctx, s, &eq_a, &eq_b, n1, n2)?;                                     // assert_equal_muled :1053-1063
        self.main_gate().assert_one(ctx, &eq_bit)?;
        Ok(prod_int)
    }

    /// big_integer/chip.rs:642-649
    pub fn square_mod(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, a: &AssignedInteger<F, Fresh>, n: &AssignedInteger<F, Fresh>)
        -> Result<AssignedInteger<F, Fresh>, Error> {
        self.mul_mod(ctx, s, a, a, n)
    }

    /// big_integer/chip.rs:710-742.  `s`: the element's stream of `h2r_pow_mod_fixed_exp_batch` + `h2r_pow_trace_emit_stream`.
    pub fn pow_mod_fixed_exp(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, a: &AssignedInteger<F, Fresh>, e: &BigUint,
                             n: &AssignedInteger<F, Fresh>) -> Result<AssignedInteger<F, Fresh>, Error> {
        let num_e_bits = BigIntChip::<F>::bits_size(e);
        let e_bits = e.to_bytes_le().into_iter().flat_map(|v| (0..8).map(move |i: u8| (v >> i) & 1u8 == 1u8)).collect::<Vec<bool>>();
        let e_bits = e_bits[0..num_e_bits].to_vec();
        let mut acc = self.cpu.assign_constant(ctx, BigUint::from(1usize), a.num_limbs())?;                  // :729 (constants: nothing in the stream)
        let mut squared = a.clone();
        for e_bit in e_bits.into_iter() {
            let cur_sq = squared;
            squared = self.square_mod(ctx, s, &cur_sq, n)?;
            if !e_bit { continue; }
            acc = self.mul_mod(ctx, s, &acc, &cur_sq, n)?;
        }
        let _result_limbs = s.bytes(a.num_limbs() * self.lo.limb_bytes as usize);   // the stream ends with x^e mod n: equal to acc's limbs
        debug_assert!(s.finished());
        Ok(acc)
    }

    /// big_integer/chip.rs:664-696.  `s`: the stream of `h2r_pow_mod_batch`: the exponent's bits first (one byte each), then per bit
    /// mul_mod(acc, squared), the selected limbs, square_mod.
    pub fn pow_mod(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, a: &AssignedInteger<F, Fresh>, e: &AssignedInteger<F, Fresh>,
                   n: &AssignedInteger<F, Fresh>, exp_limb_bits: usize) -> Result<AssignedInteger<F, Fresh>, Error> {
        let main_gate = self.main_gate();
        let mut e_bits = Vec::new();
        for e_limb in e.limbs().into_iter() {
            let _bits = s.bytes(exp_limb_bits);                                   // to_bits decomposes the limb itself
            e_bits.append(&mut main_gate.to_bits(ctx, &AssignedValue::from(e_limb), exp_limb_bits)?);        // :677
        }
        let mut acc = self.cpu.assign_constant_fresh(ctx, BigUint::from(1usize))?;                           // :682
        let mut squared = a.clone();
        for e_bit in e_bits.into_iter() {
            let muled = self.mul_mod(ctx, s, &acc, &squared, n)?;                                            // :686
            let mut selected = Vec::with_capacity(self.cpu.num_limbs());
            for j in 0..self.cpu.num_limbs() {
                let _sel = s.limb::<F>();                                         // select computes it from its inputs
                selected.push(AssignedLimb::<F, Fresh>::from(main_gate.select(ctx, &muled.limb(j), &acc.limb(j), &e_bit)?));   // :688-691
            }
            acc = AssignedInteger::new(&selected);
            squared = self.square_mod(ctx, s, &squared, n)?;                                                 // :693
        }
        let _result_limbs = s.bytes(a.num_limbs() * self.lo.limb_bytes as usize);
        debug_assert!(s.finished());
        Ok(acc)
    }

    /// The GPU side of one call: x, n (and e limbs) of `batch` independent circuits in, one flat stream per circuit out.
    /// Returns the host copy of all streams and the bytes per element.  A prover service keeps the device buffers and calls this once
    /// per batch of proofs; `status[i] != H2R_OK` is the condition on which the reference's CPU chip panics (chip.rs:566, 583-584).
    pub unsafe fn witness_streams(&self, d_x: *const c_void, d_n: *const c_void, e: &BigUint, batch: u64, stream: *mut c_void)
        -> Result<(Vec<u8>, u64), i32> {
        let e_le = e.to_bytes_le();
        let mut pl = std::mem::zeroed::<ffi::h2r_pow_layout>();
        ffi::check(ffi::h2r_pow_fixed_layout(self.ctx, e_le.as_ptr(), e_le.len(), &mut pl))?;
        let d_trace = ffi::device_alloc(batch * pl.elem_stride)?;
        let d_ws = ffi::device_alloc(ffi::h2r_workspace_bytes(self.ctx, batch, pl.num_mul_mods))?;
        let d_out = ffi::device_alloc(batch * self.cpu.num_limbs() as u64 * self.lo.limb_bytes as u64)?;
        let d_status = ffi::device_alloc(batch)?;
        ffi::check(ffi::h2r_pow_mod_fixed_exp_batch(self.ctx, d_x, d_n, e_le.as_ptr(), e_le.len(), batch, 0, d_trace.ptr(), d_out.ptr(),
                                                    d_status.ptr() as *mut u8, d_ws.ptr(), stream))?;
        let sb = ffi::h2r_pow_stream_bytes(self.ctx, &pl, ffi::H2R_STREAM_FIELD_AB);
        let d_stream = ffi::device_alloc(batch * sb)?;
        ffi::check(ffi::h2r_pow_trace_emit_stream(self.ctx, &pl, d_trace.ptr(), 0, batch, ffi::H2R_STREAM_FIELD_AB, d_stream.ptr(), sb, 0, stream))?;
        let status = d_status.to_host(stream)?;
        if let Some(bad) = status.iter().find(|&&st| st != ffi::H2R_OK as u8) { return Err(*bad as i32); }
        Ok((d_stream.to_host(stream)?, sb))
    }
}

/// `RSAInstructions::modpow_public_key` (src/chip.rs:99-114) over the GPU chip: assert_in_field stays the reference's (one call per
/// modpow: 262 range assigns; its witness also exists on the device -- h2r_modpow_public_key_batch's in_field_trace -- for a
/// columns-only backend), the exponentiation is replayed from the stream.
pub struct GpuRSAChip<F: FieldExt> { pub bigint: GpuBigIntChip<F>, pub exp_limb_bits: usize }

impl<F: FieldExt> GpuRSAChip<F> {
    pub fn modpow_public_key(&self, ctx: &mut RegionCtx<'_, F>, s: &mut WitnessStream<'_>, x: &AssignedInteger<F, Fresh>, public_key: &AssignedRSAPublicKey<F>)
        -> Result<AssignedInteger<F, Fresh>, Error> {
        self.bigint.cpu.assert_in_field(ctx, x, &public_key.n)?;                                             // src/chip.rs:106
        match &public_key.e {
            AssignedRSAPubE::Var(e) => self.bigint.pow_mod(ctx, s, x, e, &public_key.n, self.exp_limb_bits),          // :108-110
            AssignedRSAPubE::Fix(e) => self.bigint.pow_mod_fixed_exp(ctx, s, x, e, &public_key.n),                     // :111
        }
    }
}
