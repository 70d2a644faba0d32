"""TEST INFRASTRUCTURE ONLY -- independent big-int restatement of the halo2-rsa hot path.

This file is the *second*, independent checker beside ``oracle/h2r_oracle.c``: it
restates the reference's control flow with Python's arbitrary-precision ints so
that the C oracle (schoolbook + Knuth-D on 64-bit limbs) can be cross-checked
against something that shares no arithmetic code with it.  It is used by
``tests/`` and by ``tests/golden/make_golden.py`` to mint the committed fixtures.
Nothing in the product path (``halo2_rsa_amd/``) may import it.

Every function cites the reference file:line it follows (paths relative to
``/root/reference``; that tree is NOT read at run time).

Op-trace stream ("flat stream")
-------------------------------
The oracle emits, in the exact order in which the reference issues its
``assign``/``main_gate`` calls, every *witness value* of the path as a
little-endian fixed-width integer (constants bound by ``assign_constant`` are
not emitted).  Widths (bytes) depend only on ``(limb_width w, num_limbs L)``:

=========  ==========================================  (w=64,L=32)  (w=32,L=128)
LIMB       w/8                                          8            4
WIDE       8*ceil((bits(word_max)+2)/64)                24           16
CARRY      8*ceil(carry_bits/64)                        16           8
sub-limb   1                                            1            1
flag/bit   1                                            1            1
=========  ==========================================

Besides ``assign_constant`` cells, four input-independent witnesses are NOT emitted: the literal ``1`` the reference
assigns with ``main_gate.assign_bit(Value::known(F::one()))`` as the start of a running AND / comparison
(big_integer/chip.rs:326 ``sub``'s ``one``, :761 ``is_zero``, :791 ``is_equal_fresh``, :856 ``is_equal_muled``).
They are constants in all but name and are treated like the ``assign_constant`` cells around them (SURVEY 8a lists
:851-856 as the step's constant preamble; its byte counts exclude them).

``a_b = a[i]-b[i]`` (big_integer/chip.rs:859) is a *field* subtraction in the
reference; the stream stores it as a WIDE two's-complement signed integer (the
canonical field element is ``v mod p``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

NUM_LOOKUP_LIMBS = 8  # big_integer/chip.rs:1163

# Moduli of the four fields the reference instantiates its chips with (examples/rsa_example.rs:148, benches/bench.rs:35:
# bn256::Fr; tests big_integer/chip.rs:1461-1463: bn256::Fq, pasta::Fp, pasta::Fq)
FIELD_MODULI = {
    "bn254_fr": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "bn254_fq": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "pasta_fp": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
    "pasta_fq": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
}


def bits_size(v: int) -> int:
    """big_integer/chip.rs:1352-1354 (BigUint::bits)."""
    return v.bit_length()


def sublimb_bit_len(bit_len_limb: int) -> int:
    """big_integer/chip.rs:1357-1365."""
    val = bit_len_limb // NUM_LOOKUP_LIMBS
    return 1 if val == 0 else val


def compute_mul_word_max(limb_width: int, min_n: int) -> int:
    """big_integer/chip.rs:1368-1372."""
    out_base = 1 << limb_width
    return min_n * (out_base - 1) * (out_base - 1) + (out_base - 1)


def compute_range_lens(limb_width: int, num_limbs: int) -> Tuple[List[int], List[int]]:
    """big_integer/chip.rs:1220-1249."""
    out_comp_bit_len = limb_width // NUM_LOOKUP_LIMBS
    out_overflow_bit_len = limb_width % out_comp_bit_len
    out_base = 1 << limb_width
    fresh_word_max_width = (2 * out_base).bit_length()
    fresh_carry_bits = fresh_word_max_width - limb_width
    fresh_carry_comp_bit_len = sublimb_bit_len(fresh_carry_bits)
    fresh_carry_overflow_bit_len = fresh_carry_bits % fresh_carry_comp_bit_len
    mul_word_max = num_limbs * (out_base - 1) * (out_base - 1) + (out_base - 1)
    mul_word_max_width = (mul_word_max * 2).bit_length()
    mul_carry_bits = mul_word_max_width - limb_width
    mul_carry_comp_bit_len = sublimb_bit_len(mul_carry_bits)
    mul_carry_overflow_bit_len = mul_carry_bits % mul_carry_comp_bit_len
    return (
        [out_comp_bit_len, fresh_carry_comp_bit_len, mul_carry_comp_bit_len],
        [out_overflow_bit_len, fresh_carry_overflow_bit_len, mul_carry_overflow_bit_len],
    )


def rsa_compute_range_lens(num_limbs: int) -> Tuple[List[int], List[int]]:
    """src/chip.rs:249-254 (LIMB_WIDTH = 64, src/chip.rs:203)."""
    comp, over = compute_range_lens(64, num_limbs)
    comp.append(32 // NUM_LOOKUP_LIMBS)
    return comp, over


def refresh_aux_increased_limbs(limb_width: int, num_limbs_l: int, num_limbs_r: int) -> List[int]:
    """big_integer/mod.rs:428-482 (RefreshAux::new) -> increased_limbs_vec."""
    max_limb = (1 << limb_width) - 1
    d = num_limbs_l + num_limbs_r - 1
    muled = []
    for i in range(d):
        j = 0 if num_limbs_r >= i + 1 else i + 1 - num_limbs_r
        acc = 0
        while j < num_limbs_l and j <= i:
            acc += max_limb * max_limb
            j += 1
        muled.append(acc)
    out = []
    cur_d = 0
    max_d = d
    while cur_d <= max_d:
        if cur_d >= len(muled):
            muled.append(0)
        nb = muled[cur_d].bit_length()
        num_chunks = nb // limb_width if nb % limb_width == 0 else nb // limb_width + 1
        out.append(num_chunks - 1)
        chunks = []
        for _ in range(num_chunks):
            chunks.append(muled[cur_d] & max_limb)
            muled[cur_d] >>= limb_width
        assert muled[cur_d] == 0
        for j in range(num_chunks):
            if len(muled) <= cur_d + j:
                muled.append(0)
            muled[cur_d + j] += chunks[j]
        cur_d += 1
    return out


@dataclass(frozen=True)
class Params:
    """Derived constants of BigIntChip::new(config, limb_width, bits_len) (chip.rs:1174-1185)."""

    w: int
    L: int
    field_modulus: int = 0   # p of the circuit's field F: when set, a_b (chip.rs:859) is streamed as its canonical
                             # 32-byte field element (a_b mod p) instead of the WIDE two's-complement integer

    def __post_init__(self):
        assert self.w % 8 == 0 and self.L >= 1

    @property
    def B(self) -> int:
        return 1 << self.w

    @property
    def word_max(self) -> int:
        return compute_mul_word_max(self.w, self.L)

    @property
    def carry_bits(self) -> int:
        # big_integer/chip.rs:841-842
        return bits_size(self.word_max * 2) - self.w

    @property
    def LB(self) -> int:
        return self.w // 8

    @property
    def WB(self) -> int:
        return 8 * ((bits_size(self.word_max) + 2 + 63) // 64)

    @property
    def CB(self) -> int:
        return 8 * ((self.carry_bits + 63) // 64)

    @property
    def limb_sub_bits(self) -> int:
        return sublimb_bit_len(self.w)

    @property
    def carry_sub_bits(self) -> int:
        return sublimb_bit_len(self.carry_bits)

    def n_sublimbs(self, bit_len: int) -> int:
        s = sublimb_bit_len(bit_len)
        return bit_len // s + (1 if bit_len % s else 0)

    @property
    def mul_mod_stream_bytes(self) -> int:
        L, C = self.L, 2 * self.L - 1
        nl, nc = self.n_sublimbs(self.w), self.n_sublimbs(self.carry_bits)
        per_col = 5 * self.WB + 2 * self.CB + 4 * self.LB + 4
        return (2 * L * (self.LB + nl) + 2 * L * L * self.WB + L * self.WB
                + C * per_col + (C - 1) * (self.CB + nc))


class Stream:
    """Little-endian fixed-width value stream (see module docstring)."""

    def __init__(self):
        self.buf = bytearray()

    def put(self, v: int, nbytes: int, signed: bool = False):
        self.buf += int(v).to_bytes(nbytes, "little", signed=signed)

    def bytes(self) -> bytes:
        return bytes(self.buf)


def to_limbs(v: int, L: int, w: int) -> List[int]:
    """maingate decompose_big [3P]: little-endian split (cf. mod.rs:348-359 for the inverse)."""
    m = (1 << w) - 1
    out = [(v >> (w * i)) & m for i in range(L)]
    assert v >> (w * L) == 0, "value does not fit"
    return out


def from_limbs(limbs: Sequence[int], w: int) -> int:
    """big_integer/mod.rs:348-359 (to_big_uint)."""
    return sum(int(l) << (w * i) for i, l in enumerate(limbs))


def range_sublimbs(v: int, sub_bits: int, bit_len: int) -> List[int]:
    """RangeChip::assign [3P maingate]: bit_len/sub_bits composition sub-limbs (+1 overflow)."""
    n = bit_len // sub_bits + (1 if bit_len % sub_bits else 0)
    m = (1 << sub_bits) - 1
    return [(v >> (sub_bits * t)) & m for t in range(n)]


def emit_range_assign(st: Stream, v: int, sub_bits: int, bit_len: int, nbytes: int):
    """One RangeChip::assign call: the value, then its sub-limbs (LSB first)."""
    st.put(v, nbytes)
    for s in range_sublimbs(v, sub_bits, bit_len):
        st.put(s, 1)


def mul_columns(a: Sequence[int], b: Sequence[int], st: Stream | None, WB: int) -> List[int]:
    """BigIntChip::mul, big_integer/chip.rs:386-419: un-carried column sums, every partial emitted."""
    d0, d1 = len(a), len(b)
    d = d0 + d1 - 1
    cols = []
    for i in range(d):
        acc = 0  # assign_constant(0), chip.rs:402 (not emitted)
        j = 0 if d1 >= i + 1 else i + 1 - d1
        while j < d0 and j <= i:
            acc = a[j] * b[i - j] + acc  # main_gate.mul_add, chip.rs:408
            if st is not None:
                st.put(acc, WB)
            j += 1
        cols.append(acc)
    return cols


def is_equal_muled(p: Params, a: Sequence[int], b: Sequence[int], st: Stream, n_l: int | None = None, n_r: int | None = None) -> int:
    """BigIntChip::is_equal_muled, big_integer/chip.rs:822-895.  Default num_limbs_l = num_limbs_r = L; any (n_l, n_r) with
    operands of at most L limbs otherwise (the stream's WIDE / CARRY widths stay those of the chip's L)."""
    w, B = p.w, p.B
    n_l = p.L if n_l is None else n_l
    n_r = p.L if n_r is None else n_r
    W = compute_mul_word_max(w, min(n_l, n_r))          # :832-838
    num_limbs = len(a)
    assert num_limbs == len(b) == n_l + n_r - 1          # :839
    carry_bits = bits_size(W * 2) - w                    # :841-842
    carry_sub_bits = sublimb_bit_len(carry_bits)
    acc_extra = 0
    carry = [0]
    eq_bit = 1
    for i in range(num_limbs):
        a_b = a[i] - b[i]                      # :859 (field sub; signed here)
        if p.field_modulus:
            st.put(a_b % p.field_modulus, 32)  # main_gate.sub on field elements: negative -> p - |a_b|
        else:
            st.put(a_b, p.WB, signed=True)
        s = a_b + carry[i] + W                  # :860-861
        assert s >= 0
        st.put(s, p.WB)
        new_carry, c = divmod(s, B)             # :864 -> div_mod_main_gate :1323-1349
        st.put(new_carry, p.CB)
        st.put(c, p.LB)
        nq = B * new_carry
        st.put(nq, p.WB)
        st.put(s - nq, p.LB)
        carry.append(new_carry)
        acc_extra = acc_extra + W               # :869-870
        st.put(acc_extra, p.WB)
        q_acc, mod_acc = divmod(acc_extra, B)   # :871
        st.put(q_acc, p.CB)
        st.put(mod_acc, p.LB)
        nq2 = B * q_acc
        st.put(nq2, p.WB)
        st.put(acc_extra - nq2, p.LB)
        cs_acc_eq = 1 if c == mod_acc else 0    # :873
        st.put(cs_acc_eq, 1)
        eq_bit &= cs_acc_eq                     # :874
        st.put(eq_bit, 1)
        acc_extra = q_acc                       # :875
        if i < num_limbs - 1:
            # :879-887 -- range-assign the carry, compare, AND
            emit_range_assign(st, new_carry, carry_sub_bits, carry_bits, p.CB)
            range_eq = 1
            st.put(range_eq, 1)
            eq_bit &= range_eq
            st.put(eq_bit, 1)
        else:
            final_carry_eq = 1 if new_carry == acc_extra else 0   # :890
            st.put(final_carry_eq, 1)
            eq_bit &= final_carry_eq
            st.put(eq_bit, 1)
    return eq_bit


def mul_mod(p: Params, a: Sequence[int], b: Sequence[int], n: Sequence[int], st: Stream) -> List[int]:
    """BigIntChip::mul_mod, big_integer/chip.rs:542-629.  Returns the r limbs."""
    w, L = p.w, p.L
    assert len(a) == len(n) == L and len(b) == L    # :555
    a_big, b_big, n_big = from_limbs(a, w), from_limbs(b, w), from_limbs(n, w)
    if n_big == 0:
        raise ZeroDivisionError("modulus is zero (chip.rs:566)")
    full = a_big * b_big                               # :562
    q_big, r_big = divmod(full, n_big)                 # :564-567
    if q_big >> (w * L):
        raise OverflowError("quotient does not fit num_limbs limbs (chip.rs:584)")
    q = to_limbs(q_big, L, w)                           # :570-584
    r = to_limbs(r_big, L, w)
    for v in q:                                         # :588-591
        emit_range_assign(st, v, p.limb_sub_bits, w, p.LB)
    for v in r:                                         # :596-599
        emit_range_assign(st, v, p.limb_sub_bits, w, p.LB)
    ab = mul_columns(a, b, st, p.WB)                    # :608
    qn = mul_columns(q, n, st, p.WB)                    # :609
    eq_b = []
    for i in range(2 * L - 1):                          # :614-623
        if i < L:
            v = qn[i] + r[i]
            st.put(v, p.WB)
            eq_b.append(v)
        else:
            eq_b.append(qn[i])
    ok = is_equal_muled(p, ab, eq_b, st)               # :626
    assert ok == 1                                      # assert_equal_muled :1062
    return r


def fixed_exp_bits(e: int) -> List[int]:
    """big_integer/chip.rs:717-728: LSB-first bits of e, exactly e.bits() of them."""
    return [(e >> i) & 1 for i in range(e.bit_length())]


def pow_mod_fixed_exp(p: Params, x: Sequence[int], e: int, n: Sequence[int], st: Stream) -> List[int]:
    """BigIntChip::pow_mod_fixed_exp, big_integer/chip.rs:710-742."""
    acc = to_limbs(1, p.L, p.w)          # :729 assign_constant(1, a.num_limbs())
    squared = list(x)
    for bit in fixed_exp_bits(e):
        cur_sq = squared
        squared = mul_mod(p, cur_sq, cur_sq, n, st)   # :734 square_mod -> :642-649
        if not bit:
            continue
        acc = mul_mod(p, acc, cur_sq, n, st)          # :739
    for v in acc:
        st.put(v, p.LB)
    return acc


def pow_mod_var(p: Params, x: Sequence[int], e_limbs: Sequence[int], n: Sequence[int],
                exp_limb_bits: int, st: Stream) -> List[int]:
    """BigIntChip::pow_mod, big_integer/chip.rs:664-696."""
    e_bits = []
    for limb in e_limbs:   # to_bits constrains limb == sum(bit_t 2^t, t < exp_limb_bits): a wider limb cannot be assigned
        if limb >> exp_limb_bits:
            raise ValueError("e limb wider than exp_limb_bits (main_gate.to_bits, big_integer/chip.rs:677)")
    for limb in e_limbs:                                # :674-681 main_gate.to_bits LSB first
        for t in range(exp_limb_bits):
            e_bits.append((limb >> t) & 1)
    for b in e_bits:
        st.put(b, 1)
    acc = to_limbs(1, p.L, p.w)                         # :682
    squared = list(x)
    for bit in e_bits:
        muled = mul_mod(p, acc, squared, n, st)         # :686
        acc = [muled[j] if bit else acc[j] for j in range(p.L)]   # :688-691 select
        for v in acc:
            st.put(v, p.LB)
        squared = mul_mod(p, squared, squared, n, st)   # :693
    for v in acc:
        st.put(v, p.LB)
    return acc


def big_pow_mod(a: int, b: int, n: int) -> int:
    """big_integer/utils.rs:2-17 (recursive square-and-multiply; b == 0 returns 1 un-reduced)."""
    if b == 0:
        return 1
    is_odd = b % 2 == 1
    bb = b - 1 if is_odd else b
    x = big_pow_mod(a, bb // 2, n)
    x2 = (x * x) % n
    return (a * x2) % n if is_odd else x2


# ----------------------------------------------------------------------------------------------
# "next" rows of SURVEY 8(f): assert_in_field (#1) and the pkcs1v15 encoded-message check (#2)
# ----------------------------------------------------------------------------------------------

def add_fresh(p: Params, a: Sequence[int], b: Sequence[int], st: Stream) -> List[int]:
    """BigIntChip::add, big_integer/chip.rs:245-297."""
    w, B = p.w, p.B
    max_n = max(len(a), len(b))
    a = list(a) + [0] * (max_n - len(a))
    b = list(b) + [0] * (max_n - len(b))
    c_vals, carrys = [], [0]
    SB = p.LB + 8  # a_b / sum / c+carry*B : up to w+2 bits
    for i in range(max_n):
        a_b = a[i] + b[i]                                # :272
        st.put(a_b, SB)
        s = a_b + carrys[i]                              # :273
        st.put(s, SB)
        c, carry = s % B, s >> w                         # :276-277
        emit_range_assign(st, c, p.limb_sub_bits, w, p.LB)       # :279-280
        emit_range_assign(st, carry, p.limb_sub_bits, w, p.LB)   # :281-282
        st.put(carry * B + c, SB)                        # :283 mul_add
        c_vals.append(c)
        carrys.append(carry)
    c_vals.append(carrys[max_n])                         # :290
    return c_vals


def is_equal_fresh(p: Params, a: Sequence[int], b: Sequence[int], st: Stream) -> int:
    """BigIntChip::is_equal_fresh, big_integer/chip.rs:780-805."""
    n1, n2 = len(a), len(b)
    is_a_larger = n1 > n2
    max_n = n1 if is_a_larger else n2
    eq_bit = 1
    for i in range(max_n):
        if is_a_larger and i >= n2:
            flag = 1 if a[i] == 0 else 0
        elif (not is_a_larger) and i >= n1:
            flag = 1 if b[i] == 0 else 0
        else:
            flag = 1 if a[i] == b[i] else 0
        st.put(flag, 1)
        eq_bit &= flag
        st.put(eq_bit, 1)
    return eq_bit


def sub_unchecked(p: Params, a: Sequence[int], b: Sequence[int], st: Stream) -> List[int]:
    """BigIntChip::sub_unchecked, big_integer/chip.rs:1286-1318."""
    w = p.w
    assert len(a) >= len(b)                              # :1294
    max_n = len(a)
    c_big = from_limbs(a, w) - from_limbs(b, w)          # :1300 (panics on underflow)
    if c_big < 0:
        raise OverflowError("a < b in sub_unchecked (chip.rs:1300)")
    c = []
    for _ in range(max_n):                               # :1304-1311
        v = c_big % p.B
        emit_range_assign(st, v, p.limb_sub_bits, w, p.LB)
        c.append(v)
        c_big >>= w
    added = add_fresh(p, b, c, st)                       # :1315
    ok = is_equal_fresh(p, a, added, st)                 # :1316
    assert ok == 1
    return c


def sub_fresh(p: Params, a: Sequence[int], b: Sequence[int], st: Stream) -> Tuple[List[int], int]:
    """BigIntChip::sub, big_integer/chip.rs:310-373.  Returns (|a-b| limbs, is_overflowed)."""
    n2 = len(b)
    max_int = [p.B - 1] * n2                             # :319 max_value :138-154 (constants)
    inflated_a = add_fresh(p, a, max_int, st)            # :321
    inflated_subed = sub_unchecked(p, inflated_a, b, st)  # :323
    is_not_overflowed = 1 if inflated_subed[n2] == 1 else 0   # :330
    st.put(is_not_overflowed, 1)
    is_overflowed = 1 - is_not_overflowed                # :331
    st.put(is_overflowed, 1)
    num_limbs_l = len(inflated_subed)
    num_limbs_r = max(len(a), n2)
    sel_l, sel_r = [], []
    for i in range(num_limbs_l):                         # :345-357
        if i >= n2:
            v = inflated_subed[i] if is_not_overflowed else 0
        else:
            v = inflated_subed[i] if is_not_overflowed else b[i]
        st.put(v, p.LB)
        sel_l.append(v)
    for i in range(num_limbs_r):                         # :358-367
        if i >= len(a):
            v = max_int[i] if is_not_overflowed else 0
        elif i >= n2:
            v = 0 if is_not_overflowed else a[i]
        else:
            v = max_int[i] if is_not_overflowed else a[i]
        st.put(v, p.LB)
        sel_r.append(v)
    real_subed = sub_unchecked(p, sel_l, sel_r, st)      # :371
    return real_subed, is_overflowed


def is_less_than(p: Params, a: Sequence[int], b: Sequence[int], st: Stream) -> int:
    """BigIntChip::is_less_than, big_integer/chip.rs:908-919 (+ :932-941)."""
    _, is_overflowed = sub_fresh(p, a, b, st)            # :939
    is_eq = is_equal_fresh(p, a, b, st)                  # :916
    is_not_eq = 1 - is_eq                                # :917
    st.put(is_not_eq, 1)
    out = is_overflowed & is_not_eq                      # :918
    st.put(out, 1)
    return out


def assert_in_field(p: Params, a: Sequence[int], n: Sequence[int], st: Stream) -> int:
    """BigIntChip::assert_in_field, big_integer/chip.rs:1150-1158 -> is_in_field :998-1006."""
    return is_less_than(p, a, n, st)


def refresh(p: Params, a: Sequence[int], st: Stream, n_l: int | None = None, n_r: int | None = None) -> List[int]:
    """BigIntChip::refresh, big_integer/chip.rs:168-233, with aux = RefreshAux::new(w, n_l, n_r) (mod.rs:428-482; default
    n_l = n_r = L).  `a` holds the n_l + n_r - 1 Muled limbs; returns the Fresh limbs."""
    w, B, L = p.w, p.B, p.L
    n_l = L if n_l is None else n_l
    n_r = L if n_r is None else n_r
    inc = refresh_aux_increased_limbs(w, n_l, n_r)
    assert len(a) == n_l + n_r - 1                   # :181
    num_limbs_fresh = len(inc)
    r = list(a) + [0] * (num_limbs_fresh - len(a))   # :186-192
    for i in range(num_limbs_fresh):                 # :195
        limb = r[i]
        for j in range(inc[i] + 1):                  # :198
            q, n = divmod(limb, B)                   # :201 div_mod_main_gate :1323-1349
            st.put(q, p.CB)
            st.put(n, p.LB)
            st.put(B * q, p.WB)
            st.put(limb - B * q, p.LB)
            if j == 0:
                r[i] = n                             # :204
            else:
                r[i + j] = r[i + j] + n              # :207 main_gate.add
                st.put(r[i + j], p.WB)
            limb = q
        if limb != 0:                                # :213 assert_zero
            raise OverflowError("refresh: limb does not fit increased_limbs_vec (chip.rs:213)")
    for i in range(num_limbs_fresh):                 # :217-226
        emit_range_assign(st, r[i], p.limb_sub_bits, w, p.LB)
    return r


def is_zero(p: Params, a: Sequence[int], st: Stream) -> int:
    """BigIntChip::is_zero, big_integer/chip.rs:754-767."""
    bit = 1
    for v in a:
        z = 1 if v == 0 else 0
        st.put(z, 1)
        bit &= z
        st.put(bit, 1)
    return bit


def add_mod(p: Params, a: Sequence[int], b: Sequence[int], n: Sequence[int], st: Stream) -> List[int]:
    """BigIntChip::add_mod, big_integer/chip.rs:452-481.  NOTE: sub's overflow bit is 1 iff added <= n, so the
    result is `added` (un-reduced) when a + b == n -- restated as written."""
    added = add_fresh(p, a, b, st)                        # :462
    subed, is_overflowed = sub_fresh(p, added, n, st)    # :464
    num_limbs = len(subed)
    added = list(added) + [0] * (num_limbs - len(added))  # :467
    res = []
    for i in range(num_limbs):                            # :469-474 select(added, subed, is_overflowed)
        v = added[i] if is_overflowed else subed[i]
        st.put(v, p.LB)
        res.append(v)
    assert all(v == 0 for v in res[len(n):])              # :475-478 assert_zero
    return res[:len(n)]


def sub_mod(p: Params, a: Sequence[int], b: Sequence[int], n: Sequence[int], st: Stream) -> List[int]:
    """BigIntChip::sub_mod, big_integer/chip.rs:495-528."""
    subed1, is_overflowed1 = sub_fresh(p, a, b, st)       # :506
    subed2, is_overflowed2 = sub_fresh(p, n, subed1, st)  # :509
    if is_overflowed2 != 0:                                 # :510 assert_zero (fails e.g. when a == b: n - 0 <= ... see tests)
        raise ValueError("sub_mod: n <= |a - b| (chip.rs:510)")
    num_limbs = len(subed2)
    subed1 = list(subed1) + [0] * (num_limbs - len(subed1))
    res = []
    for i in range(num_limbs):                            # :516-521 select(subed2, subed1, is_overflowed1)
        v = subed2[i] if is_overflowed1 else subed1[i]
        st.put(v, p.LB)
        res.append(v)
    assert all(v == 0 for v in res[len(n):])
    return res[:len(n)]


def is_less_than_or_equal(p, a, b, st):
    """big_integer/chip.rs:932-941."""
    _, ov = sub_fresh(p, a, b, st)
    return ov


def is_greater_than(p, a, b, st):
    """big_integer/chip.rs:954-963."""
    le = is_less_than_or_equal(p, a, b, st)
    st.put(1 - le, 1)
    return 1 - le


def is_greater_than_or_equal(p, a, b, st):
    """big_integer/chip.rs:976-985."""
    lt = is_less_than(p, a, b, st)
    st.put(1 - lt, 1)
    return 1 - lt


FRESH_OPS = ("add", "sub", "add_mod", "sub_mod", "is_zero", "is_equal_fresh", "is_less_than", "is_less_than_or_equal",
             "is_greater_than", "is_greater_than_or_equal", "is_in_field")


def fresh_op(p: Params, op: str, a, b, n, st: Stream):
    """Dispatch used by the tests: returns (value limbs or None, flag or None)."""
    if op == "add":
        return add_fresh(p, a, b, st), None
    if op == "sub":
        v, ov = sub_fresh(p, a, b, st)
        return v, ov
    if op == "add_mod":
        return add_mod(p, a, b, n, st), None
    if op == "sub_mod":
        return sub_mod(p, a, b, n, st), None
    if op == "is_zero":
        return None, is_zero(p, a, st)
    if op == "is_equal_fresh":
        return None, is_equal_fresh(p, a, b, st)
    if op in ("is_less_than", "is_in_field"):
        return None, is_less_than(p, a, b, st)
    if op == "is_less_than_or_equal":
        return None, is_less_than_or_equal(p, a, b, st)
    if op == "is_greater_than":
        return None, is_greater_than(p, a, b, st)
    if op == "is_greater_than_or_equal":
        return None, is_greater_than_or_equal(p, a, b, st)
    raise ValueError(op)


PKCS1_PREFIX_64_1 = 217300885422736416   # src/chip.rs:150
PKCS1_PREFIX_64_2 = 938447882527703397   # src/chip.rs:152
PKCS1_PREFIX_32 = 3158320                # src/chip.rs:175
PKCS1_FF_32 = 4294967295                 # src/chip.rs:180
PKCS1_FF_64 = 18446744073709551615       # src/chip.rs:184
PKCS1_LAST_EM = 562949953421311          # src/chip.rs:191


def pkcs1v15_em_check(powed: Sequence[int], hashed: Sequence[int], bits_len: int, st: Stream) -> int:
    """RSAChip::verify_pkcs1v15_signature after the modpow, src/chip.rs:136-198 (LIMB_WIDTH = 64)."""
    is_eq = 1
    hash_len = 4
    for i in range(hash_len):                            # :141-144
        f = 1 if powed[i] == hashed[i] else 0
        st.put(f, 1)
        is_eq &= f
        st.put(is_eq, 1)
    f1 = 1 if powed[hash_len] == PKCS1_PREFIX_64_1 else 0          # :153
    f2 = 1 if powed[hash_len + 1] == PKCS1_PREFIX_64_2 else 0      # :154
    st.put(f1, 1)
    st.put(f2, 1)
    is_eq &= f1
    st.put(is_eq, 1)
    is_eq &= f2
    st.put(is_eq, 1)
    v = powed[hash_len + 2]
    low, high = v % (1 << 32), v >> 32                    # :159-168
    emit_range_assign(st, low, 4, 32, 4)                  # :170
    emit_range_assign(st, high, 4, 32, 4)                 # :171
    st.put(high * (1 << 32) + low, 8)                     # :173 mul_add
    f = 1 if low == PKCS1_PREFIX_32 else 0                # :176
    st.put(f, 1)
    is_eq &= f
    st.put(is_eq, 1)
    f = 1 if high == PKCS1_FF_32 else 0                   # :181
    st.put(f, 1)
    is_eq &= f
    st.put(is_eq, 1)
    nl = bits_len // 64
    for i in range(hash_len + 3, nl - 1):                 # :185-188
        f = 1 if powed[i] == PKCS1_FF_64 else 0
        st.put(f, 1)
        is_eq &= f
        st.put(is_eq, 1)
    f = 1 if powed[nl - 1] == PKCS1_LAST_EM else 0        # :192-196
    st.put(f, 1)
    is_eq &= f
    st.put(is_eq, 1)
    return is_eq


def sha256(msg: bytes) -> bytes:
    """FIPS 180-4 SHA-256, restated from the standard (the reference takes the values from the third-party sha2 0.10.6 crate
    and the halo2-dynamic-sha256 chip, src/lib.rs:205-209, :283-343); pinned by tests/golden/sha256_kat.json."""
    K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
         0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
         0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
         0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
         0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
         0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
         0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
    M = 0xffffffff
    rotr = lambda x, n: ((x >> n) | (x << (32 - n))) & M
    H = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    data = bytes(msg) + b"\x80"
    data += b"\x00" * ((56 - len(data)) % 64) + (8 * len(msg)).to_bytes(8, "big")
    for b in range(0, len(data), 64):
        W = [int.from_bytes(data[b + 4 * t:b + 4 * t + 4], "big") for t in range(16)]
        for t in range(16, 64):
            s0 = rotr(W[t - 15], 7) ^ rotr(W[t - 15], 18) ^ (W[t - 15] >> 3)
            s1 = rotr(W[t - 2], 17) ^ rotr(W[t - 2], 19) ^ (W[t - 2] >> 10)
            W.append((s1 + W[t - 7] + s0 + W[t - 16]) & M)
        a, bb, c, d, e, f, g, h = H
        for t in range(64):
            T1 = (h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & M & g)) + K[t] + W[t]) & M
            T2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c))) & M
            h, g, f, e, d, c, bb, a = g, f, e, (d + T1) & M, c, bb, a, (T1 + T2) & M
        H = [(x + y) & M for x, y in zip(H, [a, bb, c, d, e, f, g, h])]
    return b"".join(x.to_bytes(4, "big") for x in H)


def hashed_msg(digest: bytes, st: Stream | None = None) -> List[int]:
    """RSASignatureVerifier::verify_pkcs1v15_signature, src/lib.rs:210-239: the digest bytes reversed, composed eight at a time
    into 64-bit limbs by a mul_add chain.  Stream: the 32 reversed byte cells (1 byte each), then every limb_val the chain assigns."""
    hashed_bytes = list(bytes(digest))[::-1]                 # :213
    assert len(hashed_bytes) == 32
    if st is not None:
        for b in hashed_bytes:
            st.put(b, 1)
    limbs = []
    for i in range(len(hashed_bytes) // 8):                  # :225
        limb_val = 0                                         # :226
        for j in range(8):
            coeff = 1 << (8 * j)                             # :228-229
            limb_val = coeff * hashed_bytes[8 * i + j] + limb_val   # :230-235
            if st is not None:
                st.put(limb_val, 8)
        limbs.append(limb_val)                               # :237
    return limbs


def modpow_public_key_fixed(p: Params, x: Sequence[int], e: int, n: Sequence[int], st: Stream,
                            with_in_field: bool = True) -> List[int]:
    """RSAChip::modpow_public_key with RSAPubE::Fix, src/chip.rs:99-114."""
    if with_in_field:
        ok = assert_in_field(p, x, n, st)                 # :106
        if ok != 1:
            raise ValueError("x >= n (src/chip.rs:106 assert_in_field)")
    return pow_mod_fixed_exp(p, x, e, n, st)              # :111
