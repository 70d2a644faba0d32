"""TEST HELPER: the main-gate image of one mul_mod -- advice rows, fixed (selector) rows, lookup inputs, the lookup table
and halo2's permuted lookup columns -- built in plain Python from the ORACLE's flat stream and the operands.

Everything here restates THIRD-PARTY code that is not in the reference tree (maingate / halo2wrong rev 63bde545, halo2's
lookup prover): the op -> row shapes documented in DESIGN.md section 2b and the `permute_expression_pair` algorithm.  The
VALUES come from the oracle's stream (pinned); the PLACEMENT is this restatement (unpinned, self-consistent: every row
satisfies the main-gate equation with its fixed row, every lookup input is a table row -- checked by the tests).

Row kinds (include/h2r.h H2R_ROW_*):  cells a..e, gate
    sa*a + sb*b + sc*c + sd*d + se*e + s_mul_ab*a*b + s_mul_cd*c*d + se_next*e(next row) + s_const = 0."""
import numpy as np

ROW_NOP, ROW_CONST0, ROW_CONST1, ROW_CONST_B, ROW_BIT, ROW_VALUE, ROW_MUL_ADD, ROW_ADD, ROW_SUB, ROW_ADD_WM, ROW_ADDC_WM, ROW_MUL, \
    ROW_ASSERT_EQ, ROW_ISZERO_INV, ROW_ISZERO_RA = range(15)
ROW_RANGE_LIMB = 32      # + row of the assign (0 .. nr-1)
ROW_RANGE_CARRY = 40     # + row of the assign (0 .. nrc-1)

ARGS = ("composition_a", "composition_b", "composition_c", "composition_d", "overflow_a")


def word_max(w, L):
    B = 1 << w
    return L * (B - 1) * (B - 1) + (B - 1)


class LookupConfig:
    """RangeChip::configure's bit_len -> tag map and the (tag, value) table `load_table` writes: row 0 = (0, 0), then for
    every distinct nonzero bit length in ascending order (tag = 1, 2, ...) the rows (tag, 0 .. 2^bit_len - 1)."""

    def __init__(self, bit_lens, tags=None):
        self.bit_lens = sorted(set(b for b in bit_lens if b))
        self.tags = list(tags) if tags is not None else list(range(1, len(self.bit_lens) + 1))
        self.tag_of = dict(zip(self.bit_lens, self.tags))
        self.row_off = {}
        off = 1
        for b in self.bit_lens:
            self.row_off[b] = off
            off += 1 << b
        self.n_rows = off

    def table(self):
        rows = [(0, 0)]
        for b in self.bit_lens:
            rows += [(self.tag_of[b], v) for v in range(1 << b)]
        return rows


def range_lens(w, L, rsa=False):
    """BigIntChip::compute_range_lens (big_integer/chip.rs:1220-1249) [+ RSAChip's 32/8, src/chip.rs:252]: the bit lengths
    RangeChip::configure is given (composition + overflow)."""
    wm = word_max(w, L)
    carry_bits = (2 * wm).bit_length() - w
    comp = [w // 8, 1, max(1, carry_bits // 8)]
    over = [0, 0, carry_bits % max(1, carry_bits // 8)]
    if rsa:
        comp.append(32 // 8)
    return comp + over


class Image:
    """Rows of (cells[5] as canonical field elements, kind)."""

    def __init__(self, w, L, P):
        self.w, self.L, self.P = w, L, P
        self.rows, self.kinds = [], []

    def row(self, kind, *cells):
        cells = list(cells) + [0] * (5 - len(cells))
        self.rows.append([c % self.P for c in cells])
        self.kinds.append(kind)

    # ---- main-gate ops (maingate instructions, restated) -----------------------------------------------------
    def assign_constant(self, c):
        self.row({0: ROW_CONST0, 1: ROW_CONST1}.get(c, ROW_CONST_B), c)

    def assign_bit(self, v):
        self.row(ROW_BIT, v, v, v)

    def assign_value(self, v):
        self.row(ROW_VALUE, v)

    def is_equal(self, x, y, flag):
        """is_equal(x, y) = sub + is_zero: [x, y, d], assign_bit(r), [d, d', r], [r, d]; r must equal the stream's flag."""
        d = (x - y) % self.P
        r = 1 if d == 0 else 0
        assert r == flag
        inv = 1 if d == 0 else pow(d, self.P - 2, self.P)
        self.row(ROW_SUB, x, y, d)
        self.assign_bit(r)
        self.row(ROW_ISZERO_INV, d, inv, r)
        self.row(ROW_ISZERO_RA, r, d)

    def range_assign(self, value, subs, sub_bits, kind0):
        """RangeChip::assign -> main_gate.decompose: chunks of four terms (columns a..d), column e = what remains to be
        composed (row 0: the value itself); the LAST chunk is reversed so that the last (overflow) term sits in column a."""
        remaining = value
        nrows = (len(subs) + 3) // 4
        for rr in range(nrows):
            chunk = subs[4 * rr:4 * rr + 4]
            comp = sum(sv << ((4 * rr + k) * sub_bits) for k, sv in enumerate(chunk))
            cells = list(chunk)
            if rr == nrows - 1:
                cells = cells[::-1]
                assert comp == remaining
            self.row(kind0 + rr, *(cells + [0] * (4 - len(cells))), remaining)
            remaining -= comp
        assert remaining == 0


def mul_mod_image(p, a, b, n, stream, P):
    """p: oracle params (w, L, LB, WB, CB, carry_bits, carry_sub_bits, carry_nsub); a, b, n: limb lists; stream: the
    mul_mod flat stream.  Returns the Image of the whole BigIntChip::mul_mod (chip.rs:542-629), every cell included."""
    w, L = p.w, p.L
    LB, WB, CB = p.LB, p.WB, p.CB
    C = 2 * L - 1
    st = bytes(stream)
    pos = 0

    def take(nb, signed=False):
        nonlocal pos
        v = int.from_bytes(st[pos:pos + nb], "little", signed=signed)
        pos += nb
        return v

    im = Image(w, L, P)
    W = word_max(w, L)
    q, r = [], []
    for which in (q, r):                                    # T1 / T2   chip.rs:588-599
        for _ in range(L):
            v = take(LB)
            subs = [take(1) for _ in range(8)]
            which.append(v)
            im.range_assign(v, subs, w // 8, ROW_RANGE_LIMB)
    ab, qn = [], []
    for (x, y, cols) in ((a, b, ab), (q, n, qn)):           # T3 / T4   chip.rs:400-412
        for i in range(C):
            im.assign_constant(0)                            # :402
            prev = 0
            for j in range(max(0, i - L + 1), min(i, L - 1) + 1):
                acc = take(WB)
                im.row(ROW_MUL_ADD, x[j], y[i - j], prev, acc)
                prev = acc
            cols.append(prev)
    eqb = []
    for i in range(L):                                       # T5        chip.rs:617
        v = take(WB)
        im.row(ROW_ADD, qn[i], r[i], v)
        eqb.append(v)
    eqb += qn[L:]
    B = 1 << w
    im.assign_constant(B)                                    # limb_max          chip.rs:851
    im.assign_constant(0)                                    # accumulated_extra :852
    im.assign_constant(0)                                    # carry[0]          :855
    im.assign_bit(1)                                         # eq_bit            :856
    carry_prev, x_prev, eq_prev = 0, 0, 1
    for i in range(C):                                       # T6        chip.rs:857-893
        a_b = take(WB, signed=True)
        s = take(WB)
        cy = take(CB)
        c = take(LB)
        nq = take(WB)
        amnq = take(LB)
        accx = take(WB)
        qacc = take(CB)
        modacc = take(LB)
        nq2 = take(WB)
        amnq2 = take(LB)
        f1, e1 = take(1), take(1)
        im.row(ROW_SUB, ab[i], eqb[i], a_b)                                # :859
        im.row(ROW_ADD_WM, a_b, carry_prev, s)                             # :860-861
        im.assign_value(cy); im.assign_value(c)                            # div_mod_main_gate :1341-1344
        im.row(ROW_MUL, B, cy, nq); im.row(ROW_SUB, s, nq, amnq); im.row(ROW_ASSERT_EQ, c, amnq)   # :1345-1347
        im.row(ROW_ADDC_WM, x_prev, accx)                                  # :869-870
        im.assign_value(qacc); im.assign_value(modacc)
        im.row(ROW_MUL, B, qacc, nq2); im.row(ROW_SUB, accx, nq2, amnq2); im.row(ROW_ASSERT_EQ, modacc, amnq2)
        im.is_equal(c, modacc, f1)                                         # :873
        im.row(ROW_MUL, eq_prev, f1, e1)                                   # and, :874
        if i < C - 1:
            dup = take(CB)
            subs = [take(1) for _ in range(p.carry_nsub)]
            im.range_assign(dup, subs, p.carry_sub_bits, ROW_RANGE_CARRY)  # :879-885
            f2, e2 = take(1), take(1)
            im.is_equal(cy, dup, f2)                                       # :886
        else:
            f2, e2 = take(1), take(1)
            im.is_equal(cy, qacc, f2)                                      # :890
        im.row(ROW_MUL, e1, f2, e2)                                        # and, :887 / :891
        carry_prev, x_prev, eq_prev = cy, qacc, e2
    im.row(17, eq_prev)                                      # T7: assert_equal_muled's main_gate.assert_one(eq_bit) :1062 (ROW_ASSERT_ONE)
    assert pos == len(st)
    return im


def image_bytes(im):
    out = np.zeros((len(im.rows), 160), dtype=np.uint8)
    for ri, cells in enumerate(im.rows):
        out[ri] = np.frombuffer(b"".join(int(c).to_bytes(32, "little") for c in cells), dtype=np.uint8)
    return out


def advice_image_from_stream(p, a, b, n, stream, field_modulus):
    """uint8 [rows, 160]: the advice image of one mul_mod (five 32-byte little-endian cells per row)."""
    return image_bytes(mul_mod_image(p, a, b, n, stream, field_modulus))


# ---- fixed columns -------------------------------------------------------------------------------------------------
FIXED_NAMES = ("sa", "sb", "sc", "sd", "se", "s_mul_ab", "s_mul_cd", "se_next", "s_const")


def fixed_row(kind, w, L, carry_bits, carry_sub_bits, carry_nsub, cfg):
    """Selectors of one row kind as a dict (integers, reduce mod p), plus tag_composition / tag_overflow (0 = lookup off)."""
    f = dict.fromkeys(FIXED_NAMES, 0)
    f["tag_composition"] = f["tag_overflow"] = 0
    B, W = 1 << w, word_max(w, L)
    if kind in (ROW_CONST0, ROW_CONST1, ROW_CONST_B):
        f["sa"] = 1
        f["s_const"] = -{ROW_CONST0: 0, ROW_CONST1: 1, ROW_CONST_B: B}[kind]
    elif kind == ROW_BIT:
        f["s_mul_ab"], f["sc"] = 1, -1
    elif kind == ROW_MUL_ADD:
        f["s_mul_ab"], f["sc"], f["sd"] = 1, 1, -1
    elif kind == ROW_ADD:
        f["sa"], f["sb"], f["sc"] = 1, 1, -1
    elif kind == ROW_SUB:
        f["sa"], f["sb"], f["sc"] = 1, -1, -1
    elif kind == ROW_ADD_WM:
        f["sa"], f["sb"], f["sc"], f["s_const"] = 1, 1, -1, W
    elif kind == ROW_ADDC_WM:
        f["sa"], f["sb"], f["s_const"] = 1, -1, W
    elif kind == ROW_MUL:
        f["s_mul_ab"], f["sc"] = 1, -1
    elif kind == ROW_ASSERT_EQ:
        f["sa"], f["sb"] = 1, -1
    elif kind == ROW_ISZERO_INV:
        f["s_mul_ab"], f["sc"], f["s_const"] = 1, 1, -1
    elif kind == ROW_ISZERO_RA:
        f["s_mul_ab"] = 1
    elif kind == 15:      # ROW_SELECT: cond*a - cond*b + b - res = 0 on [cond, a, cond, b, res]
        f["s_mul_ab"], f["s_mul_cd"], f["sd"], f["se"] = 1, -1, 1, -1
    elif kind == 16:      # ROW_NOT: c + not_c - 1 = 0
        f["sa"], f["sb"], f["s_const"] = 1, 1, -1
    elif kind == 17:      # ROW_ASSERT_ONE
        f["sa"], f["s_const"] = 1, -1
    elif kind == 18:      # ROW_CONST_BM1: assign_constant(2^w - 1)
        f["sa"], f["s_const"] = 1, -(B - 1)
    elif kind == 19:      # ROW_ASSERT_ZERO
        f["sa"] = 1
    elif 20 <= kind < 26:  # ROW_CONST_EM + j
        f["sa"], f["s_const"] = 1, -EM_CONSTS[kind - 20]
    elif 56 <= kind < 64:  # ROW_CONST_COEFF8 + j: assign_constant(2^(8j)) (src/lib.rs:228-229)
        f["sa"], f["s_const"] = 1, -(1 << (8 * (kind - 56)))
    elif kind in (48, 49):  # ROW_RANGE_U32 + row: RangeChip::assign(value, 4, 32), eight 4-bit sub-limbs
        last = kind == 49
        for q, nm in enumerate(("sa", "sb", "sc", "sd")):
            f[nm] = 1 << (4 * ((7 - q) if last else q))
        f["se"] = -1
        if not last:
            f["se_next"] = 1
        f["tag_composition"] = cfg.tag_of.get(4, 0) if cfg is not None else 0
    elif kind >= ROW_RANGE_LIMB:
        carry = kind >= ROW_RANGE_CARRY
        rr = kind - (ROW_RANGE_CARRY if carry else ROW_RANGE_LIMB)
        s = carry_sub_bits if carry else w // 8
        nsub = carry_nsub if carry else 8
        nrows = (nsub + 3) // 4
        idx = list(range(4 * rr, min(4 * rr + 4, nsub)))
        last = rr == nrows - 1
        if last:
            idx = idx[::-1]
        for name, k in zip(("sa", "sb", "sc", "sd"), idx):
            f[name] = 1 << (k * s)
        f["se"] = -1
        f["se_next"] = 0 if last else 1
        f["tag_composition"] = cfg.tag_of[s]
        ov = (carry_bits % s) if carry else 0
        if last and ov:
            f["tag_overflow"] = cfg.tag_of[ov]
    return f


def gate_residual(cells, e_next, f, P):
    a, b, c, d, e = cells
    return (f["sa"] * a + f["sb"] * b + f["sc"] * c + f["sd"] * d + f["se"] * e + f["s_mul_ab"] * a * b + f["s_mul_cd"] * c * d +
            f["se_next"] * e_next + f["s_const"]) % P


# ---- halo2's lookup argument (prover side), restated ----------------------------------------------------------------
def lookup_inputs(rows, fixed_rows, usable_rows):
    """Per lookup argument the (tag, value) input of every usable row: composition_a..d read columns a..d under
    tag_composition, overflow_a reads column a under tag_overflow; value = selector * advice, so a row with the lookup off
    contributes (0, 0).  Rows past the image (the rest of the circuit) are (0, 0)."""
    out = {name: [(0, 0)] * usable_rows for name in ARGS}
    for ri, (cells, f) in enumerate(zip(rows, fixed_rows)):
        if f["tag_composition"]:
            for k in range(4):
                out[ARGS[k]][ri] = (f["tag_composition"], cells[k])
        if f["tag_overflow"]:
            out[ARGS[4]][ri] = (f["tag_overflow"], cells[0])
    return out


def compress(pairs, theta, P):
    """halo2 `compress_expressions`: fold(0, |acc, e| acc * theta + e) over (tag, value) = tag * theta + value."""
    return [(t * theta + v) % P for (t, v) in pairs]


def permute_expression_pair(inp, table):
    """halo2 (v0.2 / PSE fork) plonk::lookup::prover::permute_expression_pair on the usable rows, without the blinding tail:
    sorted input A'; S' = the input value on the first row of every run, the leftover table values (ascending, popped from
    the END of the list of repeated rows) elsewhere.  Field `Ord` = order of the canonical integers."""
    usable = len(inp)
    assert len(table) == usable
    a_perm = sorted(inp)
    leftover = {}
    for c in table:
        leftover[c] = leftover.get(c, 0) + 1
    s_perm = [0] * usable
    repeated = []
    for row, v in enumerate(a_perm):
        if row == 0 or v != a_perm[row - 1]:
            s_perm[row] = v
            assert leftover.get(v, 0) > 0, "input value not in the table"
            leftover[v] -= 1
        else:
            repeated.append(row)
    for coeff in sorted(leftover):
        for _ in range(leftover[coeff]):
            s_perm[repeated.pop()] = coeff
    assert not repeated
    return a_perm, s_perm


def table_column(cfg, theta, usable_rows, P):
    """The compressed table expression over the usable rows: the table's rows, then the default (first) row (0, 0)."""
    t = compress(cfg.table(), theta, P)
    assert len(t) <= usable_rows
    return t + [0] * (usable_rows - len(t))


# ---- the Fresh-integer family as advice rows: add, sub, add_mod, sub_mod, is_zero, comparisons, assert_in_field --
ROW_SELECT, ROW_NOT, ROW_ASSERT_ONE, ROW_CONST_BM1, ROW_ASSERT_ZERO = 15, 16, 17, 18, 19


def fresh_image(p, name, a, b, n, stream, P, assert_one=False):
    """Rows of one Fresh-integer op of BigIntChip (big_integer/chip.rs: add :245-297, sub :310-373, add_mod :452-481, sub_mod
    :495-528, is_zero :754-767, is_equal_fresh :780-805, is_less_than :908-919, is_less_than_or_equal :932-941, is_greater_than
    :954-963, is_greater_than_or_equal :976-985, is_in_field :998-1006; helpers max_value :138-154, sub_unchecked :1286-1318),
    every cell, built from the ORACLE's stream of the op (values in its order) and the operands' limbs.  assert_one: the op's bit
    is then given to main_gate.assert_one (assert_in_field :1150-1158 and the other assert_* of instructions.rs).
    maingate rows as in mul_mod_image, plus select(a, b, cond) = [cond, a, cond, b, res] (cond*a - cond*b + b - res = 0),
    not(c) = [c, 1 - c], assert_one(a) = [a], assert_zero(a) = [a]."""
    w, L = p.w, p.L
    LB = p.LB
    SB = LB + 8
    B = 1 << w
    st = bytes(stream)
    pos = 0

    def take(nb):
        nonlocal pos
        v = int.from_bytes(st[pos:pos + nb], "little")
        pos += nb
        return v

    im = Image(w, L, P)

    def const(c):
        im.row({0: ROW_CONST0, 1: ROW_CONST1, B: ROW_CONST_B, B - 1: ROW_CONST_BM1}[c], c)

    def range_limb():
        v = take(LB)
        subs = [take(1) for _ in range(8)]
        im.range_assign(v, subs, w // 8, ROW_RANGE_LIMB)
        return v

    def add(a, b):                                          # chip.rs:245-297
        max_n = max(len(a), len(b))
        const(0)                                            # zero_value :254
        a = list(a) + [0] * (max_n - len(a))
        b = list(b) + [0] * (max_n - len(b))
        const(B)                                            # limb_max_val :267
        carry, out = 0, []
        for i in range(max_n):
            a_b = take(SB); im.row(ROW_ADD, a[i], b[i], a_b)             # :272
            assert a_b == a[i] + b[i]
            s = take(SB); im.row(ROW_ADD, a_b, carry, s)                 # :273
            c = range_limb()                                             # :279-280
            new_carry = range_limb()                                     # :281-282
            cac = take(SB); im.row(ROW_MUL_ADD, new_carry, B, c, cac)    # :283
            im.row(ROW_ASSERT_EQ, s, cac)                                # :285
            out.append(c)
            carry = new_carry
        return out + [carry]                                # :290

    def is_zero_rows(a_val, flag):                          # main_gate.is_zero: assign_bit(r), [a, a', r], [r, a]
        r = 1 if a_val % P == 0 else 0
        assert r == flag
        inv = 1 if r else pow(a_val % P, P - 2, P)
        im.assign_bit(r)
        im.row(ROW_ISZERO_INV, a_val, inv, r)
        im.row(ROW_ISZERO_RA, r, a_val)

    def is_equal_fresh(a, b):                               # chip.rs:780-805
        n1, n2 = len(a), len(b)
        larger = n1 > n2
        max_n = n1 if larger else n2
        im.assign_bit(1)
        eq = 1
        for i in range(max_n):
            flag, run = take(1), take(1)
            if larger and i >= n2:
                is_zero_rows(a[i], flag)
            elif (not larger) and i >= n1:
                is_zero_rows(b[i], flag)
            else:
                im.is_equal(a[i], b[i], flag)
            im.row(ROW_MUL, eq, flag, run)                  # and
            assert run == (eq & flag)
            eq = run
        return eq

    def is_zero(a):                                         # chip.rs:754-767
        im.assign_bit(1)
        bit = 1
        for v in a:
            flag, run = take(1), take(1)
            is_zero_rows(v, flag)
            im.row(ROW_MUL, bit, flag, run)
            bit = run
        return bit

    def sub_unchecked(a, b):                                # chip.rs:1286-1318
        c = [range_limb() for _ in range(len(a))]
        added = add(b, c)
        ok = is_equal_fresh(a, added)
        im.row(ROW_ASSERT_ONE, ok)                          # assert_equal_fresh -> assert_one
        return c

    def sub(a, b):                                          # chip.rs:310-373
        n2 = len(b)
        max_int = [B - 1] * n2
        for _ in range(n2):
            const(B - 1)                                    # max_value :138-154
        inflated_a = add(a, max_int)
        inflated_subed = sub_unchecked(inflated_a, b)
        im.assign_bit(1)                                    # one :326
        not_ov, ov = take(1), take(1)
        im.is_equal(inflated_subed[n2], 1, not_ov)          # :330
        im.row(ROW_NOT, not_ov, ov)                         # :331
        const(0)                                            # zero_value :343
        sel_l, sel_r = [], []
        for i in range(len(inflated_subed)):                # :345-357
            v = take(LB)
            bsrc = 0 if i >= n2 else b[i]
            im.row(ROW_SELECT, not_ov, inflated_subed[i], not_ov, bsrc, v)
            sel_l.append(v)
        for i in range(max(len(a), n2)):                    # :358-367
            v = take(LB)
            if i >= len(a):
                aa, bb = max_int[i], 0
            elif i >= n2:
                aa, bb = 0, a[i]
            else:
                aa, bb = max_int[i], a[i]
            im.row(ROW_SELECT, not_ov, aa, not_ov, bb, v)
            sel_r.append(v)
        real = sub_unchecked(sel_l, sel_r)
        return real, ov

    def is_less_than(a, b):                                 # chip.rs:908-919
        _, is_ov = sub(a, b)                                # is_less_than_or_equal :932-941
        is_eq = is_equal_fresh(a, b)                        # :916
        is_not_eq, lt = take(1), take(1)
        im.row(ROW_NOT, is_eq, is_not_eq)                   # :917
        im.row(ROW_MUL, is_ov, is_not_eq, lt)               # :918
        return lt

    def select_mod(x, y, cond, n_limbs):                    # the tail of add_mod :466-478 / sub_mod :512-525
        const(0)                                            # zero_value
        num = max(len(x), len(y))
        x = list(x) + [0] * (num - len(x))
        y = list(y) + [0] * (num - len(y))
        res = []
        for i in range(num):
            v = take(LB)
            im.row(ROW_SELECT, cond, x[i], cond, y[i], v)
            res.append(v)
        for i in range(n_limbs, num):
            im.row(ROW_ASSERT_ZERO, res[i])
        return res[:n_limbs]

    a = [int(v) for v in a]
    b = [int(v) for v in b] if b is not None else None
    n = [int(v) for v in n] if n is not None else None
    bit = None
    if name == "add":
        add(a, b)
    elif name == "sub":
        sub(a, b)
    elif name == "add_mod":                                 # chip.rs:452-481
        added = add(a, b)
        subed, ov = sub(added, n)
        select_mod(added, subed, ov, len(n))
    elif name == "sub_mod":                                 # chip.rs:495-528
        subed1, ov1 = sub(a, b)
        subed2, ov2 = sub(n, subed1)
        im.row(ROW_ASSERT_ZERO, ov2)                        # :510
        select_mod(subed2, subed1, ov1, len(n))
    elif name == "is_zero":
        bit = is_zero(a)
    elif name == "is_equal_fresh":
        bit = is_equal_fresh(a, b)
    elif name in ("is_less_than", "is_in_field"):
        bit = is_less_than(a, b)
    elif name == "is_less_than_or_equal":
        bit = sub(a, b)[1]
    elif name == "is_greater_than":                         # chip.rs:954-963
        le = sub(a, b)[1]
        bit = take(1)
        im.row(ROW_NOT, le, bit)
    elif name == "is_greater_than_or_equal":                # chip.rs:976-985
        lt = is_less_than(a, b)
        bit = take(1)
        im.row(ROW_NOT, lt, bit)
    else:
        raise ValueError(name)
    if assert_one:
        im.row(ROW_ASSERT_ONE, bit)                         # e.g. assert_in_field :1157
    assert pos == len(st), (pos, len(st))
    return im


def in_field_image(p, x, n, stream, P):
    """assert_in_field(x, n) = is_less_than(x, n) + assert_one (big_integer/chip.rs:1150-1158)."""
    return fresh_image(p, "is_in_field", x, n, None, stream, P, assert_one=True)


# ---- RSAChip::verify_pkcs1v15_signature after the modpow (src/chip.rs:138-198) as advice rows ----------------------------------------
ROW_CONST_EM, ROW_RANGE_U32 = 20, 48
EM_CONSTS = [217300885422736416, 938447882527703397, 1 << 32, 3158320, 4294967295, 562949953421311]


def em_image(p, powed, hashed, stream, P):
    """The encoded-message check of RSAChip::verify_pkcs1v15_signature (src/chip.rs:138-198, 64-bit limbs) as rows, built from the
    ORACLE's EM stream: is_equal + and per compared limb, the assigned constants, the two RangeChip::assign(half, 4, 32) of limb 6
    with their mul_add recomposition and assert_equal.  (is_eq's seed, assign_constant(1) :137, sits BEFORE the modpow.)"""
    L = p.L
    st = bytes(stream)
    pos = 0

    def take(nb):
        nonlocal pos
        v = int.from_bytes(st[pos:pos + nb], "little")
        pos += nb
        return v

    im = Image(64, L, P)
    powed = [int(v) for v in powed]
    hashed = [int(v) for v in hashed]
    is_eq = 1

    def and_(flag, run):
        nonlocal is_eq
        im.row(ROW_MUL, is_eq, flag, run)
        assert run == (is_eq & flag)
        is_eq = run

    for i in range(4):                                       # :141-144
        f, r = take(1), take(1)
        im.is_equal(powed[i], hashed[i], f)
        and_(f, r)
    im.row(ROW_CONST_EM + 0, EM_CONSTS[0]); im.row(ROW_CONST_EM + 1, EM_CONSTS[1])     # :149-152
    f1, f2, r1, r2 = take(1), take(1), take(1), take(1)
    im.is_equal(powed[4], EM_CONSTS[0], f1); im.is_equal(powed[5], EM_CONSTS[1], f2)   # :153-154
    and_(f1, r1); and_(f2, r2)                               # :155-156
    halves = []
    for _ in range(2):                                       # :170-171
        v = take(4)
        subs = [take(1) for _ in range(8)]
        im.range_assign(v, subs, 4, ROW_RANGE_U32)
        halves.append(v)
    low, high = halves
    concat = take(8)
    im.row(ROW_CONST_EM + 2, EM_CONSTS[2])                   # :172
    im.row(ROW_MUL_ADD, high, EM_CONSTS[2], low, concat)     # :173
    im.row(ROW_ASSERT_EQ, powed[6], concat)                  # :174
    f, r = take(1), take(1)
    im.row(ROW_CONST_EM + 3, EM_CONSTS[3]); im.is_equal(low, EM_CONSTS[3], f); and_(f, r)     # :175-177
    f, r = take(1), take(1)
    im.row(ROW_CONST_EM + 4, EM_CONSTS[4]); im.is_equal(high, EM_CONSTS[4], f); and_(f, r)    # :180-182
    im.row(ROW_CONST_BM1, (1 << 64) - 1)                     # ff_64 :183-184
    for i in range(7, L - 1):                                # :185-188
        f, r = take(1), take(1)
        im.is_equal(powed[i], (1 << 64) - 1, f); and_(f, r)
    f, r = take(1), take(1)
    im.row(ROW_CONST_EM + 5, EM_CONSTS[5]); im.is_equal(powed[L - 1], EM_CONSTS[5], f); and_(f, r)   # :190-197
    assert pos == len(st), (pos, len(st))
    return im, is_eq


# ---- RSASignatureVerifier::verify_pkcs1v15_signature, the hashed-message limbs (src/lib.rs:225-239) as advice rows -------------------
ROW_CONST_COEFF8 = 56


def hashed_msg_image(stream, P):
    """limb_val = assign_constant(0); for j in 0..8: coeff = assign_constant(2^(8j)); limb_val = mul_add(coeff, hashed_bytes[8i+j],
    limb_val) -- built from the ORACLE's 288-byte stream (32 reversed byte cells, then the 32 limb_val cells)."""
    st = bytes(stream)
    assert len(st) == 288
    hb = list(st[:32])
    im = Image(64, 4, P)
    limbs = []
    for i in range(4):
        im.row(ROW_CONST0, 0)                                 # :226
        limb_val = 0
        for j in range(8):
            coeff = 1 << (8 * j)
            im.row(ROW_CONST_COEFF8 + j, coeff)               # :228-229
            nv = int.from_bytes(st[32 + 8 * (8 * i + j):32 + 8 * (8 * i + j) + 8], "little")
            assert nv == coeff * hb[8 * i + j] + limb_val
            im.row(ROW_MUL_ADD, coeff, hb[8 * i + j], limb_val, nv)   # :230-235
            limb_val = nv
        limbs.append(limb_val)
    return im, limbs


# ---- BigIntChip::pow_mod (variable exponent, big_integer/chip.rs:664-696) as advice rows ------------------------------------------
ROW_BITS_COMPOSE, ROW_BITS_COMPOSE_LAST = 64, 80


def to_bits_rows(im, limb, nb):
    """main_gate.to_bits(limb, nb) (maingate, restated): assign_bit per bit, `compose` = decompose's rows (four terms per row in
    columns a..d, the last row reversed and zero-padded, column e = what remains to be composed), assert_equal(result, limb)."""
    bits = [(limb >> t) & 1 for t in range(nb)]
    for b in bits:
        im.assign_bit(b)
    nc = (nb + 3) // 4
    remaining = sum(b << t for t, b in enumerate(bits))
    result = remaining
    for rr in range(nc):
        chunk = bits[4 * rr:4 * rr + 4]
        comp = sum(b << (4 * rr + k) for k, b in enumerate(chunk))
        last = rr == nc - 1
        cells = chunk[::-1] if last else chunk
        im.row((ROW_BITS_COMPOSE_LAST + 4 * rr + len(chunk) - 1) if last else (ROW_BITS_COMPOSE + rr), *(cells + [0] * (4 - len(cells))), remaining)
        remaining -= comp
    assert remaining == 0
    im.row(ROW_ASSERT_EQ, result, limb)


def pow_var_image(p, x, e_limbs, nb, n, stream, P, mul_mod_stream_bytes):
    """The whole pow_mod element from the ORACLE's Var stream (e bits, then per bit: mul_mod stream, selected limbs, square_mod
    stream; then the result limbs): [to_bits rows of every limb] [acc = assign_constant_fresh(1): CONST1, CONST0] per bit
    [mul_mod(acc, squared) rows] [select rows] [square_mod rows]."""
    w, L, LB = p.w, p.L, p.LB
    st = bytes(stream)
    nbits = len(e_limbs) * nb
    im = Image(w, L, P)
    for limb in e_limbs:
        to_bits_rows(im, limb, nb)
    bits = list(st[:nbits])
    assert bits == [(limb >> t) & 1 for limb in e_limbs for t in range(nb)]
    im.assign_constant(1)
    im.assign_constant(0)
    pos = nbits
    msb = mul_mod_stream_bytes
    B = 1 << w
    acc = [1] + [0] * (L - 1)
    squared = list(x)
    N = sum(int(v) << (w * i) for i, v in enumerate(n))

    def val(limbs):
        return sum(int(v) << (w * i) for i, v in enumerate(limbs))

    def to_l(v):
        return [(v >> (w * i)) & (B - 1) for i in range(L)]

    for bit in bits:
        sub = mul_mod_image(p, acc, squared, n, st[pos:pos + msb], P)
        im.rows += sub.rows
        im.kinds += sub.kinds
        pos += msb
        muled = to_l(val(acc) * val(squared) % N)
        sel = []
        for j in range(L):
            v = int.from_bytes(st[pos:pos + LB], "little")
            pos += LB
            assert v == (muled[j] if bit else acc[j])
            im.row(ROW_SELECT, bit, muled[j], bit, acc[j], v)
            sel.append(v)
        acc = sel
        sub = mul_mod_image(p, squared, squared, n, st[pos:pos + msb], P)
        im.rows += sub.rows
        im.kinds += sub.kinds
        pos += msb
        squared = to_l(val(squared) * val(squared) % N)
    assert pos + L * LB == len(st)
    return im


def fixed_row_bits_compose(kind):
    """Selectors of to_bits' compose rows (h2r_advice_fixed_row of H2R_ROW_BITS_COMPOSE / _LAST kinds)."""
    f = dict.fromkeys(FIXED_NAMES, 0)
    f["tag_composition"] = f["tag_overflow"] = 0
    last = kind >= ROW_BITS_COMPOSE_LAST
    rr = (kind - ROW_BITS_COMPOSE_LAST) // 4 if last else kind - ROW_BITS_COMPOSE
    terms = (kind - ROW_BITS_COMPOSE_LAST) % 4 + 1 if last else 4
    for q, nm in enumerate(("sa", "sb", "sc", "sd")[:terms]):
        f[nm] = 1 << ((4 * rr + terms - 1 - q) if last else (4 * rr + q))
    f["se"] = -1
    f["se_next"] = 0 if last else 1
    return f


# ---- the consumer's representation (include/h2r.h h2r_advice_repr) --------------------------------------------------------------
R256 = 1 << 256


def image_to_repr(rowmajor, rows, P, columns=False, montgomery=False, col_stride=0, fill=0):
    """The default image of ONE element -- uint8 array of rows * 160 bytes, row-major, canonical little-endian cells -- as the
    planar (one contiguous vector per column, `col_stride` bytes apart; 0 = packed) and / or Montgomery-form (x * R mod p,
    R = 2^256: the in-memory form of halo2curves / pasta field elements [3P]) image the library writes for a ctx created with
    H2R_ADVICE_COLUMNS / H2R_ADVICE_MONTGOMERY.  Bytes the image does not cover are `fill`."""
    rm = np.ascontiguousarray(rowmajor, dtype=np.uint8).reshape(rows, 5, 32)
    if montgomery:
        words = rm.view("<u8").reshape(rows * 5, 4)
        out = np.empty((rows * 5, 4), dtype=np.uint64)
        for k in range(rows * 5):
            v = int(words[k, 0]) | int(words[k, 1]) << 64 | int(words[k, 2]) << 128 | int(words[k, 3]) << 192
            if v:
                v = v * R256 % P
            out[k] = (v & 0xFFFFFFFFFFFFFFFF, (v >> 64) & 0xFFFFFFFFFFFFFFFF, (v >> 128) & 0xFFFFFFFFFFFFFFFF, v >> 192)
        rm = out.view(np.uint8).reshape(rows, 5, 32)
    if not columns:
        return rm.reshape(-1).copy()
    cs = col_stride or rows * 32
    buf = np.full(4 * cs + rows * 32, fill, dtype=np.uint8)
    for c in range(5):
        buf[c * cs:c * cs + rows * 32] = rm[:, c, :].reshape(-1)
    return buf
