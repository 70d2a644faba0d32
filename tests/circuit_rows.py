"""TEST HELPER: the number of main-gate rows of the reference's circuits, counted by walking the reference's own control flow.

Every method below follows the reference method of the same name line by line (file:line cited) and COUNTS the main-gate / range-chip
calls it issues; a call costs the rows the restated [3P] maingate layout gives it (tests/advice_ref.py,
halo2_rsa_amd/csrc/h2r_rowprog.hpp -- the layout the kernels emit):

    assign_constant / assign_bit / assign_value / add / sub / mul / mul_add / add_constant / add_with_constant /
    and / not / select / assert_equal / assert_zero / assert_one          1 row
    is_zero    = assign_bit + 2 rows                                      3 rows
    is_equal   = sub + is_zero                                            4 rows
    RangeChip::assign(v, s, bit_len): ceil(nsub / 4) rows, nsub = bit_len / s + (1 if bit_len % s else 0)
    to_bits(v, nb): nb assign_bit + ceil(nb / 4) composition rows + assert_equal

tests/test_reference_circuit_sizes.py first pins this walk to the LIBRARY's exported row counts on every shape the library has an
export for (mul_mod, the Fresh-op family, pow, modpow_public_key, verify), then sizes the reference's own test circuits with it."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref  # noqa: E402  (the checker's parameter functions: compute_range_lens, RefreshAux)


def bits_size(v):
    return max(v.bit_length(), 0)


class Rows:
    """BigIntChip<F>::new(config, limb_width, bits_len) as a row counter."""

    def __init__(self, limb_width, bits_len):
        assert bits_len % limb_width == 0                                   # big_integer/chip.rs:1175
        self.w, self.L = limb_width, bits_len // limb_width
        self.n = 0
        wm = pyref.compute_mul_word_max(self.w, self.L)
        self.carry_bits = bits_size(2 * wm) - self.w                        # :841-842
        self.sub_bits = pyref.sublimb_bit_len(self.w)                       # :1357-1365

    # ---- maingate / RangeChip calls [3P, restated costs] ----------------------------------------------------------
    def op(self, k=1):
        self.n += k

    def is_zero(self):
        self.n += 3

    def is_equal(self):
        self.n += 4

    def range_assign(self, sub_bits, bit_len):
        nsub = bit_len // sub_bits + (1 if bit_len % sub_bits else 0)
        self.n += (nsub + 3) // 4

    def range_limb(self):
        self.range_assign(self.sub_bits, self.w)

    def to_bits(self, nb):
        self.n += nb + (nb + 3) // 4 + 1

    def div_mod_main_gate(self):                                            # :1323-1349: q, r, n*q, a - n*q, assert_equal
        self.op(5)

    # ---- BigIntInstructions ---------------------------------------------------------------------------------------------
    def assign_integer(self, num_limbs=None):                               # :62-82
        for _ in range(self.L if num_limbs is None else num_limbs):
            self.range_limb()
        return self.L if num_limbs is None else num_limbs

    def assign_constant(self, integer, max_num_limbs):                      # :1252-1281
        nl = -(-bits_size(integer) // self.w)
        assert nl <= max_num_limbs
        self.op(nl)
        self.op(1)                                                          # the shared zero cell :1276 (assigned even when unused)
        return max_num_limbs

    def assign_constant_fresh(self, integer):                               # instructions.rs: assign_constant(integer, num_limbs)
        return self.assign_constant(integer, self.L)

    def max_value(self, num_limbs):                                         # :138-154
        self.op(num_limbs)
        return num_limbs

    def add(self, n1, n2):                                                  # :245-297
        m = max(n1, n2)
        self.op(2)                                                          # zero_value :254, limb_max_val :267
        for _ in range(m):
            self.op(2)                                                      # a_b, sum :272-273
            self.range_limb(); self.range_limb()                            # c, carry :279-282
            self.op(2)                                                      # mul_add, assert_equal :283-285
        return m + 1

    def is_equal_fresh(self, n1, n2):                                       # :780-805
        self.op(1)                                                          # eq_bit
        for i in range(max(n1, n2)):
            if i >= min(n1, n2):
                self.is_zero()
            else:
                self.is_equal()
            self.op(1)                                                      # and

    def assert_equal_fresh(self, n1, n2):                                   # instructions.rs:197-206: is_equal_fresh + assert_one
        self.is_equal_fresh(n1, n2)
        self.op(1)

    def sub_unchecked(self, n1, n2):                                        # :1286-1318
        assert n1 >= n2
        for _ in range(n1):
            self.range_limb()
        added = self.add(n2, n1)
        self.assert_equal_fresh(n1, added)
        return n1

    def sub(self, n1, n2):                                                  # :310-373
        self.max_value(n2)
        infl = self.add(n1, n2)
        subed = self.sub_unchecked(infl, n2)
        self.op(1)                                                          # one :326
        self.is_equal()                                                     # :330
        self.op(1)                                                          # not :331
        nl, nr = subed, max(n1, n2)
        self.op(1)                                                          # zero_value :343
        self.op(nl + nr)                                                    # selects :345-367
        return self.sub_unchecked(nl, nr)

    def mul(self, d0, d1):                                                  # :386-419
        for i in range(d0 + d1 - 1):
            self.op(1)                                                      # acc = assign_constant(0) :402
            j0 = 0 if d1 >= i + 1 else i + 1 - d1
            self.op(sum(1 for j in range(j0, d0) if j <= i))                # mul_add :408
        return d0 + d1 - 1

    def refresh(self, n_l, n_r):                                            # :168-233
        inc = pyref.refresh_aux_increased_limbs(self.w, n_l, n_r)           # mod.rs:428-482
        nf = len(inc)
        self.op(2)                                                          # zero_val :186, limb_max :194
        for i in range(nf):
            for j in range(inc[i] + 1):
                self.div_mod_main_gate()                                    # :201
                if j:
                    self.op(1)                                              # add :207
            self.op(1)                                                      # assert_zero :213
        for _ in range(nf):
            self.range_limb()                                               # :219-224
            self.op(1)                                                      # assert_equal :225
        return nf

    def is_equal_muled(self, n_l, n_r, cols):                               # :822-895
        self.op(4)                                                          # limb_max, acc_extra, carry, eq_bit :851-856
        for i in range(cols):
            self.op(2)                                                      # a_b :859, sum :860-861
            self.div_mod_main_gate()                                        # :864
            self.op(1)                                                      # acc_extra + word_max :869-870
            self.div_mod_main_gate()                                        # :871
            self.is_equal()                                                 # cs_acc_eq :873
            self.op(1)                                                      # and :874
            if i < cols - 1:
                self.range_assign(pyref.sublimb_bit_len(self.carry_bits), self.carry_bits)   # :880-885
                self.is_equal(); self.op(1)                                 # :886-887
            else:
                self.is_equal(); self.op(1)                                 # :888-892

    def mul_mod(self, n_a=None):                                            # :542-629 (a.num_limbs() == n.num_limbs(), :555)
        L = self.L
        for _ in range(2 * L):
            self.range_limb()                                               # q limbs :588-591, r limbs :596-599
        self.mul(L, L); self.mul(L, L)                                      # :608-609
        self.op(L)                                                          # eq_b = qn + r :617
        self.is_equal_muled(L, L, 2 * L - 1)                                # assert_equal_muled :626 ...
        self.op(1)                                                          # ... its assert_one :1062
        return L

    def square_mod(self):
        return self.mul_mod()

    def pow_mod(self, e_num_limbs, exp_limb_bits):                          # :664-696
        for _ in range(e_num_limbs):
            self.to_bits(exp_limb_bits)                                     # :674-681
        self.assign_constant_fresh(1)                                       # :682
        for _ in range(e_num_limbs * exp_limb_bits):
            self.mul_mod()                                                  # :686
            self.op(self.L)                                                 # selects :688-691
            self.square_mod()                                               # :693
        return self.L

    def pow_mod_fixed_exp(self, e):                                         # :710-742
        self.assign_constant(1, self.L)                                     # :729
        for t in range(bits_size(e)):
            self.square_mod()
            if (e >> t) & 1:
                self.mul_mod()
        return self.L

    def is_zero_int(self, n1):                                              # :754-767
        self.op(1)
        for _ in range(n1):
            self.is_zero(); self.op(1)

    def is_less_than_or_equal(self, n1, n2):                                # :932-941
        self.sub(n1, n2)

    def is_less_than(self, n1, n2):                                         # :908-919
        self.is_less_than_or_equal(n1, n2)
        self.is_equal_fresh(n1, n2)
        self.op(2)                                                          # not, and

    def is_greater_than(self, n1, n2):                                      # :954-963
        self.is_less_than_or_equal(n1, n2); self.op(1)

    def is_greater_than_or_equal(self, n1, n2):                             # :976-985
        self.is_less_than(n1, n2); self.op(1)

    def is_in_field(self, n1, n2):                                          # :998-1006
        self.is_less_than(n1, n2)

    def assert_in_field(self, n1=None, n2=None):                            # :1150-1158
        self.is_in_field(self.L if n1 is None else n1, self.L if n2 is None else n2)
        self.op(1)

    def select_tail(self, num_limbs, n_limbs):                              # tail of add_mod :466-478 / sub_mod :512-525
        self.op(1)                                                          # zero_value
        self.op(num_limbs)                                                  # selects
        self.op(num_limbs - n_limbs)                                        # assert_zero of the limbs beyond n's

    def add_mod(self, n1=None, n2=None):                                    # :452-479
        L = self.L
        added = self.add(L if n1 is None else n1, L if n2 is None else n2)
        subed = self.sub(added, L)
        self.select_tail(subed, L)
        return L

    def sub_mod(self):                                                      # :493-528
        L = self.L
        s1 = self.sub(L, L)
        s2 = self.sub(L, s1)
        self.op(1)                                                          # assert_zero(is_overflowed2) :510
        self.select_tail(s2, L)
        return L


class RsaRows(Rows):
    """RSAChip<F>::new(config, bits_len, exp_limb_bits) (src/chip.rs:203-221; LIMB_WIDTH = 64)."""

    def __init__(self, bits_len, exp_limb_bits=5):
        super().__init__(64, bits_len)
        self.exp_limb_bits = exp_limb_bits

    def assign_public_key(self, var_e_limbs=0):                             # src/chip.rs:58-70
        self.assign_integer()
        if var_e_limbs:
            self.assign_integer(var_e_limbs)

    def assign_signature(self):                                             # :80-88
        self.assign_integer()

    def modpow_public_key(self, e=None, var_e_limbs=0):                     # :99-114
        self.assert_in_field()                                              # :106
        if var_e_limbs:
            self.pow_mod(var_e_limbs, self.exp_limb_bits)                   # :108-110
        else:
            self.pow_mod_fixed_exp(e)                                       # :111

    def verify_pkcs1v15_signature(self, e=65537, var_e_limbs=0):            # :128-199
        L = self.L
        self.op(1)                                                          # is_eq = assign_constant(1) :137
        self.modpow_public_key(e, var_e_limbs)                              # :138
        for _ in range(4):
            self.is_equal(); self.op(1)                                     # :141-144
        self.op(2)                                                          # prefix_64_1, prefix_64_2 :149-152
        self.is_equal(); self.is_equal(); self.op(2)                        # :153-156
        self.range_assign(4, 32); self.range_assign(4, 32)                  # :170-171
        self.op(3)                                                          # u32_assign, mul_add, assert_equal :172-174
        self.op(1); self.is_equal(); self.op(1)                             # prefix_32 :175-177
        self.op(1); self.is_equal(); self.op(1)                             # ff_32 :180-182
        self.op(1)                                                          # ff_64 :183-184
        for _ in range(4 + 3, L - 1):
            self.is_equal(); self.op(1)                                     # :185-188
        self.op(1); self.is_equal(); self.op(1)                             # last_em :190-197
