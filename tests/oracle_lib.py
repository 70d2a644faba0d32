"""ctypes binding of oracle/libh2r_oracle.so -- TEST INFRASTRUCTURE (the checker), never the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_SO = os.path.join(ORACLE_DIR, "libh2r_oracle.so")


class OracleParams(ctypes.Structure):
    _fields_ = [("w", ctypes.c_uint32), ("L", ctypes.c_uint32), ("LB", ctypes.c_uint32),
                ("WB", ctypes.c_uint32), ("CB", ctypes.c_uint32), ("word_max_bits", ctypes.c_uint32),
                ("carry_bits", ctypes.c_uint32), ("limb_sub_bits", ctypes.c_uint32),
                ("limb_nsub", ctypes.c_uint32), ("carry_sub_bits", ctypes.c_uint32),
                ("carry_nsub", ctypes.c_uint32), ("word_max", ctypes.c_uint64 * 4),
                ("mul_mod_stream_bytes", ctypes.c_uint64)]


def build():
    """Compile the C restatement with gcc (building the checker is not using it)."""
    src = os.path.join(ORACLE_DIR, "h2r_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ORACLE_DIR, "h2r_oracle.h"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for name in ("h2ro_pow_fixed_stream_bytes", "h2ro_pow_var_stream_bytes", "h2ro_in_field_stream_bytes",
                     "h2ro_pkcs1v15_stream_bytes"):
            getattr(_lib, name).restype = ctypes.c_uint64
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    """Thin object wrapper: numpy in, numpy out.  Limbs are uint64 (w=64) / uint32 (w=32) arrays."""

    def __init__(self, w, L):
        self.p = OracleParams()
        rc = lib().h2ro_params_init(ctypes.byref(self.p), w, L)
        if rc:
            raise ValueError("h2ro_params_init failed: %d" % rc)
        self.w, self.L = w, L
        self.dtype = np.uint64 if w == 64 else np.uint32

    @property
    def mul_mod_stream_bytes(self):
        return int(self.p.mul_mod_stream_bytes)

    def limbs(self, v, L=None):
        L = self.L if L is None else L
        m = (1 << self.w) - 1
        assert v >> (self.w * L) == 0
        return np.array([(v >> (self.w * i)) & m for i in range(L)], dtype=self.dtype)

    def to_int(self, a):
        return sum(int(x) << (self.w * i) for i, x in enumerate(a))

    def mul_mod(self, a, b, n, want_stream=True):
        st = np.zeros(self.mul_mod_stream_bytes, dtype=np.uint8) if want_stream else None
        r = np.zeros(self.L, dtype=self.dtype)
        rc = lib().h2ro_mul_mod(ctypes.byref(self.p), _ptr(np.ascontiguousarray(a, self.dtype)), _ptr(np.ascontiguousarray(b, self.dtype)),
                                _ptr(np.ascontiguousarray(n, self.dtype)), _ptr(st), _ptr(r))
        return rc, r, st

    def mul_columns(self, a, b):
        cols = np.zeros((2 * self.L - 1, 4), dtype=np.uint64)
        lib().h2ro_mul_columns(ctypes.byref(self.p), _ptr(np.ascontiguousarray(a, self.dtype)), _ptr(np.ascontiguousarray(b, self.dtype)), None, _ptr(cols))
        return [sum(int(cols[i, k]) << (64 * k) for k in range(4)) for i in range(2 * self.L - 1)]

    def pow_fixed_stream_bytes(self, e):
        eb = e_bytes(e)
        return int(lib().h2ro_pow_fixed_stream_bytes(ctypes.byref(self.p), eb, len(eb)))

    def pow_mod_fixed_exp(self, x, n, e, want_stream=True):
        eb = e_bytes(e)
        st = np.zeros(self.pow_fixed_stream_bytes(e), dtype=np.uint8) if want_stream else None
        out = np.zeros(self.L, dtype=self.dtype)
        rc = lib().h2ro_pow_mod_fixed_exp(ctypes.byref(self.p), _ptr(np.ascontiguousarray(x, self.dtype)), _ptr(np.ascontiguousarray(n, self.dtype)),
                                          eb, len(eb), _ptr(st), _ptr(out))
        return rc, out, st

    def pow_var_stream_bytes(self, e_num_limbs, exp_limb_bits):
        return int(lib().h2ro_pow_var_stream_bytes(ctypes.byref(self.p), e_num_limbs, exp_limb_bits))

    def pow_mod(self, x, e_limbs, exp_limb_bits, n, want_stream=True):
        e_limbs = np.ascontiguousarray(e_limbs, self.dtype)
        st = np.zeros(self.pow_var_stream_bytes(len(e_limbs), exp_limb_bits), dtype=np.uint8) if want_stream else None
        out = np.zeros(self.L, dtype=self.dtype)
        rc = lib().h2ro_pow_mod(ctypes.byref(self.p), _ptr(np.ascontiguousarray(x, self.dtype)), _ptr(e_limbs), len(e_limbs), exp_limb_bits,
                                _ptr(np.ascontiguousarray(n, self.dtype)), _ptr(st), _ptr(out))
        return rc, out, st

    def big_pow_mod(self, a, e, n):
        eb = e_bytes(e)
        out = np.zeros(self.L, dtype=self.dtype)
        rc = lib().h2ro_big_pow_mod(ctypes.byref(self.p), _ptr(np.ascontiguousarray(a, self.dtype)), eb, len(eb), _ptr(np.ascontiguousarray(n, self.dtype)), _ptr(out))
        return rc, out

    def assert_in_field(self, a, n):
        nb = int(lib().h2ro_in_field_stream_bytes(ctypes.byref(self.p)))
        st = np.zeros(nb, dtype=np.uint8)
        lt = ctypes.c_int(-1)
        rc = lib().h2ro_assert_in_field(ctypes.byref(self.p), _ptr(np.ascontiguousarray(a, self.dtype)), _ptr(np.ascontiguousarray(n, self.dtype)), _ptr(st), ctypes.byref(lt))
        return rc, lt.value, st

    def pkcs1v15_em_check(self, powed, hashed4):
        nb = int(lib().h2ro_pkcs1v15_stream_bytes(ctypes.byref(self.p)))
        st = np.zeros(nb, dtype=np.uint8)
        ok = ctypes.c_int(-1)
        h = np.ascontiguousarray(hashed4, np.uint64)
        rc = lib().h2ro_pkcs1v15_em_check(ctypes.byref(self.p), _ptr(np.ascontiguousarray(powed, np.uint64)), _ptr(h), _ptr(st), ctypes.byref(ok))
        return rc, ok.value, st

    def pow_mod_fixed_exp_batch(self, x, n, e, nthreads=1, want_stream=False, stream_buf=None):
        """stream_buf: a reusable [batch, stream_bytes] uint8 array (timing loops: no allocation / first-touch cost)."""
        eb = e_bytes(e)
        batch = x.shape[0]
        sb = self.pow_fixed_stream_bytes(e)
        st = stream_buf if stream_buf is not None else (np.zeros((batch, sb), dtype=np.uint8) if want_stream else None)
        assert st is None or (st.shape == (batch, sb) and st.dtype == np.uint8 and st.flags.c_contiguous)
        out = np.zeros((batch, self.L), dtype=self.dtype)
        status = np.zeros(batch, dtype=np.uint8)
        lib().h2ro_pow_mod_fixed_exp_batch(ctypes.byref(self.p), _ptr(np.ascontiguousarray(x, self.dtype)), _ptr(np.ascontiguousarray(n, self.dtype)),
                                           eb, len(eb), ctypes.c_uint64(batch), _ptr(st), _ptr(out), _ptr(status), nthreads)
        return out, status, st


def pow_mod_fixed_exp_timed(o, x, n, e, passes, nthreads):
    """cpu_baseline timing leg: (seconds, failed elements) for `passes` passes over the batch on `nthreads` threads."""
    eb = e_bytes(e)
    sec, bad = ctypes.c_double(0.0), ctypes.c_uint64(0)
    x, n = np.ascontiguousarray(x, o.dtype), np.ascontiguousarray(n, o.dtype)
    rc = lib().h2ro_pow_mod_fixed_exp_timed(ctypes.byref(o.p), _ptr(x), _ptr(n), eb, len(eb), ctypes.c_uint64(x.shape[0]),
                                            ctypes.c_uint64(passes), int(nthreads), ctypes.byref(sec), ctypes.byref(bad))
    assert rc == 0
    return sec.value, bad.value


def e_bytes(e):
    """e.to_bytes_le() (reference big_integer/chip.rs:719-720); BigUint zero is one zero byte."""
    return int(e).to_bytes(max(1, (int(e).bit_length() + 7) // 8), "little")


FRESH_OPS = ["add", "sub", "add_mod", "sub_mod", "is_zero", "is_equal_fresh", "is_less_than", "is_less_than_or_equal",
             "is_greater_than", "is_greater_than_or_equal", "is_in_field"]


def fresh_op(o, name, a, b, n):
    """Oracle for the Fresh-integer family: returns (rc, value limbs, flag, stream)."""
    lib().h2ro_fresh_op_stream_bytes.restype = ctypes.c_uint64
    k = FRESH_OPS.index(name)
    nb = int(lib().h2ro_fresh_op_stream_bytes(ctypes.byref(o.p), k))
    st = np.zeros(nb, dtype=np.uint8)
    vout = np.zeros(o.L + 4, dtype=o.dtype)
    nv, fl = ctypes.c_uint32(0), ctypes.c_int(-1)
    rc = lib().h2ro_fresh_op(ctypes.byref(o.p), k, _ptr(np.ascontiguousarray(a, o.dtype)),
                             _ptr(np.ascontiguousarray(b, o.dtype)) if b is not None else None,
                             _ptr(np.ascontiguousarray(n, o.dtype)) if n is not None else None, _ptr(st), _ptr(vout),
                             ctypes.byref(nv), ctypes.byref(fl))
    return rc, vout[:nv.value].copy(), fl.value, st


def _u256(vals):
    return np.array([[(int(v) >> (64 * k)) & (2 ** 64 - 1) for k in range(4)] for v in vals], dtype=np.uint64)


def refresh(o, cols):
    """Oracle BigIntChip::refresh of 2L-1 Muled columns -> (rc, 2L fresh limbs, stream)."""
    lib().h2ro_refresh_stream_bytes.restype = ctypes.c_uint64
    nb = int(lib().h2ro_refresh_stream_bytes(ctypes.byref(o.p)))
    st = np.zeros(nb, dtype=np.uint8)
    out = np.zeros(2 * o.L, dtype=o.dtype)
    rc = lib().h2ro_refresh(ctypes.byref(o.p), _ptr(_u256(cols)), _ptr(st), _ptr(out))
    return rc, out, st


def is_equal_muled(o, a_cols, b_cols):
    lib().h2ro_is_equal_muled_stream_bytes.restype = ctypes.c_uint64
    nb = int(lib().h2ro_is_equal_muled_stream_bytes(ctypes.byref(o.p)))
    st = np.zeros(nb, dtype=np.uint8)
    eq = ctypes.c_int(-1)
    lib().h2ro_is_equal_muled(ctypes.byref(o.p), _ptr(_u256(a_cols)), _ptr(_u256(b_cols)), _ptr(st), ctypes.byref(eq))
    return eq.value, st


def mul_stream(o, a, b):
    """Oracle BigIntChip::mul: (columns, stream of the L*L partial accumulators)."""
    st = np.zeros(o.L * o.L * o.p.WB, dtype=np.uint8)
    cols = np.zeros((2 * o.L - 1, 4), dtype=np.uint64)
    lib().h2ro_mul_columns(ctypes.byref(o.p), _ptr(np.ascontiguousarray(a, o.dtype)), _ptr(np.ascontiguousarray(b, o.dtype)), _ptr(st), _ptr(cols))
    return [sum(int(cols[i, k]) << (64 * k) for k in range(4)) for i in range(2 * o.L - 1)], st


def sha256(msg: bytes) -> bytes:
    """h2ro_sha256: the C restatement of FIPS 180-4 (the caller-side step, reference src/lib.rs:205-209)."""
    d = (ctypes.c_uint8 * 32)()
    lib().h2ro_sha256(bytes(msg), ctypes.c_uint64(len(msg)), d)
    return bytes(d)


def hashed_msg(digest: bytes):
    """h2ro_hashed_msg (reference src/lib.rs:210-239): (4 limbs as uint64, the step's 288-byte flat stream)."""
    h = np.zeros(4, dtype=np.uint64)
    st = np.zeros(288, dtype=np.uint8)
    lib().h2ro_hashed_msg(bytes(digest), _ptr(h), _ptr(st))
    return h, st
