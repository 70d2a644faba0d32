"""CPU-side checks of the C ABI: every symbol include/h2r.h declares is exported, parameter and
layout queries match the goldens, and h2r_trace_flatten agrees with an independent Python placement
of the oracle's stream (pins the documented plane index maps).  No device work, no compute calls."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import H2RLayout, H2RParams, H2RPowLayout, lib
from layout_ref import unflatten_record
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(64, 4), (64, 8), (64, 16), (64, 32), (64, 64), (32, 8), (32, 32), (32, 64), (32, 128), (64, 48), (64, 24), (32, 96), (64, 12)]


def host_ctx(w, L):
    ctx = ctypes.c_void_p()
    p = H2RParams(w, w * L, 0, -1)
    rc = lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx))
    assert rc == 0, rc
    return ctx


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "h2r.h")).read()
    declared = sorted(set(re.findall(r"\b(h2r_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = ctypes.CDLL(_lib.lib_path())
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared
    # ... and nothing else: the library is built with hidden visibility and a version script, so that a Rust cdylib or a C++
    # host linking it never meets an un-prefixed internal (call_plan, launch_step, ... in round 2) or a libstdc++ weak symbol
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.lib_path()], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared, sorted(set(exported) ^ set(declared))


def test_integration_md_binds_exactly_the_header():
    """The Rust `extern "C"` block of INTEGRATION.md (what a maintainer pastes into the reference crate) lists exactly the header's
    prototypes -- the same set the library exports."""
    hdr = open(os.path.join(ROOT, "include", "h2r.h")).read()
    declared = sorted(set(re.findall(r"\b(h2r_[a-z0-9_]+)\s*\(", hdr)))
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    bound = sorted(set(re.findall(r"pub fn (h2r_[a-z0-9_]+)\(", md)))
    assert bound == declared, sorted(set(bound) ^ set(declared))


def test_ctx_create_status_codes():
    ctx = ctypes.c_void_p()
    for (w, bits, field, want) in [(64, 2048 + 32, 0, _lib.H2R_E_SHAPE),      # bits_len % limb_width (chip.rs:1175)
                                   (0, 2048, 0, _lib.H2R_E_SHAPE), (16, 2048, 0, _lib.H2R_E_UNSUPPORTED),
                                   (64, 64 * 3, 0, _lib.H2R_E_UNSUPPORTED), (64, 2048, 9, _lib.H2R_E_SHAPE)]:
        p = H2RParams(w, bits, field, -1)
        assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == want
    # num_limbs need not be a power of two (chip.rs:1174-1185 only asserts divisibility): RSA-3072 / RSA-1536 / 32-bit limbs
    for (w, bits) in [(64, 3072), (64, 1536), (64, 768), (32, 3072), (32, 768), (64, 4096), (32, 4096)]:
        p = H2RParams(w, bits, 0, -1)
        assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == _lib.H2R_OK, (w, bits)
        lo = H2RLayout()
        assert lib().h2r_trace_layout(ctx, ctypes.byref(lo)) == 0 and lo.num_limbs == bits // w and lo.record_stride % 256 == 0
        lib().h2r_ctx_destroy(ctx)
    for (w, bits) in [(64, 64 * 6), (32, 32 * 12), (64, 64 * 68), (32, 32 * 136)]:   # not a multiple of 4 / 8 limbs, or too long
        p = H2RParams(w, bits, 0, -1)
        assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == _lib.H2R_E_UNSUPPORTED, (w, bits)
    assert lib().h2r_ctx_create(None, ctypes.byref(ctx)) == _lib.H2R_E_NULL
    assert lib().h2r_status_str(_lib.H2R_E_NOT_REDUCED).decode().startswith("quotient")
    c = host_ctx(64, 32)   # host-only ctx refuses device work
    st = (ctypes.c_uint8 * 1)()
    buf = (ctypes.c_uint64 * 32)()
    assert lib().h2r_mul_mod_batch(c, buf, buf, buf, 1, 0, None, None, st, None, None) == _lib.H2R_E_UNSUPPORTED
    lib().h2r_ctx_destroy(c)


def test_range_lens_and_layout_match_goldens(golden):
    for row in golden["params"]:
        comp, over = (ctypes.c_uint32 * 3)(), (ctypes.c_uint32 * 3)()
        assert lib().h2r_compute_range_lens(row["w"], row["L"], comp, over) == 0
        assert list(comp) == row["comp"] and list(over) == row["over"]
        if (row["w"], row["L"]) in SHAPES:
            c = host_ctx(row["w"], row["L"])
            lo = H2RLayout()
            assert lib().h2r_trace_layout(c, ctypes.byref(lo)) == 0
            assert (lo.limb_bytes, lo.wide_bytes, lo.carry_bytes, lo.carry_bits, lo.word_max_bits, lo.stream_bytes) == \
                (row["LB"], row["WB"], row["CB"], row["carry_bits"], row["word_max_bits"], row["mul_mod_stream_bytes"])
            acc = {_lib.PLANES.index(n) for n in ("AB_LO", "AB_HI", "QN_LO", "QN_HI")}
            offs = [o for k, o in enumerate(lo.plane_off) if k not in acc or k == _lib.PLANES.index("AB_LO")]
            assert offs == sorted(offs) and all(o % 256 == 0 for o in offs) and lo.record_stride % 256 == 0
            assert all(lo.plane_off[k] % 16 == 0 and lo.plane_off[k] < lo.record_stride for k in acc)
            assert lo.acc_steps_per_group in (1, 2) and lo.acc_lo_group_bytes % 16 == 0
            lib().h2r_ctx_destroy(c)
    comp4, over3 = (ctypes.c_uint32 * 4)(), (ctypes.c_uint32 * 3)()
    assert lib().h2r_rsa_compute_range_lens(32, comp4, over3) == 0
    assert [list(comp4), list(over3)] == golden["rsa_range_lens_2048"]


def test_pow_layouts():
    c = host_ctx(64, 32)
    o = Oracle(64, 32)
    pl = H2RPowLayout()
    for e in (65537, 1, 0b1011011, (1 << 2047) | 12345, 0):
        eb = int(e).to_bytes(max(1, (e.bit_length() + 7) // 8), "little")
        assert lib().h2r_pow_fixed_layout(c, eb, len(eb), ctypes.byref(pl)) == 0
        assert pl.num_mul_mods == e.bit_length() + bin(e).count("1") and pl.num_exp_bits == e.bit_length()
        assert pl.stream_bytes == o.pow_fixed_stream_bytes(e)
    assert lib().h2r_pow_var_layout(c, 1, 5, ctypes.byref(pl)) == 0
    assert pl.num_mul_mods == 10 and pl.stream_bytes == o.pow_var_stream_bytes(1, 5)
    assert lib().h2r_pow_var_layout(c, 1, 65, ctypes.byref(pl)) == _lib.H2R_E_SHAPE
    lib().h2r_ctx_destroy(c)


@pytest.mark.parametrize("w,L", SHAPES)
def test_flatten_inverts_documented_layout(w, L):
    """oracle stream -> (python placement per include/h2r.h) -> h2r_trace_flatten == oracle stream."""
    c = host_ctx(w, L)
    lo = H2RLayout()
    lib().h2r_trace_layout(c, ctypes.byref(lo))
    o = Oracle(w, L)
    rng = random.Random(w + L)
    n = rng.getrandbits(w * L) | (1 << (w * L - 1))
    a, b = rng.randrange(n), rng.randrange(n)
    rc, r, st = o.mul_mod(o.limbs(a), o.limbs(b), o.limbs(n))
    assert rc == 0
    rec = unflatten_record(st, lo, _lib.PLANES)
    out = np.zeros(lo.stream_bytes, dtype=np.uint8)
    assert lib().h2r_trace_flatten(c, rec.ctypes.data, out.ctypes.data) == 0
    assert np.array_equal(out, st)
    lib().h2r_ctx_destroy(c)


@pytest.mark.parametrize("field", ["bn254_fr", "bn254_fq", "pasta_fp", "pasta_fq"])
@pytest.mark.parametrize("w,L", [(64, 32), (32, 128), (64, 12)])
def test_flatten_field_encoded_a_b(w, L, field):
    """H2R_STREAM_FIELD_AB: a_b = a[i] - b[i] is a FIELD subtraction in the reference (big_integer/chip.rs:859), so a
    negative difference is the element p - |x|.  h2r_trace_flatten_ex with the flag == the Python restatement run with
    the field's modulus (a_b streamed as a 32-byte canonical element); both signs occur in every record."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyref as R
    ctx = ctypes.c_void_p()
    p = H2RParams(w, w * L, _lib.FIELDS[field], -1)
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    lo = H2RLayout()
    lib().h2r_trace_layout(ctx, ctypes.byref(lo))
    rng = random.Random(w * L + len(field))
    n = rng.getrandbits(w * L) | (1 << (w * L - 1))
    a, b = rng.randrange(n), rng.randrange(n)
    o = Oracle(w, L)
    rc, r, st = o.mul_mod(o.limbs(a), o.limbs(b), o.limbs(n))
    rec = unflatten_record(st, lo, _lib.PLANES)
    pf = R.Params(w, L, field_modulus=R.FIELD_MODULI[field])
    want = R.Stream()
    R.mul_mod(pf, R.to_limbs(a, L, w), R.to_limbs(b, L, w), R.to_limbs(n, L, w), want)
    want = np.frombuffer(want.bytes(), dtype=np.uint8)
    nb = lib().h2r_stream_bytes(ctx, _lib.H2R_STREAM_FIELD_AB)
    assert nb == len(want) == lo.stream_bytes + lo.num_cols * (32 - lo.wide_bytes)
    out = np.zeros(nb, dtype=np.uint8)
    assert lib().h2r_trace_flatten_ex(ctx, rec.ctypes.data, _lib.H2R_STREAM_FIELD_AB, out.ctypes.data) == 0
    assert np.array_equal(out, want)
    # both signs were exercised: some a_b has its top bytes equal to p's, some is small
    plain = R.Stream()
    R.mul_mod(R.Params(w, L), R.to_limbs(a, L, w), R.to_limbs(b, L, w), R.to_limbs(n, L, w), plain)
    assert plain.bytes() == bytes(st)
    out0 = np.zeros(lo.stream_bytes, dtype=np.uint8)
    assert lib().h2r_trace_flatten_ex(ctx, rec.ctypes.data, 0, out0.ctypes.data) == 0 and np.array_equal(out0, st)
    lib().h2r_ctx_destroy(ctx)


def test_element_strides_avoid_slow_interleave_residues():
    """Element strides (pow / var-pow / verify layouts) are odd multiples of 256 bytes whose residue modulo 256 units
    stays out of 24..62 (profiles/r01_elem_stride_sweep.txt), and they cover everything they hold."""
    from halo2_rsa_amd._lib import H2RVerifyLayout
    for w, L in ((64, 32), (64, 16), (64, 64), (32, 128)):
        c = host_ctx(w, L)
        pl = H2RPowLayout()
        for e in (65537, 3, (1 << 2047) | 1, (1 << 40) - 1, 0x10001 << 7):
            eb = int(e).to_bytes(max(1, (e.bit_length() + 7) // 8), "little")
            assert lib().h2r_pow_fixed_layout(c, eb, len(eb), ctypes.byref(pl)) == 0
            u = pl.elem_stride // 256
            assert pl.elem_stride % 256 == 0 and u % 2 == 1 and not 24 <= u % 256 <= 62, (w, L, e, u)
            assert pl.elem_stride >= pl.off_result + L * (w // 8)
            if w == 64:
                vl = H2RVerifyLayout()
                assert lib().h2r_verify_layout_fixed(c, eb, len(eb), ctypes.byref(vl)) == 0
                u = vl.elem_stride // 256
                assert u % 2 == 1 and not 24 <= u % 256 <= 62 and vl.elem_stride >= vl.off_em + vl.em_stream_bytes
        assert lib().h2r_pow_var_layout(c, 2, 5, ctypes.byref(pl)) == 0
        assert (pl.elem_stride // 256) % 2 == 1
        lib().h2r_ctx_destroy(c)


def test_verify_layout_sizes(golden):
    """in-field / EM stream sizes of h2r_verify_layout_fixed equal the oracle's."""
    from halo2_rsa_amd._lib import H2RVerifyLayout
    from oracle_lib import lib as olib
    for L in (16, 32, 64):
        c = host_ctx(64, L)
        o = Oracle(64, L)
        vl = H2RVerifyLayout()
        eb = (65537).to_bytes(3, "little")
        assert lib().h2r_verify_layout_fixed(c, eb, 3, ctypes.byref(vl)) == 0
        assert vl.in_field_stream_bytes == int(olib().h2ro_in_field_stream_bytes(ctypes.byref(o.p)))
        assert vl.em_stream_bytes == int(olib().h2ro_pkcs1v15_stream_bytes(ctypes.byref(o.p))) == 2 * L + 34
        assert vl.stream_bytes == vl.in_field_stream_bytes + o.pow_fixed_stream_bytes(65537) + vl.em_stream_bytes
        assert vl.off_in_field % 256 == 0 and vl.off_em % 256 == 0 and vl.elem_stride % 256 == 0
        lib().h2r_ctx_destroy(c)
    assert golden["rsa_kats"][0]["in_field_stream_bytes"] == 9620
    c = host_ctx(32, 128)
    assert lib().h2r_verify_layout_fixed(c, (65537).to_bytes(3, "little"), 3, ctypes.byref(H2RVerifyLayout())) == _lib.H2R_E_SHAPE
    lib().h2r_ctx_destroy(c)


def test_exp_segment_plan():
    """h2r_exp_segment_plan: how a long exponent on a latency-bound batch is walked (host logic of exp_segment_count /
    exp_segment_plan): up to 16 segments of at least 128 bits, boundaries on 32-bit words of e, the mul_mod boundaries follow the
    reference's call order (pow_mod_fixed_exp, big_integer/chip.rs:731-740: one squaring per bit + one multiply per set bit; pow_mod,
    :684-694: two per bit).  Short exponents, batches of more than two elements per CU (256 CUs on a host-only ctx) are not cut."""
    import random
    c = host_ctx(64, 32)

    def plan(batch, e=None, var_bits=0):
        bits, muls, n = (ctypes.c_uint32 * 40)(), (ctypes.c_uint32 * 40)(), ctypes.c_uint32()
        eb = e.to_bytes((e.bit_length() + 7) // 8, "little") if e is not None else None
        assert lib().h2r_exp_segment_plan(c, batch, eb, len(eb) if eb else 0, var_bits, bits, muls, 40, ctypes.byref(n)) == _lib.H2R_OK
        return n.value, [int(bits[i]) for i in range(n.value + 1)], [int(muls[i]) for i in range(n.value + 1)]

    rng = random.Random(7)
    e = rng.getrandbits(2048) | (1 << 2047)
    n, bits, muls = plan(256, e)
    assert n == 16 and bits == list(range(0, 2049, 128))
    assert muls[0] == 0 and muls[-1] == 2048 + bin(e).count("1")
    for i in range(n):   # a segment's mul_mods: its bits + its set bits
        seg = (e >> bits[i]) & ((1 << (bits[i + 1] - bits[i])) - 1)
        assert muls[i + 1] - muls[i] == (bits[i + 1] - bits[i]) + bin(seg).count("1")
    e700 = rng.getrandbits(700) | (1 << 699)
    n, bits, muls = plan(5, e700)
    assert n == 5 and bits[0] == 0 and bits[-1] == 700 and all(b % 32 == 0 for b in bits[:-1]) and bits == sorted(set(bits))
    assert muls[-1] == 700 + bin(e700).count("1")
    assert plan(256, 65537)[0] == 1 and plan(256, (1 << 511) - 1)[0] == 1      # short exponents
    assert plan(513, e)[0] == 1 and plan(512, e)[0] == 16                       # more than two elements per CU: not cut
    n, bits, muls = plan(256, None, 2048)                                       # variable exponent: two mul_mods per bit
    assert n == 16 and muls == [2 * b for b in bits]
    assert plan(6, None, 600)[0] == 4
    nn = ctypes.c_uint32()
    assert lib().h2r_exp_segment_plan(None, 1, None, 0, 600, None, None, 0, ctypes.byref(nn)) == _lib.H2R_E_NULL
    assert lib().h2r_exp_segment_plan(c, 256, None, 0, 2048, None, None, 0, ctypes.byref(nn)) == _lib.H2R_OK and nn.value == 16
    lib().h2r_ctx_destroy(c)


def test_pipeline_call_plan():
    """h2r_pipeline_call_plan: how a fixed-exponent call is walked (host logic of pipeline_plan).  Record-bound shapes
    (RSA-1536/2048) grow by 3/2 from one chain-kernel grid when the pipeline is empty and are one launch pair when it is
    busy; chain-bound 64-bit-limb shapes (RSA-3072/4096) are uniform sub-batches either way; RSA-1024 and the 32-bit-limb
    shapes are never split.  The sizes always add up to the batch and every boundary is a multiple of 256 elements."""
    def plan(w, L, batch, busy):
        c = host_ctx(w, L)
        sizes = (ctypes.c_uint64 * 64)()
        n, paced = ctypes.c_uint32(), ctypes.c_uint32()
        assert lib().h2r_pipeline_call_plan(c, batch, busy, sizes, 64, ctypes.byref(n), ctypes.byref(paced)) == _lib.H2R_OK
        out = [int(sizes[i]) for i in range(n.value)]
        lib().h2r_ctx_destroy(c)
        assert sum(out) == batch and all(s % 256 == 0 for s in out[:-1])
        return out, bool(paced.value)

    assert plan(64, 32, 1024, 0) == ([1024], False)
    assert plan(64, 32, 1536, 0) == ([1536], False)
    assert plan(64, 32, 3840, 0) == ([1024, 1536, 1280], True)
    assert plan(64, 32, 8192, 0) == ([1024, 1536, 2304, 3328], True)
    assert plan(64, 32, 8192, 1) == ([8192], False)
    assert plan(64, 24, 4096, 0) == ([1024, 1536, 1536], True)
    assert plan(64, 48, 4096, 0) == ([1024] * 4, False)
    assert plan(64, 48, 4096, 1) == ([1024] * 4, False)
    assert plan(64, 48, 1536, 0) == ([1536], False)
    assert plan(64, 64, 2048, 0) == ([1024, 1024], False)
    assert plan(64, 16, 8192, 0) == ([8192], False)
    assert plan(32, 128, 4096, 0) == ([4096], False)
    assert plan(64, 32, 100000, 0)[0][:3] == [1024, 1536, 2304]
    n = ctypes.c_uint32()
    assert lib().h2r_pipeline_call_plan(None, 1, 0, None, 0, ctypes.byref(n), None) == _lib.H2R_E_NULL


FIELD_P = {
    "bn254_fr": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "bn254_fq": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "pasta_fp": 28948022309329048855892746252171976963363056481941560715954676764349967630337,
    "pasta_fq": 28948022309329048855892746252171976963363056481941647379679742748393362948097,
}


def _fe(v):
    return (ctypes.c_uint64 * 4)(*[(v >> (64 * k)) & (2 ** 64 - 1) for k in range(4)])


def _int(a):
    return sum(int(a[k]) << (64 * k) for k in range(4))


@pytest.mark.parametrize("field", sorted(FIELD_P))
def test_field_arithmetic_matches_python(field):
    """h2r_field_eval runs (on the host) the very code the kernels use for the theta-compression of the lookup inputs and for
    main_gate.is_zero's inverse witness: + - * and inversion in all four fields against Python big integers."""
    P = FIELD_P[field]
    ctx = ctypes.c_void_p()
    p = H2RParams(64, 2048, _lib.FIELDS[field], -1)
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    rng = random.Random(hash(field) & 0xffff)
    vals = [0, 1, 2, P - 1, P - 2, (1 << 64) - 1, 1 << 64, 1 << 128, (1 << 135) + 12345] + [rng.randrange(P) for _ in range(400)] + [P - (1 << k) for k in range(1, 250, 7)] + [1 << k for k in range(0, 254, 5)]
    out = (ctypes.c_uint64 * 4)()
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        for op, want in ((0, (a + b) % P), (1, (a - b) % P), (2, (a * b) % P)):
            assert lib().h2r_field_eval(ctx, op, _fe(a), _fe(b), out) == 0
            assert _int(out) == want, (field, op, a, b)
        if a:
            assert lib().h2r_field_eval(ctx, 3, _fe(a), None, out) == 0
            assert _int(out) == pow(a, P - 2, P) and (_int(out) * a) % P == 1
            assert lib().h2r_field_eval(ctx, 4, _fe(a), None, out) == 0 and _int(out) == pow(a, P - 2, P)      # Fermat cross-check
            assert lib().h2r_field_eval(ctx, 5, _fe(a), None, out) == 0 and _int(out) == pow(a, P - 2, P)      # one-word Euclid / fall-back
    # op 5's fast path: +-(one word), the differences main_gate.is_zero sees on this path (limb - limb, limb - constant, flags)
    words = [2, 3, 255, 256, (1 << 32) - 1, 1 << 32, (1 << 63) - 1, 1 << 63, (1 << 64) - 1, (1 << 64) - 2, P % (1 << 64) or 7,
             0x0304020105000420, 0x0001ffffffffffff] + [rng.getrandbits(64) | 1 for _ in range(300)] + [rng.getrandbits(rng.randrange(2, 65)) or 5 for _ in range(300)]
    for w in words:
        for a in (w, P - w):
            assert lib().h2r_field_eval(ctx, 5, _fe(a), None, out) == 0
            assert (_int(out) * a) % P == 1 and _int(out) < P, (field, a)
    assert lib().h2r_field_eval(ctx, 3, _fe(0), None, out) == _lib.H2R_E_SHAPE        # no inverse of zero
    assert lib().h2r_field_eval(ctx, 0, _fe(P), _fe(1), out) == _lib.H2R_E_SHAPE       # canonical elements only
    lib().h2r_ctx_destroy(ctx)


def test_lookup_config_and_table_image():
    """RangeChip::configure's bit_len -> tag map and the table load_table writes (restated third-party behaviour)
    against the independent Python restatement; custom tag maps; the reference's RSA-2048 table has 339 rows."""
    import advice_ref as AR
    for (w, L, rsa) in [(64, 32, True), (64, 32, False), (32, 128, False), (64, 16, True), (64, 64, False), (32, 8, False)]:
        ctx = host_ctx(w, L)
        cfg = _lib.H2RLookupConfig()
        assert lib().h2r_lookup_config_default(ctx, 1 if rsa else 0, ctypes.byref(cfg)) == 0
        ref = AR.LookupConfig(AR.range_lens(w, L, rsa=rsa))
        assert list(cfg.bit_len[:cfg.n_lens]) == ref.bit_lens and list(cfg.tag[:cfg.n_lens]) == ref.tags
        assert cfg.n_rows == ref.n_rows and [cfg.row_off[i] for i in range(cfg.n_lens)] == [ref.row_off[b] for b in ref.bit_lens]
        tag_col = np.zeros((cfg.n_rows, 4), dtype=np.uint64)
        val_col = np.zeros((cfg.n_rows, 4), dtype=np.uint64)
        assert lib().h2r_lookup_table_image(ctx, ctypes.byref(cfg), tag_col.ctypes.data, val_col.ctypes.data) == 0
        assert [(int(t[0]), int(v[0])) for t, v in zip(tag_col, val_col)] == ref.table()
        assert not tag_col[:, 1:].any() and not val_col[:, 1:].any()
        if (w, L, rsa) == (64, 32, True):
            assert ref.bit_lens == [1, 4, 6, 8] and cfg.n_rows == 339
        lib().h2r_ctx_destroy(ctx)
    # a maingate revision that tags a bit length with the bit length itself
    lens = (ctypes.c_uint32 * 5)(8, 0, 6, 8, 1)
    tags = (ctypes.c_uint32 * 5)(8, 0, 6, 8, 1)
    cfg = _lib.H2RLookupConfig()
    assert lib().h2r_lookup_config_custom(lens, tags, 5, ctypes.byref(cfg)) == 0
    assert list(cfg.bit_len[:3]) == [1, 6, 8] and list(cfg.tag[:3]) == [1, 6, 8] and cfg.n_rows == 1 + 2 + 64 + 256
    tags[3] = 9                                     # the same bit length with two tags
    assert lib().h2r_lookup_config_custom(lens, tags, 5, ctypes.byref(cfg)) == _lib.H2R_E_SHAPE
    tags[3] = 8; tags[2] = 8                        # two bit lengths with one tag
    assert lib().h2r_lookup_config_custom(lens, tags, 5, ctypes.byref(cfg)) == _lib.H2R_E_SHAPE


def test_fresh_op_row_programs_follow_the_restated_op_order():
    """The row program of every Fresh-integer op (h2r_fresh_op_row_kinds: the host's symbolic walk of the reference's control flow,
    no GPU needed) has the row kinds -- hence the op order and row count -- of the independent Python restatement built from the
    ORACLE's stream (tests/advice_ref.fresh_image), and the fixed rows of the new kinds (select / not / assert_one / assert_zero /
    the 2^w - 1 constant) agree with the Python table in the ctx's field."""
    import random
    import advice_ref as AR
    from oracle_lib import FRESH_OPS, fresh_op
    for (w, L) in [(64, 32), (32, 16), (64, 4)]:
        ctx = host_ctx(w, L)
        o = Oracle(w, L)
        rng = random.Random(w * L)
        n = rng.getrandbits(w * L) | (1 << (w * L - 1)) | 1
        a, b = rng.randrange(n), rng.randrange(n)
        P = AR.FIELD_MODULI["bn254_fr"] if hasattr(AR, "FIELD_MODULI") else 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
        for k, name in enumerate(FRESH_OPS):
            rc, ov, of, st = fresh_op(o, name, o.limbs(a), o.limbs(b), o.limbs(n))
            assert rc == 0
            for assert_one in (False, True):
                fl = _lib.H2R_ADVICE_ASSERT_ONE if assert_one else 0
                rows = int(lib().h2r_fresh_op_advice_rows(ctx, k, fl))
                if assert_one and name in ("add", "sub", "add_mod", "sub_mod"):
                    assert rows == 0          # no bit to assert
                    continue
                im = AR.fresh_image(o.p, name, o.limbs(a), None if name == "is_zero" else o.limbs(b),
                                    o.limbs(n) if name in ("add_mod", "sub_mod") else None, st, P, assert_one=assert_one)
                kinds = np.zeros(rows, dtype=np.uint8)
                assert lib().h2r_fresh_op_row_kinds(ctx, k, fl, kinds.ctypes.data) == 0
                assert kinds.tolist() == im.kinds, (w, L, name, assert_one)
        if (w, L) == (64, 32):
            assert int(lib().h2r_fresh_op_advice_rows(ctx, FRESH_OPS.index("is_in_field"), _lib.H2R_ADVICE_ASSERT_ONE)) == 1532
        cfg = AR.LookupConfig(AR.range_lens(w, L))
        for kind in (15, 16, 17, 18, 19):
            fr = _lib.H2RFixedRow()
            assert lib().h2r_advice_fixed_row(ctx, None, kind, ctypes.byref(fr)) == 0
            ref = AR.fixed_row(kind, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
            got = fr.as_dict()
            assert {nm: v % P for nm, v in ref.items() if nm in AR.FIXED_NAMES} == {nm: got[nm] for nm in AR.FIXED_NAMES}, kind
        assert int(lib().h2r_fresh_op_advice_rows(ctx, 99, 0)) == 0
        lib().h2r_ctx_destroy(ctx)
