"""Host-side mirror of the reference's RSAChip for the accelerated path (reference src/chip.rs:38-255,
src/lib.rs:25-140): RSAPublicKey / RSAPubE / RSASignature containers and
RSAInstructions::{assign_public_key, assign_signature, modpow_public_key} in batch form."""
from dataclasses import dataclass
from typing import Union

import ctypes

import numpy as np
import torch

from ._lib import H2R_ADVICE_DIRECT, H2R_HASHED_MSG_STREAM_BYTES, H2RVerifyLayout, check, lib
from .big_integer import AssignedInteger, BatchResult, BigIntChip, UnassignedInteger, _e_bytes


@dataclass
class Fix:
    """RSAPubE::Fix(BigUint) (src/lib.rs): the same fixed exponent for every element."""
    e: int


@dataclass
class Var:
    """RSAPubE::Var(UnassignedInteger): per-element exponent limbs."""
    e: Union[UnassignedInteger, AssignedInteger]


@dataclass
class RSAPublicKey:
    n: Union[UnassignedInteger, AssignedInteger]
    e: Union[Fix, Var]


@dataclass
class RSASignature:
    c: Union[UnassignedInteger, AssignedInteger]


class RSAChip:
    LIMB_WIDTH = 64  # src/chip.rs:203

    def __init__(self, bits_len: int, exp_limb_bits: int, field: str = "bn254_fr", device: int = 0, **advice_repr):
        """RSAChip::new (src/chip.rs:214-221).  advice_repr: BigIntChip's columns / montgomery / col_stride."""
        self.bits_len, self.exp_limb_bits = bits_len, exp_limb_bits
        self._bigint = BigIntChip(self.LIMB_WIDTH, bits_len, field, device, **advice_repr)

    def bigint_chip(self) -> BigIntChip:
        """src/chip.rs:224-230."""
        return self._bigint

    @staticmethod
    def compute_range_lens(num_limbs: int):
        """src/chip.rs:249-254."""
        comp, over = (ctypes.c_uint32 * 4)(), (ctypes.c_uint32 * 3)()
        check(lib().h2r_rsa_compute_range_lens(num_limbs, comp, over), "RSAChip::compute_range_lens")
        return list(comp), list(over)

    def assign_public_key(self, public_key: RSAPublicKey) -> RSAPublicKey:
        """src/chip.rs:58-70."""
        n = self._bigint.assign_integer(public_key.n)
        e = public_key.e
        if isinstance(e, Var):
            ev = e.e
            if isinstance(ev, UnassignedInteger):
                import numpy as np
                import torch
                t = torch.from_numpy(np.ascontiguousarray(ev.limbs).view(np.int64)).to(n.limbs_dev.device)
                ev = AssignedInteger(t.contiguous(), self.LIMB_WIDTH)
            e = Var(ev)
        return RSAPublicKey(n, e)

    def assign_signature(self, signature: RSASignature) -> RSASignature:
        """src/chip.rs:80-88."""
        return RSASignature(self._bigint.assign_integer(signature.c))

    def modpow_public_key(self, x: AssignedInteger, public_key: RSAPublicKey, want_trace: bool = True) -> BatchResult:
        """src/chip.rs:99-114: assert_in_field(x, n) (:106; witness in `result.in_field`, per-element status
        H2R_E_NOT_IN_FIELD where it fails), then Fix -> pow_mod_fixed_exp, Var -> pow_mod with the chip's exp_limb_bits."""
        if isinstance(public_key.e, Fix):
            return self._bigint.pow_mod_fixed_exp(x, public_key.e.e, public_key.n, want_trace, check_in_field=True)
        return self._bigint.pow_mod(x, public_key.e.e, public_key.n, self.exp_limb_bits, want_trace, check_in_field=True)


    def verify_pkcs1v15_signature(self, public_key: RSAPublicKey, hashed_msg, signature: RSASignature) -> "VerifyResult":
        """src/chip.rs:128-199 (after the SHA step): assert_in_field + modpow_public_key + encoded-message check.
        hashed_msg: per element the SHA-256 digest as an integer / 4 little-endian 64-bit limbs (src/chip.rs:141-144).
        RSAPubE::Fix -> h2r_verify_pkcs1v15_batch, RSAPubE::Var -> h2r_verify_pkcs1v15_var_batch (the chip's exp_limb_bits)."""
        chip = self._bigint
        n, sig = chip.assign_integer(public_key.n), chip.assign_integer(signature.c)
        batch, dev = sig.batch, sig.limbs_dev.device
        if isinstance(hashed_msg, AssignedInteger):
            hashed = hashed_msg.limbs_dev
        elif isinstance(hashed_msg, torch.Tensor):
            hashed = hashed_msg
        else:
            hashed = torch.from_numpy(UnassignedInteger.from_ints(list(hashed_msg), 4, 64).limbs.view(np.int64)).to(dev)
        vl = self._verify_layout(public_key)
        bufs = _VerifyBuffers(chip, batch, vl, dev)
        self._verify_call(public_key, sig, n, hashed, batch, vl, bufs)
        return VerifyResult(bufs.is_valid, AssignedInteger(bufs.powed, 64), bufs.status, bufs.trace, vl, chip, bufs.ws, (sig, n, hashed))

    def _verify_layout(self, public_key: RSAPublicKey) -> H2RVerifyLayout:
        chip, vl = self._bigint, H2RVerifyLayout()
        if isinstance(public_key.e, Fix):
            eb = _e_bytes(public_key.e.e)
            check(lib().h2r_verify_layout_fixed(chip._ctx, eb, len(eb), ctypes.byref(vl)), "h2r_verify_layout_fixed")
        else:
            check(lib().h2r_verify_layout_var(chip._ctx, public_key.e.e.num_limbs(), self.exp_limb_bits, ctypes.byref(vl)), "h2r_verify_layout_var")
        return vl

    def _verify_call(self, public_key, sig, n, hashed, batch, vl, b):
        chip = self._bigint
        if isinstance(public_key.e, Fix):
            eb = _e_bytes(public_key.e.e)
            check(lib().h2r_verify_pkcs1v15_batch(chip._ctx, sig.data_ptr(), n.data_ptr(), eb, len(eb), hashed.data_ptr(), batch,
                                                  chip._flags(n, batch), b.trace.data_ptr(), b.powed.data_ptr(), b.is_valid.data_ptr(),
                                                  b.status.data_ptr(), b.ws.data_ptr(), chip._stream()), "verify_pkcs1v15_signature")
        else:
            e = public_key.e.e
            if not isinstance(e, AssignedInteger):
                raise TypeError("RSAPubE::Var: assign the public key first (RSAChip.assign_public_key)")
            check(lib().h2r_verify_pkcs1v15_var_batch(chip._ctx, sig.data_ptr(), n.data_ptr(), e.data_ptr(), e.num_limbs(), self.exp_limb_bits,
                                                      hashed.data_ptr(), batch, chip._flags(n, batch), b.trace.data_ptr(), b.powed.data_ptr(),
                                                      b.is_valid.data_ptr(), b.status.data_ptr(), b.ws.data_ptr(), chip._stream()),
                  "verify_pkcs1v15_signature (Var)")


class _VerifyBuffers:
    def __init__(self, chip, batch, vl, dev):
        self.trace = torch.empty(batch * vl.elem_stride, dtype=torch.uint8, device=dev)
        self.powed = torch.empty((batch, chip.num_limbs), dtype=torch.int64, device=dev)
        self.is_valid = torch.zeros(batch, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        self.ws = torch.empty(chip.workspace_bytes(batch, vl.pow.num_mul_mods), dtype=torch.uint8, device=dev)   # kept: emit_advice reads it


# ---- byte-level plumbing of the reference's example / verifier (BASELINE config 1) -------------------------------------
def signature_from_bytes_be(sig_bytes, bits_len: int, limb_width: int = 64) -> RSASignature:
    """examples/rsa_example.rs:175-178: the signature comes off the wire big-endian; `sign.reverse()`, then
    BigUint::from_bytes_le and decompose_big into bits_len / limb_width little-endian limbs.  One signature or a list."""
    sigs = [sig_bytes] if isinstance(sig_bytes, (bytes, bytearray)) else list(sig_bytes)
    vals = [int.from_bytes(bytes(sb)[::-1], "little") for sb in sigs]
    return RSASignature(UnassignedInteger.from_ints(vals, bits_len // limb_width, limb_width))


def hashed_msg_from_digest(digest) -> UnassignedInteger:
    """src/lib.rs:213-239: the 32 SHA-256 digest bytes are reversed and packed eight per limb, byte j of a limb with
    coefficient 2^(8j) -- i.e. the digest read as a big-endian integer, cut into 4 little-endian 64-bit limbs
    (the operand RSAChip::verify_pkcs1v15_signature compares with the low limbs of the encoded message, src/chip.rs:141-144)."""
    digests = [digest] if isinstance(digest, (bytes, bytearray)) else list(digest)
    out = np.zeros((len(digests), 4), dtype=np.uint64)
    for r, d in enumerate(digests):
        hashed_bytes = bytes(d)[::-1]                               # hashed_bytes.reverse()  (:213)
        assert len(hashed_bytes) == 32
        for i in range(4):                                          # bytes_len / limb_bytes limbs  (:225)
            limb = 0
            for j in range(8):                                      # limb_val += 2^(8j) * hashed_bytes[8i + j]  (:227-236)
                limb += hashed_bytes[8 * i + j] << (8 * j)
            out[r, i] = limb
    return UnassignedInteger(out)


def pack_messages(msgs, device):
    """Ragged message bytes for the device: (uint8 buffer, int64 offsets [batch + 1]) -- element e's message is
    buffer[offsets[e]:offsets[e + 1]].  Host-side packing of the caller's byte strings only; no hashing happens here."""
    msgs = [bytes(m) for m in msgs]
    off = np.zeros(len(msgs) + 1, dtype=np.int64)
    np.cumsum([len(m) for m in msgs], out=off[1:])
    buf = np.frombuffer(b"".join(msgs) or b"\x00", dtype=np.uint8).copy()
    return torch.from_numpy(buf).to(device), torch.from_numpy(off).to(device)


def check_packed_messages(buf, off, batch: int, device, max_byte_size=None):
    """A caller-packed (uint8 buffer, int64 offsets [batch + 1]) pair: the kernels take message e from off[e] to off[e + 1]
    unchecked, so the pair is validated here -- shapes, dtypes, device, monotonic offsets inside the buffer, and the SHA-256
    chip's max_byte_size (src/lib.rs:321) -- with one device reduction."""
    if not (isinstance(buf, torch.Tensor) and isinstance(off, torch.Tensor)):
        raise TypeError("packed messages: (uint8 tensor, int64 offsets tensor)")
    if buf.dtype != torch.uint8 or off.dtype != torch.int64 or buf.dim() != 1 or off.dim() != 1 or not buf.is_contiguous() or not off.is_contiguous():
        raise ValueError("packed messages: a contiguous uint8 buffer and contiguous int64 offsets")
    if off.numel() != batch + 1:
        raise ValueError("packed messages: offsets must have batch + 1 entries")
    if buf.device != torch.device(device) or off.device != torch.device(device):
        raise ValueError("packed messages: buffer and offsets must be on the chip's device")
    diff = off[1:] - off[:-1]
    facts = torch.stack([(diff >= 0).all(), off[0] >= 0, off[-1] <= buf.numel(),
                         (diff.max() <= max_byte_size) if (max_byte_size is not None and batch) else torch.ones((), dtype=torch.bool, device=off.device)]).cpu().tolist()
    if not (facts[0] and facts[1] and facts[2]):
        raise ValueError("packed messages: offsets must be non-negative, non-decreasing and end inside the buffer")
    if not facts[3]:
        raise ValueError("message longer than the SHA-256 chip's max_byte_size")


class RSASignatureVerifier:
    """src/lib.rs:149-246: SHA-256 of the message, the reversed digest packed into the hashed-message limbs, then
    RSAChip::verify_pkcs1v15_signature -- all on the device (h2r_signature_verifier_batch).  The SHA-256 chip's own circuit
    cells (third-party Table16 region) are outside the accelerated path; its digest VALUES and the limb composition rows of
    the verifier's region (:225-239) are produced.  sha256_max_byte_size mirrors the chip's configured capacity
    (Sha256Config::new(..., max_byte_size), src/lib.rs:321): a longer message is refused as the reference's circuit would."""

    def __init__(self, rsa_chip: "RSAChip", sha256_max_byte_size: int = None):
        self.rsa_chip = rsa_chip
        self.sha256_max_byte_size = sha256_max_byte_size

    def verify_pkcs1v15_signature(self, public_key: RSAPublicKey, msg, signature: RSASignature) -> "VerifyResult":
        """msg: one message (bytes, signed by every element), one message per element (list of bytes), or an already packed
        pair (uint8 device tensor, int64 device offsets [batch + 1])."""
        rsa, chip = self.rsa_chip, self.rsa_chip.bigint_chip()
        n, sig = chip.assign_integer(public_key.n), chip.assign_integer(signature.c)
        batch, dev = sig.batch, sig.limbs_dev.device
        if isinstance(msg, tuple):
            buf, off = msg
            check_packed_messages(buf, off, batch, dev, self.sha256_max_byte_size)
        else:
            msgs = [msg] * batch if isinstance(msg, (bytes, bytearray)) else list(msg)
            if len(msgs) != batch:
                raise ValueError("one message per signature")
            if self.sha256_max_byte_size is not None and any(len(m) > self.sha256_max_byte_size for m in msgs):
                raise ValueError("message longer than the SHA-256 chip's max_byte_size")
            buf, off = pack_messages(msgs, dev)
        vl = rsa._verify_layout(public_key)
        b = _VerifyBuffers(chip, batch, vl, dev)
        digest = torch.empty((batch, 32), dtype=torch.uint8, device=dev)
        hashed = torch.empty((batch, 4), dtype=torch.int64, device=dev)
        hm = torch.empty((batch, H2R_HASHED_MSG_STREAM_BYTES), dtype=torch.uint8, device=dev)
        if isinstance(public_key.e, Fix):     # one call: SHA-256, limb packing, verification (h2r_signature_verifier_batch)
            eb = _e_bytes(public_key.e.e)
            check(lib().h2r_signature_verifier_batch(chip._ctx, buf.data_ptr(), off.data_ptr(), 0, sig.data_ptr(), n.data_ptr(), eb, len(eb),
                                                     batch, chip._flags(n, batch), b.trace.data_ptr(), hm.data_ptr(), hm.shape[1],
                                                     digest.data_ptr(), hashed.data_ptr(), b.powed.data_ptr(), b.is_valid.data_ptr(),
                                                     b.status.data_ptr(), b.ws.data_ptr(), chip._stream()), "h2r_signature_verifier_batch")
        else:                                  # RSAPubE::Var: the two steps in stream order
            check(lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), off.data_ptr(), 0, batch, digest.data_ptr(), hashed.data_ptr(),
                                                    hm.data_ptr(), hm.shape[1], chip._stream()), "h2r_sha256_hashed_msg_batch")
            rsa._verify_call(public_key, sig, n, hashed, batch, vl, b)
        return VerifyResult(b.is_valid, AssignedInteger(b.powed, 64), b.status, b.trace, vl, chip, b.ws, (sig, n, hashed), digest, hm)


def sha256_hashed_msg(chip: BigIntChip, msgs, want_trace: bool = True):
    """Step 1 of the verifier alone (h2r_sha256_hashed_msg_batch): (digest uint8 [batch, 32], hashed limbs int64 [batch, 4],
    the step's flat stream uint8 [batch, 288] or None), all on the device."""
    dev = torch.device("cuda", chip.device)
    buf, off = msgs if isinstance(msgs, tuple) else pack_messages(msgs, dev)
    batch = off.numel() - 1
    if isinstance(msgs, tuple):
        check_packed_messages(buf, off, batch, dev)
    digest = torch.empty((batch, 32), dtype=torch.uint8, device=dev)
    hashed = torch.empty((batch, 4), dtype=torch.int64, device=dev)
    hm = torch.empty((batch, H2R_HASHED_MSG_STREAM_BYTES), dtype=torch.uint8, device=dev) if want_trace else None
    check(lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), off.data_ptr(), 0, batch, digest.data_ptr(), hashed.data_ptr(),
                                            hm.data_ptr() if want_trace else None, H2R_HASHED_MSG_STREAM_BYTES if want_trace else 0,
                                            chip._stream()), "h2r_sha256_hashed_msg_batch")
    return digest, hashed, hm


@dataclass
class VerifyResult:
    is_valid: "torch.Tensor"     # uint8 [batch]
    powed: AssignedInteger
    status: "torch.Tensor"
    trace: "torch.Tensor"
    layout: H2RVerifyLayout
    chip: BigIntChip
    workspace: "torch.Tensor" = None
    inputs: tuple = None         # (sig, n, hashed)
    digest: "torch.Tensor" = None        # RSASignatureVerifier only: uint8 [batch, 32], the `hashed_bytes` the reference returns (src/lib.rs:243-245)
    hashed_msg_trace: "torch.Tensor" = None   # RSASignatureVerifier only: the limb composition's flat stream, uint8 [batch, 288]

    def advice_sections(self):
        """Row counts of the image's four sections: is_eq seed, assert_in_field, pow_mod_fixed_exp, encoded-message check."""
        sec = (ctypes.c_uint64 * 4)()
        total = int(lib().h2r_verify_advice_rows(self.chip._ctx, ctypes.byref(self.layout), sec))
        return total, [int(v) for v in sec]

    def row_kinds(self) -> "np.ndarray":
        total, _ = self.advice_sections()
        kinds = np.zeros(total, dtype=np.uint8)
        check(lib().h2r_verify_row_kinds(self.chip._ctx, ctypes.byref(self.layout), kinds.ctypes.data), "h2r_verify_row_kinds")
        return kinds

    def emit_advice(self, with_hashed_msg: bool = False, direct: bool = False) -> "torch.Tensor":
        """Every cell of the whole verify_pkcs1v15_signature element as rows of the main gate's five advice columns
        (h2r_verify_emit_advice): uint8 [batch, rows * 160] in HBM.  with_hashed_msg (a result of RSASignatureVerifier): the
        region of src/lib.rs:220-241 -- the hashed-message limb composition rows (h2r_hashed_msg_emit_advice) in front.
        direct=True (H2R_ADVICE_DIRECT): the pow rows are written from the operands, the records are not read."""
        sig, n, hashed = self.inputs
        batch = sig.batch
        total, _ = self.advice_sections()
        pre = int(lib().h2r_hashed_msg_advice_rows(self.chip._ctx)) if with_hashed_msg else 0
        if pre and self.chip.columns and not self.chip.col_stride:
            raise ValueError("emit_advice(with_hashed_msg): two calls write one image -- planar columns need a fixed col_stride")
        out = torch.empty((batch, self.chip.image_bytes(pre + total)), dtype=torch.uint8, device=self.trace.device)
        row_bytes = 32 if self.chip.columns else 160
        if with_hashed_msg:
            check(lib().h2r_hashed_msg_emit_advice(self.chip._ctx, self.hashed_msg_trace.data_ptr(), self.hashed_msg_trace.shape[1], batch,
                                                   None, out.data_ptr(), out.shape[1], self.chip._stream()), "h2r_hashed_msg_emit_advice")
        check(lib().h2r_verify_emit_advice(self.chip._ctx, ctypes.byref(self.layout), sig.data_ptr(), n.data_ptr(), hashed.data_ptr(),
                                           self.powed.data_ptr(), self.chip._flags(n, batch) | (H2R_ADVICE_DIRECT if direct else 0), self.trace.data_ptr(),
                                           self.workspace.data_ptr(), batch, self.status.data_ptr(), out.data_ptr() + pre * row_bytes, out.shape[1],
                                           self.chip._stream()), "h2r_verify_emit_advice")
        return out

    def flatten(self, elem: int) -> "np.ndarray":
        s = self.layout.elem_stride
        host = np.ascontiguousarray(self.trace[elem * s:(elem + 1) * s].cpu().numpy())
        out = np.zeros(self.layout.stream_bytes, dtype=np.uint8)
        check(lib().h2r_verify_trace_flatten(self.chip._ctx, ctypes.byref(self.layout), host.ctypes.data, out.ctypes.data),
              "h2r_verify_trace_flatten")
        return out
