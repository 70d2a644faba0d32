"""Host-side mirror of the reference's `big_integer` module for the accelerated path.

Same names and argument meaning as `BigIntInstructions<F>` (reference src/big_integer/instructions.rs:7-260)
for the methods on the hot path -- `assign_integer`, `mul_mod`, `square_mod`, `pow_mod`,
`pow_mod_fixed_exp` -- in batch form: every integer argument is a batch of integers (one per
independent circuit), resident in HBM.  All arithmetic happens in libh2r.so (hand-written HIP);
this module only moves pointers.  There is no CPU fallback.
"""
import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import H2RLayout, H2RParams, H2RPowLayout, check, lib


def _e_bytes(e: int) -> bytes:
    """BigUint::to_bytes_le (reference big_integer/chip.rs:719-720)."""
    return int(e).to_bytes(max(1, (int(e).bit_length() + 7) // 8), "little")


@dataclass
class UnassignedInteger:
    """reference big_integer/mod.rs:270-302: limb values about to be assigned (host side)."""
    limbs: np.ndarray  # [batch, num_limbs]

    @property
    def num_limbs(self):
        return self.limbs.shape[1]

    @staticmethod
    def from_ints(values: Sequence[int], num_limbs: int, limb_width: int) -> "UnassignedInteger":
        """maingate::decompose_big (call sites examples/rsa_example.rs:177,182): little-endian split."""
        dt = np.uint64 if limb_width == 64 else np.uint32
        m = (1 << limb_width) - 1
        out = np.zeros((len(values), num_limbs), dtype=dt)
        for r, v in enumerate(values):
            v = int(v)
            if v >> (limb_width * num_limbs):
                raise ValueError("value does not fit %d limbs of %d bits" % (num_limbs, limb_width))
            for i in range(num_limbs):
                out[r, i] = (v >> (limb_width * i)) & m
        return UnassignedInteger(out)


class AssignedInteger:
    """reference big_integer/mod.rs:306-382 (Fresh range type): a batch of limb vectors in HBM."""

    def __init__(self, limbs: torch.Tensor, limb_width: int):
        assert limbs.is_cuda and limbs.dim() == 2 and limbs.is_contiguous()
        self.limbs_dev = limbs
        self.limb_width = limb_width

    def num_limbs(self):
        return self.limbs_dev.shape[1]

    @property
    def batch(self):
        return self.limbs_dev.shape[0]

    def limbs_host(self) -> np.ndarray:
        dt = np.uint64 if self.limb_width == 64 else np.uint32
        return self.limbs_dev.cpu().numpy().view(dt)

    def to_big_uint(self) -> List[int]:
        """reference big_integer/mod.rs:348-359."""
        h = self.limbs_host()
        return [sum(int(x) << (self.limb_width * i) for i, x in enumerate(row)) for row in h]

    def data_ptr(self):
        return self.limbs_dev.data_ptr()


class Trace:
    """Witness records of a batch call, resident in HBM, plus the layout needed to walk them."""

    def __init__(self, chip: "BigIntChip", buf: torch.Tensor, batch: int, pow_layout: Optional[H2RPowLayout]):
        self.chip, self.buf, self.batch, self.pow_layout = chip, buf, batch, pow_layout
        self.layout = chip.layout
        self.elem_stride = pow_layout.elem_stride if pow_layout is not None else chip.layout.record_stride
        self.num_mul_mods = pow_layout.num_mul_mods if pow_layout is not None else 1

    @property
    def stream_bytes(self):
        return self.pow_layout.stream_bytes if self.pow_layout is not None else self.layout.stream_bytes

    def elem_host(self, elem: int) -> np.ndarray:
        s = self.elem_stride
        return self.buf[elem * s:(elem + 1) * s].cpu().numpy()

    def stream_bytes_ex(self, flags: int = 0) -> int:
        if self.pow_layout is None:
            return int(lib().h2r_stream_bytes(self.chip._ctx, flags))
        return int(lib().h2r_pow_stream_bytes(self.chip._ctx, ctypes.byref(self.pow_layout), flags))

    def flatten(self, elem: int, flags: int = 0) -> np.ndarray:
        """The element's op-trace in the reference's assignment order (host walk of one element: h2r_trace_flatten_ex)."""
        host = np.ascontiguousarray(self.elem_host(elem))
        out = np.zeros(self.stream_bytes_ex(flags), dtype=np.uint8)
        if self.pow_layout is None:
            check(lib().h2r_trace_flatten_ex(self.chip._ctx, host.ctypes.data, flags, out.ctypes.data), "h2r_trace_flatten_ex")
        else:
            check(lib().h2r_pow_trace_flatten_ex(self.chip._ctx, ctypes.byref(self.pow_layout), host.ctypes.data, flags, out.ctypes.data),
                  "h2r_pow_trace_flatten_ex")
        return out

    def emit_stream(self, flags: int = 0, out: Optional[torch.Tensor] = None, out_stride: Optional[int] = None) -> torch.Tensor:
        """Device-side flatten of EVERY element (h2r_trace_emit_stream / h2r_pow_trace_emit_stream): uint8
        [batch, out_stride] in HBM whose first stream_bytes_ex(flags) bytes per row are the element's flat stream."""
        sb = self.stream_bytes_ex(flags)
        out_stride = sb if out_stride is None else out_stride
        if out is None:
            out = torch.empty((self.batch, out_stride), dtype=torch.uint8, device=self.buf.device)
        if self.pow_layout is None:
            check(lib().h2r_trace_emit_stream(self.chip._ctx, self.buf.data_ptr(), self.batch, flags, out.data_ptr(), out_stride, 0,
                                              self.chip._stream()), "h2r_trace_emit_stream")
        else:
            check(lib().h2r_pow_trace_emit_stream(self.chip._ctx, ctypes.byref(self.pow_layout), self.buf.data_ptr(), self.elem_stride,
                                                  self.batch, flags, out.data_ptr(), out_stride, 0, self.chip._stream()),
                  "h2r_pow_trace_emit_stream")
        return out

    def plane(self, elem: int, t: int, name: str) -> np.ndarray:
        """Raw bytes [count, elem_bytes] of one plane of mul_mod record t of element elem."""
        p = _lib.PLANES.index(name)
        if name in ("AB_LO", "AB_HI", "QN_LO", "QN_HI"):
            raise ValueError("the accumulator planes are interleaved (see include/h2r.h); use flatten()")
        lo = self.layout
        off0 = (self.pow_layout.off_records if self.pow_layout is not None else 0) + t * lo.record_stride
        start = elem * self.elem_stride + off0 + lo.plane_off[p]
        n = lo.plane_elem[p] * lo.plane_count[p]
        return self.buf[start:start + n].cpu().numpy().reshape(lo.plane_count[p], lo.plane_elem[p])

    def lookup_hist(self) -> torch.Tensor:
        """Multiplicity of every lookup-table row hit by this trace's range checks, per element."""
        hl = lib().h2r_hist_len(self.chip._ctx)
        hist = torch.zeros((self.batch, hl), dtype=torch.int32, device=self.buf.device)
        off = self.pow_layout.off_records if self.pow_layout is not None else 0
        check(lib().h2r_trace_lookup_hist(self.chip._ctx, self.buf.data_ptr(), off, self.elem_stride, self.batch,
                                          self.num_mul_mods, hist.data_ptr(), self.chip._stream()), "h2r_trace_lookup_hist")
        return hist


    def lookup_permutation(self, with_hist: bool = False):
        """(perm, rows): stable grouping of every element's lookup cells by table row (h2r_trace_lookup_permutation);
        with_hist=True: (perm, rows, hist) from ONE launch (h2r_trace_lookup_permutation_hist)."""
        n = lib().h2r_lookups_per_record(self.chip._ctx) * self.num_mul_mods
        perm = torch.empty((self.batch, n), dtype=torch.int32, device=self.buf.device)
        rows = torch.empty((self.batch, n), dtype=torch.int16, device=self.buf.device)
        off = self.pow_layout.off_records if self.pow_layout is not None else 0
        if with_hist:
            hist = torch.empty((self.batch, int(lib().h2r_hist_len(self.chip._ctx))), dtype=torch.int32, device=self.buf.device)
            check(lib().h2r_trace_lookup_permutation_hist(self.chip._ctx, self.buf.data_ptr(), off, self.elem_stride, self.batch,
                                                          self.num_mul_mods, perm.data_ptr(), rows.data_ptr(), hist.data_ptr(),
                                                          self.chip._stream()), "h2r_trace_lookup_permutation_hist")
            return perm, rows, hist
        check(lib().h2r_trace_lookup_permutation(self.chip._ctx, self.buf.data_ptr(), off, self.elem_stride, self.batch,
                                                 self.num_mul_mods, perm.data_ptr(), rows.data_ptr(), self.chip._stream()),
              "h2r_trace_lookup_permutation")
        return perm, rows


class InFieldTrace:
    """The assert_in_field(x, n) witness of RSAChip::modpow_public_key (src/chip.rs:106): per element the flat stream of
    the is_in_field Fresh op, sections 16-byte aligned in the device buffer (h2r_fresh_op_flatten packs them)."""

    def __init__(self, chip: "BigIntChip", buf: torch.Tensor, batch: int, elem_stride: int, stream_bytes: int):
        self.chip, self.buf, self.batch, self.elem_stride, self.stream_bytes = chip, buf, batch, elem_stride, stream_bytes

    def emit_advice(self, x: "AssignedInteger", n: "AssignedInteger") -> torch.Tensor:
        """assert_in_field(x, n) as advice rows (is_less_than's cells + the assert_one row): uint8 [batch, rows * 160]."""
        flags = _lib.H2R_F_SHARED_MODULUS if (n.batch == 1 and self.batch != 1) else 0
        return self.chip.fresh_op_emit_advice(_lib.FRESH_OPS.index("is_in_field"), x, n, None, flags, self.buf, 0, self.elem_stride,
                                              self.batch, None, assert_one=True)

    def flatten(self, elem: int) -> np.ndarray:
        host = np.ascontiguousarray(self.buf[elem * self.elem_stride:(elem + 1) * self.elem_stride].cpu().numpy())
        out = np.zeros(self.stream_bytes, dtype=np.uint8)
        check(lib().h2r_fresh_op_flatten(self.chip._ctx, _lib.FRESH_OPS.index("is_in_field"), host.ctypes.data, out.ctypes.data),
              "h2r_fresh_op_flatten")
        return out


@dataclass
class BatchResult:
    value: AssignedInteger      # a*b mod n  /  a^e mod n
    trace: Optional[Trace]
    status: torch.Tensor        # uint8 [batch], H2R_* per element
    in_field: Optional[InFieldTrace] = None   # modpow_public_key only: the assert_in_field witness
    workspace: Optional[torch.Tensor] = None   # pow calls: the operands buffer of every mul_mod (kept for audit())
    inputs: Optional[tuple] = None             # (kind, a/x, b, n, e bytes or None) the call was made with
    chip: Optional["BigIntChip"] = None        # (results without a trace: the chip and the pow layout of the call)
    pow_layout: Optional[H2RPowLayout] = None

    def audit(self):
        """In-place device check of every record of the trace (h2r_mul_mod_trace_check / h2r_pow_trace_check):
        returns (bad, first_bad) uint32 tensors [batch]; bad == 0 everywhere for a valid witness."""
        chip = self.trace.chip
        batch, dev = self.trace.batch, self.trace.buf.device
        bad = torch.empty(batch, dtype=torch.int32, device=dev)
        first = torch.empty(batch, dtype=torch.int32, device=dev)
        kind, a, b, n, eb = self.inputs
        flags = chip._flags(n, batch)
        if kind == "mul_mod":
            check(lib().h2r_mul_mod_trace_check(chip._ctx, a.data_ptr(), b.data_ptr(), n.data_ptr(), flags, self.trace.buf.data_ptr(), batch,
                                                self.status.data_ptr(), bad.data_ptr(), first.data_ptr(), chip._stream()), "h2r_mul_mod_trace_check")
        else:
            check(lib().h2r_pow_trace_check(chip._ctx, ctypes.byref(self.trace.pow_layout), a.data_ptr(), n.data_ptr(), eb,
                                            len(eb) if eb is not None else 0, flags, self.trace.buf.data_ptr(), self.trace.elem_stride,
                                            self.workspace.data_ptr(), batch, self.status.data_ptr(), bad.data_ptr(), first.data_ptr(),
                                            chip._stream()), "h2r_pow_trace_check")
        return bad, first

    def emit_advice(self, out: Optional[torch.Tensor] = None, direct: bool = False) -> torch.Tensor:
        """The 5-column advice image of every mul_mod record (h2r_*_emit_advice): uint8 [batch, T * rows * 160] in HBM,
        row = 5 cells of 32 bytes (canonical elements of the chip's field); row shapes in DESIGN.md section 2b.
        out (optional): a uint8 buffer of at least batch * T * rows * 160 bytes to write the image into (e.g. a region of the
        placement-aware arena: where the image lies physically decides the store rate, as for the trace).
        direct=True (H2R_ADVICE_DIRECT): the cells are recomputed from the operands, the records are not read -- a pow result
        made with want_trace=False (and a workspace) has only this form."""
        kind, a, b, n, eb = self.inputs
        chip = self.trace.chip if self.trace is not None else self.chip
        batch, dev = a.batch, a.limbs_dev.device
        rows = int(lib().h2r_advice_rows(chip._ctx))
        flags = chip._flags(n, batch) | (_lib.H2R_ADVICE_DIRECT if direct else 0)
        if self.trace is None and not (direct and kind != "mul_mod" and self.workspace is not None):
            raise ValueError("emit_advice: a result without records has only the direct form of a pow call's image (direct=True, workspace kept)")
        pl = self.trace.pow_layout if self.trace is not None else self.pow_layout
        T = pl.num_mul_mods if pl is not None else 1
        if kind != "mul_mod":   # a fixed-exponent pow element starts with the two constant rows of acc = 1 (h2r_pow_advice_rows)
            nrows = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
        else:
            nrows = T * rows
        ib = chip.image_bytes(nrows)
        if out is None:
            out = torch.empty((batch, ib), dtype=torch.uint8, device=dev)
        else:
            if out.dtype != torch.uint8 or out.numel() < batch * ib or not out.is_contiguous():
                raise ValueError("emit_advice: out must be a contiguous uint8 buffer of at least batch * rows * 160 bytes")
            out = out.view(-1)[:batch * ib].view(batch, ib)
        if kind == "mul_mod":
            check(lib().h2r_mul_mod_emit_advice(chip._ctx, a.data_ptr(), b.data_ptr(), n.data_ptr(), flags, self.trace.buf.data_ptr(), batch,
                                                self.status.data_ptr(), out.data_ptr(), out.shape[1], chip._stream()), "h2r_mul_mod_emit_advice")
        else:
            check(lib().h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pl), n.data_ptr(), flags,
                                                  self.trace.buf.data_ptr() if self.trace is not None else None,
                                                  self.trace.elem_stride if self.trace is not None else 0, self.workspace.data_ptr(), batch,
                                                  self.status.data_ptr(), out.data_ptr(), out.shape[1], chip._stream()), "h2r_pow_trace_emit_advice")
        return out

    def emit_modpow_advice(self, out: Optional[torch.Tensor] = None, direct: Optional[bool] = None) -> torch.Tensor:
        """One RSAChip::modpow_public_key element as advice rows (h2r_modpow_public_key_emit_advice): [assert_in_field(x, n)]
        [pow_mod_fixed_exp], uint8 [batch, rows * 160] in HBM.  A result made with want_trace=False (chain + in-field witness only)
        gets its pow rows written directly from the operands; direct=True asks for that with records present too."""
        kind, x, _, n, eb = self.inputs
        if kind == "mul_mod" or self.in_field is None or self.workspace is None:
            raise ValueError("emit_modpow_advice: a modpow_public_key result with its in-field witness and workspace")
        chip = self.trace.chip if self.trace is not None else self.chip
        pl = self.trace.pow_layout if self.trace is not None else self.pow_layout
        batch = x.batch
        if direct is None:
            direct = self.trace is None
        nrows = int(lib().h2r_modpow_public_key_advice_rows(chip._ctx, ctypes.byref(pl), None))
        ib = chip.image_bytes(nrows)
        if out is None:
            out = torch.empty((batch, ib), dtype=torch.uint8, device=x.limbs_dev.device)
        else:
            if out.dtype != torch.uint8 or out.numel() < batch * ib or not out.is_contiguous():
                raise ValueError("emit_modpow_advice: out must be a contiguous uint8 buffer of at least batch * rows * 160 bytes")
            out = out.view(-1)[:batch * ib].view(batch, ib)
        flags = chip._flags(n, batch) | (_lib.H2R_ADVICE_DIRECT if direct else 0)
        check(lib().h2r_modpow_public_key_emit_advice(chip._ctx, ctypes.byref(pl), x.data_ptr(), n.data_ptr(), flags, self.in_field.buf.data_ptr(),
                                                      self.trace.buf.data_ptr() if (self.trace is not None and not direct) else None,
                                                      self.workspace.data_ptr(), batch, self.status.data_ptr(), out.data_ptr(), out.shape[1],
                                                      chip._stream()), "h2r_modpow_public_key_emit_advice")
        return out

    def flatten(self, elem: int) -> np.ndarray:
        """The element's witness in the reference's assignment order (modpow_public_key: in-field stream, then pow)."""
        st = self.trace.flatten(elem)
        return np.concatenate([self.in_field.flatten(elem), st]) if self.in_field is not None else st


class BigIntChip:
    """reference big_integer/chip.rs:42-51, 1161-1249."""

    NUM_LOOKUP_LIMBS = 8  # big_integer/chip.rs:1163

    def __init__(self, limb_width: int, bits_len: int, field: str = "bn254_fr", device: int = 0, columns: bool = False,
                 montgomery: bool = False, col_stride: int = 0):
        """BigIntChip::new (big_integer/chip.rs:1174-1185); raises where the reference asserts.
        columns / montgomery / col_stride: the representation of the advice images and of every field element that crosses the
        boundary (h2r_advice_repr): planar column vectors instead of 160-byte rows, x * R mod p instead of canonical integers."""
        self.limb_width, self.bits_len, self.device = limb_width, bits_len, device
        self.columns, self.montgomery, self.col_stride = bool(columns), bool(montgomery), int(col_stride)
        self._ctx = ctypes.c_void_p()
        p = H2RParams(limb_width, bits_len, _lib.FIELDS[field], device)
        rp = _lib.H2RAdviceRepr(ctypes.sizeof(_lib.H2RAdviceRepr), (_lib.H2R_ADVICE_COLUMNS if columns else 0) |
                                (_lib.H2R_ADVICE_MONTGOMERY if montgomery else 0), col_stride)
        check(lib().h2r_ctx_create_ex(ctypes.byref(p), ctypes.byref(rp), ctypes.byref(self._ctx)), "BigIntChip::new")
        self.num_limbs = bits_len // limb_width
        self.layout = H2RLayout()
        check(lib().h2r_trace_layout(self._ctx, ctypes.byref(self.layout)), "h2r_trace_layout")
        self.torch_dtype = torch.int64 if limb_width == 64 else torch.int32
        self.np_dtype = np.uint64 if limb_width == 64 else np.uint32

    def __del__(self):
        try:
            if self._ctx:
                lib().h2r_ctx_destroy(self._ctx)
                self._ctx = ctypes.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def advice_check(self, kinds, image: torch.Tensor, batch: int, status: Optional[torch.Tensor] = None, copies=None, src_a=None,
                     src_b=None, src_n=None, lookup: Optional["LookupArgument"] = None, layout=None):
        """The device-side MockProver (h2r_advice_check): every row of every element image against the main-gate equation, the lookup
        table and the copy pairs.  kinds: uint8 row kinds (numpy or device tensor); copies: H2RCopy array (ctypes) or a device tensor of
        [n, 4] int32; returns (bad int32 [batch], first int64 [batch]); bad == 0 everywhere for a satisfying assignment."""
        dev = image.device
        kd = kinds if isinstance(kinds, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(kinds, dtype=np.uint8)).to(dev)
        rows = kd.numel()
        cd, n_copies = None, 0
        if copies is not None:
            if isinstance(copies, torch.Tensor):
                cd = copies
            else:
                cd = torch.from_numpy(np.frombuffer(copies, dtype=np.uint32).view(np.int32).reshape(-1, 4).copy()).to(dev)
            n_copies = cd.shape[0]
        bad = torch.empty(batch, dtype=torch.int32, device=dev)
        first = torch.empty(batch, dtype=torch.int64, device=dev)
        flags = _lib.H2R_F_SHARED_MODULUS if (src_n is not None and src_n.batch == 1 and batch != 1) else 0
        stride = image.shape[1] if image.dim() > 1 else image.numel() // batch
        check(lib().h2r_advice_check(self._ctx, ctypes.byref(lookup.cfg) if lookup is not None else None,
                                     ctypes.byref(layout) if layout is not None else None, kd.data_ptr(), rows, image.data_ptr(), stride, batch,
                                     status.data_ptr() if status is not None else None, cd.data_ptr() if cd is not None else None, n_copies,
                                     src_a.data_ptr() if src_a is not None else None, src_b.data_ptr() if src_b is not None else None,
                                     src_n.data_ptr() if src_n is not None else None, flags, bad.data_ptr(), first.data_ptr(), self._stream()),
              "h2r_advice_check")
        self._keep = (kd, cd)
        return bad, first

    def pow_copy_map(self, pl: H2RPowLayout, e: int, row_offset: int = 0):
        """h2r_pow_copy_map: the copy pairs of one fixed-exponent pow element as a ctypes array of H2RCopy."""
        eb = _e_bytes(e)
        n = int(lib().h2r_pow_copy_map(self._ctx, ctypes.byref(pl), eb, len(eb), row_offset, None, 0))
        if n == 0:
            check(_lib.H2R_E_UNSUPPORTED, "h2r_pow_copy_map")
        arr = (_lib.H2RCopy * n)()
        assert int(lib().h2r_pow_copy_map(self._ctx, ctypes.byref(pl), eb, len(eb), row_offset, arr, n)) == n
        return arr

    def image_bytes(self, rows: int) -> int:
        """Bytes of one element's advice image of `rows` rows in the chip's representation: rows * 160 (row-major, or planar with
        the five columns packed), 5 * col_stride for planar columns of a fixed stride."""
        if self.columns and self.col_stride:
            if rows * 32 > self.col_stride:
                raise ValueError("image_bytes: %d rows do not fit a column of %d bytes" % (rows, self.col_stride))
            return 5 * self.col_stride
        return rows * 160

    @staticmethod
    def compute_range_lens(limb_width: int, num_limbs: int):
        """big_integer/chip.rs:1220-1249 -> (composition_bit_lens, overflow_bit_lens)."""
        comp, over = (ctypes.c_uint32 * 3)(), (ctypes.c_uint32 * 3)()
        check(lib().h2r_compute_range_lens(limb_width, num_limbs, comp, over), "compute_range_lens")
        return list(comp), list(over)

    # ---- assignment ---------------------------------------------------------------------------------
    def assign_integer(self, integer) -> AssignedInteger:
        """big_integer/chip.rs:62-82: here = move the limbs into HBM (the range-check sub-limbs of
        assigned inputs are the limbs' own bytes)."""
        if isinstance(integer, AssignedInteger):
            return integer
        if isinstance(integer, UnassignedInteger):
            arr = integer.limbs
        else:
            arr = UnassignedInteger.from_ints(list(integer), self.num_limbs, self.limb_width).limbs
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64 if self.limb_width == 64 else np.int32))
        return AssignedInteger(t.to("cuda:%d" % self.device).contiguous(), self.limb_width)

    def assign_constant(self, value: int, num_limbs: int, batch: int = 1) -> AssignedInteger:
        """big_integer/chip.rs:1252-1281: ceil(bits / limb_width) limbs of the constant, then the shared zero cell up to
        num_limbs (asserts that the value fits); the same constant for every element of the batch."""
        value = int(value)
        assert value >= 0 and value >> (self.limb_width * num_limbs) == 0      # :1266
        arr = UnassignedInteger.from_ints([value] * batch, num_limbs, self.limb_width).limbs
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64 if self.limb_width == 64 else np.int32))
        return AssignedInteger(t.to("cuda:%d" % self.device).contiguous(), self.limb_width)

    def assign_constant_fresh(self, value: int, batch: int = 1) -> AssignedInteger:
        """big_integer/chip.rs:95-101 (instructions.rs:16): the constant as a num_limbs-limb Fresh integer."""
        return self.assign_constant(value, self.num_limbs, batch)

    def assign_constant_muled(self, value: int, num_limbs_l: int, num_limbs_r: int, batch: int = 1) -> "MuledResult":
        """big_integer/chip.rs:119-127 (instructions.rs:23): the constant as a Muled integer of n_l + n_r - 1 limbs of
        limb_width bits each (here n_l = n_r = num_limbs: the 2L-column Muled container of this chip)."""
        assert num_limbs_l == self.num_limbs and num_limbs_r == self.num_limbs
        nl = 2 * self.num_limbs - 1
        value = int(value)
        assert value >> (self.limb_width * nl) == 0
        cols = np.zeros((batch, 2 * self.num_limbs, 4), dtype=np.uint64)
        m = (1 << self.limb_width) - 1
        for i in range(nl):
            cols[:, i, 0] = (value >> (self.limb_width * i)) & m
        return MuledResult(torch.from_numpy(cols.view(np.int64)).to("cuda:%d" % self.device), None, self)

    def max_value(self, num_limbs: Optional[int] = None, batch: int = 1) -> AssignedInteger:
        """big_integer/chip.rs:138-154 (instructions.rs:32): every limb = 2^limb_width - 1."""
        nl = self.num_limbs if num_limbs is None else num_limbs
        return self.assign_constant((1 << (self.limb_width * nl)) - 1, nl, batch)

    # ---- assert_* (instructions.rs:197-254): the predicate, then main_gate.assert_one on its bit -- a violated assertion
    # makes the reference's circuit unsatisfiable; here the element's status becomes H2R_E_ASSERTION ------------------------
    def _assert(self, res: "FreshResult") -> "FreshResult":
        viol = (res.flag == 0) & (res.status == 0)
        res.status = torch.where(viol, torch.full_like(res.status, _lib.H2R_E_ASSERTION), res.status)
        return res

    def assert_zero(self, a):
        """big_integer/chip.rs:1020-1028."""
        return self._assert(self.is_zero(a))

    def assert_equal_fresh(self, a, b):
        """big_integer/chip.rs:1034-1042."""
        return self._assert(self.is_equal_fresh(a, b))

    def assert_equal_muled(self, a: "MuledResult", b: "MuledResult"):
        """big_integer/chip.rs:1053-1063 -> (status uint8[batch], trace)."""
        eq, trace = self.is_equal_muled(a, b)
        return torch.where(eq == 0, torch.full_like(eq, _lib.H2R_E_ASSERTION), torch.zeros_like(eq)), trace

    def assert_less_than(self, a, b):
        """big_integer/chip.rs:1074-1082."""
        return self._assert(self.is_less_than(a, b))

    def assert_less_than_or_equal(self, a, b):
        """big_integer/chip.rs:1093-1101."""
        return self._assert(self.is_less_than_or_equal(a, b))

    def assert_greater_than(self, a, b):
        """big_integer/chip.rs:1112-1120."""
        return self._assert(self.is_greater_than(a, b))

    def assert_greater_than_or_equal(self, a, b):
        """big_integer/chip.rs:1131-1139."""
        return self._assert(self.is_greater_than_or_equal(a, b))

    def assert_in_field(self, a, n):
        """big_integer/chip.rs:1150-1158."""
        return self._assert(self.is_in_field(a, n))

    def range_check_sublimbs(self, integer: AssignedInteger) -> torch.Tensor:
        """The RangeChip::assign(limb, limb_width/8, limb_width) decomposition that assign_integer performs for
        every limb of an input integer (big_integer/chip.rs:71-76): uint8 [batch, num_limbs, 8] sub-limbs."""
        sub = torch.empty((integer.batch, integer.num_limbs(), 8), dtype=torch.uint8, device=integer.limbs_dev.device)
        values = integer.limbs_dev
        if self.limb_width == 32:   # h2r_range_decompose_batch takes 8- or 16-byte values
            values = (values.to(torch.int64) & 0xffffffff).contiguous()
        check(lib().h2r_range_decompose_batch(self._ctx, values.data_ptr(), 8, values.numel(), self.limb_width,
                                              self.limb_width // 8, sub.data_ptr(), 8, None, self._stream()),
              "range_check_sublimbs")
        return sub

    def _new_limbs(self, batch):
        return torch.empty((batch, self.num_limbs), dtype=self.torch_dtype, device="cuda:%d" % self.device)

    def _flags(self, n: AssignedInteger, batch: int):
        if n.batch == 1 and batch != 1:
            return _lib.H2R_F_SHARED_MODULUS
        assert n.batch == batch
        return 0

    # ---- the hot path ---------------------------------------------------------------------------------
    def mul_mod(self, a: AssignedInteger, b: AssignedInteger, n: AssignedInteger, want_trace: bool = True) -> BatchResult:
        """big_integer/chip.rs:542-629."""
        assert a.num_limbs() == n.num_limbs() == self.num_limbs  # :555
        batch = a.batch
        dev = "cuda:%d" % self.device
        trace = torch.empty(batch * self.layout.record_stride, dtype=torch.uint8, device=dev) if want_trace else None
        r = self._new_limbs(batch)
        status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        check(lib().h2r_mul_mod_batch(self._ctx, a.data_ptr(), b.data_ptr(), n.data_ptr(), batch, self._flags(n, batch),
                                      trace.data_ptr() if want_trace else None, r.data_ptr(), status.data_ptr(), None,
                                      self._stream()), "mul_mod")
        return BatchResult(AssignedInteger(r, self.limb_width), Trace(self, trace, batch, None) if want_trace else None, status,
                           inputs=("mul_mod", a, b, n, None))

    def square_mod(self, a: AssignedInteger, n: AssignedInteger, want_trace: bool = True) -> BatchResult:
        """big_integer/chip.rs:642-649."""
        return self.mul_mod(a, a, n, want_trace)

    def pow_var_layout(self, e_num_limbs: int, exp_limb_bits: int) -> H2RPowLayout:
        pl = H2RPowLayout()
        check(lib().h2r_pow_var_layout(self._ctx, e_num_limbs, exp_limb_bits, ctypes.byref(pl)), "h2r_pow_var_layout")
        return pl

    def pow_fixed_layout(self, e: int) -> H2RPowLayout:
        pl = H2RPowLayout()
        eb = _e_bytes(e)
        check(lib().h2r_pow_fixed_layout(self._ctx, eb, len(eb), ctypes.byref(pl)), "h2r_pow_fixed_layout")
        return pl

    def in_field_layout(self):
        """(element stride, flat-stream bytes) of the assert_in_field witness = the is_in_field Fresh op."""
        es, sb = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib().h2r_fresh_op_layout(self._ctx, _lib.FRESH_OPS.index("is_in_field"), ctypes.byref(es), ctypes.byref(sb), None),
              "h2r_fresh_op_layout")
        return es.value, sb.value

    def pow_mod_fixed_exp(self, a: AssignedInteger, e: int, n: AssignedInteger, want_trace: bool = True,
                          trace_buf: Optional[torch.Tensor] = None, check_in_field: bool = False,
                          workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                          status: Optional[torch.Tensor] = None, in_field_buf: Optional[torch.Tensor] = None) -> BatchResult:
        """big_integer/chip.rs:710-742.  check_in_field=True: RSAChip::modpow_public_key (src/chip.rs:99-114) -- the
        assert_in_field witness goes to `in_field` of the result (allocated here unless in_field_buf is given)."""
        batch = a.batch
        dev = "cuda:%d" % self.device
        pl = self.pow_fixed_layout(e)
        eb = _e_bytes(e)
        if want_trace and trace_buf is None:
            trace_buf = torch.empty(batch * pl.elem_stride, dtype=torch.uint8, device=dev)
        out = self._new_limbs(batch) if out is None else out
        status = torch.zeros(batch, dtype=torch.uint8, device=dev) if status is None else status
        tp = trace_buf.data_ptr() if want_trace else None
        if workspace is None and want_trace:   # kept with the result: audit() reads every mul_mod's operands from it
            workspace = torch.empty(self.workspace_bytes(batch, pl.num_mul_mods), dtype=torch.uint8, device=dev)
        wp = workspace.data_ptr() if workspace is not None else None
        in_field = None
        if check_in_field:
            in_field = self._in_field_trace(batch, in_field_buf) if (want_trace or in_field_buf is not None or workspace is not None) else None
            check(lib().h2r_modpow_public_key_batch(self._ctx, a.data_ptr(), n.data_ptr(), eb, len(eb), batch, self._flags(n, batch), tp,
                                                    in_field.buf.data_ptr() if in_field is not None else None, out.data_ptr(),
                                                    status.data_ptr(), wp, self._stream()), "modpow_public_key")
        else:
            check(lib().h2r_pow_mod_fixed_exp_batch(self._ctx, a.data_ptr(), n.data_ptr(), eb, len(eb), batch, self._flags(n, batch), tp,
                                                    out.data_ptr(), status.data_ptr(), wp, self._stream()), "pow_mod_fixed_exp")
        return BatchResult(AssignedInteger(out, self.limb_width), Trace(self, trace_buf, batch, pl) if want_trace else None, status, in_field,
                           workspace, ("pow_fixed", a, None, n, eb), self, pl)

    def _in_field_trace(self, batch, buf=None) -> "InFieldTrace":
        es, sb = self.in_field_layout()
        if buf is None:
            buf = torch.zeros(batch * es, dtype=torch.uint8, device="cuda:%d" % self.device)
        assert buf.numel() >= batch * es
        return InFieldTrace(self, buf, batch, es, sb)

    def pow_mod(self, a: AssignedInteger, e: AssignedInteger, n: AssignedInteger, exp_limb_bits: int,
                want_trace: bool = True, check_in_field: bool = False) -> BatchResult:
        """big_integer/chip.rs:664-696: variable exponent, each e-limb decomposed into `exp_limb_bits` bits.
        check_in_field=True: RSAChip::modpow_public_key with RSAPubE::Var (src/chip.rs:106-110)."""
        batch = a.batch
        dev = "cuda:%d" % self.device
        pl = H2RPowLayout()
        check(lib().h2r_pow_var_layout(self._ctx, e.num_limbs(), exp_limb_bits, ctypes.byref(pl)), "h2r_pow_var_layout")
        trace = torch.empty(batch * pl.elem_stride, dtype=torch.uint8, device=dev) if want_trace else None
        out = self._new_limbs(batch)
        status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        tp = trace.data_ptr() if want_trace else None
        ws = torch.empty(self.workspace_bytes(batch, pl.num_mul_mods), dtype=torch.uint8, device=dev) if want_trace else None
        wp = ws.data_ptr() if ws is not None else None
        in_field = None
        if check_in_field:
            in_field = self._in_field_trace(batch) if want_trace else None
            check(lib().h2r_modpow_public_key_var_batch(self._ctx, a.data_ptr(), e.data_ptr(), e.num_limbs(), exp_limb_bits, n.data_ptr(),
                                                        batch, self._flags(n, batch), tp,
                                                        in_field.buf.data_ptr() if in_field is not None else None, out.data_ptr(),
                                                        status.data_ptr(), wp, self._stream()), "modpow_public_key")
        else:
            check(lib().h2r_pow_mod_batch(self._ctx, a.data_ptr(), e.data_ptr(), e.num_limbs(), exp_limb_bits, n.data_ptr(), batch,
                                          self._flags(n, batch), tp, out.data_ptr(), status.data_ptr(), wp, self._stream()), "pow_mod")
        return BatchResult(AssignedInteger(out, self.limb_width), Trace(self, trace, batch, pl) if want_trace else None, status, in_field,
                           ws, ("pow_var", a, None, n, None))

    # ---- the Fresh-integer family (add / sub / add_mod / sub_mod / comparisons) -----------------------
    def _fresh_op(self, name: str, a: AssignedInteger, b: Optional[AssignedInteger], n: Optional[AssignedInteger]) -> "FreshResult":
        op = _lib.FRESH_OPS.index(name)
        es, sb, vl = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
        check(lib().h2r_fresh_op_layout(self._ctx, op, ctypes.byref(es), ctypes.byref(sb), ctypes.byref(vl)), name)
        batch, dev = a.batch, a.limbs_dev.device
        trace = torch.zeros(batch * es.value, dtype=torch.uint8, device=dev)
        value = torch.zeros((batch, vl.value), dtype=self.torch_dtype, device=dev) if vl.value else None
        flag = torch.zeros(batch, dtype=torch.uint8, device=dev)
        status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        flags = self._flags(n, batch) if n is not None else 0
        if b is not None and b.batch != batch:
            # one `b` for the whole batch (the modulus of is_in_field, a shared comparand): only the ops without an `n`
            if b.batch == 1 and n is None:
                flags |= _lib.H2R_F_SHARED_MODULUS
            else:
                check(_lib.H2R_E_SHAPE, name + ": operand batches differ")
        check(lib().h2r_fresh_op_batch(self._ctx, op, a.data_ptr(), b.data_ptr() if b is not None else None,
                                       n.data_ptr() if n is not None else None, batch, flags, trace.data_ptr(),
                                       value.data_ptr() if value is not None else None, flag.data_ptr(), status.data_ptr(),
                                       self._stream()), name)
        return FreshResult(AssignedInteger(value, self.limb_width) if value is not None else None, flag, status, trace,
                           es.value, sb.value, op, self, (a, b, n, flags))

    def fresh_op_emit_advice(self, op: int, a, b, n, flags: int, trace: torch.Tensor, first_off: int, elem_stride: int, batch: int,
                             status: Optional[torch.Tensor] = None, assert_one: bool = False) -> torch.Tensor:
        """h2r_fresh_op_emit_advice: the rows of a Fresh-integer op as a 5-column advice image, uint8 [batch, rows * 160] in HBM."""
        fl = flags | (_lib.H2R_ADVICE_ASSERT_ONE if assert_one else 0)
        rows = int(lib().h2r_fresh_op_advice_rows(self._ctx, op, fl))
        if rows == 0:
            check(_lib.H2R_E_UNSUPPORTED, "h2r_fresh_op_advice_rows")
        out = torch.empty((batch, self.image_bytes(rows)), dtype=torch.uint8, device=trace.device)
        check(lib().h2r_fresh_op_emit_advice(self._ctx, op, fl, a.data_ptr(), b.data_ptr() if b is not None else None,
                                             n.data_ptr() if n is not None else None, trace.data_ptr(), first_off, elem_stride, batch,
                                             status.data_ptr() if status is not None else None, out.data_ptr(), out.shape[1],
                                             self._stream()), "h2r_fresh_op_emit_advice")
        return out

    def fresh_op_row_kinds(self, op: int, assert_one: bool = False) -> np.ndarray:
        fl = _lib.H2R_ADVICE_ASSERT_ONE if assert_one else 0
        kinds = np.zeros(int(lib().h2r_fresh_op_advice_rows(self._ctx, op, fl)), dtype=np.uint8)
        check(lib().h2r_fresh_op_row_kinds(self._ctx, op, fl, kinds.ctypes.data), "h2r_fresh_op_row_kinds")
        return kinds

    def add(self, a, b):
        """big_integer/chip.rs:245-297 -> num_limbs + 1 limbs."""
        return self._fresh_op("add", a, b, None)

    def sub(self, a, b):
        """big_integer/chip.rs:310-373 -> (|a - b|, flag = is_overflowed)."""
        return self._fresh_op("sub", a, b, None)

    def add_mod(self, a, b, n):
        """big_integer/chip.rs:452-481."""
        return self._fresh_op("add_mod", a, b, n)

    def sub_mod(self, a, b, n):
        """big_integer/chip.rs:495-528."""
        return self._fresh_op("sub_mod", a, b, n)

    def is_zero(self, a):
        """big_integer/chip.rs:754-767."""
        return self._fresh_op("is_zero", a, None, None)

    def is_equal_fresh(self, a, b):
        """big_integer/chip.rs:780-805."""
        return self._fresh_op("is_equal_fresh", a, b, None)

    def is_less_than(self, a, b):
        """big_integer/chip.rs:908-919."""
        return self._fresh_op("is_less_than", a, b, None)

    def is_less_than_or_equal(self, a, b):
        """big_integer/chip.rs:932-941."""
        return self._fresh_op("is_less_than_or_equal", a, b, None)

    def is_greater_than(self, a, b):
        """big_integer/chip.rs:954-963."""
        return self._fresh_op("is_greater_than", a, b, None)

    def is_greater_than_or_equal(self, a, b):
        """big_integer/chip.rs:976-985."""
        return self._fresh_op("is_greater_than_or_equal", a, b, None)

    def is_in_field(self, a, n):
        """big_integer/chip.rs:998-1006."""
        return self._fresh_op("is_in_field", a, n, None)

    # ---- Muled integers: mul / square / is_equal_muled / refresh ----------------------------------------
    def mul(self, a: AssignedInteger, b: AssignedInteger) -> "MuledResult":
        """big_integer/chip.rs:386-419 -> AssignedInteger<Muled> (2L-1 un-carried columns) + the accumulator trace."""
        batch, dev = a.batch, a.limbs_dev.device
        trace = torch.empty(batch * self.layout.record_stride, dtype=torch.uint8, device=dev)
        cols = torch.zeros((batch, 2 * self.num_limbs, 4), dtype=torch.int64, device=dev)
        check(lib().h2r_mul_batch(self._ctx, a.data_ptr(), b.data_ptr(), batch, trace.data_ptr(), cols.data_ptr(), self._stream()), "mul")
        return MuledResult(cols, trace, self)

    def square(self, a: AssignedInteger) -> "MuledResult":
        """big_integer/chip.rs:431-437."""
        return self.mul(a, a)

    def is_equal_muled(self, a: "MuledResult", b: "MuledResult"):
        """big_integer/chip.rs:822-895 (num_limbs_l = num_limbs_r = num_limbs) -> (eq bits uint8[batch], trace tensor)."""
        batch, dev = a.cols.shape[0], a.cols.device
        trace = torch.empty(batch * self.layout.record_stride, dtype=torch.uint8, device=dev)
        eq = torch.zeros(batch, dtype=torch.uint8, device=dev)
        check(lib().h2r_is_equal_muled_batch(self._ctx, a.cols.data_ptr(), b.cols.data_ptr(), batch, trace.data_ptr(), eq.data_ptr(),
                                             self._stream()), "is_equal_muled")
        return eq, trace

    def flatten_is_equal_muled(self, trace: torch.Tensor, elem: int) -> np.ndarray:
        rs = self.layout.record_stride
        host = np.ascontiguousarray(trace[elem * rs:(elem + 1) * rs].cpu().numpy())
        out = np.zeros(int(lib().h2r_is_equal_muled_stream_bytes(self._ctx)), dtype=np.uint8)
        check(lib().h2r_is_equal_muled_flatten(self._ctx, host.ctypes.data, out.ctypes.data), "h2r_is_equal_muled_flatten")
        return out

    # ---- any operand shape (d0 != d1, RefreshAux::new(w, n_l, n_r), n_l != n_r): h2r_*_ex ----------------------------------
    def mul_ex(self, a: AssignedInteger, b: AssignedInteger) -> "MuledResult":
        """BigIntChip::mul for operands of d0 = a.num_limbs() and d1 = b.num_limbs() limbs (both <= num_limbs)."""
        batch, dev = a.batch, a.limbs_dev.device
        lo = self.layout
        trace = torch.zeros(batch * lo.record_stride, dtype=torch.uint8, device=dev)
        cols = torch.zeros((batch, 2 * self.num_limbs, 4), dtype=torch.int64, device=dev)
        check(lib().h2r_mul_batch_ex(self._ctx, a.data_ptr(), a.num_limbs(), b.data_ptr(), b.num_limbs(), batch, trace.data_ptr(),
                                     cols.data_ptr(), self._stream()), "h2r_mul_batch_ex")
        r = MuledResult(cols, trace, self)
        r.shape = (a.num_limbs(), b.num_limbs())
        return r

    def mul_ex_flatten(self, m: "MuledResult", elem: int) -> np.ndarray:
        d0, d1 = m.shape
        rs = self.layout.record_stride
        host = np.ascontiguousarray(m.trace[elem * rs:(elem + 1) * rs].cpu().numpy())
        out = np.zeros(int(lib().h2r_mul_stream_bytes_ex(self._ctx, d0, d1)), dtype=np.uint8)
        check(lib().h2r_mul_trace_flatten_ex(self._ctx, host.ctypes.data, d0, d1, out.ctypes.data), "h2r_mul_trace_flatten_ex")
        return out

    def refresh_ex(self, cols: torch.Tensor, n_l: int, n_r: int):
        """BigIntChip::refresh with RefreshAux::new(limb_width, n_l, n_r): (Fresh limbs [batch, nf], streams [batch, stride], status)."""
        nf, sb, es = ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_uint64()
        check(lib().h2r_refresh_layout(self._ctx, n_l, n_r, ctypes.byref(nf), ctypes.byref(sb), ctypes.byref(es)), "h2r_refresh_layout")
        batch, dev = cols.shape[0], cols.device
        trace = torch.zeros((batch, es.value), dtype=torch.uint8, device=dev)
        fresh = torch.zeros((batch, nf.value), dtype=self.torch_dtype, device=dev)
        status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        check(lib().h2r_refresh_batch_ex(self._ctx, cols.data_ptr(), cols.shape[1], n_l, n_r, batch, trace.data_ptr(), fresh.data_ptr(),
                                         status.data_ptr(), self._stream()), "h2r_refresh_batch_ex")
        return AssignedInteger(fresh, self.limb_width), trace[:, :sb.value], status

    def is_equal_muled_ex(self, a_cols: torch.Tensor, b_cols: torch.Tensor, n_l: int, n_r: int, flags: int = 0):
        """BigIntChip::is_equal_muled for n_l + n_r - 1 columns: (flat streams [batch, stream_bytes], eq bits)."""
        sb = int(lib().h2r_is_equal_muled_stream_bytes_ex(self._ctx, n_l, n_r, flags))
        if sb == 0:
            check(_lib.H2R_E_SHAPE, "is_equal_muled_ex")
        stride = (sb + 15) // 16 * 16
        batch, dev = a_cols.shape[0], a_cols.device
        out = torch.zeros((batch, stride), dtype=torch.uint8, device=dev)
        eq = torch.zeros(batch, dtype=torch.uint8, device=dev)
        check(lib().h2r_is_equal_muled_batch_ex(self._ctx, a_cols.data_ptr(), b_cols.data_ptr(), a_cols.shape[1], n_l, n_r, batch, flags,
                                                out.data_ptr(), stride, eq.data_ptr(), self._stream()), "h2r_is_equal_muled_batch_ex")
        return out[:, :sb], eq

    def refresh(self, a: "MuledResult"):
        """big_integer/chip.rs:168-233 with RefreshAux::new(limb_width, L, L) -> (Fresh 2L limbs, stream tensor, status)."""
        batch, dev = a.cols.shape[0], a.cols.device
        sb = int(lib().h2r_refresh_stream_bytes(self._ctx))
        stride = (sb + 255) // 256 * 256
        trace = torch.zeros(batch * stride, dtype=torch.uint8, device=dev)
        fresh = torch.zeros((batch, 2 * self.num_limbs), dtype=self.torch_dtype, device=dev)
        status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        check(lib().h2r_refresh_batch(self._ctx, a.cols.data_ptr(), batch, trace.data_ptr(), fresh.data_ptr(), status.data_ptr(),
                                      self._stream()), "refresh")
        return AssignedInteger(fresh, self.limb_width), trace.view(batch, stride)[:, :sb], status

    def pipeline(self) -> "Pipeline":
        """Opt-in two-stream pipeline (h2r_pipeline_*): consecutive modpow batches overlap chain and trace."""
        return Pipeline(self)

    def workspace_bytes(self, batch: int, num_mul_mods: int) -> int:
        return int(lib().h2r_workspace_bytes(self._ctx, batch, num_mul_mods))


class Pipeline:
    """h2r_pipeline: batch k+1's off-circuit chain overlaps batch k's record emission (side HIP streams).
    Callers rotate through `depth` buffer sets and call join() before reading the last traces."""

    def __init__(self, chip: BigIntChip, depth: int = 2, side_streams: int = 1):
        self.chip = chip
        self.depth = depth
        self._p = ctypes.c_void_p()
        check(lib().h2r_pipeline_create_ex(chip._ctx, depth, side_streams, ctypes.byref(self._p)), "h2r_pipeline_create_ex")

    def modpow_public_key(self, x: AssignedInteger, e: int, n: AssignedInteger, trace_buf, workspace, out, status, in_field_buf=None):
        """in_field_buf (optional): batch * in_field_layout()[0] bytes for the assert_in_field witness."""
        eb = _e_bytes(e)
        check(lib().h2r_pipeline_modpow_public_key(self._p, x.data_ptr(), n.data_ptr(), eb, len(eb), x.batch,
                                                   self.chip._flags(n, x.batch), trace_buf.data_ptr(),
                                                   in_field_buf.data_ptr() if in_field_buf is not None else None, out.data_ptr(),
                                                   status.data_ptr(), workspace.data_ptr(), self.chip._stream()),
              "h2r_pipeline_modpow_public_key")

    def modpow_public_key_advice(self, x: AssignedInteger, e: int, n: AssignedInteger, workspace, out, status, in_field_buf, advice_out):
        """The call WITHOUT records whose product is the element's advice image (h2r_pipeline_modpow_public_key_advice): chains, in-field
        witness and in-field rows on the current stream, the pow rows (cells_kernel) on the pipeline's side stream next to the following
        call's chains.  advice_out: uint8 [batch, rows * 160], complete after depth - 1 further calls or join()."""
        eb = _e_bytes(e)
        check(lib().h2r_pipeline_modpow_public_key_advice(self._p, x.data_ptr(), n.data_ptr(), eb, len(eb), x.batch, self.chip._flags(n, x.batch),
                                                          in_field_buf.data_ptr(), out.data_ptr(), status.data_ptr(), workspace.data_ptr(),
                                                          advice_out.data_ptr(), advice_out.shape[-1] if advice_out.dim() > 1 else advice_out.numel() // x.batch,
                                                          self.chip._stream()), "h2r_pipeline_modpow_public_key_advice")

    def modpow_public_key_var(self, x: AssignedInteger, e: AssignedInteger, exp_limb_bits: int, n: AssignedInteger, trace_buf, workspace,
                              out, status, in_field_buf=None):
        """RSAPubE::Var: per-element exponents `e` ([batch, e_num_limbs] limbs); buffers sized by pow_var_layout."""
        check(lib().h2r_pipeline_modpow_public_key_var(self._p, x.data_ptr(), e.data_ptr(), e.num_limbs(), exp_limb_bits, n.data_ptr(), x.batch,
                                                       self.chip._flags(n, x.batch), trace_buf.data_ptr(),
                                                       in_field_buf.data_ptr() if in_field_buf is not None else None, out.data_ptr(),
                                                       status.data_ptr(), workspace.data_ptr(), self.chip._stream()),
              "h2r_pipeline_modpow_public_key_var")

    def verify_pkcs1v15(self, sig: AssignedInteger, e: int, n: AssignedInteger, hashed, trace_buf, workspace, powed, is_valid, status):
        """Pipelined RSAInstructions::verify_pkcs1v15_signature (after the SHA step); `hashed`: int64 [batch, 4] on the
        device, `trace_buf` sized batch * h2r_verify_layout.elem_stride."""
        eb = _e_bytes(e)
        check(lib().h2r_pipeline_verify_pkcs1v15(self._p, sig.data_ptr(), n.data_ptr(), eb, len(eb), hashed.data_ptr(), sig.batch,
                                                 self.chip._flags(n, sig.batch), trace_buf.data_ptr(), powed.data_ptr(),
                                                 is_valid.data_ptr(), status.data_ptr(), workspace.data_ptr(), self.chip._stream()),
              "h2r_pipeline_verify_pkcs1v15")

    def modpow_public_key_var_advice(self, x: AssignedInteger, e: AssignedInteger, exp_limb_bits: int, n: AssignedInteger, workspace, out, status,
                                     in_field_buf, witness, advice_out):
        """RSAPubE::Var without records (h2r_pipeline_modpow_public_key_var_advice): `witness` uint8 [batch, pow_var_compact_layout().elem_stride]
        keeps the exponent bits, the selected operands and the result; advice_out as for modpow_public_key_advice."""
        check(lib().h2r_pipeline_modpow_public_key_var_advice(self._p, x.data_ptr(), e.data_ptr(), e.num_limbs(), exp_limb_bits, n.data_ptr(), x.batch,
                                                              self.chip._flags(n, x.batch), in_field_buf.data_ptr(), witness.data_ptr(), out.data_ptr(),
                                                              status.data_ptr(), workspace.data_ptr(), advice_out.data_ptr(),
                                                              advice_out.shape[-1] if advice_out.dim() > 1 else advice_out.numel() // x.batch,
                                                              self.chip._stream()), "h2r_pipeline_modpow_public_key_var_advice")

    def pow_var_compact_layout(self, e_num_limbs: int, exp_limb_bits: int) -> H2RPowLayout:
        """h2r_pow_layout_compact of the Var pow layout: elem_stride = witness bytes per element."""
        full, pl = H2RPowLayout(), H2RPowLayout()
        check(lib().h2r_pow_var_layout(self.chip._ctx, e_num_limbs, exp_limb_bits, ctypes.byref(full)), "h2r_pow_var_layout")
        check(lib().h2r_pow_layout_compact(self.chip._ctx, ctypes.byref(full), ctypes.byref(pl)), "h2r_pow_layout_compact")
        return pl

    def verify_pkcs1v15_advice(self, sig: AssignedInteger, e: int, n: AssignedInteger, hashed, witness, workspace, powed, is_valid, status, advice_out):
        """The whole verify_pkcs1v15_signature element as advice rows WITHOUT records (h2r_pipeline_verify_pkcs1v15_advice): chains, the
        in-field / encoded-message witness, is_valid and the three short row programs on the current stream, the pow rows (cells_kernel) on
        the pipeline's side stream next to the following call's chains.  `witness`: uint8 [batch, verify_witness_stride(e)];
        advice_out: uint8 [batch, image bytes of h2r_verify_advice_rows], complete after depth - 1 further calls or join()."""
        eb = _e_bytes(e)
        check(lib().h2r_pipeline_verify_pkcs1v15_advice(self._p, sig.data_ptr(), n.data_ptr(), eb, len(eb), hashed.data_ptr(), sig.batch,
                                                        self.chip._flags(n, sig.batch), witness.data_ptr(), powed.data_ptr(), is_valid.data_ptr(),
                                                        status.data_ptr(), workspace.data_ptr(), advice_out.data_ptr(),
                                                        advice_out.shape[-1] if advice_out.dim() > 1 else advice_out.numel() // sig.batch,
                                                        self.chip._stream()), "h2r_pipeline_verify_pkcs1v15_advice")

    def verify_pkcs1v15_var_advice(self, sig: AssignedInteger, e: AssignedInteger, exp_limb_bits: int, n: AssignedInteger, hashed, witness, workspace, powed,
                                   is_valid, status, advice_out):
        """The RSAPubE::Var arm of verify_pkcs1v15_advice (h2r_pipeline_verify_pkcs1v15_var_advice); `witness` sized by verify_compact_layout_var."""
        check(lib().h2r_pipeline_verify_pkcs1v15_var_advice(self._p, sig.data_ptr(), n.data_ptr(), e.data_ptr(), e.num_limbs(), exp_limb_bits, hashed.data_ptr(),
                                                            sig.batch, self.chip._flags(n, sig.batch), witness.data_ptr(), powed.data_ptr(), is_valid.data_ptr(),
                                                            status.data_ptr(), workspace.data_ptr(), advice_out.data_ptr(),
                                                            advice_out.shape[-1] if advice_out.dim() > 1 else advice_out.numel() // sig.batch,
                                                            self.chip._stream()), "h2r_pipeline_verify_pkcs1v15_var_advice")

    def verify_compact_layout_var(self, e_num_limbs: int, exp_limb_bits: int):
        from ._lib import H2RVerifyLayout
        full, vl = H2RVerifyLayout(), H2RVerifyLayout()
        check(lib().h2r_verify_layout_var(self.chip._ctx, e_num_limbs, exp_limb_bits, ctypes.byref(full)), "h2r_verify_layout_var")
        check(lib().h2r_verify_layout_compact(self.chip._ctx, ctypes.byref(full), ctypes.byref(vl)), "h2r_verify_layout_compact")
        return vl

    def verify_compact_layout(self, e: int):
        """h2r_verify_layout_compact of the fixed exponent's verify layout: elem_stride = bytes of witness per element; rows via
        h2r_verify_advice_rows."""
        from ._lib import H2RVerifyLayout
        eb = _e_bytes(e)
        full, vl = H2RVerifyLayout(), H2RVerifyLayout()
        check(lib().h2r_verify_layout_fixed(self.chip._ctx, eb, len(eb), ctypes.byref(full)), "h2r_verify_layout_fixed")
        check(lib().h2r_verify_layout_compact(self.chip._ctx, ctypes.byref(full), ctypes.byref(vl)), "h2r_verify_layout_compact")
        return vl

    def signature_verifier(self, msgs, msg_off, fixed_len: int, sig: AssignedInteger, e: int, n: AssignedInteger, trace_buf, workspace, powed,
                           is_valid, status, hashed, digest=None, hm_trace=None):
        """Pipelined RSASignatureVerifier::verify_pkcs1v15_signature from message bytes (src/lib.rs:183-246): `msgs` uint8 on the device,
        `msg_off` int64 [batch + 1] (or None: every message has fixed_len bytes); the SHA-256 / hashed-message step rides on the call's
        step launch.  `hashed`: int64 [batch, 4] output (the verifier's operand), digest uint8 [batch, 32] / hm_trace uint8 [batch, 288] optional."""
        eb = _e_bytes(e)
        check(lib().h2r_pipeline_signature_verifier(self._p, msgs.data_ptr(), msg_off.data_ptr() if msg_off is not None else None, fixed_len,
                                                    sig.data_ptr(), n.data_ptr(), eb, len(eb), sig.batch, self.chip._flags(n, sig.batch),
                                                    trace_buf.data_ptr(), hm_trace.data_ptr() if hm_trace is not None else None,
                                                    hm_trace.shape[1] if hm_trace is not None else 0,
                                                    digest.data_ptr() if digest is not None else None, hashed.data_ptr(), powed.data_ptr(),
                                                    is_valid.data_ptr(), status.data_ptr(), workspace.data_ptr(), self.chip._stream()),
              "h2r_pipeline_signature_verifier")

    def verify_pkcs1v15_var(self, sig: AssignedInteger, e: AssignedInteger, exp_limb_bits: int, n: AssignedInteger, hashed, trace_buf, workspace,
                            powed, is_valid, status):
        """The RSAPubE::Var arm (src/chip.rs:108-110); `trace_buf` sized batch * h2r_verify_layout_var's elem_stride."""
        check(lib().h2r_pipeline_verify_pkcs1v15_var(self._p, sig.data_ptr(), n.data_ptr(), e.data_ptr(), e.num_limbs(), exp_limb_bits,
                                                     hashed.data_ptr(), sig.batch, self.chip._flags(n, sig.batch), trace_buf.data_ptr(),
                                                     powed.data_ptr(), is_valid.data_ptr(), status.data_ptr(), workspace.data_ptr(),
                                                     self.chip._stream()), "h2r_pipeline_verify_pkcs1v15_var")

    def info(self, batch: int) -> "_lib.H2RPipelineInfo":
        """h2r_pipeline_info: the form a call of `batch` elements on the current stream takes (runs the hardware-queue probe if needed)."""
        out = _lib.H2RPipelineInfo(ctypes.sizeof(_lib.H2RPipelineInfo))
        check(lib().h2r_pipeline_info(self._p, self.chip._stream(), batch, ctypes.byref(out)), "h2r_pipeline_info")
        return out

    def join(self):
        check(lib().h2r_pipeline_join(self._p, self.chip._stream()), "h2r_pipeline_join")

    def close(self):
        if self._p:
            lib().h2r_pipeline_destroy(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TraceArena:
    """h2r_arena: `regions` trace regions of batch * elem_stride bytes each, the fastest of `candidates` mapped and measured
    ones (where a trace buffer lies physically decides how fast the record kernel writes it).
    .regions: uint8 tensors over the kept regions, fastest first (views of the arena's memory: drop them before close());
    .region_ms / .measurements_ms: record-kernel times."""

    class _Raw:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def __init__(self, chip: BigIntChip, elem_stride: int, first_record_off: int, records_per_elem: int, batch: int,
                 regions: int = 2, candidates: int = 16, max_look_bytes: int = 0):
        self.chip = chip
        self._a = ctypes.c_void_p()
        check(lib().h2r_arena_create_ex(chip._ctx, elem_stride, first_record_off, records_per_elem, batch, regions, candidates, max_look_bytes,
                                        chip._stream(), ctypes.byref(self._a)), "h2r_arena_create_ex")
        nbytes = int(lib().h2r_arena_region_bytes(self._a))
        dev = "cuda:%d" % chip.device
        self.regions = [torch.as_tensor(TraceArena._Raw(int(lib().h2r_arena_region(self._a, i)), nbytes), device=dev) for i in range(regions)]
        self.region_ms = [float(lib().h2r_arena_region_ms(self._a, i)) for i in range(regions)]
        buf = (ctypes.c_double * (12 * candidates))()   # (further candidates while the kept regions are not of one class, a second round when none stands out)
        n = int(lib().h2r_arena_measurements(self._a, buf, 12 * candidates))
        self.measurements_ms = [float(buf[i]) for i in range(n)]

    @classmethod
    def for_images(cls, chip: BigIntChip, region_bytes: int, regions: int = 2, candidates: int = 8, max_look_bytes: int = 0) -> "TraceArena":
        """h2r_image_arena_create: the same look for any large output buffer (advice images, lookup columns), timed with a streaming fill."""
        self = cls.__new__(cls)
        self.chip = chip
        self._a = ctypes.c_void_p()
        check(lib().h2r_image_arena_create(chip._ctx, region_bytes, regions, candidates, max_look_bytes, chip._stream(), ctypes.byref(self._a)),
              "h2r_image_arena_create")
        nbytes = int(lib().h2r_arena_region_bytes(self._a))
        dev = "cuda:%d" % chip.device
        self.regions = [torch.as_tensor(TraceArena._Raw(int(lib().h2r_arena_region(self._a, i)), nbytes), device=dev) for i in range(regions)]
        self.region_ms = [float(lib().h2r_arena_region_ms(self._a, i)) for i in range(regions)]
        buf = (ctypes.c_double * (12 * candidates))()
        n = int(lib().h2r_arena_measurements(self._a, buf, 12 * candidates))
        self.measurements_ms = [float(buf[i]) for i in range(n)]
        return self

    @classmethod
    def for_pow(cls, chip: BigIntChip, e: int, batch: int, regions: int = 2, candidates: int = 16) -> "TraceArena":
        pl = chip.pow_fixed_layout(e)
        return cls(chip, pl.elem_stride, pl.off_records, pl.num_mul_mods, batch, regions, candidates)

    def close(self):
        if self._a:
            self.regions = []
            lib().h2r_arena_destroy(self._a)
            self._a = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LookupArgument:
    """halo2's lookup argument for the range checks (h2r_lookup_*): RangeChip's (tag, value) table, the per-argument
    multiplicities of a batch of circuits and the permuted columns A' / S' for per-circuit challenges theta.
    Third-party behaviour restated in DESIGN.md section 2c; five arguments: composition_a..d, overflow_a."""

    ARGS = 5

    def __init__(self, chip: BigIntChip, rsa_chip: bool = True, bit_lens: Optional[Sequence[int]] = None,
                 tags: Optional[Sequence[int]] = None):
        self.chip = chip
        self.cfg = _lib.H2RLookupConfig()
        if bit_lens is None:
            check(lib().h2r_lookup_config_default(chip._ctx, 1 if rsa_chip else 0, ctypes.byref(self.cfg)), "h2r_lookup_config_default")
        else:
            n = len(bit_lens)
            check(lib().h2r_lookup_config_custom((ctypes.c_uint32 * n)(*bit_lens), (ctypes.c_uint32 * n)(*tags), n, ctypes.byref(self.cfg)),
                  "h2r_lookup_config_custom")
        self.n_rows = int(self.cfg.n_rows)

    def table_image(self):
        """[(tag, value)] as load_table writes them (canonical field elements; small integers)."""
        t = np.zeros((self.n_rows, 4), dtype=np.uint64)
        v = np.zeros((self.n_rows, 4), dtype=np.uint64)
        check(lib().h2r_lookup_table_image(self.chip._ctx, ctypes.byref(self.cfg), t.ctypes.data, v.ctypes.data), "h2r_lookup_table_image")
        return [(int(a[0]), int(b[0])) for a, b in zip(t, v)]

    def new_hist(self, batch: int) -> torch.Tensor:
        return torch.zeros((batch, self.ARGS, self.n_rows), dtype=torch.int32, device="cuda:%d" % self.chip.device)

    def hist_records(self, trace: Trace, hist: torch.Tensor, status: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Adds the lookups of every mul_mod record of every element (q / r limbs, range-assigned carries)."""
        off = trace.pow_layout.off_records if trace.pow_layout is not None else 0
        check(lib().h2r_lookup_hist_records(self.chip._ctx, ctypes.byref(self.cfg), trace.buf.data_ptr(), off, trace.elem_stride, trace.batch,
                                            trace.num_mul_mods, status.data_ptr() if status is not None else None, hist.data_ptr(),
                                            self.chip._stream()), "h2r_lookup_hist_records")
        return hist

    def hist_values(self, values: torch.Tensor, bit_len: int, sublimb_bits: int, hist: torch.Tensor) -> torch.Tensor:
        """Adds RangeChip::assign(v, sublimb_bits, bit_len) of values[elem][:] (int32 / int64 elements)."""
        assert values.dim() == 2 and values.is_contiguous()
        check(lib().h2r_lookup_hist_values(self.chip._ctx, ctypes.byref(self.cfg), values.data_ptr(), values.element_size(), values.shape[1],
                                           values.shape[0], bit_len, sublimb_bits, hist.data_ptr(), self.chip._stream()), "h2r_lookup_hist_values")
        return hist

    def hist_fresh_op(self, op_name: str, trace_buf: torch.Tensor, elem_stride: int, batch: int, hist: torch.Tensor, first_off: int = 0):
        """Adds the range assigns inside a Fresh-op witness (e.g. "is_in_field": the assert_in_field witness of modpow_public_key)."""
        check(lib().h2r_lookup_hist_fresh_op(self.chip._ctx, ctypes.byref(self.cfg), _lib.FRESH_OPS.index(op_name), trace_buf.data_ptr(), first_off,
                                             elem_stride, batch, hist.data_ptr(), self.chip._stream()), "h2r_lookup_hist_fresh_op")
        return hist

    def hist_advice(self, kinds, image: torch.Tensor, batch: int, hist: torch.Tensor, status: Optional[torch.Tensor] = None, layout=None) -> torch.Tensor:
        """The multiplicities of an advice image (h2r_lookup_hist_advice): what hist_verify / hist_records / hist_fresh_op count from a trace,
        for a witness without records.  kinds: uint8 row kinds (numpy or device tensor)."""
        kd = kinds if isinstance(kinds, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(kinds, dtype=np.uint8)).to(image.device)
        stride = image.shape[1] if image.dim() > 1 else image.numel() // batch
        check(lib().h2r_lookup_hist_advice(self.chip._ctx, ctypes.byref(self.cfg), ctypes.byref(layout) if layout is not None else None, kd.data_ptr(),
                                           kd.numel(), image.data_ptr(), stride, batch, status.data_ptr() if status is not None else None,
                                           hist.data_ptr(), self.chip._stream()), "h2r_lookup_hist_advice")
        self._keep = kd
        return hist

    def hist_verify(self, res, hist: torch.Tensor) -> torch.Tensor:
        """Every lookup inside the witness of a verify_pkcs1v15_signature batch (rsa.VerifyResult): assert_in_field's range assigns,
        the records' limbs and carries, the encoded-message check's two 4-bit range assigns (h2r_lookup_hist_verify)."""
        check(lib().h2r_lookup_hist_verify(self.chip._ctx, ctypes.byref(self.cfg), ctypes.byref(res.layout), res.trace.data_ptr(),
                                           res.is_valid.numel(), res.status.data_ptr(), hist.data_ptr(), self.chip._stream()), "h2r_lookup_hist_verify")
        return hist

    def permuted_columns(self, hist: torch.Tensor, thetas: Sequence[int], usable_rows: int, arg_mask: int = 31, out=None):
        """(A', S', status): uint8 [batch, 5, usable_rows, 32] each -- canonical little-endian field elements."""
        batch = hist.shape[0]
        dev = hist.device
        th = np.array([[(int(t) >> (64 * k)) & (2 ** 64 - 1) for k in range(4)] for t in thetas], dtype=np.uint64)
        assert th.shape == (batch, 4)
        th_dev = torch.from_numpy(th.view(np.int64)).to(dev)
        if out is None:
            a_perm = torch.empty((batch, self.ARGS, usable_rows, 32), dtype=torch.uint8, device=dev)
            s_perm = torch.empty((batch, self.ARGS, usable_rows, 32), dtype=torch.uint8, device=dev)
        else:
            a_perm, s_perm = out
        status = torch.zeros(batch, dtype=torch.uint8, device=dev)
        ws = torch.empty(int(lib().h2r_lookup_workspace_bytes(ctypes.byref(self.cfg), batch)), dtype=torch.uint8, device=dev)
        check(lib().h2r_lookup_permuted_columns(self.chip._ctx, ctypes.byref(self.cfg), hist.data_ptr(), th_dev.data_ptr(), batch, usable_rows,
                                                arg_mask, a_perm.data_ptr(), s_perm.data_ptr(), self.ARGS * usable_rows * 32, status.data_ptr(),
                                                ws.data_ptr(), self.chip._stream()), "h2r_lookup_permuted_columns")
        self._keep = (th_dev, ws)   # alive until the stream has run the kernels
        return a_perm, s_perm, status


@dataclass
class FreshResult:
    value: Optional[AssignedInteger]
    flag: torch.Tensor      # predicate / overflow bit per element
    status: torch.Tensor
    trace: torch.Tensor
    elem_stride: int
    stream_bytes: int
    op: int
    chip: BigIntChip
    inputs: Optional[tuple] = None   # (a, b, n, flags) the op was called with

    def emit_advice(self, assert_one: bool = False) -> torch.Tensor:
        """Every cell of the op as rows of the main gate's advice columns (h2r_fresh_op_emit_advice)."""
        a, b, n, flags = self.inputs
        return self.chip.fresh_op_emit_advice(self.op, a, b, n, flags, self.trace, 0, self.elem_stride, a.batch, self.status, assert_one)

    def flatten(self, elem: int) -> np.ndarray:
        host = np.ascontiguousarray(self.trace[elem * self.elem_stride:(elem + 1) * self.elem_stride].cpu().numpy())
        out = np.zeros(self.stream_bytes, dtype=np.uint8)
        check(lib().h2r_fresh_op_flatten(self.chip._ctx, self.op, host.ctypes.data, out.ctypes.data), "h2r_fresh_op_flatten")
        return out


@dataclass
class MuledResult:
    """AssignedInteger<Muled>: [batch, 2L, 4] int64 (256-bit little-endian un-carried columns) + the mul trace."""
    cols: torch.Tensor
    trace: torch.Tensor
    chip: BigIntChip

    def columns(self, elem: int) -> List[int]:
        h = self.cols[elem].cpu().numpy().view(np.uint64)
        return [sum(int(h[c, k]) << (64 * k) for k in range(4)) for c in range(2 * self.chip.num_limbs - 1)]

    def flatten(self, elem: int) -> np.ndarray:
        rs = self.chip.layout.record_stride
        host = np.ascontiguousarray(self.trace[elem * rs:(elem + 1) * rs].cpu().numpy())
        out = np.zeros(int(lib().h2r_mul_stream_bytes(self.chip._ctx)), dtype=np.uint8)
        check(lib().h2r_mul_trace_flatten(self.chip._ctx, host.ctypes.data, out.ctypes.data), "h2r_mul_trace_flatten")
        return out
