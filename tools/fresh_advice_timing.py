#!/usr/bin/env python3
"""Time the row-program interpreter (rowprog_kernel, h2r_rowprog.hpp): assert_in_field(x, n) of a batch as advice rows, and the
whole verify_pkcs1v15_signature element image (4 launches: seed row, in-field rows, pow rows, encoded-message rows).
usage: fresh_advice_timing.py [batch]"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chip = H.BigIntChip(64, 2048)
rng = random.Random(9)
n = rng.getrandbits(2048) | (1 << 2047) | 1
X = [rng.randrange(n) for _ in range(B)]
x_dev, n_dev = chip.assign_integer(X), chip.assign_integer([n])
res = chip.is_in_field(x_dev, n_dev)
img = res.emit_advice(assert_one=True); torch.cuda.synchronize()
nbytes = img.numel(); del img
_lib.profile_enable(16)
for _ in range(5):
    img = res.emit_advice(assert_one=True); del img
torch.cuda.synchronize()
ms = _lib.profile_read(_lib.KERNEL_EMIT); _lib.profile_enable(0)
avg = sum(ms) / len(ms)
print("rowprog_kernel assert_in_field rsa2048 batch %d: %.3f ms per launch (min %.3f), %d rows / element, image written %.2f TB/s (%.1f MB)"
      % (B, avg, min(ms), nbytes // B // 160, nbytes / avg / 1e9, nbytes / 1e6))
out = torch.empty((B, nbytes // B), dtype=torch.uint8, device="cuda")
op = _lib.FRESH_OPS.index("is_in_field")
fl = _lib.H2R_F_SHARED_MODULUS | _lib.H2R_ADVICE_ASSERT_ONE
def call():
    _lib.check(_lib.lib().h2r_fresh_op_emit_advice(chip._ctx, op, fl, x_dev.data_ptr(), n_dev.data_ptr(), None, res.trace.data_ptr(), 0,
                                                   res.elem_stride, B, res.status.data_ptr(), out.data_ptr(), out.shape[1], chip._stream()), "emit")
call(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record(); torch.cuda.synchronize()
both = e0.elapsed_time(e1) / 10
print("  with the inverse witnesses (rowprog_inv_kernel behind it): %.3f ms per call, %.2f TB/s of image" % (both, nbytes / both / 1e9))
# whole verify element (batch limited by the 12 MB image per element)
Bv = min(B, 512)
rsa = H.RSAChip(2048, 5)
pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints([n], 32, 64), H.Fix(65537)))
sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(X[:Bv], 32, 64)))
vres = rsa.verify_pkcs1v15_signature(pk, [rng.getrandbits(256) for _ in range(Bv)], sg)
img = vres.emit_advice(); torch.cuda.synchronize()
nbytes = img.numel(); del img
t = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = vres.emit_advice(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3); del img
total, sec = vres.advice_sections()
print("verify element image rsa2048 batch %d: %.2f ms wall (min of 3; 4 launches + torch.empty), sections %s rows, %.2f GB, %.2f TB/s"
      % (Bv, min(t), sec, nbytes / 1e9, nbytes / min(t) / 1e9))

# the same element with the pow rows written directly from the operands (H2R_ADVICE_DIRECT: cells_kernel), into a reused buffer
import ctypes
outv = torch.empty((Bv, nbytes // Bv), dtype=torch.uint8, device="cuda")
sig_d, n_d, hashed_d = vres.inputs
for direct in (0, _lib.H2R_ADVICE_DIRECT):
    def vcall():
        _lib.check(_lib.lib().h2r_verify_emit_advice(chip._ctx, ctypes.byref(vres.layout), sig_d.data_ptr(), n_d.data_ptr(), hashed_d.data_ptr(),
                                                     vres.powed.data_ptr(), chip._flags(n_d, Bv) | direct, vres.trace.data_ptr(), vres.workspace.data_ptr(),
                                                     Bv, vres.status.data_ptr(), outv.data_ptr(), outv.shape[1], chip._stream()), "verify_emit")
    vcall(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        vcall()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("verify element image, %s, reused buffer: %.3f ms per call = %.2f TB/s of image (%.2f of the HBM peak)" %
          ("pow rows DIRECT (cells_kernel)" if direct else "pow rows from the records (advice_kernel)", ms, nbytes / ms / 1e9, nbytes / ms / 8e9))

# the same DIRECT call over several candidate image buffers (the placement classes of profiles/r04_cells_placement.txt apply to the image
# like to any buffer: a consumer that reuses its image buffer keeps the best of a few), and the share of the pow rows' kernel in the call
cands = [outv] + [torch.empty_like(outv) for _ in range(3)]
def vcall_into(buf):
    _lib.check(_lib.lib().h2r_verify_emit_advice(chip._ctx, ctypes.byref(vres.layout), sig_d.data_ptr(), n_d.data_ptr(), hashed_d.data_ptr(),
                                                 vres.powed.data_ptr(), chip._flags(n_d, Bv) | _lib.H2R_ADVICE_DIRECT, vres.trace.data_ptr(), vres.workspace.data_ptr(),
                                                 Bv, vres.status.data_ptr(), buf.data_ptr(), buf.shape[1], chip._stream()), "verify_emit")
best = None
for i, buf in enumerate(cands):
    vcall_into(buf); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        vcall_into(buf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("  candidate buffer %d: %.3f ms per call = %.2f TB/s of image (%.3f of the HBM peak)" % (i, ms, nbytes / ms / 1e9, nbytes / ms / 8e9))
    if best is None or ms < best[0]:
        best = (ms, buf)
_lib.profile_enable(256)
for _ in range(5):
    vcall_into(best[1])
torch.cuda.synchronize()
cells = _lib.profile_read(_lib.KERNEL_CELLS)
emit = _lib.profile_read(_lib.KERNEL_EMIT)
_lib.profile_enable(0)
print("  best buffer, per-kernel stamps: cells_kernel %.3f ms avg of %d (%.2f TB/s on its rows); %d row-program launches, %.3f ms summed per call (they run on the ctx's side stream)"
      % (sum(cells) / len(cells), len(cells), Bv * sec[2] * 160 / (sum(cells) / len(cells)) / 1e9, len(emit), sum(emit) / 5))
