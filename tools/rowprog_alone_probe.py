"""The assert_in_field row program ALONE (no cells kernel next to it), per representation: ms per 1,024 RSA-2048 elements.
What the pipelined advice forms have to hide on the caller's stream (profiles/r05_rowprog.txt)."""
import ctypes, random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import lib

B = 1024
rng = random.Random(3)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
for kw in (dict(), dict(columns=True), dict(montgomery=True), dict(columns=True, montgomery=True)):
    rows = 1532
    cs = ((rows * 32 + 4095) // 4096) * 4096 if kw.get("columns") else 0
    chip = H.BigIntChip(64, 2048, col_stride=cs, **kw) if kw else H.BigIntChip(64, 2048)
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    res = chip.pow_mod_fixed_exp(x, 3, n, want_trace=False, check_in_field=True,
                                 workspace=torch.empty(chip.workspace_bytes(B, 3), dtype=torch.uint8, device="cuda"))
    eb = chip.image_bytes(rows)
    img = torch.empty((B, eb), dtype=torch.uint8, device="cuda")
    op = _lib.FRESH_OPS.index("is_in_field")

    def go():
        _lib.check(lib().h2r_fresh_op_emit_advice(chip._ctx, op, _lib.H2R_ADVICE_ASSERT_ONE, x.data_ptr(), n.data_ptr(), None, res.in_field.buf.data_ptr(), 0, 0, B,
                                                  res.status.data_ptr(), img.data_ptr(), eb, chip._stream()), "emit")
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        go()
    b.record()
    torch.cuda.synchronize()
    print("%-40s %.4f ms per call (rowprog + inverse kernels), %.1f GB/s" % (kw or "canonical row-major", a.elapsed_time(b) / 20, B * rows * 160 / (a.elapsed_time(b) / 20) / 1e6))
