"""(H2R_CELLS_NWV is read by the -DH2R_DEV_KNOBS build only: python -m halo2_rsa_amd._build devknobs -DH2R_DEV_KNOBS; H2R_LIB=halo2_rsa_amd/lib/variants/devknobs.so)
cells_kernel in Montgomery form, by waves per workgroup (H2R_CELLS_NWV = 1 | 2 | 4 | 8, read at ctx creation): ms per launch and TB/s
for a fixed-exponent pow image written directly from the operands.  argv: limb_width bits batch"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import lib

w, bits, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw = dict(montgomery=True) if len(sys.argv) < 5 or sys.argv[4] != "canonical" else {}
chip = H.BigIntChip(w, bits, **kw)
rng = random.Random(5)
N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
pl = chip.pow_fixed_layout(65537)
ws = torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
n_dev = chip.assign_integer(N)
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, n_dev, want_trace=False, workspace=ws)
rows = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
img = torch.empty((B, rows * 160), dtype=torch.uint8, device="cuda")


def go():
    _lib.check(lib().h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pl), n_dev.data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0, ws.data_ptr(), B, res.status.data_ptr(),
                                               img.data_ptr(), rows * 160, chip._stream()), "emit")
for _ in range(2):
    go()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    go()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print("w=%d bits=%d batch=%d %s NWV=%s: %.3f ms per launch, %.2f TB/s (%.1f GB image)" % (w, bits, B, "montgomery" if kw else "canonical", os.environ.get("H2R_CELLS_NWV", "default"), ms,
                                                                                      B * rows * 160 / ms / 1e9, B * rows * 160 / 1e9))
