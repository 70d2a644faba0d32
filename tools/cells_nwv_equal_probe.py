"""The eight-wave and the one-wave Montgomery cells kernel write the same image: a digest of a 32-element pow image of the given shape.
Run once with the product library and once with H2R_LIB=.../variants/devknobs.so H2R_CELLS_NWV=1; the two digests must be equal.  argv: limb_width bits"""
import ctypes, hashlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import lib
w, bits = int(sys.argv[1]), int(sys.argv[2])
B = 32
chip = H.BigIntChip(w, bits, columns=True, montgomery=True)
rng = random.Random(9)
N = [rng.getrandbits(bits) | (1 << (bits - 1)) | (i != 3) for i in range(B)]
X = [rng.randrange(n) for n in N]
X[5] = 0; X[6] = N[6] - 1
pl = chip.pow_fixed_layout(65537)
ws = torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
n_dev = chip.assign_integer(N)
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, n_dev, want_trace=False, workspace=ws)
rows = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
img = torch.zeros((B, rows * 160), dtype=torch.uint8, device="cuda")
_lib.check(lib().h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pl), n_dev.data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0, ws.data_ptr(), B, res.status.data_ptr(),
                                           img.data_ptr(), rows * 160, chip._stream()), "emit")
torch.cuda.synchronize()
print("w=%d bits=%d NWV=%s lib=%s sha256 %s" % (w, bits, os.environ.get("H2R_CELLS_NWV", "rule"), os.path.basename(os.environ.get("H2R_LIB", "libh2r.so")),
                                            hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()))
