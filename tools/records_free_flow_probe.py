"""The records-free flow end to end, per batch of 1,024 RSA-2048 signatures (one circuit each, k = 17): the whole verify_pkcs1v15_signature
element image (h2r_pipeline_verify_pkcs1v15_advice), its lookup multiplicities from the image (h2r_lookup_hist_advice + h2r_lookup_hist_values
for the assign_integer inputs) and halo2's permuted columns A', S' of the five lookup arguments (h2r_lookup_permuted_columns) -- everything a
column-taking prover needs of the witness, no record written.  ms per batch and bytes produced.  argv: batch [montgomery]"""
import ctypes, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mont = len(sys.argv) > 2 and sys.argv[2] == "montgomery"
usable = (1 << 17) - 6
rows_probe = H.RSAChip(2048, 5).bigint_chip()
kw = dict(columns=True, montgomery=True, col_stride=((77219 * 32 + 4095) // 4096) * 4096) if mont else {}
rsa = H.RSAChip(2048, 5, **kw)
chip = rsa.bigint_chip()
la = H.LookupArgument(chip, rsa_chip=True)
rng = random.Random(3)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
S = [rng.randrange(n) for n in N]
sig, n = chip.assign_integer(S), chip.assign_integer(N)
hashed = torch.tensor([[rng.getrandbits(63) for _ in range(4)] for _ in range(B)], dtype=torch.int64, device="cuda")
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
thetas = [rng.randrange(P) for _ in range(B)]
pipe = H.Pipeline(chip, 2, 2)
vl = pipe.verify_compact_layout(65537)
sec = (ctypes.c_uint64 * 4)()
rows = int(lib().h2r_verify_advice_rows(chip._ctx, ctypes.byref(vl), sec))
kinds = np.zeros(rows, dtype=np.uint8)
lib().h2r_verify_row_kinds(chip._ctx, ctypes.byref(vl), kinds.ctypes.data)
kd = torch.from_numpy(kinds).cuda()
eb = chip.image_bytes(rows)
sets = [dict(img=torch.empty((B, eb), dtype=torch.uint8, device="cuda"), wit=torch.zeros((B, vl.elem_stride), dtype=torch.uint8, device="cuda"),
             ws=torch.empty(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda"), powed=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
             valid=torch.zeros(B, dtype=torch.uint8, device="cuda"), st=torch.zeros(B, dtype=torch.uint8, device="cuda"), hist=la.new_hist(B)) for _ in range(2)]
cols = (torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda"), torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda"))


def step(k):
    s = sets[k & 1]
    pipe.verify_pkcs1v15_advice(sig, 65537, n, hashed, s["wit"], s["ws"], s["powed"], s["valid"], s["st"], s["img"])
    if k:   # the previous batch's image is complete (depth 2): its multiplicities and permuted columns
        p = sets[(k - 1) & 1]
        p["hist"].zero_()
        la.hist_advice(kd, p["img"], B, p["hist"], status=p["st"])
        la.hist_values(sig.limbs_dev, 64, 8, p["hist"]); la.hist_values(n.limbs_dev, 64, 8, p["hist"])      # assign_integer(sig), assign_integer(n)
        la.permuted_columns(p["hist"], thetas, usable, out=cols)


for k in range(3):
    step(k)
torch.cuda.synchronize()
K = 6
t0 = time.perf_counter()
for k in range(3, 3 + K):
    step(k)
pipe.join()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
img_b, col_b = B * rows * 160, 2 * B * 5 * usable * 32
print("%s, %d signatures per batch: %.3f ms per batch = %.0f circuits/s; %.1f GB of advice image + %.1f GB of A', S' per batch = %.2f TB/s written"
      % ("planar Montgomery" if mont else "row-major canonical", B, 1e3 * dt, B / dt, img_b / 1e9, col_b / 1e9, (img_b + col_b) / dt / 1e12))
