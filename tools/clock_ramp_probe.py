import os, sys, random, time
sys.path.insert(0, "/root/repo")
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = 1024
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537); ies = chip.in_field_layout()[0]
rng = random.Random(3)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
arena = H.TraceArena.for_pow(chip, 65537, B, regions=2, candidates=24)
print("kept", [round(t, 4) for t in arena.region_ms])
z = lambda nb: torch.zeros(nb, dtype=torch.uint8, device="cuda")
ss = [dict(ws=z(chip.workspace_bytes(B, pl.num_mul_mods)), inf=z(B * ies), out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"), status=z(B)) for _ in range(2)]
pipe = chip.pipeline()
torch.cuda.synchronize(); time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.5)
_lib.profile_enable(1024)
for k in range(400):
    s = ss[k % 2]; pipe.modpow_public_key(x, 65537, n, arena.regions[k % 2], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
pipe.join(); torch.cuda.synchronize()
st = _lib.profile_read(_lib.KERNEL_STEP)
for a in range(0, 400, 20):
    seg = st[a:a + 20]
    if seg: print("launches %3d-%3d: avg %.4f ms" % (a, a + len(seg) - 1, sum(seg) / len(seg)))
