import sys, time, random, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537); B = 1024
rng = random.Random(3)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
x, n = chip.assign_integer(X), chip.assign_integer(N)
arena = H.TraceArena.for_pow(chip, 65537, B, regions=2, candidates=16)
sets = [dict(trace=arena.regions[i], ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
             out=torch.empty((B, 32), dtype=torch.int64, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda"),
             inf=torch.zeros(B * chip.in_field_layout()[0], dtype=torch.uint8, device="cuda")) for i in range(2)]
pipe = chip.pipeline()
def run(K, timing, inf):
    _lib.profile_enable(4 * K + 8 if timing else 0)
    for k in range(4):
        s = sets[k % 2]; pipe.modpow_public_key(x, 65537, n, s["trace"], s["ws"], s["out"], s["status"], in_field_buf=s["inf"] if inf else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        s = sets[k % 2]; pipe.modpow_public_key(x, 65537, n, s["trace"], s["ws"], s["out"], s["status"], in_field_buf=s["inf"] if inf else None)
    t_host = time.perf_counter() - t0
    pipe.join(); torch.cuda.synchronize()
    t = time.perf_counter() - t0
    step = _lib.profile_read(_lib.KERNEL_STEP) if timing else []
    _lib.profile_enable(0)
    print("timing=%d in_field=%d: host issue %.1f us per call, %.4f ms per step over %d steps, step launch avg %s" %
          (timing, inf, 1e6 * t_host / K, 1e3 * t / K, K, ("%.4f" % (sum(step[-K:]) / len(step[-K:]))) if step else "-"))
print("kept regions alone ms", arena.region_ms)
for rep in range(2):
    for timing in (0, 1):
        for inf in (0, 1):
            run(200, timing, inf)
