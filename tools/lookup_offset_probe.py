"""Does the RELATIVE position of the two output columns decide the lookup call's rate?  One allocation, A' at its start, S' at
start + col_bytes + delta for a list of deltas; h2r_lookup_permuted_columns timed for each (profiles/r05_lookup_placement.txt)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H

B = 256
chip = H.BigIntChip(64, 2048)
rng = random.Random(1)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
la = H.LookupArgument(chip)
usable = (1 << 17) - 6
hist = la.new_hist(B)
la.hist_records(res.trace, hist)
torch.cuda.synchronize()
del res
torch.cuda.empty_cache()
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
thetas = [rng.randrange(P) for _ in range(B)]
col = B * 5 * usable * 32
slack = 1 << 30
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for b in range(nbuf):
    big = torch.empty(2 * col + slack + (1 << 21), dtype=torch.uint8, device="cuda")
    base = (-big.data_ptr()) % (1 << 21)
    print("allocation %d at %#x (A' at +%#x)" % (b, big.data_ptr(), base))
    for delta in (0, 32, 128, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 21, 3 << 20, 1 << 24, 1 << 27, (1 << 29) + 4096):
        a = big[base:base + col].view(B, 5, usable, 32)
        s = big[base + col + delta:base + 2 * col + delta].view(B, 5, usable, 32)
        la.permuted_columns(hist, thetas, usable, out=(a, s))
        ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ta.record()
        for _ in range(3):
            la.permuted_columns(hist, thetas, usable, out=(a, s))
        tb.record()
        torch.cuda.synchronize()
        ms = ta.elapsed_time(tb) / 3
        print("  S' at A' + col_bytes + %-10d (S' - A' = %#x): %.4f ms  %.2f TB/s" % (delta, col + delta, ms, 2 * col / ms / 1e9))
    del big, a, s
    torch.cuda.empty_cache()
    keep = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")   # (shift the next allocation)
