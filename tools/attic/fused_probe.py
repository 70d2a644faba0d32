"""Developer experiment: a pipeline step issued as ONE launch with two roles (chains of the next call + records of the
previous one), against the same work as separate kernels.  Needs the experiment build:
    python -m halo2_rsa_amd._build fused -DH2R_EXP_FUSED        (before gpurun)
    python tools/fused_probe.py [batch] [iters]
Prints ms per step for: fused launches back to back, [chain, record] serial pairs, the record kernel alone, the chain
kernel alone; then checks that the trace left behind by the fused launches is still the valid one (in-place audit)."""
import ctypes, os, random, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("H2R_LIB", os.path.join(ROOT, "halo2_rsa_amd", "lib", "variants", "fused.so"))
sys.path.insert(0, ROOT)
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd.big_integer import _e_bytes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
chip = H.BigIntChip(64, 2048)
rng = random.Random(7)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
a, n = chip.assign_integer(X), chip.assign_integer(N)
e = 65537
pl = chip.pow_fixed_layout(e)
eb = _e_bytes(e)
dev = "cuda:0"
arena = H.TraceArena.for_pow(chip, e, B, regions=1, candidates=16)
trace = arena.regions[0]
print("trace region: record kernel alone %.4f ms (best of %d candidates)" % (arena.region_ms[0], len(arena.measurements_ms)))
wsb = chip.workspace_bytes(B, pl.num_mul_mods)
ws_a = torch.empty(wsb, dtype=torch.uint8, device=dev); ws_b = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = chip._new_limbs(B); status = torch.zeros(B, dtype=torch.uint8, device=dev)
lib = _lib.lib()
fn = lib.h2r_exp_fused_period
fn.restype = ctypes.c_int32
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_void_p,
               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
               ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
def run(mode, dyn=0):
    ms = ctypes.c_double(0)
    rc = fn(chip._ctx, a.data_ptr(), n.data_ptr(), eb, len(eb), B, trace.data_ptr(), out.data_ptr(), status.data_ptr(),
            ws_a.data_ptr(), ws_b.data_ptr(), ITERS, mode, dyn, chip._stream(), ctypes.byref(ms))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return ms.value
names = {0: "fused launch (two roles)", 1: "chain kernel then record kernel (serial)", 2: "record kernel alone", 3: "chain kernel alone"}
for rep in range(2):
    for mode in (2, 3, 1, 0):
        print("%-44s %.4f ms per step" % (names[mode], run(mode)))
    for dyn in (4096, 6144, 8192, 10240, 12288, 14336):
        print("%-44s %.4f ms per step" % ("fused, + %d B dynamic LDS per workgroup" % dyn, run(0, dyn)))
# the trace the fused launches left behind
run(0)
got = H.AssignedInteger(out, 64).to_big_uint()
assert all(got[i] == pow(X[i], e, N[i]) for i in (0, 1, B // 2, B - 1)), "results differ from pow()"
tr = H.Trace(chip, trace, B, pl)
q0 = int.from_bytes(tr.plane(0, 0, "Q").tobytes(), "little"); r0 = int.from_bytes(tr.plane(0, 0, "R").tobytes(), "little")
assert X[0] * X[0] == q0 * N[0] + r0
assert int(status.max().item()) == 0
print("results and first record valid after the fused launches")
