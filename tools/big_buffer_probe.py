"""Probe: does pure-write bandwidth depend on the buffer size (TLB reach)?  torch fill_ on 1.25 .. 20 GB."""
import torch
for gb in (1.25, 2.5, 5, 10, 20):
    n = int(gb * 1e9) // 16 * 16
    x = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        x.fill_(1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        x.fill_(3)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("fill %5.2f GB  %.3f ms  %.0f GB/s" % (gb, ms, n / ms / 1e6))
    del x
