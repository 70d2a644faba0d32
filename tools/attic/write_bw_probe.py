"""Probe: practical pure-write HBM bandwidth on this GPU (torch fill / hipMemset of 1.25 GB)."""
import torch, time
n = 1251760128
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n // 8, dtype=torch.int64, device="cuda")
for name, fn in (("uint8.fill_", lambda: x.fill_(7)), ("int64.fill_", lambda: y.fill_(7)), ("zero_", lambda: x.zero_()),
                 ("copy_ (read+write)", lambda: x[: n // 2].copy_(x[n // 2: 2 * (n // 2)]))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    by = n if "copy" not in name else n  # copy moves n/2 read + n/2 written
    print("%-20s %.3f ms  %.0f GB/s" % (name, ms, by / ms / 1e6))
