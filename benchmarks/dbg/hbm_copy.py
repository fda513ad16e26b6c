"""Practical HBM ceiling of this box: torch's device-to-device copy and fill at the image pass's sizes (268 MB in, 137 MB out)."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (137, 268, 1024):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    us = t(lambda: y.copy_(x))
    print(f"copy {mb} MB: {us:.1f} us, {2 * mb * 1.048576 / us * 1e3:.0f} GB/s (read + write)")
    us = t(lambda: y.zero_())
    print(f"fill {mb} MB: {us:.1f} us, {mb * 1.048576 / us * 1e3:.0f} GB/s")
    us = t(lambda: x.sum())
    print(f"sum  {mb} MB: {us:.1f} us, {mb * 1.048576 / us * 1e3:.0f} GB/s")
