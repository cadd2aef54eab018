"""What a plain device-to-device copy of the mask stage's matrix moves on this GPU (torch's copy kernel and hipMemcpy): the yardstick
for mask_rows_fast_kernel's read + write rate.  python tools/microbench/copy_rate.py"""
import torch
for B in (1024, 4096, 16384):
    x = torch.rand((B, 29523), device="cuda")
    y = torch.empty_like(x)
    for name, fn in (("tensor.copy_", lambda: y.copy_(x)), ("x * 0.5 (elementwise)", lambda: torch.mul(x, 0.5, out=y))):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"B={B}: {name}: {ms:.4f} ms = {2 * x.numel() * 4 / ms / 1e6:.0f} GB/s (read + write)")
