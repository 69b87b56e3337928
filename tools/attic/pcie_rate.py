#!/usr/bin/env python3
"""What the host link of this box gives: pinned-memory copies H2D alone, D2H alone and both at once (two HIP streams),
256 MiB each.  The number bench.py's PCIe-inclusive `also.c5_stream_pcie` is to be read against."""
import time

import torch

n = 256 << 20
h_a, h_b = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
d_a, d_b = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(h2d, d2h, reps=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_a.copy_(h_a, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_b.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize()
    return (h2d + d2h) * reps * n / (time.perf_counter() - t0) / 1e9


run(True, True, 2)
print(f"H2D alone {run(True, False):.1f} GB/s, D2H alone {run(False, True):.1f} GB/s, both at once {run(True, True):.1f} GB/s aggregate")
