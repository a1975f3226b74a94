#!/usr/bin/env python3
"""palu_hadamard_transform (csrc/hadamard.hip): us per call and GB/s (read + write) at the widths the weight preparation
uses: n in {128, 512} x 4096 rows (one weight matrix) and a size large enough to stream (1M rows)."""
import torch
from palu_amd.kernel import hadamard_utils as hu

for dt in (torch.float16, torch.float32):
    for n, rows in ((128, 4096), (512, 4096), (128, 1 << 20), (512, 1 << 18), (4096, 1 << 15)):
        x = torch.randn(rows, n, device="cuda").to(dt)
        for _ in range(5):
            hu.hadamard_transform(x, 1.0)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                hu.hadamard_transform(x, 1.0)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        us = sorted(ts)[2]
        b = 2 * rows * n * x.element_size()
        print(f"{str(dt):14s} n={n:5d} rows={rows:8d}: {us:9.2f} us  {b / us * 1e-3:8.1f} GB/s ({b / us * 1e-3 / 8000:.3f} of 8 TB/s)")
