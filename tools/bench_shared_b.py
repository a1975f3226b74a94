#!/usr/bin/env python3
"""abx with a B shared by the heads of a group (true-GQA) vs per-head B, BASELINE config-2 size: us and HBM fraction."""
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, prepare_b, rope_inv_freq

H, G, R, D, L = 32, 8, 128, 128, 65536
torch.manual_seed(0)
a = torch.randn(H, 1, D, device="cuda").half()
bg = (torch.randn(G, 1, R, D, device="cuda") * R ** -0.5).half()
b_shared = bg.expand(G, H // G, R, D).reshape(H, R, D).contiguous()
b_free = (torch.randn(H, R, D, device="cuda") * R ** -0.5).half()
x = torch.randn(G, L, R, device="cuda").half()
out = torch.empty(H, 1, L, device="cuda", dtype=torch.float16)
byt = 2 * G * L * R + 2 * H * R * D + 2 * H * D + 2 * H * L


def t(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for r in range(3):
    us_s = t(lambda: abx(a, b_shared, x, out=out))
    us_f = t(lambda: abx(a, b_free, x, out=out))
    print(f"abx R={R} L={L}: shared-B {us_s:6.1f} us = {byt / us_s * 1e-3:6.0f} GB/s = {byt / us_s * 1e-3 / 8000:.3f} of HBM peak | "
          f"per-head B {us_f:6.1f} us = {byt / us_f * 1e-3 / 8000:.3f}")
