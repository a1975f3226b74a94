#!/usr/bin/env python3
"""us per abx launch (fp16 latents, 32 heads, 8 groups), back to back:  time_abx_loop.py R L [R L ...]"""
import sys
import torch
from palu_amd.kernel.abx_rope import abx, rope_inv_freq

H, G, D = 32, 8, 128
torch.manual_seed(0)
args = [int(v) for v in sys.argv[1:]] or [128, 65536]
for R, L in zip(args[0::2], args[1::2]):
    a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
    b = (torch.randn(H, R, D, device="cuda") * R ** -0.5).half()
    x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
    out = torch.empty(H, 1, L, device="cuda", dtype=torch.float16)
    rope_inv_freq(x.device)
    for _ in range(5):
        abx(a, b, x, out=out)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            abx(a, b, x, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 30)
    ts.sort()
    print(f"R={R} L={L}: median {ts[2]:.2f} us, min {ts[0]:.2f} us; checksum {float(out.float().abs().sum()):.3f}")
