#!/usr/bin/env python3
"""Run the abx kernel a few times at a BASELINE shape (for rocprofv3)."""
import argparse
import torch
from palu_amd.kernel.abx_rope import abx

ap = argparse.ArgumentParser()
ap.add_argument("--rank", type=int, default=1024)
ap.add_argument("--L", type=int, default=65536)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--heads", type=int, default=32)
ap.add_argument("--gs", type=int, default=4)
a = ap.parse_args()
torch.manual_seed(0)
H, G = a.heads, a.heads // a.gs
R = a.rank // G
A = torch.randn(H, 1, 128, dtype=torch.float16, device="cuda")
B = torch.randn(H, R, 128, dtype=torch.float16, device="cuda")
X = torch.randn(G, a.L, R, dtype=torch.float16, device="cuda")
for _ in range(a.iters):
    out = abx(A, B, X)
torch.cuda.synchronize()
print("ok", float(out.float().abs().max()))
