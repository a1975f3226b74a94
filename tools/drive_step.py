#!/usr/bin/env python3
"""Run the full decode step a few times at BASELINE config 2 (for rocprofv3)."""
import argparse
import torch
from palu_amd.kernel import head_parallel as hp

ap = argparse.ArgumentParser()
ap.add_argument("--rank_k", type=int, default=1024)
ap.add_argument("--rank_v", type=int, default=3072)
ap.add_argument("--L", type=int, default=65536)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
H, D, G, HID = 32, 128, 8, 4096
Rk, Rv = a.rank_k // G, a.rank_v // G
dev = torch.device("cuda", 0)
torch.manual_seed(0)
plan = hp.make_plan(1, 0, H, G, D, Rk, Rv)
w = {"wq": (torch.randn(H * D, HID, device=dev) / 64).half(), "vt_k": (torch.randn(a.rank_k, HID, device=dev) / 64).half(),
     "vt_v": (torch.randn(a.rank_v, HID, device=dev) / 64).half(), "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
     "wo": (torch.randn(HID, H * Rv, device=dev) * 0.01).half()}
cap = a.L + 128
k = torch.randn(G, cap, Rk, device=dev, dtype=torch.float16)
v = torch.randn(G, cap, Rv, device=dev, dtype=torch.float16)
hid = torch.randn(HID, device=dev, dtype=torch.float16)
dec = hp.HeadParallelDecoder(plan, w, k, v, HID)
for _ in range(a.iters):
    out = dec.step(hid, a.L, a.L)
torch.cuda.synchronize()
print("ok", float(out.float().abs().max()))
