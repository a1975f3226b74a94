#!/usr/bin/env python3
"""Run the fused decode attention kernel a few times at a BASELINE shape (for rocprofv3)."""
import argparse
import math

import torch

from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

ap = argparse.ArgumentParser()
ap.add_argument("--rank_k", type=int, default=1024)
ap.add_argument("--rank_v", type=int, default=3072)
ap.add_argument("--L", type=int, default=65537)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--heads", type=int, default=32)
ap.add_argument("--gs", type=int, default=4)
a = ap.parse_args()
torch.manual_seed(0)
H, G, D = a.heads, a.heads // a.gs, 128
Rk, Rv = a.rank_k // 8, a.rank_v // 8
q = torch.randn(H, D, dtype=torch.float16, device="cuda")
b = (torch.randn(H, Rk, D, device="cuda") * Rk ** -0.5).half()
k = torch.randn(G, a.L + 64, Rk, dtype=torch.float16, device="cuda")
v = torch.randn(G, a.L + 64, Rv, dtype=torch.float16, device="cuda")
frag = prepare_b(b, G)
inv = rope_inv_freq(k.device)
ws = torch.zeros(_lib.lib.palu_pv_workspace_bytes(H, G, a.L + 64, Rv), dtype=torch.uint8, device="cuda")
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
for _ in range(a.iters):
    _lib.check(_lib.lib.palu_decode_attn_f16(q.data_ptr(), q.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                             k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(),
                                             ws.data_ptr(), H, G, a.L, Rk, Rv, D, inv.data_ptr(), 0, math.sqrt(D),
                                             torch.cuda.current_stream().cuda_stream), "fused")
torch.cuda.synchronize()
print("ok", float(ctx.float().abs().max()))
