#!/usr/bin/env python3
"""Round-robin timing of the fused decode attention kernel under its experiment flags (one process, interleaved, so
box-to-box and warm-up differences cancel):   python tools/bench_fused_variants.py 0 8 2 ...  [--L 65537]"""
import argparse
import math

import numpy as np
import torch

from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

ap = argparse.ArgumentParser()
ap.add_argument("flags", type=int, nargs="*", default=[0])
ap.add_argument("--L", type=int, default=65537)
ap.add_argument("--G", type=int, default=8)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--calls", type=int, default=20)
a = ap.parse_args()
lib = _lib.lib
torch.manual_seed(0)
G, D, Rk, Rv, L = a.G, 128, 128, 384, a.L
H = 4 * G
q = torch.randn(H, D, dtype=torch.float16, device="cuda")
b = (torch.randn(H, Rk, D, device="cuda") * Rk ** -0.5).half()
k = torch.randn(G, L + 64, Rk, dtype=torch.float16, device="cuda")
v = torch.randn(G, L + 64, Rv, dtype=torch.float16, device="cuda")
frag = prepare_b(b, G)
inv = rope_inv_freq(k.device)
ws = torch.zeros(lib.palu_pv_workspace_bytes(H, G, L + 64, Rv), dtype=torch.uint8, device="cuda")
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
scores = torch.empty(H, L + 7, dtype=torch.float16, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def fused():
    _lib.check(lib.palu_decode_attn_f16(q.data_ptr(), q.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                        k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(),
                                        ws.data_ptr(), H, G, L, Rk, Rv, D, inv.data_ptr(), 0, math.sqrt(D), s), "fused")


def two():
    _lib.check(lib.palu_abx_rope_f16(q.data_ptr(), q.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                     k.stride(1), scores.data_ptr(), scores.stride(0), H, G, L, Rk, D, inv.data_ptr(),
                                     0, s), "abx")
    _lib.check(lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1),
                                       ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, math.sqrt(D), s), "pv")


def t(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


lib.palu_decode_attn_set_exp.argtypes = [_lib.i32]
res = {f: [] for f in a.flags}
res["two-kernel"] = []
for r in range(a.rounds):
    for f in a.flags:
        lib.palu_decode_attn_set_exp(f)
        res[f].append(t(fused, a.calls))
    res["two-kernel"].append(t(two, a.calls))
lib.palu_decode_attn_set_exp(0)
for key, vals in res.items():
    print(f"exp {key!s:>10}: min {min(vals):7.1f}  median {np.median(vals):7.1f}  max {max(vals):7.1f} us   (G={G} L={L})")
