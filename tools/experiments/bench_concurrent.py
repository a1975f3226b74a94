#!/usr/bin/env python3
"""Experiment: abx (MFMA/power-bound) and softmax.PV (HBM-bound) launched CONCURRENTLY on two streams over independent
data (no dependency) -- how much of the sum of their times does co-residency on the same CUs recover?"""
import math
import os
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

H, G, D, R, Rv, L = 32, 8, 128, 128, 384, 65536
torch.manual_seed(0)
a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
b = torch.randn(H, R, D, device="cuda", dtype=torch.float16)
x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
v = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16)
scores = torch.empty(H, L, device="cuda", dtype=torch.float16)
scores2 = (torch.randn(H, L, device="cuda") * 10).half()
frag = prepare_b(b, G)
inv = rope_inv_freq(x.device)
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def abx(stream):
    _lib.check(_lib.lib.palu_abx_rope_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), x.data_ptr(), x.stride(0),
                                          x.stride(1), scores.data_ptr(), scores.stride(0), H, G, L, R, D,
                                          inv.data_ptr(), 0, stream.cuda_stream), "abx")


def pv(stream):
    _lib.check(_lib.lib.palu_softmax_pv_f16(scores2.data_ptr(), scores2.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1),
                                            ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, math.sqrt(128.0),
                                            stream.cuda_stream), "pv")


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def both():
    s2.wait_stream(s1)
    abx(s1)
    pv(s2)
    s1.wait_stream(s2)


print(f"W4={os.environ.get('PALU_ABX_W4', '0')} PV_WGS_PER_CU={os.environ.get('PALU_PV_WGS_PER_CU', 'default')}: "
      f"abx alone {timeit(lambda: abx(s1)):6.1f} us   pv alone {timeit(lambda: pv(s1)):6.1f} us   "
      f"abx || pv {timeit(both):6.1f} us")
