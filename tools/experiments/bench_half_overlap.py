#!/usr/bin/env python3
"""Experiment (DESIGN 4.7 follow-up): can HALF of the CUs stream V at twice the per-CU rate while the other half runs the
score kernel?  abx restricted to 128 workgroups (PALU_ABX_CUS=128, 4 groups, full per-workgroup work) on one stream,
softmax.PV of 4 other groups on a second stream -- versus each alone.  If `both` ~ max(alone) the premise of a
phase-staggered persistent kernel (score phase | V phase alternating per CU) holds."""
import math
import os
import torch
os.environ.setdefault("PALU_ABX_CUS", "128")
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

H, G, D, R, Rv, L = 16, 4, 128, 128, 384, 65536
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


def both_pv_first():
    s2.wait_stream(s1)
    pv(s2)
    abx(s1)
    s1.wait_stream(s2)


for r in range(3):
    print(f"PALU_ABX_CUS={os.environ['PALU_ABX_CUS']} PV_WGS_PER_CU={os.environ.get('PALU_PV_WGS_PER_CU', 'default')}: "
          f"abx(4 groups, 128 WGs) alone {timeit(lambda: abx(s1)):6.1f} us   pv(4 groups) alone {timeit(lambda: pv(s1)):6.1f} us   "
          f"abx || pv {timeit(both):6.1f} us   pv || abx {timeit(both_pv_first):6.1f} us")
