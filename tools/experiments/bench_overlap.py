#!/usr/bin/env python3
"""Experiment: does the MFMA/power-bound abx kernel overlap with the HBM-bound softmax.PV kernel when the two run on
different halves of the cache from two streams?  (C2 shapes; timing only.)"""
import math
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
frag = prepare_b(b, G)
inv = rope_inv_freq(x.device)
ctx = torch.empty(2, H, Rv, dtype=torch.float16, device="cuda")
ws = [torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda") for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def abx(l0, l1, stream):
    xs = x[:, l0:l1]
    _lib.check(_lib.lib.palu_abx_rope_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), xs.data_ptr(), x.stride(0),
                                          x.stride(1), scores[:, l0:].data_ptr(), scores.stride(0), H, G, l1 - l0, R, D,
                                          inv.data_ptr(), l0, stream.cuda_stream), "abx")


def pv(l0, l1, k, stream):
    vs, sc = v[:, l0:l1], scores[:, l0:l1]
    _lib.check(_lib.lib.palu_softmax_pv_f16(sc.data_ptr(), scores.stride(0), 0, vs.data_ptr(), v.stride(0), v.stride(1),
                                            ctx[k].data_ptr(), 0, 0, ws[k].data_ptr(), H, G, l1 - l0, Rv, math.sqrt(128.0),
                                            stream.cuda_stream), "pv")


def timeit(fn, n=50):
    """hipGraph replay of one capture of fn (issued from s1, side work on s2): no host launch cost in the timing."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def sequential():
    abx(0, L, s1)
    pv(0, L, 0, s1)


def make_split(nparts):
    cuts = [L * i // nparts // 1024 * 1024 for i in range(nparts)] + [L]

    def run():
        s2.wait_stream(s1)
        for i in range(nparts):
            abx(cuts[i], cuts[i + 1], s1)
            ev = torch.cuda.Event()
            ev.record(s1)
            s2.wait_event(ev)
            pv(cuts[i], cuts[i + 1], i % 2, s2)      # timing only: partial contexts are not merged here
        s1.wait_stream(s2)
    return run


print(f"sequential abx + softmax.PV : {timeit(sequential):7.1f} us")
for n in (2, 3, 4):
    print(f"{n}-way split, two streams    : {timeit(make_split(n)):7.1f} us")
