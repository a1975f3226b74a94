#!/usr/bin/env python3
"""The two kernels of the attention core for ONE latent group (an 8-GPU head-group shard, BASELINE config 5): us each."""
import math, sys
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, rope_inv_freq, one_band

L = int(sys.argv[1]) if len(sys.argv) > 1 else 262145
H, G, R, Rv, D = 4, 1, 128, 384, 128
torch.manual_seed(0)
a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
b = (torch.randn(H, R, D, device="cuda") * R ** -0.5).half()
x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
v = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16)
out = torch.empty(H, 1, L + 7, device="cuda", dtype=torch.float16)[:, :, :L]
inv = rope_inv_freq(x.device)
ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
sc = out.reshape(H, L)


def t(fn, n=30):
    for _ in range(5):
        fn()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    return min(best)


def pv():
    _lib.check(_lib.lib.palu_softmax_pv_f16(sc.data_ptr(), sc.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(), 0, 0,
                                            ws.data_ptr(), H, G, L, Rv, math.sqrt(D), _lib.current_stream()), "pv")


print("L =", L, " two-band selected:", _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, 0))
print("abx          %.1f us" % t(lambda: abx(a, b, x, out=out)))
with one_band():
    print("abx one-band %.1f us" % t(lambda: abx(a, b, x, out=out)))
print("softmax.PV   %.1f us (nsplit %d)" % (t(pv), _lib.lib.palu_pv_nsplit(G, L)))
