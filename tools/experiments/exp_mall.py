#!/usr/bin/env python3
"""Experiment: does the memory-side cache (Infinity Cache / MALL) serve a repeated stream?  softmax.PV rate at several
cache sizes when the same V is re-read back to back, and after a "toucher" pass over part of V."""
import math
import torch
from palu_amd import _lib
H, G, Rv = 32, 8, 384
lib = _lib.lib


def rate(L, touch_frac=None, n=50):
    scores = (torch.randn(H, L + 7, device="cuda") * 10).half()
    v = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16)
    other = torch.randn(256 * 1024 * 1024 // 2, device="cuda", dtype=torch.float16)   # 512 MB flusher
    ws = torch.empty(lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
    ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
    s = _lib.current_stream()

    def f():
        _lib.check(lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1),
                                           ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, math.sqrt(128.0), s), "pv")
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        if touch_frac is not None:
            other.abs_()                      # flush: stream 512 MB through the caches (read+write)
            if touch_frac > 0:
                rows = int(L * touch_frac)
                v[:, :rows].amax()            # toucher: read the first part of every group's V
        evs[i][0].record()
        f()
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    us = ts[len(ts) // 2]
    return us, 2 * G * L * Rv / us * 1e-3


for L in (4096, 8192, 16384, 32768, 65536):
    us, gbps = rate(L)
    print(f"back-to-back re-read  L={L:6d} ({2*G*L*Rv/1e6:6.1f} MB): {us:7.1f} us  {gbps:7.0f} GB/s")
for frac in (0.0, 0.25, 0.5, 0.75, 1.0):
    us, gbps = rate(65536, touch_frac=frac, n=20)
    print(f"flush, touch {frac:4.2f} of V, then PV (L=65536): {us:7.1f} us  {gbps:7.0f} GB/s")
