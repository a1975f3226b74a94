#!/usr/bin/env python3
"""debug: quantised P.V kernel vs fp64 on dequantised values; prints the mismatching column pattern"""
import math, sys, torch, numpy as np
from palu_amd import _lib
from palu_amd.kernel import quant as q
DEV = "cuda"
def run(bits, Rv, L, H=32, gs=4):
    rng = np.random.default_rng(bits + Rv + L)
    G = H // gs
    scores = torch.from_numpy((rng.standard_normal((H, L)) * 20).astype(np.float16)).to(DEV)
    v = torch.from_numpy((rng.standard_normal((G, L, Rv)) * rng.uniform(0.2, 3, (G, L, 1))).astype(np.float16)).to(DEV)
    codes, meta, deq = q.quantize_pack(v, bits, want_dequant=True)
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, codes.data_ptr(), codes.stride(0),
                                          codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1), ctx.data_ptr(),
                                          0, 0, ws.data_ptr(), H, G, L, Rv, bits, math.sqrt(128.0), _lib.current_stream()), "pv_q")
    x = (scores.cpu() / math.sqrt(128.0))
    p64 = torch.softmax(x.double(), dim=-1)
    c64 = torch.matmul(p64.reshape(G, gs, L), deq.cpu().double()).reshape(H, Rv)
    err = (ctx.cpu().double() - c64).abs()
    bad = (err > 2e-3 * max(1.0, c64.abs().max().item()))
    cols = sorted(set(np.nonzero(bad.numpy())[1].tolist()))
    print(f"bits={bits} Rv={Rv} L={L}: max err {err.max():.3e}  bad cols ({len(cols)}): {cols[:48]}")
for bits in (4, 3):
    for Rv in (128, 192, 256, 384):
        run(bits, Rv, 700)
run(3, 384, 1500); run(3, 192, 40)
