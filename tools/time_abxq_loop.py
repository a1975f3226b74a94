#!/usr/bin/env python3
"""us per palu_abx_rope_q launch (packed latents, 32 heads, 8 groups):  time_abxq_loop.py BITS R L [BITS R L ...]"""
import sys
import torch
from palu_amd import _lib
from palu_amd.kernel import quant as q
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

H, G, D = 32, 8, 128
torch.manual_seed(0)
args = [int(v) for v in sys.argv[1:]] or [3, 64, 131072]
for bits, R, L in zip(args[0::3], args[1::3], args[2::3]):
    a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
    b = (torch.randn(H, R, D, device="cuda") * R ** -0.5).half()
    x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
    codes, meta = q.quantize_pack(x, bits)
    frag = prepare_b(b, G)
    inv = rope_inv_freq(x.device)
    out = torch.empty(H, 1, L, device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream().cuda_stream
    f = lambda: _lib.check(_lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                                    codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                                    out.data_ptr(), out.stride(0), H, G, L, R, D, bits, inv.data_ptr(), 0, s), "abx_q")
    for _ in range(5):
        f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 30)
    ts.sort()
    two = _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, 0)
    print(f"bits={bits} R={R} L={L}: median {ts[2]:.2f} us, min {ts[0]:.2f} us (two-band rules met: {two})")
