#!/usr/bin/env python3
"""The score kernel at ranks outside {32, 64, 128} -- the ranks the rank search emits (palu/rank_search.py:11-17) and the
reference test's 512: us, MFMA TFLOP/s on the useful flops and HBM fraction, fp16 and packed latents.  Round 2 ran all of
them on the chunked kernel (128-column chunks, fragments re-read per chunk, spilling); since round 3 they are column
windows of the 128-column fast kernel (masked staging, fp32 accumulation across windows).  VERDICT r2 item 8."""
import math
import torch
from palu_amd import _lib
from palu_amd.kernel import quant as q
from palu_amd.kernel.abx_rope import abx, prepare_b, rope_inv_freq

H, G, D, L = 32, 8, 128, 65536
torch.manual_seed(0)
dev = "cuda"
a = torch.randn(H, 1, D, device=dev, dtype=torch.float16)
inv = rope_inv_freq(torch.device(dev))
lib = _lib.lib


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for R in (96, 128, 160, 224, 256, 512):
    b = (torch.randn(H, R, D, device=dev) * R ** -0.5).half()
    x = torch.randn(G, L, R, device=dev, dtype=torch.float16)
    out = torch.empty(H, 1, L, device=dev, dtype=torch.float16)
    us = t(lambda: abx(a, b, x, out=out))
    fl = 2.0 * H * L * R * D
    by = 2 * G * L * R + 2 * H * R * D + 2 * H * L
    line = f"R={R:4d} fp16: {us:8.1f} us = {fl / us * 1e-6:7.1f} TFLOP/s, {by / us * 1e-3 / 8000:.3f} of HBM peak"
    frag = prepare_b(b, G)
    for bits in (4, 3):
        if bits == 3 and R % 32:
            continue
        codes, meta = q.quantize_pack(x, bits)
        nscr = lib.palu_abx_scratch_bytes(H, G, L, R)
        scr = torch.empty(max(nscr, 16), dtype=torch.uint8, device=dev)
        fq = lambda: _lib.check(lib.palu_abx_rope_qg(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                                     codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0),
                                                     meta.stride(1), out.data_ptr(), out.stride(0), H, G, L, R, 128, bits, 0,
                                                     inv.data_ptr(), 0, scr.data_ptr() if nscr else 0,
                                                     _lib.current_stream()), "abx_qg")
        uq = t(fq)
        line += f" | {bits}-bit {uq:8.1f} us"
    print(line, flush=True)
