#!/usr/bin/env python3
"""Score kernel, us per launch back to back, by form: fp16 rows through the position-split and the pair-split two-band kernels, packed
4- / 3-bit rows (pair-split) -- per (rank per group, cached positions); 32 heads, 8 groups.   time_abx_forms.py [R L ...]"""
import sys
import torch
from palu_amd import _lib
from palu_amd.kernel import quant as q
from palu_amd.kernel.abx_rope import abx, pair_split, prepare_b, rope_inv_freq

H, G, D = 32, 8, 128
torch.manual_seed(0)
args = [int(v) for v in sys.argv[1:]] or [64, 131073, 128, 65537]


def t(fn, n=30):
    for _ in range(5):
        fn()
    best = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    return min(best)


for R, L in zip(args[0::2], args[1::2]):
    a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
    b = (torch.randn(H, R, D, device="cuda") * R ** -0.5).half()
    x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
    out = torch.empty(H, 1, (L + 7) // 8 * 8, device="cuda", dtype=torch.float16)[:, :, :L]
    inv = rope_inv_freq(x.device)
    frag = prepare_b(b, G)
    s = torch.cuda.current_stream().cuda_stream
    row = "R = %3d  L = %6d:  fp16 position-split %.1f" % (R, L, t(lambda: abx(a, b, x, out=out)))
    with pair_split():
        row += "  fp16 pair-split %.1f" % t(lambda: abx(a, b, x, out=out))
    for bits in (4, 3):
        if bits == 3 and R % 32:
            continue
        codes, meta = q.quantize_pack(x, bits)

        def fq():
            _lib.check(_lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(), codes.stride(0),
                                                codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1), out.data_ptr(), out.stride(0),
                                                H, G, L, R, D, bits, inv.data_ptr(), 0, s), "abx_q")
        row += "  %d-bit %.1f" % (bits, t(fq))
        del codes, meta
    print(row, flush=True)
