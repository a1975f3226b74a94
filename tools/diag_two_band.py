#!/usr/bin/env python3
"""Two-band score kernel (csrc/abx_rope2_kernel.h) against an fp64 evaluation of the same scores on the GPU and against the
one-band kernel, band by band, plus timings of both at the bench shapes.

    python tools/diag_two_band.py [quick]

Isolation without a debug build: a query whose low-band components (d in 32..63, 96..127) are zero exercises the high band
alone, and vice versa.  Errors are printed as max |diff| / max |score| with the position (mod 128) and head of the worst
element, so a wrong fragment / polynomial term / pipeline slot shows up as a pattern."""
import sys
import numpy as np
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, prepare_b, rope_inv_freq, _rope_tables, ROPE_TABLE_POSITIONS
from palu_amd.kernel import quant as pq

dev = torch.device("cuda:0")
D = 128
inv = rope_inv_freq(dev)
inv_host = inv.cpu()
props = torch.cuda.get_device_properties(0)
print("device:", props.name, "CUs", props.multi_processor_count, "shared/block", getattr(props, "shared_memory_per_block", "?"),
      "optin", getattr(props, "shared_memory_per_block_optin", "?"))


def set_two_band(on: bool):
    if on:
        tab = _rope_tables[inv.data_ptr()][0]
        _lib.check(_lib.lib.palu_rope_table_register(inv.data_ptr(), tab.data_ptr(), 0, ROPE_TABLE_POSITIONS, float(inv_host[32])), "reg")
    else:
        _lib.lib.palu_rope_table_unregister(inv.data_ptr())


def scores_f64(a, b, x):
    """fp64 scores with the oracle's fp32-rounded angles (oracle.abx_scores_f64, evaluated on the GPU in row chunks)."""
    H, R, _ = b.shape
    G, L, _ = x.shape
    gs = H // G
    out = torch.empty(H, L, dtype=torch.float64, device=dev)
    bd = b.double().reshape(G, gs, R, D)
    ad = a.double().reshape(G, gs, D)
    for l0 in range(0, L, 8192):
        l1 = min(L, l0 + 8192)
        keys = torch.matmul(x[:, None, l0:l1].double(), bd)                       # [G,gs,l,D]
        pos = torch.arange(l0, l1, device=dev, dtype=torch.int64).to(torch.float32)
        ang = torch.outer(pos, inv).double()
        c, s = ang.cos(), ang.sin()
        k1, k2 = keys[..., :64], keys[..., 64:]
        rot = torch.cat((k1 * c - k2 * s, k2 * c + k1 * s), dim=-1)
        out[:, l0:l1] = torch.einsum("ghd,ghld->ghl", ad, rot).reshape(H, l1 - l0)
    return out


def report(tag, y, ref):
    d = (y.double() - ref).abs()
    mx = float(ref.abs().max())
    i = int(d.argmax())
    h, l = divmod(i, ref.shape[1])
    bad = d > 2e-3 * mx
    nb = int(bad.sum())
    msg = f"{tag}: max|d|/max = {float(d.max()) / mx:.2e} at head {h} pos {l} (pos%128={l % 128})  n(>2e-3)={nb}"
    if nb:
        bl = torch.nonzero(bad)
        msg += f"  bad heads {sorted(set(bl[:, 0].tolist()))[:8]} bad pos%128 blocks {sorted(set(((bl[:, 1] % 128) // 32).tolist()))}" \
               f" first bad pos {int(bl[:, 1].min())} last {int(bl[:, 1].max())}"
    print(msg, flush=True)
    return nb == 0


def run_case(H, G, R, L, seed, bits=0, band="all"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(H, 1, D, generator=g).half()
    if band == "high":
        a[:, :, 32:64] = 0
        a[:, :, 96:128] = 0
    elif band == "low":
        a[:, :, 0:32] = 0
        a[:, :, 64:96] = 0
    b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half()
    x = torch.randn(G, L, R, generator=g).half()
    a, b, x = a.to(dev), b.to(dev), x.to(dev)
    if bits:
        codes, meta = pq.quantize_pack(x, bits)
        x = pq.unpack_dequant(codes, meta, bits, R)
    ref = scores_f64(a, b, x)

    def launch():
        if not bits:
            return abx(a, b, x).reshape(H, L)
        out = torch.empty(H, 1, L, dtype=torch.float16, device=dev)
        frag = prepare_b(b, G)
        _lib.check(_lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                            codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                            out.data_ptr(), out.stride(0), H, G, L, R, 128, bits, inv.data_ptr(), 0,
                                            torch.cuda.current_stream().cuda_stream), "abx_q")
        return out.reshape(H, L)
    sel = _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, 0)
    y2 = launch()
    torch.cuda.synchronize()
    set_two_band(False)
    y1 = launch()
    torch.cuda.synchronize()
    set_two_band(True)
    tag = f"H={H} G={G} R={R} L={L} bits={bits} band={band} two_band_selected={sel}"
    ok1 = report(tag + " | one-band vs f64", y1, ref)
    ok2 = report(tag + " | TWO-band vs f64", y2, ref)
    e1 = float(((y1.double() - ref) ** 2).mean().sqrt())
    e2 = float(((y2.double() - ref) ** 2).mean().sqrt())
    print(f"    rms err one-band {e1:.3e}  two-band {e2:.3e}  ratio {e2 / max(e1, 1e-30):.2f}", flush=True)
    return ok2


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return min(ts), sorted(ts)[len(ts) // 2]


def time_case(H, G, R, L, bits=0):
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn(H, 1, D, generator=g).half().to(dev)
    b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half().to(dev)
    x = torch.randn(G, L, R, generator=g).half().to(dev)
    out = torch.empty(H, 1, L, dtype=torch.float16, device=dev)
    frag = prepare_b(b, G)
    S = torch.cuda.current_stream().cuda_stream
    if bits:
        codes, meta = pq.quantize_pack(x, bits)
        fn = lambda: _lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                              codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                              out.data_ptr(), out.stride(0), H, G, L, R, 128, bits, inv.data_ptr(), 0, S)
    else:
        fn = lambda: _lib.lib.palu_abx_rope_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), x.data_ptr(), x.stride(0),
                                                x.stride(1), out.data_ptr(), out.stride(0), H, G, L, R, 128, inv.data_ptr(), 0, S)
    res = {}
    for rep in range(2):
        for on in (True, False):
            set_two_band(on)
            res.setdefault(on, []).append(timeit(fn))
    set_two_band(True)
    print(f"time H={H} G={G} R={R} L={L} bits={bits}: two-band (min, med) {res[True]}  one-band {res[False]}", flush=True)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
if len(sys.argv) > 1 and sys.argv[1] == "time":
    time_case(32, 8, 128, 65537)
    time_case(32, 8, 64, 131073, bits=4)
    time_case(32, 8, 128, 65537, bits=3)
    sys.exit(0)
ok = True
for band in ("high", "low", "all"):
    ok &= run_case(32, 8, 128, 384, 0, band=band)
for (H, G, R, L) in ((32, 8, 128, 64), (32, 8, 128, 129), (32, 8, 128, 4096 + 33), (4, 1, 128, 1000), (32, 8, 64, 2048 + 65), (32, 8, 32, 2048),
                     (32, 8, 128, 65537), (32, 8, 64, 131073)):
    ok &= run_case(H, G, R, L, 1)
if not quick:
    ok &= run_case(32, 8, 128, 65537, 2, band="low")
    ok &= run_case(32, 8, 128, 4096 + 97, 3, bits=3)
    ok &= run_case(32, 8, 64, 4096 + 1, 3, bits=4)
    ok &= run_case(32, 8, 128, 65537, 4, bits=3)
    ok &= run_case(32, 8, 64, 131073, 4, bits=4)
print("ALL OK" if ok else "FAILURES", flush=True)
time_case(32, 8, 128, 65537)
time_case(32, 8, 64, 131073, bits=4)
time_case(32, 8, 128, 65537, bits=3)
