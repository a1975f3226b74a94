#!/usr/bin/env python3
"""Position-split two-band score kernel (csrc/abx_rope3_kernel.h) against an fp64 evaluation of the same scores on the GPU
and against the pair-split kernel (csrc/abx_rope2_kernel.h), band by band; A/B timings at the bench shapes.

    python tools/diag_split.py [check] [time] [timeline]

`timeline` needs a -DPALU_EXPERIMENTS build of the library (PALU_HIP_LIB=...): per-wave s_memtime stamps of one launch."""
import ctypes as C
import sys
import numpy as np
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, rope_inv_freq, pair_split, one_band

_lib.lib.palu_abx_set_position_split(-1)      # every shape the kernel takes, not only where it is the default

dev = torch.device("cuda:0")
D = 128
inv = rope_inv_freq(dev)
what = sys.argv[1:] or ["check", "time"]


def scores_f64(a, b, x, pos0=0):
    """fp64 scores with the oracle's fp32-rounded angles (oracle.abx_scores_f64, evaluated on the GPU in row chunks)."""
    H, R, _ = b.shape
    G, L, _ = x.shape
    gs = H // G
    out = torch.empty(H, L, dtype=torch.float64, device=dev)
    bd = b.double().reshape(G, gs, R, D)
    ad = a.double().reshape(G, gs, D)
    for l0 in range(0, L, 8192):
        l1 = min(L, l0 + 8192)
        keys = torch.matmul(x[:, None, l0:l1].double(), bd)
        pos = torch.arange(pos0 + l0, pos0 + l1, device=dev, dtype=torch.int64).to(torch.float32)
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
    nb = int((d > 2e-3 * mx).sum())
    rms = float((d ** 2).mean().sqrt()) / mx
    flag = "" if nb == 0 else f"   <-- {nb} elements off by > 2e-3 (first bad l: {int((d > 2e-3 * mx).any(dim=0).nonzero()[0])})"
    print(f"  {tag:11s} max {float(d.max()) / mx:.2e} (h {h}, l {l}, l%128 {l % 128})  rms {rms:.2e}{flag}")
    return nb


def inputs(H, G, R, L, seed, band="all"):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(H, 1, D, generator=g)
    if band == "high":
        a[:, :, 32:64] = 0
        a[:, :, 96:128] = 0
    elif band == "low":
        a[:, :, 0:32] = 0
        a[:, :, 64:96] = 0
    b = torch.randn(H, R, D, generator=g) * R ** -0.5
    x = torch.randn(G, L, R, generator=g)
    return a.half().to(dev), b.half().to(dev), x.half().to(dev)


bad = 0
if "check" in what:
    for (H, G, R, L) in [(32, 8, 128, 128), (32, 8, 128, 1), (32, 8, 128, 33), (32, 8, 128, 129), (32, 8, 128, 255), (32, 8, 128, 1000),
                         (32, 8, 128, 4096 + 97), (4, 1, 128, 8191), (32, 8, 128, 65537), (32, 8, 64, 2113), (32, 8, 64, 131073),
                         (32, 8, 32, 2048), (8, 2, 64, 5000), (4, 1, 128, 200000)]:
        for band in (("high", "low", "all") if L <= 1000 else ("all",)):
            a, b, x = inputs(H, G, R, L, 7 * L + R, band)
            assert _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, 0) == 1
            ref = scores_f64(a, b, x)
            print(f"H={H} G={G} R={R} L={L} band={band}")
            y3 = abx(a, b, x).reshape(H, L)
            bad += report("split", y3, ref)
            with pair_split():
                y2 = abx(a, b, x).reshape(H, L)
            report("pair-split", y2, ref)
            y3b = abx(a, b, x).reshape(H, L)
            if not torch.equal(y3, y3b):
                print("  !! second launch differs from the first:", int((y3 != y3b).sum()), "elements")
                bad += 1
    # position offset in whole tiles
    a, b, x = inputs(32, 8, 128, 1500, 3)
    full = abx(a, b, x)
    part = abx(a, b, x[:, 384:].contiguous(), pos_offset=384)
    d = float((part.float() - full[:, :, 384:].float()).abs().max()) / float(full.float().abs().max())
    print(f"pos_offset=384 vs full: {d:.2e}")
    bad += d > 5e-4
    # strided query / cache view (row stride > R)
    xb = torch.randn(8, 700, 160, device=dev).half()
    xv = xb[:, :, :128]
    ref = scores_f64(a, b, xv.contiguous())
    print("strided rows (sx_l = 160)")
    bad += report("split", abx(a, b, xv).reshape(32, 700), ref)
    print("CHECK", "FAILED" if bad else "ok")

if "time" in what:
    def timeit(fn, n=30, reps=5):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        ts.sort()
        return ts[0], ts[len(ts) // 2]
    for (H, G, R, L) in [(32, 8, 128, 65537), (32, 8, 128, 65536), (32, 8, 64, 131073), (32, 8, 32, 65536), (32, 8, 128, 32768), (32, 8, 128, 16384),
                         (32, 8, 128, 4096), (4, 1, 128, 200000)]:
        a, b, x = inputs(H, G, R, L, 1)
        out = torch.empty(H, 1, L, device=dev, dtype=torch.float16)
        for rep in range(2):
            t3 = timeit(lambda: abx(a, b, x, out=out))
            with pair_split():
                t2 = timeit(lambda: abx(a, b, x, out=out))
            print(f"time H={H} G={G} R={R} L={L}: split (min, med) {t3[0]:.2f} {t3[1]:.2f} us   pair-split {t2[0]:.2f} {t2[1]:.2f} us")
    # zero-filled latents: the same instruction stream at (much) lower switching power -- how far the real run is from the clock ceiling
    a, b, x = inputs(32, 8, 128, 65536, 1)
    out = torch.empty(32, 1, 65536, device=dev, dtype=torch.float16)
    xz = torch.zeros_like(x)
    t3 = timeit(lambda: abx(a, b, xz, out=out))
    with pair_split():
        t2 = timeit(lambda: abx(a, b, xz, out=out))
    print(f"time ZERO latents C2: split {t3[0]:.2f} {t3[1]:.2f} us   pair-split {t2[0]:.2f} {t2[1]:.2f} us")

if "timeline" in what:
    fn = getattr(_lib.lib, "palu_abx3_timeline_buffer", None)
    if fn is None:
        print("timeline: the library has no palu_abx3_timeline_buffer (build with -DPALU_EXPERIMENTS)")
        sys.exit(0)
    fn.restype = None
    fn.argtypes = [C.c_void_p]
    from palu_amd.kernel.abx_rope import in_kernel_fold
    import contextlib
    for (H, G, R, L, prefold) in [(32, 8, 128, 65536, True), (32, 8, 128, 65536, False), (32, 8, 64, 131072, True)]:
      with (contextlib.nullcontext() if prefold else in_kernel_fold()):
        a, b, x = inputs(H, G, R, L, 1)
        out = torch.empty(H, 1, L, device=dev, dtype=torch.float16)
        nwg = 256
        print("fold:", "once per launch (fold kernel in front)" if prefold else "in every workgroup's prologue")
        dbg = torch.zeros(nwg * 4 * 64, dtype=torch.int64, device=dev)
        for _ in range(3):
            abx(a, b, x, out=out)
        for it in range(3):
            dbg.zero_()
            fn(dbg.data_ptr())
            abx(a, b, x, out=out)
            torch.cuda.synchronize()
            fn(None)
        d = dbg.cpu().numpy().reshape(nwg, 4, 64).astype(np.int64)
        nst = int((d[0, 0] != 0).sum())
        rel = d - d[:, :1, :1]
        names = (["start", "low frags requested", "low frags + tables in", "high frags requested, rope", "barrier A", "first W image",
                  "high frags in (barrier B)", "frags in AGPRs (barrier C)"] if prefold else
                 ["start", "loads issued", "query in LDS", "rope init", "folds done", "frags in AGPRs", "first W image", "first block landed"])
        print(f"== timeline R={R} L={L}: {nst} stamps per wave; s_memtime ticks relative to the workgroup's wave 0 start, mean over workgroups [min .. max]")
        for i in range(nst):
            nm = names[i] if i < len(names) else (f"block {i - 8}" if i < nst - 2 else ("drain" if i == nst - 2 else "end"))
            col = rel[:, :, i]
            print(f"  {i:2d} {nm:18s} " + " ".join(f"w{k} {col[:, k].mean():8.0f}" for k in range(4)) + f"   [{col.min()} .. {col.max()}]")
        blk = np.diff(rel[:, :, 8:nst - 1], axis=2)
        print("  block durations (ticks), mean over waves:", np.round(blk.mean(axis=(0, 1))).astype(int).tolist())
        print("  per block-in-tile (0..3):", [int(blk[:, :, k::4].mean()) for k in range(4)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(dbg.data_ptr())
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            abx(a, b, x, out=out)
        e1.record()
        torch.cuda.synchronize()
        fn(None)
        us = e0.elapsed_time(e1) * 1e3 / 20
        tot = rel[:, :, nst - 1].max(axis=1).mean()
        print(f"  wall {us:.1f} us per launch (timing build) -> {tot / us * 1e-3:.2f} ticks/ns; total ticks (slowest wave of a workgroup, mean) {tot:.0f}")
