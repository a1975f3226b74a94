#!/usr/bin/env python3
"""GPU check of the fused decode attention kernel (palu_decode_attn_f16) against the two-kernel path
(palu_abx_rope_f16 -> palu_softmax_pv_f16) and an fp64 evaluation of the same math, plus timings.

    python tools/check_fused.py [--time]
"""
import argparse
import math
import sys

import torch

from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

lib = _lib.lib
D = 128


def ref_fp64(a, b, x, v, pos0, scores16):
    """softmax(fp16(fp16 scores / sqrt(D))) . V in fp64 from the two-kernel path's fp16 scores (isolates the P.V part)"""
    H, G = a.shape[0], x.shape[0]
    gs = H // G
    xs = (scores16.float() / math.sqrt(D)).half().double()
    p = torch.softmax(xs, dim=-1)
    return torch.einsum("ghl,glr->ghr", p.view(G, gs, -1), v.double()).reshape(H, -1)


_BIG = None


def run(H, G, Rk, Rv, L, pos0=0, scale=1.0, seed=0, time_it=False, ldk=None, ldv=None, cold=False, quiet=False):
    dev = "cuda"
    torch.manual_seed(seed)
    a = (torch.randn(H, D, device=dev) * scale).half()
    b = (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half()
    cap = L + 70
    ldk = ldk or Rk
    ldv = ldv or Rv
    xk = torch.randn(G, cap, ldk, device=dev).half()
    xv = torch.randn(G, cap, ldv, device=dev).half()
    # rows past L are poison: they must never influence the result
    xk[:, L:] = float("nan")
    xv[:, L:] = float("nan")
    k = xk[:, :, :Rk]
    v = xv[:, :, :Rv]
    frag = prepare_b(b, G)
    inv = rope_inv_freq(torch.device(dev))
    s = _lib.current_stream()
    scores = torch.empty(H, (L + 7) // 8 * 8, dtype=torch.float16, device=dev)
    ws = torch.zeros(lib.palu_pv_workspace_bytes(H, G, cap, Rv), dtype=torch.uint8, device=dev)
    ctx0 = torch.empty(H, Rv, dtype=torch.float16, device=dev)
    ctx1 = torch.full((H, Rv), float("nan"), dtype=torch.float16, device=dev)

    def old():
        _lib.check(lib.palu_abx_rope_f16(a.data_ptr(), a.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                         k.stride(1), scores.data_ptr(), scores.stride(0), H, G, L, Rk, D,
                                         inv.data_ptr(), pos0, s), "abx")
        if (H // G) in (1, 2, 4, 8):
            _lib.check(lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0, v.data_ptr(), v.stride(0),
                                               v.stride(1), ctx0.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv,
                                               math.sqrt(D), s), "pv")
        else:       # the two-kernel P.V has no gs = 3 instance: its stand-in is the fp64 evaluation itself
            ctx0.copy_(ref_fp64(a, b, k[:, :L], v[:, :L], pos0, scores[:, :L]).half())

    def new():
        _lib.check(lib.palu_decode_attn_f16(a.data_ptr(), a.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                            k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), ctx1.data_ptr(),
                                            ws.data_ptr(), H, G, L, Rk, Rv, D, inv.data_ptr(), pos0, math.sqrt(D), s),
                   "fused")

    old()
    torch.cuda.synchronize()
    ref = ref_fp64(a, b, k[:, :L], v[:, :L], pos0, scores[:, :L])
    if cold:                                   # 512 MB through the caches: the fused launch starts cold
        global _BIG
        if _BIG is None:
            _BIG = torch.empty(1 << 28, device=dev, dtype=torch.float16)
        _BIG[: 1 << 27].copy_(_BIG[1 << 27:])
    new()
    torch.cuda.synchronize()
    e_old = (ctx0.double() - ref).abs().max().item()
    e_new = (ctx1.double() - ref).abs().max().item()
    d = (ctx1.float() - ctx0.float()).abs().max().item()
    # (cold stress: a stale 32-row unit shows as >= 1e-2 of the largest output; the kernel's own fp16 rounding stays below 3e-3)
    ok = bool(torch.isfinite(ctx1).all()) and e_new <= max((5e-3 if cold else 2e-3) * ref.abs().max().item(), 3 * e_old + 1e-4)
    line = (f"H={H} G={G} Rk={Rk} Rv={Rv} L={L} pos0={pos0} scale={scale}: |ref|max {ref.abs().max():.3f}  "
            f"err two-kernel {e_old:.2e}  err fused {e_new:.2e}  fused-vs-two {d:.2e}  {'OK' if ok else 'FAIL'}")
    if time_it:
        def t(fn, n=50):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        line += f"   two-kernel {t(old):.1f} us  fused {t(new):.1f} us"
    if not (quiet and ok):
        print(line, flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--cold", type=int, default=0, help="N repetitions of ragged shapes on launches that start cold (fresh inputs, "
                    "caches turned over): the static vmcnt schedule of the kernel at the end of a workgroup's range")
    args = ap.parse_args()
    if args.cold:
        bad = n = 0
        for rep in range(args.cold):
            for c in ((4, 1, 128, 384, 3000 + 7 * rep), (4, 1, 128, 384, 65537 + rep), (32, 8, 128, 384, 2900 + 13 * rep),
                      (4, 1, 64, 192, 20011 + rep), (8, 2, 128, 384, 9973 + 5 * rep), (32, 8, 64, 128, 5003 + rep)):
                n += 1
                bad += not run(*c, seed=100 + rep, cold=True, quiet=True)
        print(f"check_fused --cold: {bad} bad of {n} launches")
        sys.exit(1 if bad else 0)
    ok = True
    cases = [
        (32, 8, 128, 384, 64), (32, 8, 128, 384, 65), (32, 8, 128, 384, 1), (32, 8, 128, 384, 129),
        (32, 8, 128, 384, 200), (32, 8, 128, 384, 1000), (32, 8, 128, 384, 4097),
        (4, 1, 128, 384, 3000), (8, 2, 128, 384, 20000), (32, 8, 64, 192, 2500), (32, 8, 128, 256, 777),
        (32, 8, 64, 128, 5000), (24, 8, 128, 384, 1500),
    ]
    for c in cases:
        ok &= run(*c)
    ok &= run(32, 8, 128, 384, 3001, pos0=5000)
    ok &= run(32, 8, 128, 384, 3001, scale=4.0)          # peaked softmax: running-max rescales matter
    ok &= run(32, 8, 128, 384, 2049, ldk=136, ldv=392)   # strided cache rows
    if not args.quick:
        ok &= run(32, 8, 128, 384, 65537, time_it=args.time)
        ok &= run(4, 1, 128, 384, 262144, time_it=args.time)
        ok &= run(32, 8, 64, 192, 131073, time_it=args.time)
    print("check_fused:", "ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
