#!/usr/bin/env python3
"""softmax.PV on fp16 latent rows: the VALU streaming kernel vs the register-direct matrix-core kernel (PALU_PV_DIRECT), per
(latent groups, cached positions); 4 heads per group, rank_v / G = 384.   python tools/time_pv_forms.py [--child G L]"""
import math, os, subprocess, sys


def child(G, L):
    import torch
    from palu_amd import _lib
    H, Rv, D = 4 * G, 384, 128
    torch.manual_seed(0)
    sc = torch.randn(H, L + 7, device="cuda", dtype=torch.float16)[:, :L]
    v = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16)
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
    ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")

    def pv():
        _lib.check(_lib.lib.palu_softmax_pv_f16(sc.data_ptr(), sc.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(), 0, 0,
                                                ws.data_ptr(), H, G, L, Rv, math.sqrt(D), _lib.current_stream()), "pv")
    for _ in range(5):
        pv()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            pv()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / 30)
    print("%.1f" % min(best))


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)
print("G      L      VALU us  direct us   (default selection: us)")
for G in (1, 2, 4, 8):
    for L in (16385, 65537, 131073, 262145):
        if G * L > 8 * 65537 * 2:
            continue
        r = []
        for d in ("0", "1", None):
            env = dict(os.environ)
            env.pop("PALU_PV_DIRECT", None)
            if d is not None:
                env["PALU_PV_DIRECT"] = d
            out = subprocess.run([sys.executable, __file__, "--child", str(G), str(L)], env=env, capture_output=True, text=True)
            r.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "fail")
        print("%d %8d %8s %8s   %8s" % (G, L, r[0], r[1], r[2]), flush=True)
