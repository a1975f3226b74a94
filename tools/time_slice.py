#!/usr/bin/env python3
"""One rank's share of a head-group-parallel decode step (qkv of its heads + attention core, before the all-gather), for a
tensor-parallel degree tp in {8, 4, 2} of the 8-group model (G = 8 / tp latent groups per rank), under PALU_PV_DIRECT unset / 0 / 1.
   python tools/time_slice.py            (runs every (tp, L, setting) in a child process)"""
import os, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(tp, Lp):
    import torch
    import bench
    from palu_amd.kernel import head_parallel as hp
    dev = torch.device("cuda:0")
    H, G, D, HIDDEN, Rk, Rv = 32, 8, 128, 4096, 128, 384
    plan = hp.make_plan(tp, 0, H, G, D, Rk, Rv)
    torch.manual_seed(99)
    full = {"wq": (torch.randn(H * D, HIDDEN, device=dev) / 64).half(), "vt_k": (torch.randn(G * Rk, HIDDEN, device=dev) / 64).half(),
            "vt_v": (torch.randn(G * Rv, HIDDEN, device=dev) / 64).half(), "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": torch.zeros(1, 1, device=dev).half()}
    w = {k_: v_.contiguous() for k_, v_ in hp.shard_weights(plan, full).items()}
    cap = Lp + 64
    kc = torch.randn(G // tp, cap, Rk, device=dev, dtype=torch.float16)
    vc = torch.randn(G // tp, cap, Rv, device=dev, dtype=torch.float16)
    hidden = torch.randn(HIDDEN, device=dev, dtype=torch.float16)
    dec = hp.HeadParallelDecoder(plan, w, kc, vc, HIDDEN)
    us, form = bench.best_form(lambda: dec.local_step(hidden, Lp, Lp), 60)
    print("%.1f" % us)


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)
print("tp  G/rank       L    auto us   VALU us  direct us")
for tp in (8, 4, 2):
    for Lp in (65536, 262144):
        if (8 // tp) * Lp > 4 * 65536 * 2:
            continue
        r = []
        for d in (None, "0", "1"):
            env = dict(os.environ)
            env.pop("PALU_PV_DIRECT", None)
            if d is not None:
                env["PALU_PV_DIRECT"] = d
            out = subprocess.run([sys.executable, __file__, "--child", str(tp), str(Lp)], env=env, capture_output=True, text=True)
            r.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "fail:" + out.stderr.strip()[-120:])
        print("%d %6d %9d %9s %9s %9s" % (tp, 8 // tp, Lp, r[0], r[1], r[2]), flush=True)
