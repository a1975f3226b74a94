#!/usr/bin/env python3
"""The score kernel as a whole model sees it: L2 / MALL flushed between launches (a 1 GB copy stands in for the ~400 MB of
weights and ~270 MB of V latents that pass between two layers' score launches), one launch between two events.

    python tools/time_abx_cold.py [L]

Back-to-back loops (tools/diag_split.py time, bench.py's per-kernel numbers) keep the B fragments, the coefficient table and
the page-table entries of both hot; a decode step of a 32-layer model does not.  Prints the median / min of 60 single launches
per form, cold and hot (no flush), on the same inputs."""
import sys
import torch
from palu_amd.kernel import abx_rope as AR
from palu_amd.kernel.abx_rope import abx, rope_inv_freq

dev = torch.device("cuda:0")
H, G, R, D = 32, 8, 128, 128
L = int(sys.argv[1]) if len(sys.argv) > 1 else 65537
g = torch.Generator().manual_seed(1)
a = torch.randn(H, 1, D, generator=g).half().to(dev)
b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half().to(dev)
x = torch.randn(G, L, R, generator=g).half().to(dev)
out = torch.empty(H, 1, L, device=dev, dtype=torch.float16)
rope_inv_freq(dev)
big = torch.empty(1 << 29, device=dev, dtype=torch.float16)
half = big.numel() // 2


def one(fn, flush):
    ts = []
    for _ in range(64):
        if flush:
            big[:half].copy_(big[half:])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[4:])
    return ts[len(ts) // 2], ts[0]


def run(name):
    fn = lambda: abx(a, b, x, out=out)
    c = one(fn, True)
    h = one(fn, False)
    print(f"{name:28s} cold (median, min) {c[0]:7.2f} {c[1]:7.2f} us   hot {h[0]:7.2f} {h[1]:7.2f} us", flush=True)


print(f"H={H} G={G} R={R} L={L}")
for rep in range(2):
    run("default")
    if hasattr(AR, "pair_split"):
        with AR.pair_split():
            run("pair-split")
    with AR.one_band():
        run("one-band")
