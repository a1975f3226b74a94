#!/usr/bin/env python3
"""Round-3 experiment: since the score kernel (abx) went from 248 to 192 VGPRs, ONE softmax.PV workgroup (256 threads,
126 VGPRs, < 9 KB LDS after the per-head phase C) fits on a CU beside an abx workgroup.  Do the two kernels then overlap
(abx is MFMA/issue-bound and leaves HBM 3/4 idle, P.V is a pure stream)?
  seq        abx(L) -> pv(L) on one stream (what the step does)
  indep      abx(L) on stream 1 || pv(L) on stream 2 over independent data (upper bound of the overlap)
  pipeN      the step's REAL dependency chain cut into N row slices: abx(s) on stream 1, pv(s) on stream 2 after abx(s);
             pv(s) therefore runs beside abx(s+1).  (A production version would LSE-merge the N partial contexts.)
Direct launches only: hipGraph replays serialise branches on this stack (tools/experiments/README.md, round 2)."""
import math
import sys
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

H, G, D, R, Rv = 32, 8, 128, 128, 384
L = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(0)
dev = "cuda"
a = torch.randn(H, 1, D, device=dev, dtype=torch.float16)
b = (torch.randn(H, R, D, device=dev) * R ** -0.5).half()
x = torch.randn(G, L, R, device=dev, dtype=torch.float16)
v = torch.randn(G, L, Rv, device=dev, dtype=torch.float16)
scores = torch.empty(H, L + 8, device=dev, dtype=torch.float16)
scores2 = (torch.randn(H, L + 8, device=dev) * 3).half()
frag = prepare_b(b, G)
inv = rope_inv_freq(torch.device(dev))
lib = _lib.lib
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NS = 8
ctx = [torch.empty(H, Rv, dtype=torch.float16, device=dev) for _ in range(NS)]
ws = [torch.empty(lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=dev) for _ in range(NS)]


def abx(stream, l0, n, out=scores):
    xs = x[:, l0:l0 + n]
    _lib.check(lib.palu_abx_rope_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), xs.data_ptr(), x.stride(0),
                                     x.stride(1), out[:, l0:].data_ptr(), out.stride(0), H, G, n, R, D, inv.data_ptr(), l0,
                                     stream.cuda_stream), "abx")


def pv(stream, l0, n, k=0, sc=scores):
    vs = v[:, l0:l0 + n]
    _lib.check(lib.palu_softmax_pv_f16(sc[:, l0:].data_ptr(), sc.stride(0), 0, vs.data_ptr(), v.stride(0), v.stride(1),
                                       ctx[k].data_ptr(), 0, 0, ws[k].data_ptr(), H, G, n, Rv, math.sqrt(128.0),
                                       stream.cuda_stream), "pv")


def seq():
    abx(s1, 0, L)
    pv(s1, 0, L)


def indep():
    s2.wait_stream(s1)
    abx(s1, 0, L)
    pv(s2, 0, L, sc=scores2)
    s1.wait_stream(s2)


def pipe(n):
    bounds = [(-(-L // n) + 127) // 128 * 128 * i for i in range(n)] + [L]
    bounds = [min(b_, L) for b_ in bounds]
    def run():
        s2.wait_stream(s1)
        for i in range(n):
            l0, l1 = bounds[i], bounds[i + 1]
            if l1 <= l0:
                continue
            abx(s1, l0, l1 - l0)
            s2.wait_stream(s1)
            pv(s2, l0, l1 - l0, k=i)
        s1.wait_stream(s2)
    return run


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000)          # (on the default stream; s1 waits for it below)
        s1.wait_stream(torch.cuda.current_stream())
        e0.record(s1)
        for _ in range(n):
            fn()
        e1.record(s1)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


for rep in range(2):
    print(f"L={L}: abx alone {timeit(lambda: abx(s1, 0, L)):6.1f}  pv alone {timeit(lambda: pv(s1, 0, L)):6.1f}  "
          f"seq {timeit(seq):6.1f}  indep(abx || pv) {timeit(indep):6.1f}  "
          f"pipe2 {timeit(pipe(2)):6.1f}  pipe3 {timeit(pipe(3)):6.1f}  pipe4 {timeit(pipe(4)):6.1f}  pipe6 {timeit(pipe(6)):6.1f} us",
          flush=True)
