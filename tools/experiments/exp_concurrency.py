#!/usr/bin/env python3
"""Do two kernels on two streams overlap at all on this box?  (1) two small-footprint torch kernels, (2) abx on 128 CUs
with softmax.PV on another stream, WITHOUT graph capture, timed with wall clock around a full sync."""
import math, os, time
os.environ.setdefault("PALU_ABX_CUS", "128")
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x1 = torch.randn(64, 1024, device="cuda")      # tiny grids: 64 WGs each, long-running through repetition
x2 = torch.randn(64, 1024, device="cuda")


def spin(x, stream, n=200):
    with torch.cuda.stream(stream):
        for _ in range(n):
            x.mul_(1.0001)


def wall(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return min(ts)


a1 = wall(lambda: spin(x1, s1))
b1 = wall(lambda: spin(x2, s2))
ab = wall(lambda: (spin(x1, s1), spin(x2, s2)))
print(f"tiny kernels: stream1 {a1:.0f} us, stream2 {b1:.0f} us, both {ab:.0f} us (launch-bound, only indicative)")

H, G, D, R, Rv, L = 16, 4, 128, 128, 384, 65536
a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
b = torch.randn(H, R, D, device="cuda", dtype=torch.float16)
x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
v = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16)
scores = torch.empty(H, L, device="cuda", dtype=torch.float16)
scores2 = (torch.randn(H, L, device="cuda") * 10).half()
frag = prepare_b(b, G)
inv = rope_inv_freq(x.device)
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
N = 20


def abx(stream):
    for _ in range(N):
        _lib.check(_lib.lib.palu_abx_rope_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), x.data_ptr(), x.stride(0),
                                              x.stride(1), scores.data_ptr(), scores.stride(0), H, G, L, R, D,
                                              inv.data_ptr(), 0, stream.cuda_stream), "abx")


def pv(stream):
    for _ in range(N):
        _lib.check(_lib.lib.palu_softmax_pv_f16(scores2.data_ptr(), scores2.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1),
                                                ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, math.sqrt(128.0),
                                                stream.cuda_stream), "pv")


ta = wall(lambda: abx(s1)) / N
tp = wall(lambda: pv(s2)) / N
tb = wall(lambda: (abx(s1), pv(s2))) / N
tb2 = wall(lambda: (pv(s2), abx(s1))) / N
print(f"no graph, {N} launches per stream: abx(128 WGs) {ta:.1f} us  pv {tp:.1f} us  both {tb:.1f} us  (pv first {tb2:.1f})  "
      f"sum {ta + tp:.1f}  max {max(ta, tp):.1f}")
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), " ROC/HIP env:", {k: v for k, v in os.environ.items() if k.startswith(("HIP_", "ROC", "HSA_", "AMD_"))})
