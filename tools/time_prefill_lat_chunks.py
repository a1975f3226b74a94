#!/usr/bin/env python3
"""prefill_lat_kernel on a 64k-token causal prompt: ONE launch for all queries against the same work in query chunks (the module's
form: the transients of a chunk bound the memory), back to back with nothing between the launches -- what the chunking itself costs.
   python tools/time_prefill_lat_chunks.py [T] [chunk ...]"""
import math, sys
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import rope_inv_freq

lib, S = _lib.lib, _lib.current_stream
H, G, D, Rk, Rv = 32, 8, 128, 128, 384
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
chunks = [int(a) for a in sys.argv[2:]] or [T, 16384, 8192, 4096, 3072, 2048]
dev = torch.device("cuda:0")
inv = rope_inv_freq(dev)
torch.manual_seed(0)
q = torch.randn(H, T, D, device=dev, dtype=torch.float16)
xk = torch.randn(G, T, Rk, device=dev, dtype=torch.float16)
xv = torch.randn(G, T, Rv, device=dev, dtype=torch.float16)
bt = (torch.randn(H, D, Rk, device=dev) * Rk ** -0.5).half()
cs = torch.empty(lib.palu_rope_cs_table_bytes(T), dtype=torch.uint8, device=dev)
_lib.check(lib.palu_rope_cs_table_build(inv.data_ptr(), 0, T, cs.data_ptr(), S()), "cs")
out = torch.empty(T, H * Rv, dtype=torch.float16, device=dev)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]
TWO = False


def run(qc):
    if TWO:                                   # consecutive chunk launches on alternating streams: no dependency between them
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
    for i, c0 in enumerate(range(0, T, qc)):
        sid = streams[i & 1].cuda_stream if TWO else S()
        t = min(T, c0 + qc) - c0
        qq = q[:, c0:c0 + t]
        oo = out[c0:c0 + t]
        _lib.check(lib.palu_prefill_attn_lat_f16(qq.data_ptr(), qq.stride(0), qq.stride(1), xk.data_ptr(), xk.stride(0), xk.stride(1),
                                                 xv.data_ptr(), xv.stride(0), xv.stride(1), bt.data_ptr(), cs.data_ptr(), oo.data_ptr(),
                                                 oo.stride(0), H, G, D, t, c0 + t, Rk, Rv, c0, 1, 1.0 / math.sqrt(D), sid), "lat")
    if TWO:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)


for qc in chunks + [-c for c in chunks if c < T]:
    TWO = qc < 0
    qc = abs(qc)
    run(qc)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(qc)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("chunk %6d%s (%3d launches, %5d workgroups each): %.2f ms" % (qc, " on two streams" if TWO else "", (T + qc - 1) // qc, H * ((min(qc, T) + 127) // 128), min(ts)), flush=True)
