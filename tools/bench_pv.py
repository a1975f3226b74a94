#!/usr/bin/env python3
"""softmax.PV kernel timing at BASELINE config 2 (fp16) -- for tuning experiments."""
import math, sys, torch
from palu_amd import _lib
H, G, L, Rv = 32, 8, 65537, 384
torch.manual_seed(0)
scores = (torch.randn(H, L + 7, device="cuda") * 10).half()
v = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16)
ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
def f():
    _lib.check(_lib.lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0, v.data_ptr(), v.stride(0), v.stride(1),
                                            ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, math.sqrt(128.0), _lib.current_stream()), "pv")
for _ in range(10): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 10
print(f"softmax_pv: {us:.1f} us  {(2*G*L*Rv)/us*1e-3:.0f} GB/s  nsplit={_lib.lib.palu_pv_nsplit(G, L)}")
