#!/usr/bin/env python3
"""Prefill attention kernel timing: palu_prefill_attn_f16 at Llama-2-7B geometry (H=32, D=128, gs=4)."""
import math
import sys
import torch
from palu_amd import _lib

H, G, D = 32, 8, 128
Rv = int(sys.argv[2]) if len(sys.argv) > 2 else 384
for T in [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "16384"])]:
    q = torch.randn(H, T, D, device="cuda", dtype=torch.float16)
    k = torch.randn(H, T, D, device="cuda", dtype=torch.float16)
    pad = (T + 63) // 64 * 64
    vt = torch.randn(G, Rv, pad, device="cuda", dtype=torch.float16)
    out = torch.empty(T, H * Rv, device="cuda", dtype=torch.float16)

    def run():
        _lib.check(_lib.lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                                  vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(), out.stride(0),
                                                  H, G, D, T, T, Rv, 0, 1, 1.0 / math.sqrt(D), torch.cuda.current_stream().cuda_stream), "pf")
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5 if T <= 16384 else 2
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    useful = H * (T * T / 2) * (D + Rv) * 2           # causal half of q.k^T and P.V
    print(f"T={T:6d} Rv={Rv}: {ms:9.3f} ms  {useful / ms * 1e-9:7.1f} TFLOP/s useful (causal)")
