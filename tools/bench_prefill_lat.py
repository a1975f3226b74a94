#!/usr/bin/env python3
"""palu_prefill_attn_lat_f16 (keys rebuilt per tile in the kernel, V from the cache rows) against palu_prefill_attn_f16 (K~ / V^T
workspaces) at Llama-2-7B geometry, causal prompt of T tokens:  bench_prefill_lat.py [T ...]"""
import math, sys
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import rope_inv_freq

lib, S = _lib.lib, _lib.current_stream
H, G, D, Rk, Rv = 32, 8, 128, 128, 384
dev = torch.device("cuda:0")
inv = rope_inv_freq(dev)
for T in [int(a) for a in sys.argv[1:]] or [16384, 65536]:
    torch.manual_seed(0)
    q = torch.randn(H, T, D, device=dev, dtype=torch.float16)
    xk = torch.randn(G, T, Rk, device=dev, dtype=torch.float16)
    xv = torch.randn(G, T, Rv, device=dev, dtype=torch.float16)
    bt = (torch.randn(H, D, Rk, device=dev) * Rk ** -0.5).half()
    cs = torch.empty(lib.palu_rope_cs_table_bytes(T), dtype=torch.uint8, device=dev)
    _lib.check(lib.palu_rope_cs_table_build(inv.data_ptr(), 0, T, cs.data_ptr(), S()), "cs")
    out = torch.empty(T, H * Rv, dtype=torch.float16, device=dev)

    def lat():
        _lib.check(lib.palu_prefill_attn_lat_f16(q.data_ptr(), q.stride(0), q.stride(1), xk.data_ptr(), xk.stride(0), xk.stride(1),
                                                 xv.data_ptr(), xv.stride(0), xv.stride(1), bt.data_ptr(), cs.data_ptr(), out.data_ptr(),
                                                 out.stride(0), H, G, D, T, T, Rk, Rv, 0, 1, 1.0 / math.sqrt(D), S()), "lat")
    keys = torch.randn(H, T, D, device=dev, dtype=torch.float16)
    vt = xv.transpose(1, 2).contiguous()

    def ws():
        _lib.check(lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), keys.data_ptr(), keys.stride(0), keys.stride(1),
                                             vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(), out.stride(0), H, G, D, T, T, Rv,
                                             0, 1, 1.0 / math.sqrt(D), S()), "ws")
    flops = 2.0 * H * (T * (T + 1) / 2) * (D + Rv)
    for name, fn in (("workspace form", ws), ("latent form", lat)):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"T={T} {name:15s} {min(ts):8.2f} ms  {flops / min(ts) * 1e-9:7.1f} causal TFLOP/s", flush=True)
