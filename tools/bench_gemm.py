#!/usr/bin/env python3
"""Prefill down-projection GEMM throughput (palu_lowrank_project_gemm) vs torch.matmul (rocBLAS/hipBLASLt)."""
import sys
import torch
from palu_amd import _lib

def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for M, N, K, R in [(8192, 1024, 4096, 128), (8192, 3072, 4096, 384), (65536, 1024, 4096, 128), (65536, 3072, 4096, 384)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = (torch.randn(N, K, device="cuda") / 64).half()
    G = N // R
    out = torch.empty(G, M, R, device="cuda", dtype=torch.float16)
    f = lambda: _lib.check(_lib.lib.palu_lowrank_project_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(),
                                                              out.stride(0), out.stride(1), M, N, K, R, 0, _lib.current_stream()), "g")
    us = bench(f)
    ut = bench(lambda: torch.nn.functional.linear(x, w))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: ours {us:9.1f} us = {fl / us * 1e-6:7.1f} TF ({100 * fl / us * 1e-6 / 2500:.1f}% of 2.5 PF)   torch {ut:9.1f} us = {fl / ut * 1e-6:7.1f} TF")
