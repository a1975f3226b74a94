#!/usr/bin/env python3
"""Per-kernel timing of the quantised-latent decode kernels (BASELINE configs 3 and 4)."""
import math
import sys
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

H, G, D = 32, 8, 128
lib = _lib.lib


def run(bits, rank_k, rank_v, L):
    Rk, Rv = rank_k // G, rank_v // G
    torch.manual_seed(0)
    a = torch.randn(H, 1, D, device="cuda", dtype=torch.float16)
    b = (torch.randn(H, Rk, D, device="cuda") * Rk ** -0.5).half()
    frag = prepare_b(b, G)
    inv = rope_inv_freq(a.device)
    kb, vb = lib.palu_packed_row_bytes(Rk, bits), lib.palu_packed_row_bytes(Rv, bits)
    kc = torch.randint(0, 256, (G, L, kb), device="cuda", dtype=torch.uint8)
    vc = torch.randint(0, 256, (G, L, vb), device="cuda", dtype=torch.uint8)
    km = torch.rand(G, L, 2, device="cuda").half() * 0.1 + 0.05
    vm = torch.rand(G, L, 2, device="cuda").half() * 0.1 + 0.05
    km[..., 1] = 7.0
    vm[..., 1] = 7.0
    scores = torch.empty(H, (L + 7) // 8 * 8, device="cuda", dtype=torch.float16)
    ctx = torch.empty(H, Rv, device="cuda", dtype=torch.float16)
    ws = torch.empty(lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
    s = _lib.current_stream

    def k_abx():
        _lib.check(lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), kc.data_ptr(), kc.stride(0),
                                       kc.stride(1), km.data_ptr(), km.stride(0), km.stride(1), scores.data_ptr(),
                                       scores.stride(0), H, G, L, Rk, D, bits, inv.data_ptr(), 0, s()), "abx_q")

    def k_pv():
        _lib.check(lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, vc.data_ptr(), vc.stride(0), vc.stride(1),
                                         vm.data_ptr(), vm.stride(0), vm.stride(1), ctx.data_ptr(), 0, 0, ws.data_ptr(),
                                         H, G, L, Rv, bits, math.sqrt(D), s()), "pv_q")

    for name, fn, nbytes in (("abx_q", k_abx, G * L * (kb + 4) + 2 * H * L), ("softmax_pv_q", k_pv, G * L * (vb + 4) + 2 * H * L)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"bits={bits} rank {rank_k}/{rank_v} L={L}: {name:13s} {us:7.1f} us   {nbytes / us * 1e-3:7.0f} GB/s algorithmic")


run(3, 1024, 3072, 65537)
run(4, 512, 1536, 131073)
