"""Prompt attention straight from the latent caches (csrc/prefill_lat.hip): the keys are rebuilt per kv tile inside
the kernel, K~ = RoPE(X_k . B_h) (kernel/palu_attention.py:67-77, :199-205), the latent values come from the cache's row-major
rows.  Compared with (a) the workspace form -- keys reconstructed by the projection GEMM (fp16 rounding) + the rotary kernel with
the reference's fp16 arithmetic (csrc/rope.hip), then palu_prefill_attn_f16 -- and (b) an fp32 evaluation of the branch."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
D = 128


def _mods():
    from palu_amd import _lib
    from palu_amd.kernel import abx_rope
    return _lib, abx_rope


def _workspace_form(q, xk, xv, b, past, causal, inv):
    """K~ through the existing pieces: X_g . U_g^T rounded to fp16 (palu_lowrank_project_gemm), palu_rope_f16, V^T copy."""
    _lib, _ = _mods()
    lib, S = _lib.lib, torch.cuda.current_stream().cuda_stream
    H, Tq, _ = q.shape
    G, Tk, Rk = xk.shape
    Rv = xv.shape[2]
    gs = H // G
    keys = torch.empty(H, Tk, D, dtype=torch.float16, device=DEV)
    for g in range(G):
        u = b[g * gs:(g + 1) * gs].transpose(1, 2).reshape(gs * D, Rk).contiguous()           # U_g [gs*D, Rk]
        _lib.check(lib.palu_lowrank_project_gemm(xk[g].data_ptr(), xk.stride(1), u.data_ptr(), u.stride(0), keys[g * gs].data_ptr(),
                                                 keys.stride(0), keys.stride(1), Tk, gs * D, Rk, D, 0, S), "gemm")
    _lib.check(lib.palu_rope_f16(keys.data_ptr(), keys.stride(0), keys.stride(1), H, Tk, D, 0, inv.data_ptr(), S), "rope")
    pad = (Tk + 63) // 64 * 64
    vt = torch.zeros(G, Rv, pad, dtype=torch.float16, device=DEV)
    vt[:, :, :Tk].copy_(xv[:, :Tk].transpose(1, 2))
    out = torch.empty(Tq, H * Rv, dtype=torch.float16, device=DEV)
    _lib.check(lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), keys.data_ptr(), keys.stride(0), keys.stride(1),
                                         vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(), out.stride(0), H, G, D, Tq, Tk, Rv,
                                         past, 1 if causal else 0, 1.0 / math.sqrt(D), S), "prefill_attn")
    return out, keys


def _lat_form(q, xk, xv, b, past, causal, inv, Tk):
    _lib, _ = _mods()
    lib, S = _lib.lib, torch.cuda.current_stream().cuda_stream
    H, Tq, _ = q.shape
    G, _, Rk = xk.shape
    Rv = xv.shape[2]
    bt = b.transpose(1, 2).contiguous()                                                       # [H, D, Rk]
    cs = torch.empty(lib.palu_rope_cs_table_bytes(Tk), dtype=torch.uint8, device=DEV)
    _lib.check(lib.palu_rope_cs_table_build(inv.data_ptr(), 0, Tk, cs.data_ptr(), S), "cs")
    out = torch.empty(Tq, H * Rv, dtype=torch.float16, device=DEV)
    _lib.check(lib.palu_prefill_attn_lat_f16(q.data_ptr(), q.stride(0), q.stride(1), xk.data_ptr(), xk.stride(0), xk.stride(1),
                                             xv.data_ptr(), xv.stride(0), xv.stride(1), bt.data_ptr(), cs.data_ptr(), out.data_ptr(),
                                             out.stride(0), H, G, D, Tq, Tk, Rk, Rv, past, 1 if causal else 0, 1.0 / math.sqrt(D), S),
               "prefill_attn_lat")
    return out


def _ref_f32(q, keys, xv, past, causal):
    H, Tq, _ = q.shape
    G, Tk, Rv = xv.shape
    s = torch.matmul(q.float(), keys.float().transpose(1, 2)) / math.sqrt(D)
    if causal:
        i = torch.arange(Tq, device=DEV).unsqueeze(1) + past
        j = torch.arange(Tk, device=DEV).unsqueeze(0)
        s = s.masked_fill((j > i).unsqueeze(0), float("-inf"))
    p = torch.softmax(s, dim=-1)
    ctx = torch.matmul(p.reshape(G, (H // G) * Tq, Tk), xv.float()).reshape(H, Tq, Rv)
    return ctx.transpose(0, 1).reshape(Tq, H * Rv)


@pytest.mark.parametrize("H,gs,Tq,Tk,Rv,causal", [
    (4, 4, 128, 128, 384, True), (8, 4, 130, 130, 384, True), (8, 4, 257, 257, 128, True), (32, 4, 300, 300, 384, True),
    (4, 1, 1, 200, 256, False), (8, 4, 129, 1000, 384, True), (4, 4, 128, 192, 384, False), (8, 4, 333, 333, 256, True),
    (32, 4, 1100, 1100, 384, True), (8, 2, 64, 4097, 384, True), (4, 4, 200, 70, 384, False),
])
def test_latent_prefill_kernel_vs_workspace_form_and_fp32(H, gs, Tq, Tk, Rv, causal):
    _lib, ar = _mods()
    G, Rk = H // gs, 128
    g = torch.Generator().manual_seed(H * 1000 + Tq + Tk + Rv)
    past = Tk - Tq if causal and Tk >= Tq else 0
    q = torch.randn(H, Tq, D, generator=g).half().to(DEV)
    cap = Tk + 5                                                            # cache rows beyond Tk exist and hold garbage
    xk = torch.randn(G, cap, Rk, generator=g).half().to(DEV)
    xv = torch.randn(G, cap, Rv, generator=g).half().to(DEV)
    xk[:, Tk:] = float("nan")
    xv[:, Tk:] = float("nan")
    b = (torch.randn(H, Rk, D, generator=g) * Rk ** -0.5).half().to(DEV)
    inv = ar.rope_inv_freq(torch.device(DEV))
    ws, keys = _workspace_form(q, xk[:, :Tk], xv[:, :Tk], b, past, causal, inv)
    lat = _lat_form(q, xk, xv, b, past, causal, inv, Tk)
    ref = _ref_f32(q, keys, xv[:, :Tk], past, causal)
    scale = ref.abs().max().item()
    assert torch.isfinite(lat).all()
    assert (lat.float() - ref).abs().max().item() <= 2e-3 * scale + 1e-3
    # the two kernels see the same fp16 keys up to the last-bit differences of two MFMA accumulation orders
    assert (lat.float() - ws.float()).abs().max().item() <= 1e-3 * scale + 1e-3


def test_latent_prefill_rejects_unsupported_shapes():
    _lib, _ = _mods()
    lib = _lib.lib
    assert lib.palu_prefill_attn_lat_supported(32, 8, 128, 128, 384) == 1
    assert lib.palu_prefill_attn_lat_supported(32, 8, 128, 64, 192) == 0
    assert lib.palu_prefill_attn_lat_supported(32, 8, 64, 128, 384) == 0
    t = torch.zeros(1024, dtype=torch.float16, device=DEV)
    rc = lib.palu_prefill_attn_lat_f16(t.data_ptr(), 128, 128, t.data_ptr(), 128, 128, t.data_ptr(), 192, 192, t.data_ptr(), t.data_ptr(),
                                       t.data_ptr(), 192, 4, 1, 128, 1, 1, 64, 192, 0, 1, 0.1, torch.cuda.current_stream().cuda_stream)
    assert rc != 0
