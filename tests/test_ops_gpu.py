"""torch.ops.palu.* custom ops: registered, dispatchable, fake kernels consistent, results = direct calls."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_custom_ops_roundtrip():
    import palu_amd.ops  # noqa: F401  (registers the ops)
    from palu_amd.kernel.abx_rope import abx
    torch.manual_seed(0)
    a = torch.randn(32, 1, 128, dtype=torch.float16, device="cuda")
    b = (torch.randn(32, 64, 128, device="cuda") / 8).half()
    x = torch.randn(8, 333, 64, dtype=torch.float16, device="cuda")
    s = torch.ops.palu.abx(a, b, x)
    assert torch.equal(s, abx(a, b, x))
    v = torch.randn(8, 333, 96, dtype=torch.float16, device="cuda")
    ctx = torch.ops.palu.softmax_pv(s[:, 0], v)
    p = torch.softmax((s[:, 0] / math.sqrt(128.0)).float(), -1)
    ref = torch.matmul(p.reshape(8, 4, 333), v.float()).reshape(32, 96)
    torch.testing.assert_close(ctx.float(), ref, rtol=1e-3, atol=1e-3)
    w = (torch.randn(100, 256, device="cuda") / 16).half()
    xv = torch.randn(256, device="cuda", dtype=torch.float16)
    torch.testing.assert_close(torch.ops.palu.gemv(w, xv).float(), (w.float() @ xv.float()), rtol=2e-3, atol=2e-3)
    codes, meta = torch.ops.palu.quantize_pack(x, 4)
    deq = torch.ops.palu.unpack_dequant(codes, meta, 4, 64)
    assert deq.shape == x.shape and (deq.float() - x.float()).abs().max() < 0.6
    h = torch.ops.palu.hadamard_transform(x.float(), 0.125)
    torch.testing.assert_close(torch.ops.palu.hadamard_transform(h, 0.125), x.float(), rtol=1e-4, atol=1e-4)
    # fake kernels (meta shapes) agree with the real ones
    torch.library.opcheck(torch.ops.palu.abx.default, (a, b, x), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.palu.quantize_pack.default, (x, 3), test_utils=("test_schema", "test_faketensor"))


def test_prefill_ops():
    import palu_amd.ops  # noqa: F401
    torch.manual_seed(1)
    H, G, T, Rv = 8, 2, 150, 128
    q = torch.randn(H, T, 128, dtype=torch.float16, device="cuda")
    k = torch.randn(H, T, 128, dtype=torch.float16, device="cuda")
    v = torch.randn(G, T, Rv, dtype=torch.float16, device="cuda")
    q0 = q.clone()
    torch.ops.palu.rope_(q, 5)
    assert not torch.equal(q, q0)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, device="cuda").float() / 128))
    ang = torch.outer(torch.arange(5, 5 + T, device="cuda").float(), inv)
    ang = torch.cat((ang, ang), -1)
    ref = q0.float() * ang.cos() + torch.cat((-q0[..., 64:], q0[..., :64]), -1).float() * ang.sin()
    torch.testing.assert_close(q.float(), ref, rtol=5e-3, atol=5e-3)
    out = torch.ops.palu.prefill_attn(q, k, v, 0, True)
    s = torch.matmul(q.float(), k.float().transpose(1, 2)) / math.sqrt(128.0)
    s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    refo = torch.matmul(torch.softmax(s, -1).reshape(G, (H // G) * T, T), v.float()).reshape(H, T, Rv).transpose(0, 1).reshape(T, H * Rv)
    assert (out.float() - refo).abs().max().item() <= 2e-3 * max(1.0, refo.abs().max().item())
    torch.library.opcheck(torch.ops.palu.prefill_attn.default, (q, k, v, 0, True), test_utils=("test_schema", "test_faketensor"))
