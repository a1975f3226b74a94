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


def _step_operands(H=32, G=8, D=128, hidden=1024, Rk=128, Rv=384, L=700, seed=0):
    from palu_amd import _lib
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq
    torch.manual_seed(seed)
    dev = "cuda"
    w = {"wq": (torch.randn(H * D, hidden, device=dev) / 32).half(), "vt_k": (torch.randn(G * Rk, hidden, device=dev) / 32).half(),
         "vt_v": (torch.randn(G * Rv, hidden, device=dev) / 32).half(), "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
         "wo": (torch.randn(hidden, H * Rv, device=dev) * 0.02).half()}
    cap = L + 64
    kc = torch.randn(G, cap, Rk, device=dev, dtype=torch.float16)
    vc = torch.randn(G, cap, Rv, device=dev, dtype=torch.float16)
    tok = torch.randn(hidden, device=dev, dtype=torch.float16)
    frag = prepare_b(w["b"], G)
    inv = rope_inv_freq(torch.device(dev))
    ws = torch.empty(_lib.lib.palu_decode_workspace_bytes(H, G, D, cap + 8, Rv), dtype=torch.uint8, device=dev)
    return w, kc, vc, tok, frag, inv, ws, cap


def test_decode_step_op_eager_and_compiled():
    """torch.ops.palu.decode_step: equals the CPU oracle step, mutates the caches in place (row L appended), passes
    opcheck (schema + fake tensor), and a decode step traces under torch.compile(fullgraph=True)."""
    import oracle
    import palu_amd.ops  # noqa: F401
    H, G, D, hidden, Rk, Rv, L = 32, 8, 128, 1024, 128, 384, 700
    w, kc, vc, tok, frag, inv, ws, cap = _step_operands(H, G, D, hidden, Rk, Rv, L)
    wc = {n: t.cpu() for n, t in w.items()}
    ref, _, k_all, v_all = oracle.decode_step(tok.cpu(), L, wc, kc[:, :L].cpu(), vc[:, :L].cpu())
    k0, v0 = kc.clone(), vc.clone()
    out = torch.ops.palu.decode_step(tok, w["wq"], w["vt_k"], w["vt_v"], frag, w["wo"], kc, vc, inv, ws, cap + 8, H, L, L)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-3)
    assert torch.equal(kc[:, :L], k0[:, :L]) and not torch.equal(kc[:, L], k0[:, L])          # appended in place
    torch.testing.assert_close(kc[:, L].cpu(), k_all[:, L], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(vc[:, L].cpu(), v_all[:, L], rtol=2e-3, atol=2e-3)
    torch.library.opcheck(torch.ops.palu.decode_step.default,
                          (tok, w["wq"], w["vt_k"], w["vt_v"], frag, w["wo"], k0, v0, inv, ws, cap + 8, H, L, L),
                          test_utils=("test_schema", "test_faketensor"))

    def step(tok_, kc_, vc_, ws_):
        return torch.ops.palu.decode_step(tok_, w["wq"], w["vt_k"], w["vt_v"], frag, w["wo"], kc_, vc_, inv, ws_, cap + 8, H, L, L)
    compiled = torch.compile(step, fullgraph=True, backend="aot_eager")
    k1, v1 = k0.clone(), v0.clone()
    out_c = compiled(tok, k1, v1, ws)
    torch.testing.assert_close(out_c, out, rtol=0, atol=0)
    assert torch.equal(k1, kc) and torch.equal(v1, vc)                                        # same in-place append


def test_module_decode_goes_through_the_dispatcher():
    """LlamaPaluAttention's decode branch calls torch.ops.palu.decode_step (one dispatcher node per token)."""
    import palu_amd.ops  # noqa: F401
    from torch.utils._python_dispatch import TorchDispatchMode
    from palu_amd.kernel.palu_attention import LatentCache
    from tests.test_decode_gpu import _module_from_palu_weights
    from tests.golden import inputs as gi
    hidden, H, D, gs, rank_k, rank_v, L = 512, 4, 128, 2, 64, 128, 96
    w, k_lat, v_lat, tok, _ = gi.step_inputs(10, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = LatentCache()
    cache.update(k_lat.unsqueeze(0).cuda(), v_lat.unsqueeze(0).cuda(), 0)
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))
    with torch.no_grad(), Spy():
        out, _, _ = m(tok.reshape(1, 1, hidden).cuda(), position_ids=torch.arange(L, L + 1), past_key_value=cache)
    assert any("palu.decode_step" in s for s in seen), seen
    assert cache.get_seq_length(0) == L + 1 and out.shape == (1, 1, hidden)


def test_new_ops_roundtrip():
    import palu_amd.ops  # noqa: F401
    H, G, D, hidden, Rk, Rv, L = 32, 8, 128, 1024, 128, 384, 500
    w, kc, vc, tok, frag, inv, ws, cap = _step_operands(H, G, D, hidden, Rk, Rv, L, seed=3)
    # decode_attn (single-kernel attention core) == abx + softmax_pv ops
    q = torch.randn(H, D, device="cuda", dtype=torch.float16)
    ctx = torch.ops.palu.decode_attn(q, frag, kc, vc, inv, ws, H, L)
    s = torch.ops.palu.abx(q.reshape(H, 1, D), w["b"], kc[:, :L].contiguous())
    ctx2 = torch.ops.palu.softmax_pv(s[:, 0], vc[:, :L].contiguous())
    torch.testing.assert_close(ctx, ctx2, rtol=1e-3, atol=2e-4)
    # prefill projection written into cache rows
    x = torch.randn(37, hidden, device="cuda", dtype=torch.float16)
    cache = torch.zeros(G, 64, Rk, device="cuda", dtype=torch.float16)
    torch.ops.palu.lowrank_project_gemm(x, w["vt_k"], cache, 5)
    ref = (x.float() @ w["vt_k"].float().t()).reshape(37, G, Rk).transpose(0, 1)
    torch.testing.assert_close(cache[:, 5:42].float(), ref, rtol=2e-3, atol=2e-3)
    assert cache[:, :5].abs().max() == 0
    codes = torch.randint(0, 8, (5, 7, 64), device="cuda", dtype=torch.uint8)
    assert torch.equal(torch.ops.palu.unpack_codes(torch.ops.palu.pack_codes(codes, 3), 3, 64), codes)
    # quantised step op vs the fp16 step op on the dequantised cache (same semantics, codes decoded on the fly)
    from palu_amd.kernel import quant as qz
    kcod, kmeta, kdeq = qz.quantize_pack(kc, 4, want_dequant=True)
    vcod, vmeta, vdeq = qz.quantize_pack(vc, 4, want_dequant=True)
    oq = torch.ops.palu.decode_step_q(tok, w["wq"], w["vt_k"], w["vt_v"], frag, w["wo"], kcod, kmeta, vcod, vmeta, inv, ws,
                                      cap + 8, H, Rk, Rv, 4, L, L)
    of = torch.ops.palu.decode_step(tok, w["wq"], w["vt_k"], w["vt_v"], frag, w["wo"], kdeq.contiguous(), vdeq.contiguous(),
                                    inv, ws, cap + 8, H, L, L)
    # the packed step also quantises the NEW latent row (4-bit), the fp16 step keeps it exact: agreement to that error
    torch.testing.assert_close(oq, of, rtol=2e-2, atol=6e-3)
