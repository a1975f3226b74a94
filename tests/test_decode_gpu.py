"""GPU parity of the decode-step kernels (softmax.PV, GEMVs, full step) against the CPU oracle and
the golden vectors generated from the reference.  Criterion P1 of SURVEY.md 8(c):
assert_close(rtol=1e-3, atol=1e-3) on attention weights and attention output."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

import oracle
from tests.golden import inputs as gi

DEV = "cuda"


def _lib():
    from palu_amd import _lib
    return _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def softmax_pv(scores, v, mask=None, want_probs=False):
    """scores [H,L] fp16, v [G,L,Rv] fp16 (cuda) -> ctx [H,Rv], probs [H,L]"""
    lib = _lib()
    H, L = scores.shape
    G, _, Rv = v.shape
    ws = torch.empty(lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    probs = torch.empty(H, L, dtype=torch.float16, device=DEV) if want_probs else None
    lib.check(lib.lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0 if mask is None else mask.data_ptr(),
                                          v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(),
                                          0 if probs is None else probs.data_ptr(), 0 if probs is None else probs.stride(0),
                                          ws.data_ptr(), H, G, L, Rv, math.sqrt(128.0), _stream()), "softmax_pv")
    return ctx, probs


def ref_softmax_pv(scores, v, mask=None):
    """CPU restatement of kernel/palu_attention.py:219,229-251 on fp16 tensors."""
    H, L = scores.shape
    G = v.shape[0]
    x = scores / math.sqrt(128.0)
    if mask is not None:
        x = x + mask.reshape(1, L)
    p = torch.softmax(x, dim=-1, dtype=torch.float32).to(torch.float16)
    ctx = torch.matmul(p.reshape(G, H // G, L), v)
    return ctx.reshape(H, -1), p


@pytest.mark.parametrize("H,gs,L,Rv", [(32, 4, 2049, 96), (32, 4, 1, 384), (32, 4, 129, 384), (8, 2, 700, 64),
                                       (4, 1, 333, 192), (16, 8, 5000, 128), (32, 4, 70000, 64)])
@pytest.mark.parametrize("masked", [False, True])
def test_softmax_pv(H, gs, L, Rv, masked):
    rng = np.random.default_rng(H + L + Rv)
    G = H // gs
    scores = torch.from_numpy((rng.standard_normal((H, L)) * 30).astype(np.float16))
    v = torch.from_numpy(rng.standard_normal((G, L, Rv)).astype(np.float16))
    mask = None
    if masked:
        m = np.zeros(L, dtype=np.float16)
        m[rng.random(L) < 0.3] = np.float16(-65504.0)
        m[L - 1] = 0
        mask = torch.from_numpy(m)
    ctx, probs = softmax_pv(scores.to(DEV), v.to(DEV), None if mask is None else mask.to(DEV), want_probs=True)
    rctx, rp = ref_softmax_pv(scores, v, mask)
    torch.testing.assert_close(probs.cpu(), rp, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(ctx.cpu(), rctx, rtol=1e-3, atol=1e-3)
    # against exact fp64 softmax of the same fp16 logits: tighter
    x = (scores / math.sqrt(128.0))
    x = x if mask is None else x + mask.reshape(1, L)
    p64 = torch.softmax(x.double(), dim=-1)
    c64 = torch.matmul(p64.reshape(G, gs, L), v.double()).reshape(H, Rv)
    assert (ctx.cpu().double() - c64).abs().max().item() <= 1e-3 * max(1.0, c64.abs().max().item())


def test_softmax_pv_fully_masked_split_and_strided_cache():
    """A whole split masked out (-inf after the fp16 add) must not poison the merge; V as a view of a
    larger pre-allocated cache."""
    rng = np.random.default_rng(3)
    H, G, L, Rv, cap = 32, 8, 3000, 96, 4096
    scores = torch.from_numpy((rng.standard_normal((H, L)) * 5).astype(np.float16))
    cache = torch.from_numpy(rng.standard_normal((G, cap, Rv)).astype(np.float16)).to(DEV)
    m = np.zeros(L, dtype=np.float16)
    m[:2000] = -np.inf
    mask = torch.from_numpy(m)
    ctx, probs = softmax_pv(scores.to(DEV), cache[:, :L], mask.to(DEV), want_probs=True)
    rctx, rp = ref_softmax_pv(scores, cache[:, :L].cpu(), mask)
    assert torch.isfinite(ctx.float()).all()
    torch.testing.assert_close(probs.cpu(), rp, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(ctx.cpu(), rctx, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 12288), (100, 768), (7, 264), (4096, 3072)])
def test_gemv(N, K):
    lib = _lib()
    rng = np.random.default_rng(N + K)
    W = torch.from_numpy((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal(K).astype(np.float16)).to(DEV)
    y = torch.empty(N, dtype=torch.float16, device=DEV)
    lib.check(lib.lib.palu_gemv_f16(W.data_ptr(), W.stride(0), x.data_ptr(), y.data_ptr(), N, K, _stream()), "gemv")
    ref = (W.double().cpu() @ x.double().cpu())
    assert (y.cpu().double() - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
    cpu16 = torch.nn.functional.linear(x.cpu().reshape(1, -1), W.cpu()).reshape(-1)
    torch.testing.assert_close(y.cpu(), cpu16, rtol=2e-3, atol=2e-3)


def _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w):
    from palu_amd.kernel.palu_attention import LlamaPaluAttention, build_b

    class Cfg:
        pass
    cfg = Cfg()
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = hidden, H, False
    cfg.group_size, cfg.num_groups = gs, H // gs
    cfg.total_rank_k, cfg.total_rank_v = rank_k, rank_v
    m = LlamaPaluAttention(cfg, 0)
    with torch.no_grad():
        m.q_proj.weight.copy_(w["wq"])
        m.k_proj.VT.weight.copy_(w["vt_k"])
        m.v_proj.VT.weight.copy_(w["vt_v"])
        for i, u in enumerate(w["u_k"]):
            m.k_proj.U_list[i].weight.copy_(u)
        m.o_proj.weight.copy_(w["wo"])
    m.k_proj.B = nn.Parameter(build_b(w["u_k"], gs, D))
    return m.eval().to(DEV, torch.float16)


@pytest.mark.parametrize("case", gi.STEP_CASES, ids=[c[0] for c in gi.STEP_CASES])
def test_decode_step_golden(golden_dir, case):
    """Whole decode step through LlamaPaluAttention.forward (one C-ABI call) vs the reference's outputs."""
    from palu_amd.kernel.palu_attention import LatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask = case
    g = np.load(os.path.join(golden_dir, "g3_decode_step.npz"))
    w, k_lat, v_lat, tok, mask = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask)
    flat = [w["wq"], w["vt_k"], w["vt_v"], w["wo"], *w["u_k"], k_lat, v_lat, tok]
    assert gi.digest(*flat) == str(g[tag + "/digest"])
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = LatentCache()
    cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
    am = None if mask is None else mask.reshape(1, 1, 1, L + 1).to(DEV)
    with torch.no_grad():
        out, probs, cache2 = m(tok.reshape(1, 1, hidden).to(DEV), attention_mask=am,
                               position_ids=torch.arange(L, L + 1), past_key_value=cache, output_attentions=True)
    assert cache2 is cache and cache.get_seq_length(0) == L + 1
    assert out.shape == (1, 1, hidden) and probs.shape == (1, H, 1, L + 1)
    torch.testing.assert_close(probs.cpu().reshape(H, L + 1), torch.from_numpy(g[tag + "/attn_weights"]),
                               rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu().reshape(-1), torch.from_numpy(g[tag + "/attn_output"]), rtol=1e-3, atol=1e-3)
    kb, vb = cache.buffers(0)
    torch.testing.assert_close(kb[0, :, L].cpu(), torch.from_numpy(g[tag + "/k_new"]), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(vb[0, :, L].cpu(), torch.from_numpy(g[tag + "/v_new"]), rtol=2e-3, atol=2e-3)
    # and against the oracle run on the same inputs (tighter on the output scale)
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    o2, p2, _, _ = oracle.decode_step(tok, L, wd, k_lat, v_lat, mask)
    assert (out.cpu().reshape(-1).float() - o2.float()).abs().max().item() <= 1e-3


def test_decode_steps_grow_the_cache_in_place():
    """Several consecutive steps: same result as feeding the oracle step by step; buffers never move."""
    from palu_amd.kernel.palu_attention import LatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, _ = gi.STEP_CASES[0]
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = LatentCache(capacity=256)
    cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
    ptr = cache.buffers(0)[0].data_ptr()
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    kc, vc = k_lat, v_lat
    rng = np.random.default_rng(0)
    for step in range(5):
        t = torch.from_numpy(rng.standard_normal(hidden).astype(np.float16))
        with torch.no_grad():
            out, _, _ = m(t.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L + step, L + step + 1),
                          past_key_value=cache)
        ref, _, kc, vc = oracle.decode_step(t, L + step, wd, kc, vc)
        assert (out.cpu().reshape(-1).float() - ref.float()).abs().max().item() <= 1e-3
    assert cache.buffers(0)[0].data_ptr() == ptr and cache.get_seq_length(0) == L + 5


def test_reference_test_scenario(golden_dir):
    """kernel/test_palu_attention.py:158-195: full-rank (4096/4096) Palu from a dense attention via
    per-group SVD, prefill 63 tokens into the cache, decode 1 token: attention weights and output vs the
    values the reference module produced (and vs vanilla attention), rtol=atol=1e-3."""
    from palu_amd.kernel.palu_attention import LatentCache, LlamaPaluAttention
    g = np.load(os.path.join(golden_dir, "g3b_reftest.npz"))
    rng = np.random.default_rng(4242)
    hidden, H, D, gs = 4096, 32, 128, 4

    class Dense(nn.Module):
        def __init__(self):
            super().__init__()
            self.layer_idx, self.head_dim = 0, D
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                setattr(self, n, nn.Linear(hidden, hidden, bias=False))
    attn = Dense()
    ws = []
    with torch.no_grad():
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            wt = torch.from_numpy(rng.uniform(-1 / 64, 1 / 64, (hidden, hidden)).astype(np.float32))
            getattr(attn, n).weight.copy_(wt)
            ws.append(wt)
    prompt = torch.from_numpy(rng.standard_normal((1, 63, hidden)).astype(np.float16))
    tok = torch.from_numpy(rng.standard_normal((1, 1, hidden)).astype(np.float16))
    assert gi.digest(*ws, prompt, tok) == str(g["digest"])

    class Cfg:
        pass
    cfg = Cfg()
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = hidden, H, False
    cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = gs, H // gs, 4096, 4096
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))   # 64 SVDs of 512x4096: oversubscription hurts
    palu = LlamaPaluAttention.from_attention(attn, cfg).eval().to(DEV, torch.float16)
    cache = LatentCache()
    with torch.no_grad():
        po, _, _ = palu(prompt.to(DEV), past_key_value=cache, position_ids=torch.arange(63).unsqueeze(0))
        do, dp, _ = palu(tok.to(DEV), output_attentions=True, past_key_value=cache,
                         position_ids=torch.arange(63, 64).unsqueeze(0))
    torch.testing.assert_close(po[0, -1].cpu(), torch.from_numpy(g["prefill_output_last"]), rtol=2e-3, atol=2e-3)
    for name_w, name_o in (("decode_weights", "decode_output"), ("vanilla_weights", "vanilla_output")):
        torch.testing.assert_close(dp[0, :, 0].cpu().float(), torch.from_numpy(g[name_w]).float(), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(do.reshape(-1).cpu().float(), torch.from_numpy(g[name_o]).float(), rtol=1e-3, atol=1e-3)


def test_decode_step_under_graph_capture():
    """run_latency_attention.py:81-90 captures one forward in a graph and replays it."""
    from palu_amd.kernel.palu_attention import LatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, _ = gi.STEP_CASES[0]
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)

    def fresh_cache():
        c = LatentCache(capacity=512)
        c.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
        return c
    x = tok.reshape(1, 1, hidden).to(DEV)
    pos = torch.arange(L, L + 1)
    with torch.no_grad():
        eager, _, _ = m(x, position_ids=pos, past_key_value=fresh_cache())          # also warms the caches
    cache = fresh_cache()
    static_x = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(s):
        m(static_x, position_ids=pos, past_key_value=fresh_cache())
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        out, _, _ = m(static_x, position_ids=pos, past_key_value=cache)
    for _ in range(3):
        static_x.copy_(x)
        graph.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(out, eager, rtol=0, atol=0)


def test_decode_step_full_size_c2_vs_oracle():
    """BASELINE config 2 at FULL size (rank 1024/3072, gs 4, 65536 cached positions): the whole HIP step
    (palu_decode_attend_f16 + o_proj) against oracle.decode_step on the same inputs, P1 tolerance.  The oracle needs
    ~13 s of CPU per step at this size (it is also what bench.py's cpu_baseline leg times and checks)."""
    from palu_amd.kernel import head_parallel as hp
    H, G, D, HID, Rk, Rv, L = 32, 8, 128, 4096, 128, 384, 65536
    torch.manual_seed(0)
    w = {"wq": torch.randn(H * D, HID).mul_(1 / 64).half(), "vt_k": torch.randn(G * Rk, HID).mul_(1 / 64).half(),
         "vt_v": torch.randn(G * Rv, HID).mul_(1 / 64).half(), "b": torch.randn(H, Rk, D).mul_(Rk ** -0.5).half(),
         "wo": torch.randn(HID, H * Rv).mul_(0.01).half()}
    k = torch.randn(G, L, Rk).half()
    v = torch.randn(G, L, Rv).half()
    tok = torch.randn(HID).half()
    with torch.no_grad():
        ref, probs, _, _ = oracle.decode_step(tok, L, w, k, v)
    plan = hp.make_plan(1, 0, H, G, D, Rk, Rv)
    wd = {n: t.to(DEV) for n, t in w.items()}
    kc = torch.zeros(G, L + 64, Rk, dtype=torch.float16, device=DEV)
    vc = torch.zeros(G, L + 64, Rv, dtype=torch.float16, device=DEV)
    kc[:, :L] = k.to(DEV)
    vc[:, :L] = v.to(DEV)
    dec = hp.HeadParallelDecoder(plan, wd, kc, vc, HID)
    out = dec.step(tok.to(DEV), L, L)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-3)
    # the appended latent rows equal the oracle's projections (fp16 GEMV with fp32 accumulation)
    k_new = torch.nn.functional.linear(tok.reshape(1, -1), w["vt_k"]).reshape(G, Rk)
    torch.testing.assert_close(kc[:, L].cpu(), k_new, rtol=2e-3, atol=2e-3)
    assert abs(probs.float().sum(-1) - 1).max().item() < 5e-2
    # the attention WEIGHTS too, so that a score-side error cannot hide behind an averaging softmax -- the
    # step's scores (the workspace rows the score kernel wrote, through the one-call step with probs requested) against
    # the oracle's softmax weights at all 65 537 positions, and the raw scores against an fp64 evaluation
    lib = _lib()
    kc2 = torch.zeros_like(kc)
    vc2 = torch.zeros_like(vc)
    kc2[:, :L] = k.to(DEV)
    vc2[:, :L] = v.to(DEV)
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq
    frag = prepare_b(wd["b"], G)
    inv = rope_inv_freq(torch.device(DEV))
    ws = torch.zeros(lib.lib.palu_decode_workspace_bytes(H, G, D, L + 64, Rv), dtype=torch.uint8, device=DEV)
    out2 = torch.empty(HID, dtype=torch.float16, device=DEV)
    pr = torch.empty(H, L + 1, dtype=torch.float16, device=DEV)
    x = tok.to(DEV)
    lib.check(lib.lib.palu_decode_step_f16(
        x.data_ptr(), wd["wq"].data_ptr(), wd["wq"].stride(0), wd["vt_k"].data_ptr(), wd["vt_k"].stride(0),
        wd["vt_v"].data_ptr(), wd["vt_v"].stride(0), frag.data_ptr(), wd["wo"].data_ptr(), wd["wo"].stride(0),
        kc2.data_ptr(), kc2.stride(0), kc2.stride(1), vc2.data_ptr(), vc2.stride(0), vc2.stride(1), 0, inv.data_ptr(),
        out2.data_ptr(), pr.data_ptr(), pr.stride(0), ws.data_ptr(), L + 64, H, G, D, HID, Rk, Rv, L, L, _stream()),
        "palu_decode_step_f16")
    torch.cuda.synchronize()
    torch.testing.assert_close(out2.cpu(), ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(pr.cpu().float(), probs.float(), rtol=1e-3, atol=1e-3)
    # relative check on the weights that matter (the largest 1 % of each head: 1e-3 absolute says little at 1 / 65537)
    top = probs.float().topk(655, dim=-1)
    got = pr.cpu().float().gather(-1, top.indices)
    assert ((got - top.values).abs() / top.values).max().item() < 2e-2


def test_softmax_pv_config5_slice():
    """BASELINE config 5, the per-GPU slice of the 8-GPU sharding: ONE latent group (G=1, gs=4), L = 262144, Rv = 384.
    Size-independent checks: fp32 torch-GPU recomputation from the same fp16 scores, and the split-L statistics."""
    torch.manual_seed(1)
    H, G, L, Rv = 4, 1, 262144, 384
    scores = (torch.randn(H, L, device=DEV) * 12).half()
    v = torch.randn(G, L, Rv, device=DEV, dtype=torch.float16)
    ctx, probs = softmax_pv(scores, v, want_probs=True)
    x = (scores / math.sqrt(128.0)).float()
    p = torch.softmax(x, dim=-1)
    ref = torch.matmul(p.reshape(G, 4, L), v.float()).reshape(H, Rv)
    torch.testing.assert_close(ctx.float(), ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(probs.float(), p, rtol=1e-3, atol=1e-3)
    # two half ranges merged with the (max, sum) statistics == the whole range (what SplitLDecoder does across GPUs)
    lib = _lib()
    halves = []
    for l0, l1 in ((0, L // 2), (L // 2, L)):
        sc, vv = scores[:, l0:l1].contiguous(), v[:, l0:l1].contiguous()
        ws = torch.empty(lib.lib.palu_pv_workspace_bytes(H, G, l1 - l0, Rv), dtype=torch.uint8, device=DEV)
        c = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
        lib.check(lib.lib.palu_softmax_pv_f16(sc.data_ptr(), sc.stride(0), 0, vv.data_ptr(), vv.stride(0), vv.stride(1),
                                              c.data_ptr(), 0, 0, ws.data_ptr(), H, G, l1 - l0, Rv, math.sqrt(128.0),
                                              _stream()), "pv")
        off = lib.lib.palu_pv_stats_offset(H, G, l1 - l0, Rv)
        halves.append((c.float(), ws[off:off + H * 8].view(torch.float32).reshape(H, 2).clone()))
    (c0, s0), (c1, s1) = halves
    M = torch.maximum(s0[:, 0], s1[:, 0])
    w0, w1 = s0[:, 1] * torch.exp(s0[:, 0] - M), s1[:, 1] * torch.exp(s1[:, 0] - M)
    merged = (w0[:, None] * c0 + w1[:, None] * c1) / (w0 + w1)[:, None]
    torch.testing.assert_close(merged, ctx.float(), rtol=2e-3, atol=2e-3)


def test_softmax_pv_full_size_c2_properties():
    """BASELINE config 2 (rank 1024/3072, gs 4, L=64k), the softmax.PV kernel alone: fp32 torch-GPU recomputation from
    the same fp16 scores, plus linearity of the context in V (the full step at this size is checked against the
    oracle in test_decode_step_full_size_c2_vs_oracle)."""
    torch.manual_seed(0)
    H, G, L, Rv = 32, 8, 65536, 384
    scores = (torch.randn(H, L, device=DEV) * 12).half()
    v = torch.randn(G, L, Rv, device=DEV, dtype=torch.float16)
    ctx, probs = softmax_pv(scores, v, want_probs=True)
    x = (scores / math.sqrt(128.0)).float()
    p = torch.softmax(x, dim=-1)
    ref = torch.matmul(p.reshape(G, 4, L), v.float()).reshape(H, Rv)
    torch.testing.assert_close(ctx.float(), ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(probs.float(), p, rtol=1e-3, atol=1e-3)
    assert abs(probs.float().sum(-1) - 1).max().item() < 5e-2
    v2 = torch.randn(G, L, Rv, device=DEV, dtype=torch.float16)
    c2, _ = softmax_pv(scores, v2)
    c12, _ = softmax_pv(scores, (v.float() * 0.5 + v2.float() * 0.25).half())
    torch.testing.assert_close(c12.float(), ctx.float() * 0.5 + c2.float() * 0.25, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M,N,K,R", [(1000, 1024, 4096, 128), (63, 4096, 4096, 512), (129, 768, 512, 96), (4096, 3072, 4096, 384),
                                     (1, 256, 4096, 32),
                                     # the 256x256 LDS-DMA kernel (M >= 512, N >= 256, K >= 512): ragged token tiles, a half-used
                                     # column tile, more token tiles than one round of the 8 XCDs, a single k-tile pair
                                     (600, 384, 512, 96), (777, 1280, 1024, 160), (2304 + 17, 512, 4096, 128), (512, 256, 512, 256),
                                     (8192, 1024, 4096, 256),
                                     # enough tiles for the 256-column form (>= 1 per CU), ragged in both directions
                                     (8192 + 5, 2048 + 128, 512, 128)])
def test_lowrank_project_gemm(M, N, K, R):
    """Prefill down-projection (MFMA GEMM) vs fp64, written into the [G, L, R] cache layout at a row offset."""
    lib = _lib()
    rng = np.random.default_rng(M + N)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float16)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((N, K)) / math.sqrt(K)).astype(np.float16)).to(DEV)
    G, row0, cap = N // R, 5, M + 16
    cache = torch.zeros(G, cap, R, dtype=torch.float16, device=DEV)
    lib.check(lib.lib.palu_lowrank_project_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), cache.data_ptr(),
                                                cache.stride(0), cache.stride(1), M, N, K, R, row0, _stream()), "gemm")
    ref = (x.double() @ w.double().t()).reshape(M, G, R).transpose(0, 1)          # [G, M, R]
    got = cache[:, row0:row0 + M].double()
    assert (got - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    assert cache[:, :row0].abs().max() == 0 and cache[:, row0 + M:].abs().max() == 0      # nothing else touched
    t16 = torch.nn.functional.linear(x, w).reshape(M, G, R).transpose(0, 1)
    torch.testing.assert_close(cache[:, row0:row0 + M], t16, rtol=2e-3, atol=2e-3)


def test_split_l_decoders_merge_to_full_step():
    """Split-L (fewer groups than GPUs): two SplitLDecoder instances stand in for two ranks on one GPU; their
    (context, max, sum) partials LSE-merge to the oracle's full decode step.  Exercises abx at pos0 != 0 and the
    statistics the P.V kernel leaves in its workspace."""
    from palu_amd.kernel import head_parallel as hp
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, _ = gi.STEP_CASES[0]
    G = H // gs
    L = 700
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)
    full = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
            "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    ref, _, _, _ = oracle.decode_step(tok, L, full, k_lat, v_lat)
    wd = {k: v.to(DEV).contiguous() for k, v in full.items()}
    ranges = hp.split_ranges(L, 2)
    parts = []
    for r, (l0, l1) in enumerate(ranges):
        cap = (l1 - l0) + 64
        kc = torch.zeros(G, cap, rank_k // G, dtype=torch.float16, device=DEV)
        vc = torch.zeros(G, cap, rank_v // G, dtype=torch.float16, device=DEV)
        kc[:, :l1 - l0] = k_lat[:, l0:l1].to(DEV)
        vc[:, :l1 - l0] = v_lat[:, l0:l1].to(DEV)
        dec = hp.SplitLDecoder(2, r, H, G, D, wd, kc, vc, l1 - l0, l0, r == 1, hidden)
        ctx, stats = dec.local_step(tok.to(DEV), L)
        parts.append((ctx.clone(), stats.clone()))
    torch.cuda.synchronize()
    ctx = torch.stack([p[0] for p in parts])
    st = torch.stack([p[1] for p in parts])
    merged = hp.merge_partials(ctx, st[..., 0], st[..., 1]).half()
    out = torch.nn.functional.linear(merged.reshape(1, -1).cpu(), full["wo"]).reshape(-1)
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=1e-3)


def test_whole_llama_model_decodes_through_latent_caches():
    """a whole transformers-5 LlamaForCausalLM with every attention replaced by the low-rank module
    (palu_amd.hf.convert_llama_to_palu) and ONE PaluCacheHF carrying the latent caches of all layers.  With full ranks
    the decomposition is exact up to fp16, so prompt logits and three decode steps must agree with the vanilla model
    (the reference's own check, test_palu_attention.py:158-195, lifted to the model level); then the same with a packed
    4-bit latent cache at reduced ranks runs end to end."""
    transformers = pytest.importorskip("transformers")
    from transformers import LlamaConfig, LlamaForCausalLM
    from palu_amd.hf import PaluCacheHF, convert_llama_to_palu, PaluAttentionHF
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=128, hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, head_dim=128, max_position_embeddings=512, rope_theta=10000.0,
                      attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    ref = LlamaForCausalLM(cfg).to(DEV, torch.float16).eval()
    import copy
    palu = convert_llama_to_palu(copy.deepcopy(ref), rank_k=512, rank_v=512, group_size=2)      # full rank: gs * D = 256 per group
    assert all(isinstance(l.self_attn, PaluAttentionHF) for l in palu.model.layers)
    ids = torch.randint(0, 128, (1, 37), device=DEV)
    with torch.no_grad():
        r = ref(ids, use_cache=True)
        cache = PaluCacheHF(bits=16)
        p = palu(ids, past_key_values=cache, use_cache=True)
    torch.testing.assert_close(p.logits.float(), r.logits.float(), rtol=3e-2, atol=3e-2)
    assert cache.get_seq_length() == 37
    rc = r.past_key_values
    tok = r.logits[:, -1:].argmax(-1)
    for step in range(3):
        with torch.no_grad():
            r = ref(tok, past_key_values=rc, use_cache=True)
            p = palu(tok, past_key_values=cache, use_cache=True)
        torch.testing.assert_close(p.logits.float(), r.logits.float(), rtol=3e-2, atol=3e-2)
        assert cache.get_seq_length() == 38 + step
        tok = r.logits[:, -1:].argmax(-1)
    # packed 4-bit latents at reduced rank: runs through prompt + decode (accuracy is the compression's, not checked here)
    small = convert_llama_to_palu(copy.deepcopy(ref), rank_k=128, rank_v=256, group_size=2)
    qc = PaluCacheHF(bits=4)
    with torch.no_grad():
        o = small(ids, past_key_values=qc, use_cache=True)
        o2 = small(tok, past_key_values=qc, use_cache=True)
    assert qc.get_seq_length() == 38 and torch.isfinite(o2.logits).all() and o.logits.shape == (1, 37, 128)


@pytest.mark.parametrize("kv_heads,group_size", [(2, 1), (2, 2), (1, 1)], ids=["kv2_g1", "kv2_g2", "mqa"])
def test_whole_llama_model_gqa(kv_heads, group_size):
    """VERDICT r2 missing #2: GQA checkpoints (num_key_value_heads < num_attention_heads) through the kernel-path module.
    Grouping as in the reference's GQA wrapper (palu/model/svd_mistral/modeling_palu_mistral.py:37-59): `group_size`
    KV heads per low-rank group; a latent group serves group_size * n_rep query heads.  Full ranks -> prompt logits and
    decode steps equal the vanilla GQA model's.  With one KV head per group the query heads of a group share B: the step
    runs on the shared-B score kernel."""
    transformers = pytest.importorskip("transformers")
    from transformers import LlamaConfig, LlamaForCausalLM
    from palu_amd.hf import PaluCacheHF, convert_llama_to_palu
    from palu_amd.kernel.abx_rope import shared_b
    import copy
    torch.manual_seed(2)
    H, D = 4, 128
    cfg = LlamaConfig(vocab_size=128, hidden_size=H * D, intermediate_size=512, num_hidden_layers=2, num_attention_heads=H,
                      num_key_value_heads=kv_heads, head_dim=D, max_position_embeddings=512, rope_theta=10000.0,
                      attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    ref = LlamaForCausalLM(cfg).to(DEV, torch.float16).eval()
    full = kv_heads * D                                           # full rank of the [kv*D, hidden] projections
    palu = convert_llama_to_palu(copy.deepcopy(ref), rank_k=full, rank_v=full, group_size=group_size)
    inner = palu.model.layers[0].self_attn.inner
    G = kv_heads // group_size
    assert inner.num_groups == G and inner.group_size == group_size * (H // kv_heads) and inner.k_proj.B.shape[0] == H
    assert (shared_b(inner.k_proj.B, G) is not None) == (group_size == 1 and H // kv_heads in (2, 3, 4))
    ids = torch.randint(0, 128, (1, 33), device=DEV)
    with torch.no_grad():
        r = ref(ids, use_cache=True)
        cache = PaluCacheHF(bits=16)
        p = palu(ids, past_key_values=cache, use_cache=True)
    torch.testing.assert_close(p.logits.float(), r.logits.float(), rtol=3e-2, atol=3e-2)
    rc, tok = r.past_key_values, r.logits[:, -1:].argmax(-1)
    for step in range(3):
        with torch.no_grad():
            r = ref(tok, past_key_values=rc, use_cache=True)
            p = palu(tok, past_key_values=cache, use_cache=True)
        torch.testing.assert_close(p.logits.float(), r.logits.float(), rtol=3e-2, atol=3e-2)
        assert cache.get_seq_length() == 34 + step
        tok = r.logits[:, -1:].argmax(-1)
    kb, _ = cache.latent.buffers(0)
    assert kb.shape[1] == G and kb.shape[3] == full // G          # latent rows per KV-head group, never reconstructed K


def test_whole_llama_model_padded_prompt_default_sdpa():
    """ADVICE r2 (palu_amd/hf.py): with a PADDED prompt transformers 5.x builds BOOLEAN masks (True = attend) under its
    default sdpa implementation -- [1,1,q,kv] for the prompt, [1,1,1,kv] for every decode step.  The adapter must convert
    them to the reference's additive convention (kernel/palu_attention.py:229-234) and must not take the causal flash
    path for a mask that carries padding.  Full ranks: logits of the valid positions equal the vanilla model's."""
    transformers = pytest.importorskip("transformers")
    from transformers import LlamaConfig, LlamaForCausalLM
    from palu_amd.hf import PaluCacheHF, convert_llama_to_palu
    import copy
    torch.manual_seed(1)
    cfg = LlamaConfig(vocab_size=128, hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, head_dim=128, max_position_embeddings=512, rope_theta=10000.0,
                      attention_bias=False, tie_word_embeddings=False)          # _attn_implementation: default (sdpa)
    ref = LlamaForCausalLM(cfg).to(DEV, torch.float16).eval()
    palu = convert_llama_to_palu(copy.deepcopy(ref), rank_k=512, rank_v=512, group_size=2)
    T, pad = 29, 6
    ids = torch.randint(0, 128, (1, T), device=DEV)
    am = torch.ones(1, T, dtype=torch.long, device=DEV)
    am[0, :pad] = 0                                                          # left-padded prompt
    seen = []
    inner = palu.model.layers[0].self_attn.inner
    orig = inner.forward

    def spy(*a, **k):
        m = k.get("attention_mask")
        seen.append((None if m is None else m.dtype, k.get("is_causal")))
        return orig(*a, **k)
    inner.forward = spy
    cache = PaluCacheHF(bits=16)
    with torch.no_grad():
        r = ref(ids, attention_mask=am, use_cache=True)
        p = palu(ids, attention_mask=am, past_key_values=cache, use_cache=True)
    torch.testing.assert_close(p.logits[:, pad:].float(), r.logits[:, pad:].float(), rtol=3e-2, atol=3e-2)
    assert seen[0] == (torch.float16, False)            # additive mask, NOT declared causal: padding columns are honoured
    last_prompt = p.logits[:, -1].float().clone()
    rc, tok = r.past_key_values, r.logits[:, -1:].argmax(-1)
    for step in range(2):
        am = torch.cat((am, torch.ones(1, 1, dtype=torch.long, device=DEV)), dim=1)
        with torch.no_grad():
            r = ref(tok, attention_mask=am, past_key_values=rc, use_cache=True)
            p = palu(tok, attention_mask=am, past_key_values=cache, use_cache=True)
        torch.testing.assert_close(p.logits.float(), r.logits.float(), rtol=3e-2, atol=3e-2)
        assert seen[-1][0] == torch.float16 and cache.get_seq_length() == T + 1 + step
        tok = r.logits[:, -1:].argmax(-1)
    # the same prompt WITHOUT padding must differ (the mask really masked something)
    with torch.no_grad():
        r_nopad = ref(ids, use_cache=False)
    assert (r_nopad.logits[:, -1].float() - last_prompt).abs().max() > 1e-3
    cache.reset()
    assert cache.get_seq_length() == 0


def test_softmax_pv_fp16_rows_through_the_register_direct_kernel():
    """PALU_PV_DIRECT=1 routes plain fp16 latent rows through pv_partial_qr_kernel<..., 16, ...> (opt-in: DESIGN 4.4).
    The switch is read once per process, so the parametrised cases of test_softmax_pv are re-run in a child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PALU_PV_DIRECT="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_decode_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_softmax_pv and not register_direct and not full_size and not config5"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


@pytest.mark.parametrize("with_probs", [False, True])
def test_decode_step_with_shared_b_goes_through_the_shared_kernel(with_probs):
    """True-GQA weights (every head of a group has the same U_k block): the module's one-call decode step
    detects the tied B, hands the [G, R, D] fragments to palu_decode_step_sharedb_f16 (scores by the shared-B kernel) and
    still equals the oracle's step on the expanded per-head weights."""
    from palu_amd.kernel.abx_rope import shared_b
    from palu_amd.kernel.palu_attention import LatentCache
    hidden, H, D, gs, rank_k, rank_v, L = 4096, 32, 128, 4, 1024, 3072, 777
    w, k_lat, v_lat, tok, _ = gi.step_inputs(77, hidden, H, D, gs, rank_k, rank_v, L, False)
    # tie the heads of each group: U_k block of head 0 of the group for all its heads
    u_tied = []
    for u in w["u_k"]:                                              # u: [gs * D, R] per group
        blk = u[:D].clone()
        u_tied.append(blk.repeat(gs, 1))
    w = dict(w, u_k=u_tied)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    assert shared_b(m.k_proj.B, H // gs) is not None
    cache = LatentCache()
    cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
    seen = []
    if not with_probs:
        from torch.utils._python_dispatch import TorchDispatchMode

        class Spy(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                if "palu.decode_step" in str(func):
                    seen.append(args[-1] if args else None)
                return func(*args, **(kwargs or {}))
        ctxm = Spy()
    else:
        import contextlib
        ctxm = contextlib.nullcontext()
    with torch.no_grad(), ctxm:
        out, probs, _ = m(tok.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L, L + 1), past_key_value=cache,
                          output_attentions=with_probs)
    if not with_probs:
        assert seen == [True], seen                                  # the op was called with shared_b=True
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    ref_out, ref_p, _, _ = oracle.decode_step(tok, L, wd, k_lat, v_lat)
    torch.testing.assert_close(out.cpu().reshape(-1), ref_out, rtol=1e-3, atol=1e-3)
    if with_probs:
        torch.testing.assert_close(probs.cpu().reshape(H, L + 1), ref_p, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("bits", [16, 4])
def test_decode_step_with_attention_bias(bits):
    """config.attention_bias = True (kernel/palu_attention.py:142-145): q_proj.bias inside the qkv kernel, o_proj.bias inside
    the last GEMV, fp16 and packed caches.  Oracle: the same decode step on weights augmented by one bias column and a
    token augmented by a constant 1 (fp16 matmul with fp32 accumulation: W x + b rounded once), o_proj.bias added last."""
    from palu_amd.kernel.palu_attention import LatentCache, LlamaPaluAttention, QuantLatentCache, build_b
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, _ = gi.STEP_CASES[0]
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)

    class Cfg:
        pass
    cfg = Cfg()
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = hidden, H, True
    cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = gs, H // gs, rank_k, rank_v
    m = LlamaPaluAttention(cfg, 0)
    g = torch.Generator().manual_seed(3)
    qb = (torch.randn(H * D, generator=g) * 0.5).half()
    ob = (torch.randn(hidden, generator=g) * 0.02).half()
    with torch.no_grad():
        m.q_proj.weight.copy_(w["wq"])
        m.q_proj.bias.copy_(qb)
        m.k_proj.VT.weight.copy_(w["vt_k"])
        m.v_proj.VT.weight.copy_(w["vt_v"])
        m.o_proj.weight.copy_(w["wo"])
        m.o_proj.bias.copy_(ob)
    m.k_proj.B = nn.Parameter(build_b(w["u_k"], gs, D))
    m = m.eval().to(DEV, torch.float16)
    if bits == 16:
        cache = LatentCache()
        cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
        k_ref, v_ref, lb = k_lat, v_lat, None
    else:
        cache = QuantLatentCache(bits)
        cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
        k_ref = oracle.quantize_rows(k_lat.reshape(-1, k_lat.shape[-1]), bits)[0].reshape(k_lat.shape)
        v_ref = oracle.quantize_rows(v_lat.reshape(-1, v_lat.shape[-1]), bits)[0].reshape(v_lat.shape)
        lb = bits
    with torch.no_grad():
        out, probs, _ = m(tok.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L, L + 1), past_key_value=cache,
                          output_attentions=True)
    assert cache.get_seq_length(0) == L + 1
    pad = 8
    tok_a = torch.cat((tok.reshape(-1), torch.ones(1, dtype=tok.dtype), torch.zeros(pad - 1, dtype=tok.dtype)))
    aug = lambda wt, b=None: torch.cat((wt.half(), (torch.zeros(wt.shape[0], 1) if b is None else b.reshape(-1, 1)).half(),
                                        torch.zeros(wt.shape[0], pad - 1).half()), dim=1)
    wd = {"wq": aug(w["wq"], qb), "vt_k": aug(w["vt_k"]), "vt_v": aug(w["vt_v"]),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    kw = {} if lb is None else {"latent_bits": lb}
    o2, p2, _, _ = oracle.decode_step(tok_a, L, wd, k_ref, v_ref, **kw)
    o2 = (o2.float() + ob.float())
    torch.testing.assert_close(probs.cpu().reshape(H, L + 1).float(), p2.float(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu().reshape(-1).float(), o2, rtol=1e-3, atol=1e-3)
    # the bias really took part
    assert (qb.abs().max() > 0.1) and float((out.cpu().reshape(-1).float() - (o2 - ob.float())).abs().max()) > 1e-2


def test_hip_decode_linears_match_the_torch_mlp_and_lm_head():
    """palu_amd.hf.use_hip_decode_linears: the one-token calls of every gated MLP (gate / up GEMVs with the SiLU product in
    the epilogue, then the down GEMV) and of lm_head on the HIP GEMV kernels; prompt passes keep the torch modules.  Logits of
    a prompt and of three decode steps against the same model without the substitution."""
    transformers = pytest.importorskip("transformers")
    from transformers import LlamaConfig, LlamaForCausalLM
    from palu_amd.hf import PaluCacheHF, convert_llama_to_palu, use_hip_decode_linears, _DecodeGatedMLP, _DecodeGemvLinear
    import copy
    torch.manual_seed(3)
    cfg = LlamaConfig(vocab_size=1000, hidden_size=512, intermediate_size=1376, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, head_dim=128, max_position_embeddings=512, rope_theta=10000.0,
                      attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    base = convert_llama_to_palu(LlamaForCausalLM(cfg).to(DEV, torch.float16).eval(), rank_k=512, rank_v=512, group_size=2)
    fast = use_hip_decode_linears(copy.deepcopy(base))
    assert all(isinstance(l.mlp, _DecodeGatedMLP) and l.mlp._ok for l in fast.model.layers)
    assert isinstance(fast.lm_head, _DecodeGemvLinear) and fast.lm_head._ok
    ids = torch.randint(0, 1000, (1, 21), device=DEV)
    c1, c2 = PaluCacheHF(bits=16), PaluCacheHF(bits=16)
    with torch.no_grad():
        a = base(ids, past_key_values=c1, use_cache=True).logits
        b = fast(ids, past_key_values=c2, use_cache=True).logits
    assert torch.equal(a, b)                                   # a prompt pass takes the wrapped torch modules
    tok = a[:, -1:].argmax(-1)
    for _ in range(3):
        with torch.no_grad():
            a = base(tok, past_key_values=c1, use_cache=True).logits
            b = fast(tok, past_key_values=c2, use_cache=True).logits
        torch.testing.assert_close(b.float(), a.float(), rtol=2e-2, atol=2e-2)
        tok = a[:, -1:].argmax(-1)
    # the fused gate/up kernel alone against the torch composition
    mlp = base.model.layers[0].mlp
    x = torch.randn(1, 1, 512, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        want = mlp(x)
        got = fast.model.layers[0].mlp(x)
    torch.testing.assert_close(got.float(), want.float(), rtol=5e-3, atol=5e-3)
