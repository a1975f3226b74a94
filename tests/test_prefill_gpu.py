"""GPU parity of the flash-style prefill attention kernel (palu_prefill_attn_f16) and of the prompt pass through
LlamaPaluAttention.forward against the golden vectors generated from the reference (tests/golden/g7_prefill.npz),
the CPU oracle, and -- for the kernel alone at sizes the oracle cannot hold -- a plain fp32 torch restatement of
softmax(q.k^T * scale [+causal]) . V.  Criterion P1 of SURVEY.md 8(c): assert_close(rtol=1e-3, atol=1e-3)."""
import math
import os

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

import oracle
from tests.golden import inputs as gi
from tests.test_decode_gpu import _module_from_palu_weights

DEV = "cuda"


def _lib():
    from palu_amd import _lib
    return _lib


def prefill_attn(q, k, v_lat, past, causal, scale=None):
    """q [H,Tq,D], k [H,Tk,D], v_lat [G,Tk,Rv] fp16 cuda -> ctx [Tq, H*Rv] through the C ABI."""
    lib = _lib()
    H, Tq, D = q.shape
    Tk = k.shape[1]
    G, _, Rv = v_lat.shape
    pad = (Tk + 63) // 64 * 64
    vt = torch.zeros(G, Rv, pad, dtype=torch.float16, device=DEV)
    vt[:, :, :Tk].copy_(v_lat.transpose(1, 2))
    out = torch.empty(Tq, H * Rv, dtype=torch.float16, device=DEV)
    lib.check(lib.lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                            vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(), out.stride(0),
                                            H, G, D, Tq, Tk, Rv, past, 1 if causal else 0,
                                            1.0 / math.sqrt(D) if scale is None else scale,
                                            torch.cuda.current_stream().cuda_stream), "prefill_attn")
    return out


def ref_attn_f32(q, k, v_lat, past, causal, scale):
    H, Tq, D = q.shape
    Tk = k.shape[1]
    G, _, Rv = v_lat.shape
    s = torch.matmul(q.float(), k.float().transpose(1, 2)) * scale
    if causal:
        i = torch.arange(Tq, device=q.device).unsqueeze(1) + past
        j = torch.arange(Tk, device=q.device).unsqueeze(0)
        s = s.masked_fill((j > i).unsqueeze(0), float("-inf"))
    p = torch.softmax(s, dim=-1)
    ctx = torch.matmul(p.reshape(G, (H // G) * Tq, Tk), v_lat.float()).reshape(H, Tq, Rv)
    return ctx.transpose(0, 1).reshape(Tq, H * Rv)


@pytest.mark.parametrize("H,gs,Tq,Tk,Rv,causal", [
    (4, 2, 48, 48, 64, True), (4, 2, 70, 70, 64, False), (8, 4, 130, 130, 96, True), (8, 4, 257, 257, 192, True),
    (32, 4, 300, 300, 384, True), (4, 1, 1, 200, 32, False), (8, 4, 129, 1000, 384, True), (8, 2, 640, 640, 128, True),
    (4, 4, 128, 192, 384, False), (8, 4, 333, 333, 256, True), (32, 4, 1100, 1100, 384, True),
    (8, 4, 200, 200, 320, True), (8, 2, 500, 700, 448, True), (4, 2, 77, 77, 160, False)])
def test_prefill_kernel_vs_fp32(H, gs, Tq, Tk, Rv, causal):
    rng = np.random.default_rng(H * 1000 + Tq + Tk + Rv)
    G = H // gs
    past = Tk - Tq
    q = torch.from_numpy(rng.standard_normal((H, Tq, 128)).astype(np.float16)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((H, Tk, 128)).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((G, Tk, Rv)).astype(np.float16)).to(DEV)
    scale = 1.0 / math.sqrt(128.0)
    out = prefill_attn(q, k, v, past, causal)
    ref = ref_attn_f32(q, k, v, past, causal, scale)
    # tolerance of the north star: 1e-3 relative to the output scale (P is rounded to fp16 inside the kernel like :238)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    assert torch.isfinite(out.float()).all()


def test_prefill_kernel_strided_and_peaked():
    """Non-contiguous q/k (head-interleaved [T, H, D] storage) and a peaked distribution (large logits)."""
    rng = np.random.default_rng(7)
    H, gs, T, Rv = 8, 4, 200, 96
    qs = torch.from_numpy((rng.standard_normal((T, H, 128)) * 4).astype(np.float16)).to(DEV)
    ks = torch.from_numpy((rng.standard_normal((T, H, 128)) * 4).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((H // gs, T, Rv)).astype(np.float16)).to(DEV)
    q, k = qs.transpose(0, 1), ks.transpose(0, 1)
    out = prefill_attn(q, k, v, 0, True)
    ref = ref_attn_f32(q, k, v, 0, True, 1.0 / math.sqrt(128.0))
    assert (out.float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_prefill_kernel_rejects_bad_args():
    lib = _lib()
    q = torch.zeros(4, 8, 128, dtype=torch.float16, device=DEV)
    v = torch.zeros(2, 64, 64, dtype=torch.float16, device=DEV)
    out = torch.zeros(8, 4 * 64, dtype=torch.float16, device=DEV)
    rc = lib.lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), q.data_ptr(), q.stride(0), q.stride(1),
                                       v.data_ptr(), v.stride(0), 8, out.data_ptr(), out.stride(0), 4, 2, 128, 8, 8, 64,
                                       0, 1, 0.1, 0)
    assert rc != 0 and b"zero-padded" in lib.lib.palu_last_error()
    rc = lib.lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), q.data_ptr(), q.stride(0), q.stride(1),
                                       v.data_ptr(), v.stride(0), 64, out.data_ptr(), out.stride(0), 4, 2, 64, 8, 8, 64,
                                       0, 1, 0.1, 0)
    assert rc != 0 and b"head_dim" in lib.lib.palu_last_error()


@pytest.mark.parametrize("case", gi.PREFILL_CASES, ids=[c[0] for c in gi.PREFILL_CASES])
@pytest.mark.parametrize("form", ["default", "latent"])
def test_prefill_module_golden(golden_dir, case, form):
    """Prompt pass through LlamaPaluAttention.forward (flash kernel) vs the reference's own outputs.  form = "latent": the kernel that
    rebuilds the keys per kv tile itself (csrc/prefill_lat.hip) forced onto the fixtures whose ranks it takes (32 / 64 and 32 / 96 per group: all three)."""
    from palu_amd.kernel.palu_attention import LatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, T, causal = case
    g = np.load(os.path.join(golden_dir, "g7_prefill.npz"))
    w, prompt, mask = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, causal)
    flat = [w["wq"], w["vt_k"], w["vt_v"], w["wo"], *w["u_k"], prompt]
    assert gi.digest(*flat) == str(g[tag + "/digest"])
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = LatentCache()
    am = None if mask is None else mask.reshape(1, 1, T, T).to(DEV)
    calls = []
    if form == "latent":
        G = H // gs
        if not _lib().lib.palu_prefill_attn_lat_supported(H, G, D, rank_k // G, rank_v // G):
            pytest.skip("ranks the latent kernel does not take")
        m.PREFILL_LATENT_ABOVE = 0
        inner = m._prefill_latent
        m._prefill_latent = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    with torch.no_grad():
        out, probs, _ = m(prompt.reshape(1, T, hidden).to(DEV), attention_mask=am,
                          position_ids=torch.arange(T).unsqueeze(0), past_key_value=cache)
    assert probs is None and cache.get_seq_length(0) == T
    assert (form != "latent") or calls == [1]
    torch.testing.assert_close(out[0].cpu(), torch.from_numpy(g[tag + "/attn_output"]), rtol=1e-3, atol=1e-3)
    kbuf, vbuf = cache.buffers(0)
    torch.testing.assert_close(kbuf[0, :, :T].cpu(), torch.from_numpy(g[tag + "/k_lat"]), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(vbuf[0, :, :T].cpu(), torch.from_numpy(g[tag + "/v_lat"]), rtol=1e-3, atol=1e-3)
    # the general (torch-composition) path of the same module agrees as well, incl. the attention weights
    cache2 = LatentCache()
    with torch.no_grad():
        out2, probs2, _ = m(prompt.reshape(1, T, hidden).to(DEV), attention_mask=am,
                            position_ids=torch.arange(T).unsqueeze(0), past_key_value=cache2, output_attentions=True)
    torch.testing.assert_close(out2[0].cpu(), torch.from_numpy(g[tag + "/attn_output"]), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(probs2[0].cpu(), torch.from_numpy(g[tag + "/attn_weights"]), rtol=1e-3, atol=1e-3)


def test_prefill_chunked_equals_whole_and_feeds_decode():
    """Causal prompt in two chunks (second chunk sees `past` rows) == one pass; the cache it leaves behind drives the
    fused decode step to the same answer as the oracle's decode on the oracle's prefill latents."""
    from palu_amd.kernel.palu_attention import LatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, T, _ = gi.PREFILL_CASES[0]
    w, prompt, mask = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, True)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    x = prompt.reshape(1, T, hidden).to(DEV)
    c1, c2 = LatentCache(), LatentCache()
    with torch.no_grad():
        whole, _, _ = m(x, past_key_value=c1, is_causal=True)
        t1 = 20
        a, _, _ = m(x[:, :t1], past_key_value=c2, is_causal=True)
        b, _, _ = m(x[:, t1:], past_key_value=c2, is_causal=True, position_ids=torch.arange(t1, T).unsqueeze(0))
    torch.testing.assert_close(torch.cat((a, b), dim=1), whole, rtol=1e-3, atol=1e-3)
    # reference semantics of the whole pass
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "u_k": [u.half() for u in w["u_k"]], "wo": w["wo"].half()}
    o_ref, _, k_ref, v_ref = oracle.prefill(prompt, wd, mask)
    torch.testing.assert_close(whole[0].cpu(), o_ref, rtol=1e-3, atol=1e-3)
    # decode one token on top of the chunked cache
    rng = np.random.default_rng(77)
    tok = torch.from_numpy(rng.standard_normal(hidden).astype(np.float16))
    with torch.no_grad():
        d, _, _ = m(tok.reshape(1, 1, hidden).to(DEV), past_key_value=c2, position_ids=torch.tensor([[T]]))
    wd2 = dict(wd, b=oracle.build_b_from_u(w["u_k"], gs, D).half())
    d_ref, _, _, _ = oracle.decode_step(tok, T, wd2, k_ref, v_ref)
    torch.testing.assert_close(d.reshape(-1).cpu(), d_ref, rtol=1e-3, atol=1e-3)


def test_prefill_long_prompt_properties():
    """8k-token causal prompt at the C2 ranks (no oracle at this size): finite, and causal prefix invariance --
    the first 1000 output rows do not depend on the tokens that follow."""
    from palu_amd.kernel.palu_attention import LatentCache
    hidden, H, D, gs, rank_k, rank_v = 4096, 32, 128, 4, 1024, 3072
    w, _, _, _, _ = gi.step_inputs(5, hidden, H, D, gs, rank_k, rank_v, 1, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 8192, hidden, generator=g).half().to(DEV)
    with torch.no_grad():
        full, _, _ = m(x, past_key_value=LatentCache(), is_causal=True)
        head, _, _ = m(x[:, :1000], past_key_value=LatentCache(), is_causal=True)
    assert torch.isfinite(full.float()).all()
    torch.testing.assert_close(full[:, :1000], head, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("bits,rank_k,rank_v", [(4, 64, 128), (3, 256, 256)])   # 3-bit decode needs R_k = 128
def test_prefill_quantised_cache(bits, rank_k, rank_v):
    """Prompt pass into a packed 3/4-bit latent cache: the flash path attends over the de-quantised rows like the
    accuracy path (oracle.prefill(latent_bits=...)); the packed cache then serves the fused quantised decode step."""
    from palu_amd.kernel.palu_attention import QuantLatentCache
    tag, seed, hidden, H, D, gs, _, _, T, _ = gi.PREFILL_CASES[0]
    w, prompt, mask = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, True)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = QuantLatentCache(bits)
    with torch.no_grad():
        out, _, _ = m(prompt.reshape(1, T, hidden).to(DEV), past_key_value=cache, is_causal=True)
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "u_k": [u.half() for u in w["u_k"]], "wo": w["wo"].half()}
    o_ref, _, k_ref, v_ref = oracle.prefill(prompt, wd, mask, latent_bits=bits)
    # latents come from a different GEMM (MFMA vs CPU) so a row may land in the neighbouring quantisation bin:
    # compare against the output scale, not element-wise at 1e-3
    err = (out[0].cpu().float() - o_ref.float()).abs().max().item()
    assert err <= 2e-2 * o_ref.float().abs().max().item(), err
    assert cache.get_seq_length(0) == T
    rng = np.random.default_rng(78)
    tok = torch.from_numpy(rng.standard_normal(hidden).astype(np.float16))
    with torch.no_grad():
        d, _, _ = m(tok.reshape(1, 1, hidden).to(DEV), past_key_value=cache, position_ids=torch.tensor([[T]]))
    wd2 = dict(wd, b=oracle.build_b_from_u(w["u_k"], gs, D).half())
    d_ref, _, _, _ = oracle.decode_step(tok, T, wd2, k_ref, v_ref, latent_bits=bits)
    err = (d.reshape(-1).cpu().float() - d_ref.float()).abs().max().item()
    assert err <= 2e-2 * d_ref.float().abs().max().item(), err


@pytest.mark.parametrize("H,T,pos0,strided", [(4, 130, 0, False), (32, 257, 1000, True), (8, 64, 65000, False)])
def test_rope_inplace_matches_reference_rounding(H, T, pos0, strided):
    """palu_rope_f16 == the fp16 tensor-op sequence of the prompt branch (oracle.rope_cos_sin cast to fp16,
    x*cos + rotate_half(x)*sin evaluated in fp16): identical up to the last fp16 bit of a few table entries."""
    from palu_amd.kernel.abx_rope import rope_inv_freq
    lib = _lib()
    rng = np.random.default_rng(T + pos0)
    if strided:
        base = torch.from_numpy(rng.standard_normal((T, H, 128)).astype(np.float16))
        x_cpu = base.transpose(0, 1)
    else:
        x_cpu = torch.from_numpy(rng.standard_normal((H, T, 128)).astype(np.float16))
    cos, sin = oracle.rope_cos_sin(pos0 + T, 128, 10000.0, start=pos0)
    cos, sin = cos.half(), sin.half()
    ref = x_cpu * cos + torch.cat((-x_cpu[..., 64:], x_cpu[..., :64]), dim=-1) * sin
    xg = (base.to(DEV).transpose(0, 1) if strided else x_cpu.to(DEV))
    inv = rope_inv_freq(torch.device(DEV), 128, 10000.0)
    lib.check(lib.lib.palu_rope_f16(xg.data_ptr(), xg.stride(0), xg.stride(1), H, T, 128, pos0, inv.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "rope")
    got = xg.cpu()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)
    assert (got != ref).float().mean().item() < 0.02          # the odd fp16 ulp from cos/sin of large angles


@pytest.mark.parametrize("quant_bits", [16, 4])
def test_three_layers_share_one_cache(quant_bits):
    """A small stack of attention modules (layer_idx 0..2) threading one cache object, as a decoder would: prompt pass
    then two decode steps; every layer's rows land in its own buffers and the outputs equal those of the same modules
    run with private caches."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, T, _ = gi.PREFILL_CASES[0]
    mods = []
    for li in range(3):
        w, prompt, _ = gi.prefill_inputs(seed + li, hidden, H, D, gs, rank_k, rank_v, T, True)
        m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
        m.layer_idx = li
        mods.append(m)
    mk = (lambda: LatentCache()) if quant_bits == 16 else (lambda: QuantLatentCache(quant_bits))
    shared, private = mk(), [mk() for _ in mods]
    x = prompt.reshape(1, T, hidden).to(DEV)
    rng = np.random.default_rng(5)
    toks = [torch.from_numpy(rng.standard_normal((1, 1, hidden)).astype(np.float16)).to(DEV) for _ in range(2)]
    with torch.no_grad():
        hs, hp_ = x, x
        for li, m in enumerate(mods):                        # prompt pass, layer by layer
            hs, _, _ = m(hs, past_key_value=shared, is_causal=True)
            m.layer_idx = 0
            hp_, _, _ = m(hp_, past_key_value=private[li], is_causal=True)
            m.layer_idx = li
        torch.testing.assert_close(hs, hp_, rtol=0, atol=0)
        for step, tok in enumerate(toks):
            hs, hp_ = tok, tok
            pos = torch.tensor([[T + step]])
            for li, m in enumerate(mods):
                hs, _, _ = m(hs, past_key_value=shared, position_ids=pos)
                m.layer_idx = 0
                hp_, _, _ = m(hp_, past_key_value=private[li], position_ids=pos)
                m.layer_idx = li
            torch.testing.assert_close(hs, hp_, rtol=0, atol=0)
    assert [shared.get_seq_length(li) for li in range(3)] == [T + 2] * 3


@pytest.mark.parametrize("mode", ["chunks", "panels"])
def test_prefill_without_causal_mask_in_chunks_equals_one_launch(mode):
    """with no mask and is_causal unset the prompt pass applies NO mask (the reference's semantics,
    palu_attention.py:229-234): every query attends every key of the pass.  Forced query chunks / kv panels must project the
    latents of ALL chunks before the first chunk attends -- same output as the one-launch form."""
    from palu_amd.kernel.palu_attention import LatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, T, _ = gi.PREFILL_CASES[2]       # T = 130: three chunks of 48 / two of 128
    w, prompt, _ = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, True)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    x = prompt.reshape(1, T, hidden).to(DEV)
    c1, c2 = LatentCache(), LatentCache()
    with torch.no_grad():
        ref, _, _ = m(x, past_key_value=c1, is_causal=False)
        causal_ref, _, _ = m(x, past_key_value=LatentCache(), is_causal=True)
        if mode == "chunks":
            m.PREFILL_WORKSPACE_BUDGET, m.PREFILL_QUERY_CHUNK = 0, 48
        else:
            m.PREFILL_PANEL_ROWS, m.PREFILL_PANEL_QUERY = 64, 128
        try:
            out, _, _ = m(x, past_key_value=c2, is_causal=False)
        finally:
            if mode == "chunks":
                del m.PREFILL_WORKSPACE_BUDGET, m.PREFILL_QUERY_CHUNK
            else:
                del m.PREFILL_PANEL_ROWS, m.PREFILL_PANEL_QUERY
    assert c2.get_seq_length(0) == T
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-3, atol=2e-3)
    assert (ref.float() - causal_ref.float()).abs().max() > 1e-2          # (the two mask semantics really differ on this prompt)
    for a, b in zip(c1.buffers(0), c2.buffers(0)):
        assert torch.equal(a[:, :, :T], b[:, :, :T])


@pytest.mark.parametrize("bits", [16, 4])
def test_prefill_in_query_chunks_and_groups_equals_one_launch(bits):
    """Long prompts run in query chunks x latent groups with bounded workspaces (LlamaPaluAttention._prefill_flash); forced
    here on a golden-sized prompt: same output and same cache contents as the one-launch form, fp16 and packed caches."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache
    tag, seed, hidden, H, D, gs, rank_k, rank_v, T, _ = gi.PREFILL_CASES[0]
    if bits < 16:
        rank_k, rank_v = 64 * (H // gs), 128 * (H // gs)
    w, prompt, _ = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, True)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    x = prompt.reshape(1, T, hidden).to(DEV)
    mk = (lambda: LatentCache()) if bits == 16 else (lambda: QuantLatentCache(bits))
    c1, c2 = mk(), mk()
    with torch.no_grad():
        ref, _, _ = m(x, past_key_value=c1, is_causal=True)
        m.PREFILL_WORKSPACE_BUDGET, m.PREFILL_QUERY_CHUNK = 0, 48          # 48: not a multiple of the kernel's 128-query tile
        try:
            out, _, _ = m(x, past_key_value=c2, is_causal=True)
        finally:
            del m.PREFILL_WORKSPACE_BUDGET, m.PREFILL_QUERY_CHUNK
    assert c1.get_seq_length(0) == c2.get_seq_length(0) == T
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-3, atol=2e-3)
    if bits == 16:
        for a, b in zip(c1.buffers(0), c2.buffers(0)):
            assert torch.equal(a[:, :, :T], b[:, :, :T])
    else:
        for k in ("kc", "km", "vc", "vm"):
            assert torch.equal(c1.buffers(0)[k][:, :, :T], c2.buffers(0)[k][:, :, :T])


def test_prefill_transient_memory_is_bounded():
    """a long prompt pass need not materialise K~ for every head, V^T for every group, the context of every
    query and (packed cache) a dequantised copy of the whole cache at once: 32k tokens at the config-2 ranks, 4-bit cache.
    The bounded-workspace form (query chunks x latent groups; the default once the one-launch form would need more than
    PREFILL_WORKSPACE_BUDGET = 6 GiB) allocates less than half of the one-launch form's transients and the same output."""
    from palu_amd.kernel.palu_attention import LlamaPaluAttention, QuantLatentCache, build_b
    hidden, H, D, gs, rank_k, rank_v, T = 4096, 32, 128, 4, 1024, 3072, 32768

    class Cfg:
        pass
    cfg = Cfg()
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = hidden, H, False
    cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = gs, H // gs, rank_k, rank_v
    torch.manual_seed(0)
    with torch.device(DEV):
        m = LlamaPaluAttention(cfg, 0).half()
        with torch.no_grad():
            for lin in (m.q_proj, m.k_proj.VT, m.v_proj.VT, m.o_proj):
                lin.weight.normal_(0.0, 0.02)
            for u in m.k_proj.U_list:
                u.weight.normal_(0.0, 128 ** -0.5)
        m.k_proj.B = nn.Parameter(build_b([u.weight for u in m.k_proj.U_list], gs, D))
    m = m.eval().prepare_decode()
    x = torch.randn(1, T, hidden, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        m(x[:, :256], past_key_value=QuantLatentCache(4), is_causal=True)       # warm-up: library handles, fragments

    m.PREFILL_LATENT_ABOVE = None               # (the workspace forms: the latent form has its own bound, tests/test_prefill_lat_gpu.py)

    def run(budget):
        cache = QuantLatentCache(4, capacity=T + 512)
        cache.reserve(0, T + 512, H // gs, rank_k // (H // gs), rank_v // (H // gs), x.device)
        m.PREFILL_WORKSPACE_BUDGET = budget
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            with torch.no_grad():
                out, _, _ = m(x, past_key_value=cache, is_causal=True)
            torch.cuda.synchronize()
            extra = torch.cuda.max_memory_allocated() - base - out.numel() * 2
        finally:
            del m.PREFILL_WORKSPACE_BUDGET
        assert cache.get_seq_length(0) == T
        return out, extra
    one, mem_one = run(1 << 50)
    bounded, mem_b = run(0)
    torch.testing.assert_close(bounded.float(), one.float(), rtol=2e-3, atol=2e-3)
    assert mem_b < 0.5 * mem_one and mem_b < 768 << 20, (mem_b / 2 ** 20, mem_one / 2 ** 20)


def prefill_attn_panels(q, k, v_lat, past, causal, C):
    """The same through palu_prefill_attn_panel_f16: kv panels of C rows, fp32 state between the launches."""
    lib = _lib()
    H, Tq, D = q.shape
    Tk = k.shape[1]
    G, _, Rv = v_lat.shape
    out = torch.full((Tq, H * Rv), float("nan"), dtype=torch.float16, device=DEV)
    st_o = torch.full((lib.lib.palu_prefill_state_bytes(H, Tq, Rv, 0) // 4,), float("nan"), dtype=torch.float32, device=DEV)
    st_ml = torch.full((lib.lib.palu_prefill_state_bytes(H, Tq, Rv, 1) // 4,), float("nan"), dtype=torch.float32, device=DEV)
    starts = list(range(0, Tk, C))
    for pi, k0 in enumerate(starts):
        n = min(Tk, k0 + C) - k0
        kp = k[:, k0:k0 + n].contiguous()
        pad = (n + 63) // 64 * 64
        vt = torch.zeros(G, Rv, pad, dtype=torch.float16, device=DEV)
        vt[:, :, :n].copy_(v_lat[:, k0:k0 + n].transpose(1, 2))
        lib.check(lib.lib.palu_prefill_attn_panel_f16(q.data_ptr(), q.stride(0), q.stride(1), kp.data_ptr(), kp.stride(0),
                                                      kp.stride(1), vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(),
                                                      out.stride(0), H, G, D, Tq, n, Rv, past - k0, 1 if causal else 0,
                                                      1.0 / math.sqrt(D), st_o.data_ptr(), st_ml.data_ptr(),
                                                      1 if pi == 0 else 0, 1 if pi == len(starts) - 1 else 0,
                                                      torch.cuda.current_stream().cuda_stream), "prefill_attn_panel")
    return out


@pytest.mark.parametrize("H,gs,Tq,Tk,Rv,causal,C", [
    (8, 4, 300, 300, 384, True, 64), (8, 4, 300, 300, 384, True, 128), (32, 4, 257, 1000, 384, True, 448),
    (8, 4, 130, 130, 96, True, 64), (8, 2, 200, 700, 448, True, 192), (4, 2, 70, 333, 64, False, 100 // 64 * 64 + 64),
    (8, 4, 129, 129, 192, True, 1024), (4, 4, 1, 500, 384, True, 128), (8, 4, 640, 640, 256, True, 320),
    # grids of several workgroups per CU whose latent columns are split over blockIdx.z (Rv = 448: 7 column blocks, Rv = 160:
    # 5): every column block carries its own (m, l) slice of the state (they used to share one entry)
    (32, 4, 2100, 2100, 448, True, 512), (32, 4, 1500, 1500, 160, True, 384)])
def test_prefill_panels_equal_one_launch(H, gs, Tq, Tk, Rv, causal, C):
    """kv panels with the carried online-softmax state (palu_prefill_attn_panel_f16) against the one-launch kernel and the
    fp32 restatement: panels that end inside a 64-row tile, panels wholly in a query tile's causal future (its state passes
    through), a single panel, both kernel variants (two waves per 32 queries for Rv <= 384, column chunks above)."""
    rng = np.random.default_rng(H + Tq + Tk + Rv + C)
    G = H // gs
    past = Tk - Tq
    q = torch.from_numpy(rng.standard_normal((H, Tq, 128)).astype(np.float16)).to(DEV)
    k = torch.from_numpy(rng.standard_normal((H, Tk, 128)).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((G, Tk, Rv)).astype(np.float16)).to(DEV)
    scale = 1.0 / math.sqrt(128.0)
    one = prefill_attn(q, k, v, past, causal)
    pan = prefill_attn_panels(q, k, v, past, causal, C)
    ref = ref_attn_f32(q, k, v, past, causal, scale)
    assert torch.isfinite(pan.float()).all()
    tol = 2e-3 * max(1.0, ref.abs().max().item())
    assert (pan.float() - ref).abs().max().item() <= tol
    assert (pan.float() - one.float()).abs().max().item() <= tol


def _c2_module(T_warm=256):
    from palu_amd.kernel.palu_attention import LlamaPaluAttention, QuantLatentCache, build_b
    hidden, H, D, gs, rank_k, rank_v = 4096, 32, 128, 4, 1024, 3072

    class Cfg:
        pass
    cfg = Cfg()
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = hidden, H, False
    cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = gs, H // gs, rank_k, rank_v
    torch.manual_seed(0)
    with torch.device(DEV):
        m = LlamaPaluAttention(cfg, 0).half()
        with torch.no_grad():
            for lin in (m.q_proj, m.k_proj.VT, m.v_proj.VT, m.o_proj):
                lin.weight.normal_(0.0, 0.02)
            for u in m.k_proj.U_list:
                u.weight.normal_(0.0, 128 ** -0.5)
        m.k_proj.B = nn.Parameter(build_b([u.weight for u in m.k_proj.U_list], gs, D))
    m = m.eval().prepare_decode()
    with torch.no_grad():
        m(torch.randn(1, T_warm, hidden, device=DEV, dtype=torch.float16), past_key_value=QuantLatentCache(4), is_causal=True)
    return m


@pytest.mark.parametrize("bits", [16, 4])
def test_prefill_module_in_panels_equals_one_launch(bits):
    """The prompt pass of a config-2 module in kv panels (PREFILL_PANEL_ROWS) -- two prompt chunks, so the second one has a
    past -- equals the one-launch pass and leaves the same cache rows."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache
    m = _c2_module()
    T = 2300
    x = torch.randn(1, T, 4096, device=DEV, dtype=torch.float16)
    mk = (lambda: LatentCache()) if bits == 16 else (lambda: QuantLatentCache(bits))

    def run(panel):
        cache = mk()
        if panel:
            m.PREFILL_PANEL_ROWS, m.PREFILL_PANEL_QUERY, m.PREFILL_PANEL_GROUPS = 448, 256, 3
        try:
            with torch.no_grad():
                a, _, _ = m(x[:, :1500], past_key_value=cache, is_causal=True)
                b, _, _ = m(x[:, 1500:], past_key_value=cache, is_causal=True)
        finally:
            if panel:
                del m.PREFILL_PANEL_ROWS, m.PREFILL_PANEL_QUERY, m.PREFILL_PANEL_GROUPS
        return torch.cat((a, b), dim=1), cache
    ref, c1 = run(False)
    out, c2 = run(True)
    assert c1.get_seq_length(0) == c2.get_seq_length(0) == T
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-3, atol=2e-3)
    if bits == 16:
        for a, b in zip(c1.buffers(0), c2.buffers(0)):
            assert torch.equal(a[:, :, :T], b[:, :, :T])
    else:
        # the packed path projects each query chunk with a torch GEMM, whose rounding depends on the chunk's row count (1500
        # rows at once vs 256-row chunks): (scale, zero) pairs agree to fp16 rounding, codes almost everywhere
        for k in ("km", "vm"):
            torch.testing.assert_close(c1.buffers(0)[k][:, :, :T].float(), c2.buffers(0)[k][:, :, :T].float(), rtol=1e-2, atol=1.0)
        for k in ("kc", "vc"):
            same = (c1.buffers(0)[k][:, :, :T] == c2.buffers(0)[k][:, :, :T]).float().mean().item()
            assert same > 0.97, same


def test_prefill_in_panels_needs_64_mib_at_64k_tokens():
    """peak EXTRA memory of a 64k-token prompt pass into a packed (4-bit) cache <= 64 MiB -- the kv-panel
    form at its defaults (512 queries x 4 latent groups x 2048-row panels): no transient grows with the prompt."""
    from palu_amd.kernel.palu_attention import QuantLatentCache
    m = _c2_module()
    H, gs, rank_k, rank_v, T = 32, 4, 1024, 3072, 65536
    x = torch.randn(1, T, 4096, device=DEV, dtype=torch.float16)
    cache = QuantLatentCache(4, capacity=T + 512)
    cache.reserve(0, T + 512, H // gs, rank_k // (H // gs), rank_v // (H // gs), x.device)
    m.PREFILL_PANEL_ROWS = 2048
    try:
        with torch.no_grad():
            m(x[:, :1024], past_key_value=QuantLatentCache(4), is_causal=True)       # the panel path's own warm-up
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        with torch.no_grad():
            out, _, _ = m(x, past_key_value=cache, is_causal=True)
        torch.cuda.synchronize()
        extra = torch.cuda.max_memory_allocated() - base - out.numel() * 2
    finally:
        del m.PREFILL_PANEL_ROWS
    assert cache.get_seq_length(0) == T
    assert torch.isfinite(out.float()).all()
    assert extra <= 64 << 20, extra / 2 ** 20
    # the last rows against the one-launch kernel on the same cache contents: rows T-256.. as a chunk with a past
    ref_cache = QuantLatentCache(4, capacity=T + 512)
    ref_cache.reserve(0, T + 512, H // gs, rank_k // (H // gs), rank_v // (H // gs), x.device)
    for k in ("kc", "km", "vc", "vm"):
        ref_cache.buffers(0)[k][:, :, :T - 256].copy_(cache.buffers(0)[k][:, :, :T - 256])
    ref_cache.advance(0, T - 256)
    with torch.no_grad():
        tail, _, _ = m(x[:, T - 256:], past_key_value=ref_cache, is_causal=True)
    torch.testing.assert_close(out[:, T - 256:].float(), tail.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("bits", [4, 3])
@pytest.mark.parametrize("M,N,K,R,row0", [(1024, 1024, 4096, 128, 0), (1000, 3072, 4096, 384, 37), (8192, 1024, 4096, 128, 0),
                                          (777, 512, 1024, 64, 5), (600, 1536, 2048, 192, 0), (2048, 768, 512, 96, 3),
                                          (4096, 2048, 1024, 256, 0), (530, 256, 512, 32, 1)])
def test_projection_gemm_fused_quantise_epilogue_is_bit_exact(bits, M, N, K, R, row0):
    """palu_lowrank_project_gemm_q (projection + quantise + pack in the GEMM's epilogue) against
    palu_lowrank_project_gemm followed by palu_quantize_pack: codes and (scale, zero) pairs bit for bit; and the dequantised
    rows against the oracle's quantize_rows on the fp16 GEMM output (the g5 semantics)."""
    lib = _lib()
    from palu_amd.kernel import quant as pq
    assert lib.lib.palu_lowrank_project_gemm_q_supported(M, N, K, R, bits) == 1
    rng = np.random.default_rng(M + N + K + R + bits)
    G = N // R
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float16)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((N, K)) * K ** -0.5 * rng.uniform(0.3, 3.0, (N, 1))).astype(np.float16)).to(DEV)
    cap = row0 + M + 9
    lat = torch.zeros(G, cap, R, dtype=torch.float16, device=DEV)
    lib.check(lib.lib.palu_lowrank_project_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), lat.data_ptr(),
                                                lat.stride(0), lat.stride(1), M, N, K, R, row0,
                                                torch.cuda.current_stream().cuda_stream), "gemm")
    rows = lat[:, row0:row0 + M].contiguous()
    codes_ref, meta_ref = pq.quantize_pack(rows, bits)
    rb = R * bits // 8
    codes = torch.full((G, cap, rb), 0xEE, dtype=torch.uint8, device=DEV)
    meta = torch.full((G, cap, 2), -7.0, dtype=torch.float16, device=DEV)
    lib.check(lib.lib.palu_lowrank_project_gemm_q(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), codes.data_ptr(),
                                                  codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0),
                                                  meta.stride(1), M, N, K, R, row0, bits,
                                                  torch.cuda.current_stream().cuda_stream), "gemm_q")
    assert torch.equal(meta[:, row0:row0 + M], meta_ref)
    assert torch.equal(codes[:, row0:row0 + M], codes_ref)
    # rows outside [row0, row0 + M) untouched
    assert bool((codes[:, :row0] == 0xEE).all()) and bool((codes[:, row0 + M:] == 0xEE).all())
    assert bool((meta[:, :row0] == -7.0).all()) and bool((meta[:, row0 + M:] == -7.0).all())
    deq = pq.unpack_dequant(codes[:, row0:row0 + M].contiguous(), meta[:, row0:row0 + M].contiguous(), bits, R)
    want = oracle.quantize_rows(rows[:, :64].cpu().reshape(-1, R), bits)[0].reshape(G, 64, R)
    assert torch.equal(deq[:, :64].cpu(), want)


def test_projection_gemm_q_unsupported_shapes_say_so():
    lib = _lib()
    sup = lib.lib.palu_lowrank_project_gemm_q_supported
    assert sup(256, 1024, 4096, 128, 4) == 0        # short chunks run the small-tile kernel
    assert sup(4096, 1280, 4096, 160, 4) == 0       # 160 columns per group do not tile
    assert sup(4096, 1024, 4096, 128, 8) == 0
    assert sup(4096, 1024, 4096, 128, 3) == 1 and sup(4096, 3072, 4096, 384, 4) == 1
