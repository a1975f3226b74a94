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
    if Rk % 64:                                                  # the GEMM takes K in whole 64s: zero columns change nothing
        pad_k = 64 - Rk % 64
        xk = torch.nn.functional.pad(xk, (0, pad_k))
        b = torch.nn.functional.pad(b, (0, 0, 0, pad_k))
        Rk += pad_k
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


@pytest.mark.parametrize("H,gs,Tq,Tk,Rk,Rv,causal", [
    (4, 4, 128, 128, 128, 384, True), (8, 4, 130, 130, 128, 384, True), (8, 4, 257, 257, 128, 128, True), (32, 4, 300, 300, 128, 384, True),
    (4, 1, 1, 200, 128, 256, False), (8, 4, 129, 1000, 128, 384, True), (4, 4, 128, 192, 128, 384, False), (8, 4, 333, 333, 128, 256, True),
    (32, 4, 1100, 1100, 128, 384, True), (8, 2, 64, 4097, 128, 384, True), (4, 4, 200, 70, 128, 384, False),
    (32, 4, 700, 700, 64, 192, True), (8, 4, 129, 1000, 64, 192, False), (8, 4, 300, 300, 64, 384, True), (8, 4, 257, 520, 128, 192, True),   # config-4 ranks
    (4, 2, 48, 48, 32, 64, True), (4, 2, 70, 333, 32, 64, False), (8, 4, 130, 130, 32, 96, True), (8, 2, 77, 400, 32, 96, False),   # config-1 and the golden fixtures' ranks
])
def test_latent_prefill_kernel_vs_workspace_form_and_fp32(H, gs, Tq, Tk, Rk, Rv, causal):
    _lib, ar = _mods()
    G = H // gs
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
    assert lib.palu_prefill_attn_lat_supported(32, 8, 128, 64, 192) == 1           # config-4 ranks
    assert lib.palu_prefill_attn_lat_supported(32, 8, 128, 32, 96) == 1            # config-1 ranks (fp16 rows)
    assert lib.palu_prefill_attn_lat_supported(32, 8, 128, 32, 128) == 0
    assert lib.palu_prefill_attn_lat_supported(32, 8, 64, 128, 384) == 0
    for shape, want in (((128, 384), (1, 1, 1)), ((64, 192), (1, 1, 0)), ((32, 96), (1, 0, 0)), ((128, 192), (1, 1, 0)), ((128, 256), (1, 1, 1))):
        assert tuple(lib.palu_prefill_attn_lat_supported_bits(32, 8, 128, *shape, b) for b in (16, 4, 3)) == want
    assert lib.palu_prefill_attn_lat_supported_bits(32, 8, 128, 128, 384, 8) == 0
    t = torch.zeros(1024, dtype=torch.float16, device=DEV)
    rc = lib.palu_prefill_attn_lat_f16(t.data_ptr(), 128, 128, t.data_ptr(), 128, 128, t.data_ptr(), 192, 192, t.data_ptr(), t.data_ptr(),
                                       t.data_ptr(), 192, 4, 1, 128, 1, 1, 32, 160, 0, 1, 0.1, torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    # the packed entry refuses what the query says it does not take: 3-bit rows at rank_k / G = 64, 4-bit rows at rank_v / G = 96
    u8 = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    for Rk, Rv, bits in ((64, 192, 3), (32, 96, 4), (128, 384, 5)):
        rc = lib.palu_prefill_attn_lat_q(t.data_ptr(), 128, 128, u8.data_ptr(), 4096, 64, t.data_ptr(), 64, 2, u8.data_ptr(), 4096, 192,
                                         t.data_ptr(), 64, 2, t.data_ptr(), t.data_ptr(), t.data_ptr(), 4 * Rv, 4, 1, 128, 1, 1, Rk, Rv, bits,
                                         0, 1, 0.1, torch.cuda.current_stream().cuda_stream)
        assert rc != 0, (Rk, Rv, bits)


# ------------------------------------------------------------------------------------------------- module level
def _module(hidden, H, gs, Rk, Rv, seed=0, n_rep=1):
    """n_rep > 1: grouped-query attention -- H query heads over H / n_rep KV heads, gs KV heads per latent group."""
    from torch import nn
    from palu_amd.kernel.palu_attention import LlamaPaluAttention, build_b

    class Cfg:
        pass
    cfg = Cfg()
    kv = H // n_rep
    G = kv // gs
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = hidden, H, False
    cfg.num_key_value_heads = kv
    cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = gs, G, Rk * G, Rv * G
    torch.manual_seed(seed)
    with torch.device(DEV):
        m = LlamaPaluAttention(cfg, 0).half()
        with torch.no_grad():
            for lin in (m.q_proj, m.k_proj.VT, m.v_proj.VT, m.o_proj):
                lin.weight.normal_(0.0, 0.03)
            for u in m.k_proj.U_list:
                u.weight.normal_(0.0, Rk ** -0.5)
        m.k_proj.B = nn.Parameter(build_b([u.weight for u in m.k_proj.U_list], gs, D, n_rep))
    return m.eval().prepare_decode()


@pytest.mark.parametrize("bits", [16, 4])
def test_gqa_prompt_pass_in_latent_form(bits):
    """Grouped-query attention (the Mistral shape of BASELINE config 4: 4 query heads per KV head, one KV head per latent group, ranks
    64 / 192): a latent group serves n_rep query heads with one B per KV head -- latent form against workspace form, then a decode step."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache
    hidden, H, T = 1024, 8, 600
    m = _module(hidden, H, 1, 64, 192, n_rep=4)
    assert m.n_rep == 4 and m.num_groups == 2 and m.group_size == 4
    x = torch.randn(1, T, hidden, device=DEV, dtype=torch.float16)
    xd = torch.randn(1, 1, hidden, device=DEV, dtype=torch.float16)
    outs = {}
    for mode, above in (("workspace", None), ("latent", 0)):
        cache = LatentCache() if bits == 16 else QuantLatentCache(bits)
        m.PREFILL_LATENT_ABOVE, m.PREFILL_LATENT_QUERY_CHUNK = above, 256
        calls = []
        inner = m._prefill_latent
        m._prefill_latent = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
        try:
            with torch.no_grad():
                o, _, _ = m(x, past_key_value=cache, is_causal=True)
                od, _, _ = m(xd, past_key_value=cache, position_ids=torch.tensor([[T]]))
        finally:
            del m.PREFILL_LATENT_ABOVE, m.PREFILL_LATENT_QUERY_CHUNK, m._prefill_latent
        assert bool(calls) == (mode == "latent")
        outs[mode] = (o, od)
    for a, b in zip(outs["latent"], outs["workspace"]):
        torch.testing.assert_close(a.float(), b.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("ranks", [(128, 384), (64, 192)], ids=["config2_ranks", "config4_ranks"])
@pytest.mark.parametrize("bits", [16, 4, 3])
@pytest.mark.parametrize("causal", [True, False])
def test_prompt_pass_in_latent_form_equals_workspace_form_and_feeds_decode(causal, bits, ranks):
    """LlamaPaluAttention.forward with the latent kernel forced (PREFILL_LATENT_ABOVE = 0, query chunks of 256 -- and 200, not a
    multiple of the kernel's 128-query tile) against the workspace form (None): same output, identical cache contents, a second
    prompt pass on top of the first (past > 0), and a decode step from either cache."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache
    hidden, H, gs, T1, T2 = 1024, 8, 4, 700, 333
    Rk, Rv = ranks
    if bits == 3 and Rk != 128:
        pytest.skip("3-bit rows in the latent kernel: rank_k / G = 128")
    m = _module(hidden, H, gs, Rk, Rv)
    x1 = torch.randn(1, T1, hidden, device=DEV, dtype=torch.float16)
    x2 = torch.randn(1, T2, hidden, device=DEV, dtype=torch.float16)
    xd = torch.randn(1, 1, hidden, device=DEV, dtype=torch.float16)
    outs = {}
    for mode, above, chunk in (("workspace", None, 2048), ("latent", 0, 256), ("latent200", 0, 200)):
        cache = LatentCache() if bits == 16 else QuantLatentCache(bits)
        m.PREFILL_LATENT_ABOVE, m.PREFILL_LATENT_QUERY_CHUNK = above, chunk
        m.PREFILL_LATENT_PROJECT_ROWS = 500 if mode == "latent200" else 16384    # projections ahead of the attention: 500-row blocks / all at once
        try:
            with torch.no_grad():
                o1, _, _ = m(x1, past_key_value=cache, is_causal=causal)
                o2, _, _ = m(x2, past_key_value=cache, is_causal=causal, position_ids=torch.arange(T1, T1 + T2).unsqueeze(0))
                od, _, _ = m(xd, past_key_value=cache, position_ids=torch.tensor([[T1 + T2]]))
        finally:
            del m.PREFILL_LATENT_ABOVE, m.PREFILL_LATENT_QUERY_CHUNK, m.PREFILL_LATENT_PROJECT_ROWS
        bufs = cache.buffers(0) if bits == 16 else [cache.buffers(0)[k] for k in ("kc", "km", "vc", "vm")]
        outs[mode] = (o1, o2, od, [b[:, :, :T1 + T2 + 1].clone() for b in bufs])
    ref = outs["workspace"]
    for mode in ("latent", "latent200"):
        got = outs[mode]
        for a, b in zip(got[:3], ref[:3]):
            torch.testing.assert_close(a.float(), b.float(), rtol=2e-3, atol=2e-3)
        for a, b in zip(got[3], ref[3]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("bits", [16, 4, 3])
def test_latent_prompt_pass_needs_128_mib_of_transients_at_32k_tokens(bits):
    """the prompt pass without the [H, kv, D] key workspace and the transposed value copy.  32k tokens at the
    config-2 ranks into an fp16 cache: the latent form's transients (rotated queries + context rows of one 2048-query chunk, the
    chunk's q_proj / o_proj outputs) stay below 128 MiB; the one-launch workspace form needs > 1 GiB; same output."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache
    hidden, H, gs, Rk, Rv, T = 4096, 32, 4, 128, 384, 32768
    m = _module(hidden, H, gs, Rk, Rv)
    x = torch.randn(1, T, hidden, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        m(x[:, :256], past_key_value=LatentCache() if bits == 16 else QuantLatentCache(bits), is_causal=True)   # warm-up: handles, fragments
    from palu_amd.kernel.abx_rope import rope_cs_table
    rope_cs_table(torch.device(DEV), D, m.rope_theta, T)                         # (persistent, like HF's cos / sin cache)

    def run(above):
        if bits == 16:
            cache = LatentCache(capacity=T + 512)
            cache.reserve(0, T + 512, torch.empty((1, H // gs, 0, Rk), dtype=torch.float16, device=DEV),
                          torch.empty((1, H // gs, 0, Rv), dtype=torch.float16, device=DEV))
        else:
            cache = QuantLatentCache(bits, capacity=T + 512)
            cache.reserve(0, T + 512, H // gs, Rk, Rv, torch.device(DEV))
        m.PREFILL_LATENT_ABOVE = above
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
            del m.PREFILL_LATENT_ABOVE
        assert cache.get_seq_length(0) == T
        return out, extra
    lat, mem_lat = run(0)
    one, mem_one = run(None)
    torch.testing.assert_close(lat.float(), one.float(), rtol=2e-3, atol=2e-3)
    assert mem_lat <= 128 << 20 and mem_one > 1 << 30, (mem_lat / 2 ** 20, mem_one / 2 ** 20)


# ------------------------------------------------------------------------------------------------- packed 4-bit caches
def bt_perm_of(b):
    """B^T [H, D, R] with the columns of every group of 8 in the order the nibbles of a dword leave it: 0 4 1 5 2 6 3 7."""
    H, R, _ = b.shape
    idx = torch.tensor([0, 4, 1, 5, 2, 6, 3, 7], device=b.device)
    return b.transpose(1, 2).reshape(H, D, R // 8, 8)[..., idx].reshape(H, D, R).contiguous()


@pytest.mark.parametrize("shape", [
    (4, 4, 128, 128, 128, 384, True), (8, 4, 257, 257, 128, 128, True), (32, 4, 300, 300, 128, 384, True), (8, 4, 129, 1000, 128, 384, True),
    (4, 4, 200, 70, 128, 384, False), (8, 4, 333, 333, 128, 256, True), (8, 2, 64, 4097, 128, 384, True),
    (32, 4, 700, 700, 64, 192, True), (8, 4, 129, 1000, 64, 192, False), (8, 4, 257, 520, 128, 192, True),                                     # config-4 ranks
    (4, 2, 130, 130, 32, 64, True),
    # 3-bit rows (config 3: 3-bit latents + Hadamard -- the rotation lives in the weights): rank_k / G = 128, rank_v / G in {128, 256, 384}
    (4, 4, 128, 128, 128, 384, True, 3), (8, 4, 257, 257, 128, 128, True, 3), (32, 4, 300, 300, 128, 384, True, 3), (8, 4, 129, 1000, 128, 384, True, 3),
    (4, 4, 200, 70, 128, 384, False, 3), (8, 4, 333, 333, 128, 256, True, 3), (8, 2, 64, 4097, 128, 384, True, 3), (4, 1, 1, 200, 128, 256, False, 3),
])
def test_latent_prefill_kernel_on_packed_caches(shape):
    """palu_prefill_attn_lat_q de-quantises the codes inside the kernel: same result as the fp16 kernel on unpack_dequant()'s rows
    (identical fp16 values enter the MFMAs; only the rebuild's summation order differs through the permuted B^T columns), and the
    fp32 evaluation on those rows."""
    from palu_amd.kernel.quant import quantize_pack, unpack_dequant
    H, gs, Tq, Tk, Rk, Rv, causal = shape[:7]
    bits = shape[7] if len(shape) > 7 else 4
    _lib, ar = _mods()
    lib, S = _lib.lib, torch.cuda.current_stream().cuda_stream
    G = H // gs
    g = torch.Generator().manual_seed(7 * H + Tq + Tk + Rv)
    past = Tk - Tq if causal and Tk >= Tq else 0
    q = torch.randn(H, Tq, D, generator=g).half().to(DEV)
    cap = Tk + 5
    xk = torch.randn(G, cap, Rk, generator=g).half().to(DEV)
    xv = torch.randn(G, cap, Rv, generator=g).half().to(DEV)
    b = (torch.randn(H, Rk, D, generator=g) * Rk ** -0.5).half().to(DEV)
    kc, km = quantize_pack(xk, bits)
    vc, vm = quantize_pack(xv, bits)
    kc[:, Tk:], vc[:, Tk:] = 0xFF, 0xFF                                      # rows beyond Tk hold garbage
    km[:, Tk:], vm[:, Tk:] = float("nan"), float("nan")
    xkd = unpack_dequant(kc[:, :Tk].contiguous(), km[:, :Tk].contiguous(), bits, Rk)
    xvd = unpack_dequant(vc[:, :Tk].contiguous(), vm[:, :Tk].contiguous(), bits, Rv)
    inv = ar.rope_inv_freq(torch.device(DEV))
    ref16 = _lat_form(q, xkd, xvd, b, past, causal, inv, Tk)
    cs = torch.empty(lib.palu_rope_cs_table_bytes(Tk), dtype=torch.uint8, device=DEV)
    _lib.check(lib.palu_rope_cs_table_build(inv.data_ptr(), 0, Tk, cs.data_ptr(), S), "cs")
    btp = bt_perm_of(b)
    out = torch.empty(Tq, H * Rv, dtype=torch.float16, device=DEV)
    _lib.check(lib.palu_prefill_attn_lat_q(q.data_ptr(), q.stride(0), q.stride(1), kc.data_ptr(), kc.stride(0), kc.stride(1),
                                           km.data_ptr(), km.stride(0), km.stride(1), vc.data_ptr(), vc.stride(0), vc.stride(1),
                                           vm.data_ptr(), vm.stride(0), vm.stride(1), btp.data_ptr(), cs.data_ptr(), out.data_ptr(),
                                           out.stride(0), H, G, D, Tq, Tk, Rk, Rv, bits, past, 1 if causal else 0, 1.0 / math.sqrt(D), S),
               "prefill_attn_lat_q")
    assert torch.isfinite(out).all()
    scale = ref16.float().abs().max().item()
    assert (out.float() - ref16.float()).abs().max().item() <= 1e-3 * scale + 1e-3
    _, keys = _workspace_form(q, xkd, xvd, b, past, causal, inv)
    ref = _ref_f32(q, keys, xvd, past, causal)
    assert (out.float() - ref).abs().max().item() <= 2e-3 * scale + 1e-3
