"""CPU tests of the host-side mirror (no GPU): the reference's own CPU tests
(kernel/test_palu_attention.py:34-133) restated against palu_amd, plus the latent cache."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from palu_amd.kernel.palu_attention import (HeadwiseLowRankModule, LatentCache, LlamaPaluAttention, build_b,
                                            fuse_wo)


class Cfg:
    def __init__(self, hidden=512, heads=4, gs=2, rk=None, rv=None):
        self.hidden_size = hidden
        self.num_attention_heads = heads
        self.attention_bias = False
        self.group_size = gs
        self.num_groups = heads // gs
        self.total_rank_k = hidden if rk is None else rk
        self.total_rank_v = hidden if rv is None else rv


class DenseAttn(nn.Module):
    """Minimal stand-in for LlamaAttention: what from_attention reads (q/k/v/o_proj, layer_idx, head_dim)."""

    def __init__(self, cfg, layer_idx=0):
        super().__init__()
        d = cfg.hidden_size
        self.layer_idx = layer_idx
        self.head_dim = d // cfg.num_attention_heads
        self.q_proj, self.k_proj = nn.Linear(d, d, bias=False), nn.Linear(d, d, bias=False)
        self.v_proj, self.o_proj = nn.Linear(d, d, bias=False), nn.Linear(d, d, bias=False)


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(0)


def test_lr_layer_init():                                  # test_palu_attention.py:34-53
    m = HeadwiseLowRankModule([2, 2], 6, 6, False)
    x = torch.randn(1, 5, 6)
    torch.testing.assert_close(m(x), m.reconstruct(m.project_to_latent(x)))
    with pytest.raises(ValueError):
        HeadwiseLowRankModule([2, 2, 2, 2], 6, 6, False)   # out_features % num_groups (:27-31)
    with pytest.raises(AssertionError):
        m(torch.randn(5, 6))


def test_lr_layer_from_linear():                           # :55-74 full rank -> lossless
    lin = nn.Linear(10, 6, False)
    svd = HeadwiseLowRankModule.from_linear(lin, [3, 3])
    x = torch.randn(1, 5, 10)
    torch.testing.assert_close(lin(x), svd(x))


def test_inherit_no_fusion():                              # :76-90
    cfg = Cfg()
    attn = DenseAttn(cfg)
    palu = LlamaPaluAttention.from_attention(attn, cfg, no_fusion=True)
    torch.testing.assert_close(attn.q_proj.weight, palu.q_proj.weight)
    torch.testing.assert_close(attn.o_proj.weight, palu.o_proj.weight)
    x = torch.randn(1, 16, cfg.hidden_size)
    torch.testing.assert_close(attn.k_proj(x), palu.k_proj(x), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(attn.v_proj(x), palu.v_proj(x), rtol=1e-4, atol=1e-5)


def test_inherit_fusion_algebra():                         # :92-133
    cfg = Cfg()
    H, gs, D = cfg.num_attention_heads, cfg.group_size, cfg.hidden_size // cfg.num_attention_heads
    G, Rv = H // gs, cfg.total_rank_v // (H // gs)
    attn = DenseAttn(cfg)
    palu = LlamaPaluAttention.from_attention(attn, cfg)
    assert palu.o_proj.weight.shape == (cfg.hidden_size, H * Rv)
    assert palu.k_proj.B.shape == (H, cfg.total_rank_k // G, D)
    q_len = 12
    x = torch.randn(1, q_len, cfg.hidden_size)
    w = torch.randn(1, H, q_len, q_len)
    v = attn.v_proj(x).view(1, q_len, H, D).transpose(1, 2)
    ref = attn.o_proj(torch.matmul(w, v).transpose(1, 2).reshape(1, q_len, -1))
    vh = palu.v_proj.project_to_latent(x).reshape(1, q_len, G, Rv).transpose(1, 2)
    lat = torch.matmul(w.reshape(1, G, q_len * gs, q_len), vh).reshape(1, H, q_len, Rv)
    got = palu.o_proj(lat.transpose(1, 2).reshape(1, q_len, -1))
    torch.testing.assert_close(ref, got, rtol=1e-3, atol=1e-4)


def test_prefill_forward_matches_dense_attention_cpu_fp32():
    """Full-rank Palu prefill (torch-composed branch) == vanilla attention, fp32 on CPU."""
    cfg = Cfg()
    H, D = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    attn = DenseAttn(cfg)
    palu = LlamaPaluAttention.from_attention(attn, cfg).eval()
    T = 9
    x = torch.randn(1, T, cfg.hidden_size)
    out, probs, _ = palu(x, output_attentions=True)
    # vanilla
    import oracle
    q = attn.q_proj(x).view(T, H, D).transpose(0, 1)
    k = attn.k_proj(x).view(T, H, D).transpose(0, 1)
    v = attn.v_proj(x).view(T, H, D).transpose(0, 1)
    cos, sin = oracle.rope_cos_sin(T, D)
    q, k = oracle.rope_rotate(q, cos, sin), oracle.rope_rotate(k, cos, sin)
    p = torch.softmax(q @ k.transpose(1, 2) / D ** 0.5, dim=-1)
    ref = attn.o_proj((p @ v).transpose(0, 1).reshape(1, T, H * D))
    torch.testing.assert_close(probs[0], p, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-4)
    # no_fusion variant takes the dense o_proj branch
    palu2 = LlamaPaluAttention.from_attention(attn, cfg, no_fusion=True).eval()
    out2, _, _ = palu2(x)
    torch.testing.assert_close(out2, ref, rtol=1e-3, atol=1e-4)


def test_b_layout_and_fusion_against_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_layout.npz"))
    gs, D = int(g["gs"]), int(g["D"])
    np.testing.assert_array_equal(build_b([torch.from_numpy(u) for u in g["u_k"]], gs, D).numpy(), g["b"])
    fused = fuse_wo(torch.from_numpy(g["wo"]), [torch.from_numpy(u) for u in g["u_v"]], D)
    np.testing.assert_allclose(fused.numpy(), g["wo_fused"], rtol=0, atol=1e-6)


def test_forward_argument_validation():
    cfg = Cfg()
    palu = LlamaPaluAttention(cfg, None)
    x = torch.randn(1, 3, cfg.hidden_size)
    with pytest.raises(ValueError):          # cache without layer index (:179-184)
        palu(x, past_key_value=LatentCache())
    palu = LlamaPaluAttention(cfg, 0)
    with pytest.raises(ValueError):          # mask shape (:230-233)
        palu(x, attention_mask=torch.zeros(1, 1, 3, 4))
    with pytest.raises(RuntimeError):        # decode on CPU: no fallback for the HIP op
        palu.k_proj.B = nn.Parameter(torch.zeros(4, 128, 128))
        c = LatentCache()
        palu(x, past_key_value=c)
        palu(x[:, :1].half(), past_key_value=c)


def test_latent_cache_protocol():
    c = LatentCache(capacity=8, headroom=2)
    assert c.get_usable_length(1, 0) == 0
    k0, v0 = torch.randn(1, 2, 5, 4), torch.randn(1, 2, 5, 6)
    ka, va = c.update(k0, v0, 0)
    assert ka.shape == (1, 2, 5, 4) and va.shape == (1, 2, 5, 6) and c.get_usable_length(1, 0) == 5
    torch.testing.assert_close(ka, k0)
    base = c.buffers(0)[0].data_ptr()
    k1, v1 = torch.randn(1, 2, 1, 4), torch.randn(1, 2, 1, 6)
    ka, va = c.update(k1, v1, 0)
    assert c.buffers(0)[0].data_ptr() == base, "append must be in place while capacity lasts"
    torch.testing.assert_close(ka, torch.cat((k0, k1), dim=2))
    torch.testing.assert_close(va, torch.cat((v0, v1), dim=2))
    for _ in range(20):                      # growth keeps the prefix
        ka, va = c.update(k1, v1, 0)
    assert ka.shape[2] == 26 and c.capacity(0) >= 26
    torch.testing.assert_close(ka[:, :, :5], k0)
    c.update(k0, v0, 2)                      # sparse layer indices
    assert c.get_seq_length(2) == 5 and c.get_seq_length(1) == 0 and len(c) == 3
    with pytest.raises(ValueError):
        c.update(k0[0], v0[0], 0)


def test_cache_save_load_roundtrip(tmp_path):
    """On-disk form of the latent caches (safetensors container): valid rows, layout and
    metadata survive a round trip for the fp16 and the packed cache, several layers, empty layers included."""
    from palu_amd.kernel.palu_attention import LatentCache, QuantLatentCache, load_cache, save_cache
    from palu_amd.kernel.quant import packed_row_bytes
    g = torch.Generator().manual_seed(5)
    c = LatentCache(capacity=100)
    for layer, n in ((0, 37), (2, 5)):
        c.update(torch.randn(1, 2, n, 32, generator=g).half(), torch.randn(1, 2, n, 64, generator=g).half(), layer)
    c.update(torch.randn(1, 2, 3, 32, generator=g).half(), torch.randn(1, 2, 3, 64, generator=g).half(), 0)
    path = str(tmp_path / "fp16.safetensors")
    save_cache(c, path)
    d = load_cache(path)
    assert isinstance(d, LatentCache) and len(d) == 3
    assert [d.get_seq_length(i) for i in range(3)] == [40, 0, 5]
    for i in (0, 2):
        n = c.get_seq_length(i)
        for a, b in zip(c.buffers(i), d.buffers(i)):
            assert torch.equal(a[:, :, :n], b[:, :, :n])
    # packed cache: fill the buffers directly (quantisation itself is a GPU kernel, tested in the gpu suite)
    q = QuantLatentCache(3)
    q.reserve(0, 50, 2, 64, 128, "cpu")
    st = q.buffers(0)
    assert st["kc"].shape[-1] == packed_row_bytes(64, 3) == 24
    for key in ("kc", "vc"):
        st[key][:, :, :41] = torch.randint(0, 256, st[key][:, :, :41].shape, generator=g, dtype=torch.uint8)
    for key in ("km", "vm"):
        st[key][:, :, :41] = torch.randn(st[key][:, :, :41].shape, generator=g).half()
    q.advance(0, 41)
    path = str(tmp_path / "packed.safetensors")
    save_cache(q, path)
    r = load_cache(path)
    assert isinstance(r, QuantLatentCache) and r.n_bits == 3 and r.get_seq_length(0) == 41
    rt = r.buffers(0)
    assert (rt["Rk"], rt["Rv"]) == (64, 128)
    for key in ("kc", "km", "vc", "vm"):
        assert torch.equal(st[key][:, :, :41], rt[key][:, :, :41])
    with pytest.raises(ValueError):
        from safetensors.torch import save_file
        save_file({"x": torch.zeros(1)}, str(tmp_path / "other.safetensors"))
        load_cache(str(tmp_path / "other.safetensors"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_from_linear_whiten_against_reference_fixture(golden_dir, tag):
    """HeadwiseLowRankModule.from_linear_whiten (svd_linear.py:170-204) on the inputs the reference ran (g9_whiten.npz): the
    decomposed module reproduces the reference module's forward, its factors up to SVD signs, the bias stays on the U side;
    a layer without `scaling_diag_matrix` raises the reference's error."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, "g9_whiten.npz"))
    ranks = [int(r) for r in g[f"{tag}/ranks"]]
    w = torch.from_numpy(g[f"{tag}/w"])
    has_bias = f"{tag}/b" in g
    lin = nn.Linear(w.shape[1], w.shape[0], bias=has_bias)
    with torch.no_grad():
        lin.weight.copy_(w)
        if has_bias:
            lin.bias.copy_(torch.from_numpy(g[f"{tag}/b"]))
    with pytest.raises(FileExistsError):
        HeadwiseLowRankModule.from_linear_whiten(lin, ranks)
    lin.scaling_diag_matrix = torch.from_numpy(g[f"{tag}/scaling"])
    mod = HeadwiseLowRankModule.from_linear_whiten(lin, ranks)
    vt_ref = torch.from_numpy(g[f"{tag}/vt"])
    r0 = 0
    for i, r in enumerate(ranks):
        u_ref = torch.from_numpy(g[f"{tag}/u{i}"])
        got = mod.U_list[i].weight.data @ mod.VT.weight.data[r0:r0 + r]
        torch.testing.assert_close(got, u_ref @ vt_ref[r0:r0 + r], rtol=0, atol=2e-5)
        if has_bias:
            assert torch.equal(mod.U_list[i].bias.data, torch.from_numpy(g[f"{tag}/bias{i}"]))
        r0 += r
    x = torch.from_numpy(g[f"{tag}/x"])
    torch.testing.assert_close(mod(x), torch.from_numpy(g[f"{tag}/y"]), rtol=1e-5, atol=1e-5)


def test_hf_adapter_converts_a_mask_once_per_forward_pass(monkeypatch):
    """palu_amd.hf.PaluAttentionHF: the boolean mask transformers hands to every layer is converted (and, for prompts, tested
    for causality) once per forward PASS -- decode steps included: three small launches per layer otherwise -- keyed on the
    mask's identity and version; the last layer drops the entry, an entry of another pass is never used."""
    import types
    from palu_amd import hf

    calls = []
    real = hf.additive_mask
    monkeypatch.setattr(hf, "additive_mask", lambda m, dt: (calls.append(m), real(m, dt))[1])

    class Inner(nn.Module):
        def __init__(self, idx):
            super().__init__()
            self.layer_idx, self.config, self.seen = idx, types.SimpleNamespace(num_hidden_layers=3), []

        def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                    is_causal=None):
            self.seen.append(attention_mask)
            return hidden_states, None, None

    class Cache:
        _mask_memo = None

        def get_seq_length(self, layer_idx=0):
            return 7

    layers = [hf.PaluAttentionHF(Inner(i)) for i in range(3)]
    cache = Cache()
    x = torch.zeros(1, 1, 8, dtype=torch.float16)

    def one_pass(mask):
        for l in layers:
            l(x, attention_mask=mask, past_key_values=cache)

    m1 = torch.ones(1, 1, 1, 8, dtype=torch.bool)
    one_pass(m1)
    assert len(calls) == 1 and cache._mask_memo is None                        # once for three layers; dropped by the last
    assert all(l.inner.seen[-1] is layers[0].inner.seen[-1] for l in layers)   # the same converted tensor
    assert layers[0].inner.seen[-1].dtype == torch.float16
    m2 = torch.ones(1, 1, 1, 8, dtype=torch.bool)
    one_pass(m2)
    assert len(calls) == 2
    # a pass that stopped early leaves an entry: a different mask, or the same one mutated in place, must not inherit it
    layers[0](x, attention_mask=m2, past_key_values=cache)
    assert len(calls) == 3 and cache._mask_memo is not None
    m3 = torch.zeros(1, 1, 1, 8, dtype=torch.bool)
    layers[1](x, attention_mask=m3, past_key_values=cache)
    assert len(calls) == 4 and float(layers[1].inner.seen[-1].max()) < 0
    m3[..., 0] = True
    layers[2](x, attention_mask=m3, past_key_values=cache)
    assert len(calls) == 5 and float(layers[2].inner.seen[-1][..., 0]) == 0 and cache._mask_memo is None
    layers[0](x, attention_mask=None, past_key_values=cache)                   # no mask: nothing converted, nothing kept
    assert len(calls) == 5 and layers[0].inner.seen[-1] is None and cache._mask_memo is None
