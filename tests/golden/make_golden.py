#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE on CPU.

Run only in the build container (needs /root/reference, read-only):

    python tests/golden/make_golden.py

Nothing of the reference's source travels: this script imports it, feeds it the seeded
inputs of tests/golden/inputs.py and stores inputs-digests + outputs as small .npz files.
The transformers-4.37.2 behaviours the reference module relies on (attributes set by
LlamaAttention.__init__, the cached rotary table, the 5-argument apply_rotary_pos_emb and
the DynamicCache protocol) are re-created below as a shim because this image ships
transformers 5.x; the shim is this build's own text.
"""
from __future__ import annotations

import math
import os
import sys
import types

import numpy as np
import scipy.linalg
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from tests.golden import inputs as gi  # noqa: E402
import oracle  # noqa: E402  (only to cross-check the restatement while generating)

torch.set_num_threads(8)


# ------------------------------------------------------------------ import shims
def _install_fht_stub():
    """fast_hadamard_transform is an un-vendored CUDA submodule; give the importer a dense
    scipy Sylvester matmul (independent of this build's FWHT) so hadamard_utils imports."""
    stub = types.ModuleType("fast_hadamard_transform")

    def hadamard_transform(x, scale=1.0):
        n = x.shape[-1]
        h = torch.from_numpy(scipy.linalg.hadamard(n).astype(np.float32)).to(x.dtype)
        return (x @ h) * scale

    stub.hadamard_transform = hadamard_transform
    sys.modules["fast_hadamard_transform"] = stub


class _Rotary437(nn.Module):
    """transformers 4.37.2 LlamaRotaryEmbedding behaviour: fp32 table, sliced and cast."""

    def __init__(self, dim, max_pos, base):
        super().__init__()
        inv = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        t = torch.arange(max_pos, dtype=inv.dtype)
        emb = torch.outer(t, inv)
        emb = torch.cat((emb, emb), dim=-1)
        self.register_buffer("cos_cached", emb.cos(), persistent=False)
        self.register_buffer("sin_cached", emb.sin(), persistent=False)

    def forward(self, x, seq_len=None):
        return self.cos_cached[:seq_len].to(x.dtype), self.sin_cached[:seq_len].to(x.dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _apply_rotary_437(q, k, cos, sin, position_ids, unsqueeze_dim=1):
    cos = cos[position_ids].unsqueeze(unsqueeze_dim)
    sin = sin[position_ids].unsqueeze(unsqueeze_dim)
    return q * cos + _rot_half(q) * sin, k * cos + _rot_half(k) * sin


class _Cache437:
    """DynamicCache protocol of 4.37.2 as used by palu_attention.py:185,193."""

    def __init__(self):
        self.k, self.v = None, None

    def get_usable_length(self, new_len, layer_idx=0):
        return 0 if self.k is None else self.k.shape[-2]

    def update(self, k, v, layer_idx, cache_kwargs=None):
        self.k = k if self.k is None else torch.cat((self.k, k), dim=-2)
        self.v = v if self.v is None else torch.cat((self.v, v), dim=-2)
        return self.k, self.v


def _import_reference():
    _install_fht_stub()
    from transformers.models.llama import modeling_llama as ml

    orig_init = ml.LlamaAttention.__init__

    def init437(self, config, layer_idx=None):
        orig_init(self, config, layer_idx)
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.max_position_embeddings = config.max_position_embeddings
        self.rotary_emb = _Rotary437(self.head_dim, config.max_position_embeddings, 10000.0)

    ml.LlamaAttention.__init__ = init437
    import kernel.palu_attention as pa
    import kernel.abx_rope as ar
    import kernel.pytorch_reference as pr
    pa.apply_rotary_pos_emb = _apply_rotary_437
    pa.recompute_k_gemv = ar.torch_abx          # Triton cannot launch here; the kernel's own oracle
    from palu.model.modules import quant as rq
    from palu.model.modules import hadamard_utils as rh
    from palu.model.modules import svd_linear as rs
    return ml, pa, ar, pr, rq, rh, rs


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------ G1 abx
def gen_abx(ar):
    print("G1 abx (torch_abx, kernel/abx_rope.py:152-171)")
    out = {}
    for tag, seed, H, D, gs, R, L, regime in gi.ABX_CASES:
        a, b, x = gi.abx_inputs(seed, H, D, gs, R, L, regime)
        ref = ar.torch_abx(a, b, x)                                   # [H,1,L] fp16
        mine = oracle.abx_scores(a, b, x)
        d = (ref.float() - mine.float()).abs().max().item()
        print(f"  {tag}: max|ref-oracle| = {d:.3e}  max|ref| = {ref.float().abs().max():.1f}")
        out[tag + "/out"] = _np(ref)
        out[tag + "/digest"] = np.array(gi.digest(a, b, x))
    _save("g1_abx", **out)


# ----------------------------------------------------------------------- G2 rope
def gen_rope(pr):
    print("G2 RoPE tables (kernel/pytorch_reference.py:3-9)")
    positions = [0, 1, 63, 2047, 65535, 262143]
    cos, sin = pr.LlamaRotaryEmbedding(dim=128, end=262144)
    idx = torch.tensor(positions)
    x = torch.from_numpy(np.random.default_rng(7).standard_normal((3, 6, 128)).astype(np.float32))
    rot = pr.apply_rotary_pos_emb_pytorch(x, cos[idx], sin[idx])
    _save("g2_rope", positions=np.array(positions), cos=_np(cos[idx]), sin=_np(sin[idx]),
          x=_np(x), rotated=_np(rot))


# -------------------------------------------------------------- G3 decode steps
def _ref_module_from_palu_weights(ml, pa, hidden, H, D, gs, rank_k, rank_v, w):
    cfg = ml.LlamaConfig(hidden_size=hidden, num_attention_heads=H, num_key_value_heads=H,
                         max_position_embeddings=4096)
    cfg.group_size, cfg.num_groups = gs, H // gs
    cfg.total_rank_k, cfg.total_rank_v = rank_k, rank_v
    m = pa.LlamaPaluAttention(cfg, 0)
    with torch.no_grad():
        m.q_proj.weight.copy_(w["wq"])
        m.k_proj.VT.weight.copy_(w["vt_k"])
        m.v_proj.VT.weight.copy_(w["vt_v"])
        for i, u in enumerate(w["u_k"]):
            m.k_proj.U_list[i].weight.copy_(u)
        m.o_proj.weight.copy_(w["wo"])
    # B layout: restated in oracle.build_b_from_u and verified against the reference in G4
    m.k_proj.B = nn.Parameter(oracle.build_b_from_u(w["u_k"], gs, D))
    return m.eval().to(torch.float16), cfg


def gen_steps(ml, pa):
    print("G3 decode steps (kernel/palu_attention.py:147-263, decode branch)")
    out = {}
    for tag, seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask in gi.STEP_CASES:
        w, k_lat, v_lat, tok, mask = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask)
        m, cfg = _ref_module_from_palu_weights(ml, pa, hidden, H, D, gs, rank_k, rank_v, w)
        cache = _Cache437()
        cache.update(k_lat.unsqueeze(0), v_lat.unsqueeze(0), 0)
        pos = torch.arange(L, L + 1)
        am = None if mask is None else mask.reshape(1, 1, 1, L + 1)
        with torch.no_grad():
            o, p, _ = m(tok.reshape(1, 1, hidden), attention_mask=am, position_ids=pos,
                        past_key_value=cache, output_attentions=True)
        # cross-check the restatement
        wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
              "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
        o2, p2, k2, v2 = oracle.decode_step(tok, L, wd, k_lat, v_lat, mask)
        print(f"  {tag}: |out-oracle| {(o.float().reshape(-1) - o2.float()).abs().max():.3e} "
              f"|P-oracle| {(p.float().reshape(H, -1) - p2.float()).abs().max():.3e} rms(out) "
              f"{o.float().pow(2).mean().sqrt():.3e}")
        assert torch.equal(cache.k[0], k2) and torch.equal(cache.v[0], v2)
        out[tag + "/attn_output"] = _np(o.reshape(-1))
        out[tag + "/attn_weights"] = _np(p.reshape(H, L + 1))
        out[tag + "/k_new"] = _np(cache.k[0, :, L])
        out[tag + "/v_new"] = _np(cache.v[0, :, L])
        flat = [w["wq"], w["vt_k"], w["vt_v"], w["wo"], *w["u_k"], k_lat, v_lat, tok]
        out[tag + "/digest"] = np.array(gi.digest(*flat))
    _save("g3_decode_step", **out)


def gen_prefill(ml, pa):
    print("G7 prefill (kernel/palu_attention.py:147-263, prompt branch :196-206)")
    out = {}
    for tag, seed, hidden, H, D, gs, rank_k, rank_v, T, causal in gi.PREFILL_CASES:
        w, prompt, mask = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, causal)
        m, cfg = _ref_module_from_palu_weights(ml, pa, hidden, H, D, gs, rank_k, rank_v, w)
        cache = _Cache437()
        am = None if mask is None else mask.reshape(1, 1, T, T)
        with torch.no_grad():
            o, p, _ = m(prompt.reshape(1, T, hidden), attention_mask=am, position_ids=torch.arange(T).unsqueeze(0),
                        past_key_value=cache, output_attentions=True)
        wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
              "u_k": [u.half() for u in w["u_k"]], "wo": w["wo"].half()}
        o2, p2, k2, v2 = oracle.prefill(prompt, wd, mask)
        print(f"  {tag}: |out-oracle| {(o[0].float() - o2.float()).abs().max():.3e} "
              f"|P-oracle| {(p[0].float() - p2.float()).abs().max():.3e} rms(out) {o.float().pow(2).mean().sqrt():.3e}")
        assert torch.equal(cache.k[0], k2) and torch.equal(cache.v[0], v2)
        out[tag + "/attn_output"] = _np(o[0])
        out[tag + "/attn_weights"] = _np(p[0])
        out[tag + "/k_lat"] = _np(cache.k[0])
        out[tag + "/v_lat"] = _np(cache.v[0])
        flat = [w["wq"], w["vt_k"], w["vt_v"], w["wo"], *w["u_k"], prompt]
        out[tag + "/digest"] = np.array(gi.digest(*flat))
    _save("g7_prefill", **out)


def gen_reftest(ml, pa):
    """The scenario of kernel/test_palu_attention.py:158-195 (full rank 4096/4096, prefill 63
    tokens into the cache, decode 1) run through from_attention (per-group SVD)."""
    print("G3b reference-test scenario (test_palu_attention.py:158-195)")
    rng = np.random.default_rng(4242)
    hidden, H, D, gs = 4096, 32, 128, 4
    cfg = ml.LlamaConfig()
    cfg.group_size, cfg.num_groups = gs, H // gs
    cfg.total_rank_k = cfg.total_rank_v = 4096
    attn = ml.LlamaAttention(cfg, 0)
    ws = {}
    with torch.no_grad():
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            wt = torch.from_numpy(rng.uniform(-1 / 64, 1 / 64, (hidden, hidden)).astype(np.float32))
            getattr(attn, name).weight.copy_(wt)
            ws[name] = wt
    prompt = torch.from_numpy(rng.standard_normal((1, 63, hidden)).astype(np.float16))
    tok = torch.from_numpy(rng.standard_normal((1, 1, hidden)).astype(np.float16))
    palu = pa.LlamaPaluAttention.from_attention(attn, cfg).eval().to(torch.float16)
    cache = _Cache437()
    with torch.no_grad():
        po, pp, _ = palu(prompt, output_attentions=True, past_key_value=cache,
                         position_ids=torch.arange(63).unsqueeze(0))
        do, dp, _ = palu(tok, output_attentions=True, past_key_value=cache,
                         position_ids=torch.arange(63, 64).unsqueeze(0))
    # independent fp32 vanilla attention (what the reference test compares against)
    x = torch.cat((prompt, tok), dim=1)[0].float()
    q = (x @ ws["q_proj"].t()).reshape(64, H, D).transpose(0, 1)
    k = (x @ ws["k_proj"].t()).reshape(64, H, D).transpose(0, 1)
    v = (x @ ws["v_proj"].t()).reshape(64, H, D).transpose(0, 1)
    cos, sin = oracle.rope_cos_sin(64, D)
    q, k = oracle.rope_rotate(q, cos, sin), oracle.rope_rotate(k, cos, sin)
    s = (q[:, 63:64] @ k.transpose(1, 2)) / math.sqrt(D)
    p = torch.softmax(s, dim=-1)
    o = ((p @ v).transpose(0, 1).reshape(1, H * D)) @ ws["o_proj"].t()
    dw = (dp[0, :, 0].float() - p[:, 0]).abs().max().item()
    dout = (do.reshape(-1).float() - o.reshape(-1)).abs().max().item()
    print(f"  palu(ref, shimmed) vs fp32 vanilla: |P| {dw:.3e} |out| {dout:.3e}")
    assert dw < 1e-3 and dout < 1e-3
    _save("g3b_reftest", decode_weights=_np(dp[0, :, 0]), decode_output=_np(do.reshape(-1)),
          vanilla_weights=_np(p[:, 0]), vanilla_output=_np(o.reshape(-1)),
          prefill_output_last=_np(po[0, -1]),
          digest=np.array(gi.digest(*[ws[n] for n in ("q_proj", "k_proj", "v_proj", "o_proj")],
                                    prompt, tok)))


# ------------------------------------------------------------- G4 layout / fusion
def gen_layout(ml, pa):
    print("G4 B layout + U_v->W_o fusion (palu_attention.py:79-122, :285-306)")
    rng = np.random.default_rng(99)
    hidden, H, D, gs = 128, 4, 32, 2
    G = H // gs
    cfg = ml.LlamaConfig(hidden_size=hidden, num_attention_heads=H, num_key_value_heads=H)
    cfg.group_size, cfg.num_groups = gs, G
    cfg.total_rank_k, cfg.total_rank_v = 32, 64
    attn = ml.LlamaAttention(cfg, 0)
    with torch.no_grad():
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            getattr(attn, name).weight.copy_(
                torch.from_numpy(rng.uniform(-0.1, 0.1, (H * D, hidden) if name != "o_proj"
                                             else (hidden, H * D)).astype(np.float32)))
    palu = pa.LlamaPaluAttention.from_attention(attn, cfg)
    u_k = [u.weight.data for u in palu.k_proj.U_list]
    u_v = [u.weight.data for u in palu.v_proj.U_list]
    b_mine = oracle.build_b_from_u(u_k, gs, D)
    wo_mine = oracle.fuse_uv_into_wo(attn.o_proj.weight.data, u_v, gs, D)
    print(f"  |B-oracle| {(palu.k_proj.B.data - b_mine).abs().max():.2e}  "
          f"|Wo'-oracle| {(palu.o_proj.weight.data - wo_mine).abs().max():.2e}")
    _save("g4_layout", u_k=np.stack([_np(u) for u in u_k]), u_v=np.stack([_np(u) for u in u_v]),
          b=_np(palu.k_proj.B.data), wo=_np(attn.o_proj.weight.data),
          wo_fused=_np(palu.o_proj.weight.data), gs=np.array(gs), D=np.array(D))


# ----------------------------------------------------------------- G5 quantiser
def gen_quant(rq):
    print("G5 quantize_tensor (palu/model/modules/quant.py:5-41)")
    out = {}
    for R in gi.QUANT_R:
        x = gi.quant_inputs(0, R)
        out[f"R{R}/digest"] = np.array(gi.digest(x))
        for bits in (3, 4):
            for sym in (False, True):
                for gsz in (0, 32):
                    for clip in ((1.0, 0.9) if (gsz == 0 and not sym) else (1.0,)):
                        ref = rq.quantize_tensor(x.clone(), bits, gsz, sym, clip)
                        deq, codes, sc, zp = oracle.quantize_rows(x.clone(), bits, gsz, sym, clip)
                        assert torch.equal(ref, deq), (R, bits, sym, gsz, clip)
                        key = f"R{R}/b{bits}_sym{int(sym)}_g{gsz}_c{clip}"
                        out[key] = _np(ref)
    _save("g5_quant", **out)
    # Quantizer wrapper (quant.py:60-79): 3-D input flattened to rows of the last dim
    q = rq.Quantizer(4, 0, False, 1.0)
    x3 = gi.quant_inputs(1, 64).reshape(2, 12, 64)
    assert torch.equal(q(x3.clone()), oracle.quantize_rows(x3.reshape(-1, 64).clone(), 4)[0].reshape(2, 12, 64))


# ------------------------------------------------------------------ G6 Hadamard
def gen_hadamard(rh, rs):
    print("G6 Hadamard (hadamard_utils.py:85-113,138-147; svd_linear.py:156-168)")
    out = {"had12": _np(rh.get_had12())}
    assert torch.equal(rh.get_had12(), oracle.had12())
    rng = np.random.default_rng(5)
    for n in (32, 64, 128, 256, 512, 192, 384):
        x = torch.from_numpy(rng.standard_normal((5, n)).astype(np.float32))
        y_loop = rh.matmul_hadU(x)                     # in-tree butterfly
        y_app = rh.apply_hadamard(x)                   # CUDA-ext form through the scipy stub
        mine = oracle.apply_hadamard(x)
        print(f"  n={n}: |hadU-apply| {(y_loop - y_app).abs().max():.2e} |apply-oracle| "
              f"{(y_app - mine).abs().max():.2e}")
        out[f"n{n}/x"], out[f"n{n}/hadU"], out[f"n{n}/apply"] = _np(x), _np(y_loop), _np(y_app)
    # fused_hadamard_matrix on a small module
    mod = rs.HeadwiseLowRankModule([32, 32], 64, 2 * 48, bias=False)
    with torch.no_grad():
        mod.VT.weight.copy_(torch.from_numpy(rng.standard_normal((64, 64)).astype(np.float32)))
        for u in mod.U:
            u.weight.copy_(torch.from_numpy(rng.standard_normal((48, 32)).astype(np.float32)))
    vt0 = mod.VT.weight.data.clone()
    u0 = [u.weight.data.clone() for u in mod.U]
    mod.fused_hadamard_matrix()
    vt1, u1 = oracle.fuse_hadamard_into_weights(vt0, u0)
    print(f"  fused: |VT-oracle| {(mod.VT.weight.data - vt1).abs().max():.2e}")
    out["fuse/vt0"], out["fuse/vt1"] = _np(vt0), _np(mod.VT.weight.data)
    out["fuse/u0"] = np.stack([_np(u) for u in u0])
    out["fuse/u1"] = np.stack([_np(u.weight.data) for u in mod.U])
    _save("g6_hadamard", **out)


# ------------------------------------------------------------------ G9 whitened decomposition
def gen_whiten(rs):
    print("G9 from_linear_whiten (palu/model/modules/svd_linear.py:6-34,170-204)")
    rng = np.random.default_rng(31)
    out = {}
    for tag, fin, per_group, ranks, bias in (("a", 64, 48, [16, 24], True), ("b", 96, 32, [32, 8, 20], False)):
        lin = nn.Linear(fin, per_group * len(ranks), bias=bias)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(rng.standard_normal((per_group * len(ranks), fin)).astype(np.float32) * 0.2))
            if bias:
                lin.bias.copy_(torch.from_numpy(rng.standard_normal(per_group * len(ranks)).astype(np.float32)))
        # a calibration-style scaling matrix: lower-triangular (Cholesky-like), well conditioned
        a = rng.standard_normal((fin, fin)).astype(np.float32) * 0.05
        sc = torch.from_numpy(np.linalg.cholesky(a @ a.T + np.eye(fin, dtype=np.float32)).astype(np.float32))
        lin.scaling_diag_matrix = sc
        mod = rs.HeadwiseLowRankModule.from_linear_whiten(lin, ranks)
        us = [u.weight.data.clone() for u in mod.U]
        vt = mod.VT.weight.data.clone()
        mine_u, mine_vt, mine_b = oracle.from_linear_whiten(lin.weight.data, lin.bias.data if bias else None, sc, ranks)
        r0 = 0
        for g, r in enumerate(ranks):
            d = (us[g] @ vt[r0:r0 + r] - mine_u[g] @ mine_vt[r0:r0 + r]).abs().max()
            print(f"  {tag} group {g}: |U VT - oracle| {d:.2e}")
            out[f"{tag}/u{g}"] = _np(us[g])
            if bias:
                out[f"{tag}/bias{g}"] = _np(mod.U[g].bias.data)
            r0 += r
        out[f"{tag}/w"], out[f"{tag}/scaling"], out[f"{tag}/vt"] = _np(lin.weight.data), _np(sc), _np(vt)
        out[f"{tag}/ranks"] = np.array(ranks)
        if bias:
            out[f"{tag}/b"] = _np(lin.bias.data)
        x = torch.from_numpy(rng.standard_normal((1, 5, fin)).astype(np.float32))
        out[f"{tag}/x"], out[f"{tag}/y"] = _np(x), _np(mod(x))          # forward of the decomposed module
    _save("g9_whiten", **out)


def main():
    ml, pa, ar, pr, rq, rh, rs = _import_reference()
    only = set(sys.argv[1:])
    if not only or "abx" in only:
        gen_abx(ar)
    if not only or "rope" in only:
        gen_rope(pr)
    if not only or "layout" in only:
        gen_layout(ml, pa)
    if not only or "steps" in only:
        gen_steps(ml, pa)
    if not only or "quant" in only:
        gen_quant(rq)
    if not only or "hadamard" in only:
        gen_hadamard(rh, rs)
    if not only or "prefill" in only:
        gen_prefill(ml, pa)
    if not only or "reftest" in only:
        gen_reftest(ml, pa)
    if not only or "whiten" in only:
        gen_whiten(rs)


if __name__ == "__main__":
    main()
