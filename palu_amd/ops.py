"""PyTorch-ROCm custom ops (`torch.ops.palu.*`) over the C ABI, so that the kernels are reachable through the
dispatcher (torch.compile / export see opaque ops with fake kernels).  Importing this module registers:

    palu::abx(a, b, x, theta=1e4, pos_offset=0) -> Tensor            kernel/abx_rope.py:114-150
    palu::softmax_pv(scores, v, mask=None, sqrt_d) -> Tensor          kernel/palu_attention.py:219-251
    palu::gemv(w, x) -> Tensor                                        nn.Linear at batch 1 (:257)
    palu::quantize_pack(x, bits) -> (codes, meta)                     palu/model/modules/quant.py:5-41
    palu::unpack_dequant(codes, meta, bits, rank) -> Tensor
    palu::hadamard_transform(x, scale) -> Tensor                      fast_hadamard_transform.hadamard_transform
    palu::rope_(x, pos0, theta=1e4) -> ()   (in place)                 rotary_emb + apply_rotary_pos_emb (:204-205)
    palu::prefill_attn(q, k, v_lat, past, causal, scale) -> Tensor    prompt branch :205-255, flash-style
    palu::decode_step(hidden, wq, vt_k, vt_v, bfrag, wo, k_cache!, v_cache!, inv_freq, workspace!, ...) -> Tensor
                                                                      the decode branch :207-257 in one node (! = mutated)
    palu::decode_step_q(... k_codes!, k_meta!, v_codes!, v_meta! ...) -> Tensor   the same on a packed 3/4-bit cache
    palu::decode_attn(q, bfrag, k, v, inv_freq, workspace!, H, L, pos0) -> Tensor  single-kernel scores+softmax+P.V
    palu::lowrank_project_gemm(x, w, cache!, row0) -> ()              prefill down-projection into the cache rows
    palu::pack_codes(codes, bits) / palu::unpack_codes(packed, bits, rank)

All run on the current stream and allocate only through torch's caching allocator (graph-capturable).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from .kernel import abx_rope as _abx
from .kernel import hadamard_utils as _had
from .kernel import quant as _quant


@torch.library.custom_op("palu::abx", mutates_args=())
def abx(a: torch.Tensor, b: torch.Tensor, x: torch.Tensor, theta: float = 10000.0, pos_offset: int = 0) -> torch.Tensor:
    return _abx.abx(a, b, x, theta=theta, pos_offset=pos_offset)


@abx.register_fake
def _(a, b, x, theta=10000.0, pos_offset=0):
    return x.new_empty((a.shape[0], 1, x.shape[1]))


@torch.library.custom_op("palu::softmax_pv", mutates_args=())
def softmax_pv(scores: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor] = None,
               sqrt_d: float = math.sqrt(128.0)) -> torch.Tensor:
    H, L = scores.shape
    G, Lv, Rv = v.shape
    assert Lv == L and scores.dtype == v.dtype == torch.float16 and scores.is_cuda
    if scores.stride(1) != 1:
        scores = scores.contiguous()
    if v.stride(2) != 1 or v.stride(1) % 8 or v.stride(0) % 8:
        v = v.contiguous()
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=v.device)
    ctx = torch.empty((H, Rv), dtype=torch.float16, device=v.device)
    m = None if mask is None else mask.reshape(-1).to(torch.float16).contiguous()
    _lib.check(_lib.lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0 if m is None else m.data_ptr(),
                                            v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(), 0, 0, ws.data_ptr(),
                                            H, G, L, Rv, float(sqrt_d), _lib.current_stream()), "palu_softmax_pv_f16")
    return ctx


@softmax_pv.register_fake
def _(scores, v, mask=None, sqrt_d=math.sqrt(128.0)):
    return v.new_empty((scores.shape[0], v.shape[2]))


@torch.library.custom_op("palu::gemv", mutates_args=())
def gemv(w: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    N, K = w.shape
    assert x.numel() == K and w.dtype == x.dtype == torch.float16 and w.is_cuda and w.stride(1) == 1
    xc = x.reshape(-1).contiguous()
    y = torch.empty(N, dtype=torch.float16, device=w.device)
    _lib.check(_lib.lib.palu_gemv_f16(w.data_ptr(), w.stride(0), xc.data_ptr(), y.data_ptr(), N, K, _lib.current_stream()),
               "palu_gemv_f16")
    return y


@gemv.register_fake
def _(w, x):
    return x.new_empty((w.shape[0],))


@torch.library.custom_op("palu::quantize_pack", mutates_args=())
def quantize_pack(x: torch.Tensor, bits: int) -> Tuple[torch.Tensor, torch.Tensor]:
    codes, meta = _quant.quantize_pack(x, bits)
    return codes, meta


@quantize_pack.register_fake
def _(x, bits):
    nb = x.shape[-1] * bits // 8
    return (x.new_empty((*x.shape[:-1], nb), dtype=torch.uint8), x.new_empty((*x.shape[:-1], 2)))


@torch.library.custom_op("palu::unpack_dequant", mutates_args=())
def unpack_dequant(codes: torch.Tensor, meta: torch.Tensor, bits: int, rank: int) -> torch.Tensor:
    return _quant.unpack_dequant(codes, meta, bits, rank)


@unpack_dequant.register_fake
def _(codes, meta, bits, rank):
    return meta.new_empty((*codes.shape[:-1], rank))


@torch.library.custom_op("palu::hadamard_transform", mutates_args=())
def hadamard_transform(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    return _had.hadamard_transform(x, scale)


@hadamard_transform.register_fake
def _(x, scale=1.0):
    return torch.empty_like(x)


@torch.library.custom_op("palu::rope_", mutates_args=("x",))
def rope_(x: torch.Tensor, pos0: int, theta: float = 10000.0) -> None:
    """In-place rotary embedding of x [H, T, 128] fp16 (D contiguous, any H/T strides), row t at position pos0 + t."""
    assert x.dim() == 3 and x.dtype == torch.float16 and x.is_cuda and x.stride(2) == 1
    inv = _abx.rope_inv_freq(x.device, x.shape[2], theta)
    _lib.check(_lib.lib.palu_rope_f16(x.data_ptr(), x.stride(0), x.stride(1), x.shape[0], x.shape[1], x.shape[2], pos0,
                                      inv.data_ptr(), _lib.current_stream()), "palu_rope_f16")


@rope_.register_fake
def _(x, pos0, theta=10000.0):
    return None


@torch.library.custom_op("palu::prefill_attn", mutates_args=())
def prefill_attn(q: torch.Tensor, k: torch.Tensor, v_lat: torch.Tensor, past: int = 0, causal: bool = True,
                 scale: float = 1.0 / math.sqrt(128.0)) -> torch.Tensor:
    """q [H,Tq,D], k [H,Tk,D] (both RoPE'd), v_lat [G,Tk,Rv] fp16 -> [Tq, H*Rv]: softmax(q.k^T*scale [causal]) . V_lat
    per head, the head's group supplying V (kernel/palu_attention.py:205-255 without the [Tq,Tk] matrix)."""
    H, Tq, D = q.shape
    Tk = k.shape[1]
    G, _, Rv = v_lat.shape
    assert q.dtype == k.dtype == v_lat.dtype == torch.float16 and q.is_cuda and k.shape[0] == H and v_lat.shape[1] == Tk
    if q.stride(2) != 1 or q.stride(0) % 8 or q.stride(1) % 8:
        q = q.contiguous()
    if k.stride(2) != 1 or k.stride(0) % 8 or k.stride(1) % 8:
        k = k.contiguous()
    pad = (Tk + 63) // 64 * 64
    vt = torch.zeros((G, Rv, pad), dtype=torch.float16, device=q.device)
    vt[:, :, :Tk].copy_(v_lat.transpose(1, 2))
    out = torch.empty((Tq, H * Rv), dtype=torch.float16, device=q.device)
    _lib.check(_lib.lib.palu_prefill_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                              vt.data_ptr(), vt.stride(0), vt.stride(1), out.data_ptr(), out.stride(0),
                                              H, G, D, Tq, Tk, Rv, int(past), 1 if causal else 0, float(scale),
                                              _lib.current_stream()), "palu_prefill_attn_f16")
    return out


@prefill_attn.register_fake
def _(q, k, v_lat, past=0, causal=True, scale=1.0 / math.sqrt(128.0)):
    return q.new_empty((q.shape[1], q.shape[0] * v_lat.shape[2]))


# ------------------------------------------------------------------------------------------------------------
# The decode step itself as dispatcher ops (SURVEY.md 8(b) "new ops"): the caches and the workspace are mutated in
# place (mutates_args), so torch.compile / export see ONE opaque node per token and functionalisation keeps the
# in-place cache append.  LlamaPaluAttention._decode_fused / _decode_fused_q call these.
@torch.library.custom_op("palu::decode_step", mutates_args=("k_cache", "v_cache", "workspace"))
def decode_step(hidden: torch.Tensor, wq: torch.Tensor, vt_k: torch.Tensor, vt_v: torch.Tensor, bfrag: torch.Tensor,
                wo: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, inv_freq: torch.Tensor,
                workspace: torch.Tensor, ws_capacity: int, num_heads: int, cache_len: int, pos: int,
                mask: Optional[torch.Tensor] = None, shared_b: bool = False) -> torch.Tensor:
    """One decode token on an fp16 latent cache (palu_decode_step_f16; kernel/palu_attention.py:207-257).
    hidden [hidden_size]; k_cache [G, cap, Rk] / v_cache [G, cap, Rv] hold cache_len rows and receive row cache_len;
    bfrag from palu_abx_prepare_b; workspace of palu_decode_workspace_bytes(H, G, D, ws_capacity, Rv) bytes.
    shared_b: bfrag holds the fragments of the [G, R, D] factor all heads of a group share (palu_decode_step_sharedb_f16)."""
    G, cap, Rk = k_cache.shape
    Rv = v_cache.shape[2]
    hidden_size = wq.shape[1]
    D = wq.shape[0] // num_heads
    x = hidden.reshape(-1).contiguous()
    out = torch.empty(hidden_size, dtype=torch.float16, device=hidden.device)
    m = None if mask is None else mask.reshape(-1).to(torch.float16).contiguous()
    fn = _lib.lib.palu_decode_step_sharedb_f16 if shared_b else _lib.lib.palu_decode_step_f16
    with _lib.on_device(hidden):
        _lib.check(fn(
            x.data_ptr(), wq.data_ptr(), wq.stride(0), vt_k.data_ptr(), vt_k.stride(0), vt_v.data_ptr(), vt_v.stride(0),
            bfrag.data_ptr(), wo.data_ptr(), wo.stride(0),
            k_cache.data_ptr(), k_cache.stride(0), k_cache.stride(1), v_cache.data_ptr(), v_cache.stride(0), v_cache.stride(1),
            0 if m is None else m.data_ptr(), inv_freq.data_ptr(), out.data_ptr(), 0, 0,
            workspace.data_ptr(), int(ws_capacity), num_heads, G, D, hidden_size, Rk, Rv, int(cache_len), int(pos),
            _lib.current_stream()), "palu_decode_step_f16")
    return out


@decode_step.register_fake
def _(hidden, wq, vt_k, vt_v, bfrag, wo, k_cache, v_cache, inv_freq, workspace, ws_capacity, num_heads, cache_len, pos,
      mask=None, shared_b=False):
    return hidden.new_empty((wq.shape[1],))


@torch.library.custom_op("palu::decode_step_q", mutates_args=("k_codes", "k_meta", "v_codes", "v_meta", "workspace"))
def decode_step_q(hidden: torch.Tensor, wq: torch.Tensor, vt_k: torch.Tensor, vt_v: torch.Tensor, bfrag: torch.Tensor,
                  wo: torch.Tensor, k_codes: torch.Tensor, k_meta: torch.Tensor, v_codes: torch.Tensor,
                  v_meta: torch.Tensor, inv_freq: torch.Tensor, workspace: torch.Tensor, ws_capacity: int,
                  num_heads: int, rank_k: int, rank_v: int, bits: int, cache_len: int, pos: int,
                  mask: Optional[torch.Tensor] = None, group_size: int = 0) -> torch.Tensor:
    """One decode token on a packed 3/4-bit latent cache (palu_decode_step_qg): codes [G, cap, R*bits/8] uint8,
    meta [G, cap, 2] fp16 = (scale, zero) -- or [G, cap, 2 R / group_size] with `group_size` columns per pair
    (quant.py:11-13); the new latent rows are quantised + packed into row cache_len."""
    G = k_codes.shape[0]
    hidden_size = wq.shape[1]
    D = wq.shape[0] // num_heads
    x = hidden.reshape(-1).contiguous()
    out = torch.empty(hidden_size, dtype=torch.float16, device=hidden.device)
    m = None if mask is None else mask.reshape(-1).to(torch.float16).contiguous()
    with _lib.on_device(hidden):
        _lib.check(_lib.lib.palu_decode_step_qg(
            x.data_ptr(), wq.data_ptr(), wq.stride(0), vt_k.data_ptr(), vt_k.stride(0), vt_v.data_ptr(), vt_v.stride(0),
            bfrag.data_ptr(), wo.data_ptr(), wo.stride(0),
            k_codes.data_ptr(), k_codes.stride(0), k_codes.stride(1), k_meta.data_ptr(), k_meta.stride(0), k_meta.stride(1),
            v_codes.data_ptr(), v_codes.stride(0), v_codes.stride(1), v_meta.data_ptr(), v_meta.stride(0), v_meta.stride(1),
            0 if m is None else m.data_ptr(), inv_freq.data_ptr(), out.data_ptr(), 0, 0,
            workspace.data_ptr(), int(ws_capacity), num_heads, G, D, hidden_size, int(rank_k), int(rank_v), int(bits),
            int(group_size), int(cache_len), int(pos), _lib.current_stream()), "palu_decode_step_qg")
    return out


@decode_step_q.register_fake
def _(hidden, wq, vt_k, vt_v, bfrag, wo, k_codes, k_meta, v_codes, v_meta, inv_freq, workspace, ws_capacity, num_heads,
      rank_k, rank_v, bits, cache_len, pos, mask=None, group_size=0):
    return hidden.new_empty((wq.shape[1],))


@torch.library.custom_op("palu::lowrank_project_gemm", mutates_args=("cache",))
def lowrank_project_gemm(x: torch.Tensor, w: torch.Tensor, cache: torch.Tensor, row0: int) -> None:
    """Prefill down-projection (MFMA GEMM) written into the latent-cache layout: cache[g, row0 + m, :] = x[m] . w[g*R:(g+1)*R]^T
    for x [M, K], w = VT [G*R, K], cache [G, cap, R] (HeadwiseLowRankModule.project_to_latent, :59-65, :167-168)."""
    M, K = x.shape
    N = w.shape[0]
    G, cap, R = cache.shape
    assert N == G * R and row0 + M <= cap and x.dtype == w.dtype == cache.dtype == torch.float16
    if x.stride(1) != 1 or x.stride(0) % 8:
        x = x.contiguous()
    with _lib.on_device(x):
        _lib.check(_lib.lib.palu_lowrank_project_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), cache.data_ptr(),
                                                      cache.stride(0), cache.stride(1), M, N, K, R, int(row0),
                                                      _lib.current_stream()), "palu_lowrank_project_gemm")


@lowrank_project_gemm.register_fake
def _(x, w, cache, row0):
    return None


@torch.library.custom_op("palu::pack_codes", mutates_args=())
def pack_codes(codes_u8: torch.Tensor, bits: int) -> torch.Tensor:
    return _quant.pack_codes(codes_u8, bits)


@pack_codes.register_fake
def _(codes_u8, bits):
    return codes_u8.new_empty((*codes_u8.shape[:-1], codes_u8.shape[-1] * bits // 8))


@torch.library.custom_op("palu::unpack_codes", mutates_args=())
def unpack_codes(packed: torch.Tensor, bits: int, rank: int) -> torch.Tensor:
    return _quant.unpack_codes(packed, bits, rank)


@unpack_codes.register_fake
def _(packed, bits, rank):
    return packed.new_empty((*packed.shape[:-1], rank))


@torch.library.custom_op("palu::decode_attn", mutates_args=("workspace",))
def decode_attn(q: torch.Tensor, bfrag: torch.Tensor, k: torch.Tensor, v: torch.Tensor, inv_freq: torch.Tensor,
                workspace: torch.Tensor, num_heads: int, length: int, pos0: int = 0,
                mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Single-kernel attention core (palu_decode_attn_mask_f16): scores -> /sqrt(D) (+ additive `mask` [length] fp16) ->
    softmax -> latent P.V over the first `length` rows of k [G, cap, Rk] / v [G, cap, Rv]; q [H, D] rotated query ->
    ctx [H, Rv]."""
    G, _, Rk = k.shape
    Rv = v.shape[2]
    D = q.shape[-1]
    q2 = q.reshape(num_heads, D)
    ctx = torch.empty((num_heads, Rv), dtype=torch.float16, device=q.device)
    if mask is not None:
        mask = mask.reshape(-1).to(torch.float16).contiguous()
        if mask.numel() != int(length):
            raise ValueError("decode_attn: mask must have `length` elements")
    with _lib.on_device(q):
        _lib.check(_lib.lib.palu_decode_attn_mask_f16(q2.data_ptr(), q2.stride(0), q2.stride(1), bfrag.data_ptr(),
                                                      k.data_ptr(), k.stride(0), k.stride(1), v.data_ptr(), v.stride(0),
                                                      v.stride(1), 0 if mask is None else mask.data_ptr(), ctx.data_ptr(),
                                                      workspace.data_ptr(), num_heads, G, int(length), Rk, Rv, D,
                                                      inv_freq.data_ptr(), int(pos0), math.sqrt(D), _lib.current_stream()),
                   "palu_decode_attn_mask_f16")
    return ctx


@decode_attn.register_fake
def _(q, bfrag, k, v, inv_freq, workspace, num_heads, length, pos0=0, mask=None):
    return q.new_empty((num_heads, v.shape[2]))
