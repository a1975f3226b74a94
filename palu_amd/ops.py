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
