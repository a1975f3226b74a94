"""HIP-backed latent quantisation with the reference's Python surface.

`quantize_tensor` / `Quantizer` keep the call signatures of palu/model/modules/quant.py:5-41,46-79
(fake-quant: returns the dequantised tensor) but run on the packed-format kernels; the additional
functions expose the real packed cache format (the reference never materialises codes, README.md:24).
Supported: n_bits in {3,4}, asymmetric, whole-row groups (group_size 0) or group_size dividing the
row, clip_ratio 1.0 -- the reference defaults (utils.py:103-108).  Anything else raises
NotImplementedError; n_bits >= 16 is the passthrough of quant.py:61-62.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib


def packed_row_bytes(R: int, n_bits: int) -> int:
    n = _lib.lib.palu_packed_row_bytes(R, n_bits)
    if n == 0:
        raise ValueError(f"unsupported packed row: R={R}, bits={n_bits} (4-bit: R%8==0, 3-bit: R%32==0)")
    return n


def _as_gl(x: torch.Tensor):
    """view [..., R] as [G=1? , rows, R] with uniform strides"""
    R = x.shape[-1]
    x2 = x.reshape(-1, R)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    return x2


def quantize_pack(x: torch.Tensor, n_bits: int, want_dequant: bool = False, sym: bool = False, clip_ratio: float = 1.0):
    """x [..., R] fp16 (cuda) -> (codes uint8 [..., R*b/8], meta fp16 [..., 2] = (scale, zero)[, dequant]).
    sym / clip_ratio: the other modes of quantize_tensor (quant.py:18-36); symmetric rows are stored in offset binary
    (zero = 2^(b-1)), so unpack_dequant and the decode kernels need no flag."""
    if not x.is_cuda or x.dtype != torch.float16:
        raise TypeError("quantize_pack: fp16 ROCm tensor required (no CPU fallback)")
    R = x.shape[-1]
    nb = packed_row_bytes(R, n_bits)
    x2 = _as_gl(x)
    n = x2.shape[0]
    codes = torch.empty((n, nb), dtype=torch.uint8, device=x.device)
    meta = torch.empty((n, 2), dtype=torch.float16, device=x.device)
    deq = torch.empty((n, R), dtype=torch.float16, device=x.device) if want_dequant else None
    _lib.check(_lib.lib.palu_quantize_pack_ex(x2.data_ptr(), 0, x2.stride(0), codes.data_ptr(), 0, nb, meta.data_ptr(), 0, 2,
                                              0 if deq is None else deq.data_ptr(), 0, R, 1, n, R, n_bits,
                                              1 if sym else 0, float(clip_ratio), _lib.current_stream()),
               "palu_quantize_pack")
    lead = x.shape[:-1]
    out = (codes.reshape(*lead, nb), meta.reshape(*lead, 2))
    return out + ((deq.reshape(x.shape),) if want_dequant else ())


def unpack_dequant(codes: torch.Tensor, meta: torch.Tensor, n_bits: int, R: int) -> torch.Tensor:
    nb = packed_row_bytes(R, n_bits)
    c2 = codes.reshape(-1, nb).contiguous()
    m2 = meta.reshape(-1, 2).contiguous()
    n = c2.shape[0]
    out = torch.empty((n, R), dtype=torch.float16, device=codes.device)
    _lib.check(_lib.lib.palu_unpack_dequant(c2.data_ptr(), 0, nb, m2.data_ptr(), 0, 2, out.data_ptr(), 0, R, 1, n, R,
                                            n_bits, _lib.current_stream()), "palu_unpack_dequant")
    return out.reshape(*codes.shape[:-1], R)


def pack_codes(codes_u8: torch.Tensor, n_bits: int) -> torch.Tensor:
    R = codes_u8.shape[-1]
    nb = packed_row_bytes(R, n_bits)
    c = codes_u8.contiguous()
    out = torch.empty((*c.shape[:-1], nb), dtype=torch.uint8, device=c.device)
    _lib.check(_lib.lib.palu_pack_codes(c.data_ptr(), out.data_ptr(), c.numel(), n_bits, _lib.current_stream()), "palu_pack_codes")
    return out


def unpack_codes(packed: torch.Tensor, n_bits: int, R: int) -> torch.Tensor:
    nb = packed_row_bytes(R, n_bits)
    assert packed.shape[-1] == nb
    p = packed.contiguous()
    out = torch.empty((*p.shape[:-1], R), dtype=torch.uint8, device=p.device)
    _lib.check(_lib.lib.palu_unpack_codes(p.data_ptr(), out.data_ptr(), out.numel(), n_bits, _lib.current_stream()),
               "palu_unpack_codes")
    return out


@torch.no_grad()
def quantize_tensor(w: torch.Tensor, n_bits, group_size, sym, clip_ratio=1.0) -> torch.Tensor:
    """Fake-quant with the reference's signature (quant.py:5): returns (code - zero) * scale, same shape."""
    assert w.dim() == 2
    assert n_bits < 16
    if n_bits not in (3, 4):
        raise NotImplementedError("HIP latent quantiser: 3- and 4-bit codes (the packed cache formats)")
    if not 0.0 < clip_ratio <= 1.0:
        raise ValueError("clip_ratio must be in (0, 1]")
    shape = w.shape
    if group_size > 0:
        assert shape[-1] % group_size == 0
        w = w.reshape(-1, group_size)
    *_, deq = quantize_pack(w.half() if w.dtype != torch.float16 else w, n_bits, want_dequant=True, sym=sym,
                            clip_ratio=clip_ratio)
    return deq.reshape(shape).to(w.dtype)


class Quantizer(nn.Module):
    """quant.py:46-83: passthrough for n_bits >= 16, else row-wise fake-quant over the last dim."""

    def __init__(self, n_bits: int, group_size: int, sym: bool, clip_ratio: float) -> None:
        super().__init__()
        self.n_bits, self.group_size, self.sym, self.clip_ratio = n_bits, group_size, sym, clip_ratio

    @torch.no_grad()
    def forward(self, x):
        if self.n_bits >= 16:
            return x
        shape = x.shape
        assert self.group_size == 0 or shape[-1] % self.group_size == 0, "Group size should be divisible by (dim)."
        y = quantize_tensor(x.reshape(-1, shape[-1]), self.n_bits, self.group_size, self.sym, self.clip_ratio)
        return y.view(shape)
