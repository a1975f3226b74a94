"""Hadamard rotation helpers with the reference's names (palu/model/modules/hadamard_utils.py:85-90,
138-147; svd_linear.py:156-168), backed by the HIP FWHT (`palu_hadamard_transform`).

`hadamard_transform(x, scale)` is the drop-in for the external `fast_hadamard_transform` op.  Sizes:
n = 2^m, and n = 12 * 2^m through the had12 (x) H_{n/12} Kronecker form (the n % 12 branch of get_hadK,
:75-78, which BASELINE configs 3/4 hit with R_v = 384 / 192).  The other 19 literal tables are out of
scope.  The 12x12 mixing is a tiny torch matmul: this is offline weight preparation.
"""
from __future__ import annotations

import math

import torch

from .. import _lib


def is_pow2(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


def hadamard_transform(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """(x @ H_n) * scale over the last dim, Sylvester order, n a power of two; fp16 or fp32 on a ROCm device."""
    if not x.is_cuda:
        raise RuntimeError("hadamard_transform: ROCm tensor required (no CPU fallback)")
    n = x.shape[-1]
    if not is_pow2(n):
        raise ValueError(f"hadamard_transform: last dim must be a power of two, got {n}")
    if x.dtype == torch.float16:
        dt = 0
    elif x.dtype == torch.float32:
        dt = 1
    else:
        raise TypeError("hadamard_transform: fp16 or fp32")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    _lib.check(_lib.lib.palu_hadamard_transform(xc.data_ptr(), y.data_ptr(), xc.numel() // n, n, float(scale), dt,
                                                _lib.current_stream()), "palu_hadamard_transform")
    return y


def get_had12(device=None) -> torch.Tensor:
    """12x12 Hadamard matrix of hadamard_utils.py:196-211 as a bordered circulant (same entries)."""
    c = torch.tensor([1, -1, 1, -1, -1, -1, 1, 1, 1, -1, 1], dtype=torch.float32)
    m = torch.ones(12, 12, dtype=torch.float32)
    m[0, 1:] = -1.0
    for r in range(11):
        m[r + 1, 1:] = torch.roll(c, r)
    return m if device is None else m.to(device)


def apply_hadamard(x: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """x -> x . Had_n / sqrt(n) over the last dim (hadamard_utils.py:85-90 + :138-147)."""
    dtype = x.dtype
    n = x.shape[-1]
    work = x.float() if dtype not in (torch.float16, torch.float32) else x
    if is_pow2(n):
        return hadamard_transform(work.contiguous(), 1.0 / math.sqrt(n)).to(dtype)
    if n % 12 != 0 or not is_pow2(n // 12):
        raise NotImplementedError(f"apply_hadamard: n={n} needs a Hadamard table that is out of scope (2^m, 12*2^m only)")
    h12 = get_had12(x.device)
    if transpose:
        h12 = h12.t().contiguous()
    y = hadamard_transform(work.reshape(-1, 12, n // 12).contiguous(), 1.0 / math.sqrt(n))
    y = torch.matmul(h12.to(y.dtype), y)
    return y.reshape(x.shape).to(dtype)


def fuse_hadamard_into_weights(vt_weight: torch.Tensor, u_weights):
    """svd_linear.py:156-168 (`fused_hadamard_matrix`): VT_g <- (had(VT_g^T))^T, U_g <- had(U_g), in place
    on the given tensors.  U'.VT' == U.VT up to rounding, latents are born rotated."""
    r0 = 0
    for u in u_weights:
        R = u.shape[1]
        vt_weight[r0:r0 + R] = apply_hadamard(vt_weight[r0:r0 + R].t().contiguous()).t()
        u.copy_(apply_hadamard(u.contiguous()))
        r0 += R
    return vt_weight, u_weights
