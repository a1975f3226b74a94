"""Hadamard rotation helpers with the reference's names (palu/model/modules/hadamard_utils.py:85-90,
138-147; svd_linear.py:156-168), backed by the HIP FWHT (`palu_hadamard_transform`).

`hadamard_transform(x, scale)` is the drop-in for the external `fast_hadamard_transform` op.  Sizes:
n = 2^m, and n = K * 2^m for every K of get_hadK (:5-83: 12 ... 244) through the hadK (x) H_{n/K} Kronecker form of
matmul_hadU_cuda (:138-147) -- BASELINE configs 3/4 hit K = 12 with R_v = 384 / 192, the Fisher rank search
(rank_search.py:11-17) produces widths such as 160, 224, 320 (K = 40, 28, 40).  The K x K factors are numeric data
(`hadk_tables.npz`, bit-packed, written by tests/golden/make_hadk.py from the reference's get_hadK) -- they must be
the reference's own matrices for rotated weights to agree.  The K x K mixing is a small torch matmul: this is offline
weight preparation.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .. import _lib

# get_hadK's test order (hadamard_utils.py:7-80): the first K with n % K == 0 wins (e.g. 160 takes 40, not 20)
HAD_K_ORDER = (244, 180, 172, 156, 140, 108, 92, 84, 76, 68, 60, 52, 44, 36, 28, 40, 20, 12)
_tables = None


def _load_tables():
    global _tables
    if _tables is None:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hadk_tables.npz"))
        _tables = {}
        for K in HAD_K_ORDER:
            bits = np.unpackbits(z[f"had{K}"])[:K * K].reshape(K, K)
            _tables[K] = torch.from_numpy(bits.astype(np.float32) * 2.0 - 1.0)
    return _tables


def get_hadK(n: int, transpose: bool = False):
    """(hadK [K, K] fp32 or None, K) for a width n = K * 2^m, the selection rule of hadamard_utils.py:5-83."""
    for K in HAD_K_ORDER:
        if n % K == 0:
            if not is_pow2(n // K):
                raise ValueError(f"get_hadK: n={n} is {K} * {n // K}, not {K} * 2^m (hadamard_utils.py asserts the same)")
            h = _load_tables()[K]
            return (h.t().contiguous() if transpose else h), K
    if not is_pow2(n):
        raise ValueError(f"get_hadK: n={n} is neither 2^m nor K * 2^m for a known Hadamard factor K")
    return None, 1


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
    hadK, K = get_hadK(n, transpose)
    if K == 1:
        return hadamard_transform(work.contiguous(), 1.0 / math.sqrt(n)).to(dtype)
    # matmul_hadU_cuda (:142-147): Sylvester transform over the inner n/K axis, hadK over the outer axis, one 1/sqrt(n)
    if n == K:
        y = work.reshape(-1, K, 1) * (1.0 / math.sqrt(n))
    else:
        y = hadamard_transform(work.reshape(-1, K, n // K).contiguous(), 1.0 / math.sqrt(n))
    y = torch.matmul(hadK.to(device=y.device, dtype=y.dtype), y)
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
