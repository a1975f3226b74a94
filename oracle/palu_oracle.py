"""CPU restatement of the reference's low-rank-KV decode path (test infrastructure).

Every function cites the reference file:line (relative to the reference repo root)
whose arithmetic it follows.  torch-CPU is used for the fp16 pieces because the
reference itself is torch code and its fp16 rounding points are part of the spec;
numpy is used for the integer pack/unpack layout (which the reference does not
define -- SURVEY.md F2 -- so the layout documented in DESIGN.md is the spec).

Do not import this from the product package (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

__all__ = [
    "rope_inv_freq", "rope_cos_sin", "rope_rotate", "abx_scores", "abx_scores_f64",
    "build_b_from_u", "fuse_uv_into_wo", "decode_step", "prefill", "quantize_rows",
    "pack_codes", "unpack_codes", "packed_row_bytes", "dequant_codes",
    "fwht", "had12", "hadK_for", "HAD_K_ORDER", "apply_hadamard", "fuse_hadamard_into_weights",
    "whiten_decompose", "from_linear_whiten",
]


# --------------------------------------------------------------------------- RoPE
def rope_inv_freq(dim: int = 128, theta: float = 10000.0) -> torch.Tensor:
    """fp32 inverse frequencies 1/theta^(2i/dim), i<dim/2.

    Follows kernel/pytorch_reference.py:4 (int64 arange -> float -> /dim -> pow -> reciprocal).
    """
    expo = torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float32) / dim
    return 1.0 / (theta ** expo)


def rope_cos_sin(end: int, dim: int = 128, theta: float = 10000.0,
                 start: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin tables [end-start, dim] in fp32 for positions start..end-1.

    Follows kernel/pytorch_reference.py:3-9: the angle is the *fp32-rounded* product
    fl32(t) * inv_freq (torch.outer in fp32), duplicated over both halves of the head.
    """
    pos = torch.arange(start, end, dtype=torch.int64).to(torch.float32)
    ang = torch.outer(pos, rope_inv_freq(dim, theta))
    ang = torch.cat((ang, ang), dim=-1)
    return ang.cos(), ang.sin()


def rope_rotate(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Half-split rotary embedding x*cos + rot(x)*sin, rot(x) = (-x[d/2:], x[:d/2]).

    Follows kernel/pytorch_reference.py:11-21 (cos/sin broadcast over the leading dim;
    fp32 tables promote an fp16 ``x`` to fp32).
    """
    half = x.shape[-1] // 2
    rot = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return x * cos.unsqueeze(0) + rot * sin.unsqueeze(0)


# ------------------------------------------------------------------ abx (scores)
def abx_scores(a: torch.Tensor, b: torch.Tensor, x: torch.Tensor,
               theta: float = 10000.0) -> torch.Tensor:
    """Oracle for the fused reconstruct-K -> RoPE -> q.K^T kernel.

    a: [H,1,D] fp16 (already-rotated query), b: [H,R,D] fp16, x: [G,L,R] fp16
    -> [H,1,L] fp16 (no 1/sqrt(D) scaling; key position = row index of x).

    Follows kernel/abx_rope.py:152-171 (torch_abx) including its rounding points:
    K = x@b in fp16 tensors, RoPE with fp32 tables, cast to fp16, then a@K^T in fp16.
    """
    H, R, D = b.shape
    G, L, _ = x.shape
    gs = H // G
    keys = torch.matmul(x[:, None, :, :], b.reshape(G, gs, R, D))       # [G,gs,L,D] fp16
    keys = keys.reshape(H, L, D)
    cos, sin = rope_cos_sin(L, D, theta)
    keys = rope_rotate(keys, cos, sin).to(torch.float16)                 # fp32 math, one rounding
    return torch.matmul(a, keys.transpose(-1, -2))


def abx_scores_f64(a, b, x, theta: float = 10000.0) -> torch.Tensor:
    """Same math as :func:`abx_scores` with every product/sum in fp64 (no intermediate
    rounding).  The rotation angle is still the oracle's fp32-rounded ``fl32(l*inv_freq)``
    (pytorch_reference.py:5-6) so that this is the exact value the oracle approximates."""
    H, R, D = b.shape
    G, L, _ = x.shape
    gs = H // G
    keys = torch.matmul(x.double()[:, None], b.double().reshape(G, gs, R, D)).reshape(H, L, D)
    pos = torch.arange(L, dtype=torch.int64).to(torch.float32)
    ang = torch.outer(pos, rope_inv_freq(D, theta)).double()
    ang = torch.cat((ang, ang), dim=-1)
    keys = rope_rotate(keys, ang.cos(), ang.sin())
    return torch.matmul(a.double(), keys.transpose(-1, -2))


# ------------------------------------------------------------- weight preparation
def build_b_from_u(u_weights, group_size: int, head_dim: int) -> torch.Tensor:
    """B[h] = U_{h//gs}.weight[(h%gs)*D:(h%gs+1)*D, :]^T  -> [H,R,D].

    Follows kernel/palu_attention.py:108-114 (stack U^T, split the gs*D axis, swap).
    ``u_weights``: list of G tensors [gs*D, R].
    """
    out = []
    for u in u_weights:
        R = u.shape[1]
        out.append(u.t().reshape(R, group_size, head_dim).permute(1, 0, 2))   # [gs,R,D]
    return torch.cat(out, dim=0).contiguous()


def fuse_uv_into_wo(wo: torch.Tensor, uv_weights, group_size: int, head_dim: int) -> torch.Tensor:
    """W_o'[:, h*Rv:(h+1)*Rv] = W_o[:, h*D:(h+1)*D] @ U_v[h//gs][(h%gs)*D:(h%gs+1)*D, :].

    Follows kernel/palu_attention.py:285-306.  wo: [hidden, H*D]; returns [hidden, H*Rv].
    """
    cols = []
    h = 0
    for u in uv_weights:
        for j in range(group_size):
            cols.append(wo[:, h * head_dim:(h + 1) * head_dim] @ u[j * head_dim:(j + 1) * head_dim, :])
            h += 1
    return torch.cat(cols, dim=1)


def whiten_decompose(weight: torch.Tensor, scaling: torch.Tensor, rank: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rank-`rank` factors (L [out, rank], R [rank, in]) of one group's weight with the activation-whitening matrix S.

    Follows palu/model/modules/svd_linear.py:6-34: SVD of W S in fp32, V = Vt S^-1, truncate, sqrt(Sigma) on BOTH factors,
    cast back to the weight's dtype."""
    dt = weight.dtype
    s32 = scaling.to(torch.float32)
    inv = torch.linalg.inv(s32)
    u, sig, vt = torch.linalg.svd(torch.matmul(weight.to(torch.float32), s32), full_matrices=False)
    v = torch.matmul(vt, inv)
    root = torch.sqrt(torch.diag(sig[:rank]))
    return torch.matmul(u[:, :rank], root).to(dt), torch.matmul(root, v[:rank, :]).to(dt)


def from_linear_whiten(weight: torch.Tensor, bias: Optional[torch.Tensor], scaling: torch.Tensor, ranks):
    """Group-wise whitened decomposition of a dense projection: (u_list [out/G, r_g], vt [sum r_g, in], bias_list or None).

    Follows palu/model/modules/svd_linear.py:170-204: the weight rows are split into len(ranks) groups, each decomposed by
    whiten_decompose; the bias (if any) is split the same way and stays on the U side."""
    G = len(ranks)
    w = weight.reshape(G, -1, weight.shape[1])
    us, rs = [], []
    for g, r in enumerate(ranks):
        left, right = whiten_decompose(w[g], scaling, r)
        us.append(left.contiguous())
        rs.append(right)
    b = None if bias is None else [x for x in bias.reshape(G, -1)]
    return us, torch.cat(rs, dim=0).contiguous(), b


# ------------------------------------------------------------------- prefill (q_len > 1)
def prefill(hidden: torch.Tensor, weights: Dict[str, torch.Tensor], attention_mask: Optional[torch.Tensor] = None,
            theta: float = 10000.0, latent_bits: int = 16):
    """Prompt pass of the low-rank attention module (batch 1, empty cache), fp16 tensors on CPU.

    hidden [T, hidden] fp16; weights: wq [H*D,hidden], vt_k [G*Rk,hidden], vt_v [G*Rv,hidden], u_k: list of G
    tensors [gs*D, Rk], wo [hidden, H*Rv] (U_v already folded in); attention_mask: additive [T, T] or None.
    Returns (attn_output [T, hidden] fp16, attn_weights [H, T, T] fp16, k_lat [G,T,Rk], v_lat [G,T,Rv]).

    Follows kernel/palu_attention.py:147-263, prompt branch :196-206:
      projections :164-174; key reconstruction through U (HeadwiseLowRankModule.reconstruct :67-77) :199-201;
      HF-4.37.2 RoPE on q and k in the activation dtype :204-205; q.k^T / sqrt(D) :206; mask :229-234;
      softmax fp32 -> fp16 :238; latent P.V with the [G, gs*q, kv] reshape :246-251; o_proj :257.
    """
    wq, vt_k, vt_v, u_k, wo = (weights[k] for k in ("wq", "vt_k", "vt_v", "u_k", "wo"))
    T = hidden.shape[0]
    G = len(u_k)
    Rk = u_k[0].shape[1]
    D = 128 if "head_dim" not in weights else int(weights["head_dim"])
    gs = u_k[0].shape[0] // D
    H = G * gs
    Rv = vt_v.shape[0] // G
    lin = torch.nn.functional.linear
    q = lin(hidden, wq).reshape(T, H, D).transpose(0, 1)                       # [H,T,D]
    k_lat = lin(hidden, vt_k).reshape(T, G, Rk).transpose(0, 1).contiguous()   # [G,T,Rk]
    v_lat = lin(hidden, vt_v).reshape(T, G, Rv).transpose(0, 1).contiguous()   # [G,T,Rv]
    if latent_bits < 16:
        # accuracy-path semantics: fake-quantise each (token, group) latent row before it is used
        # (palu/model/modules/svd_linear.py:84-90,124-139)
        k_lat = quantize_rows(k_lat.reshape(G * T, Rk), latent_bits)[0].reshape(G, T, Rk)
        v_lat = quantize_rows(v_lat.reshape(G * T, Rv), latent_bits)[0].reshape(G, T, Rv)
    keys = torch.cat([lin(k_lat[g], u_k[g]) for g in range(G)], dim=-1)        # [T, H*D]  (:67-77)
    keys = keys.reshape(T, H, D).transpose(0, 1)                               # [H,T,D]
    cos, sin = rope_cos_sin(T, D, theta)
    cos, sin = cos.to(q.dtype), sin.to(q.dtype)

    def rot(x):
        return x * cos + torch.cat((-x[..., D // 2:], x[..., :D // 2]), dim=-1) * sin

    q, keys = rot(q), rot(keys)
    scores = torch.matmul(q, keys.transpose(1, 2)) / math.sqrt(D)             # [H,T,T]
    if attention_mask is not None:
        scores = scores + attention_mask.reshape(1, T, T)
    probs = torch.softmax(scores, dim=-1, dtype=torch.float32).to(q.dtype)
    ctx = torch.matmul(probs.reshape(G, gs * T, T), v_lat)                     # [G, gs*T, Rv]
    ctx = ctx.reshape(H, T, Rv).transpose(0, 1).reshape(T, H * Rv)
    out = lin(ctx, wo)
    return out, probs, k_lat, v_lat


# ------------------------------------------------------------------- decode step
def decode_step(hidden: torch.Tensor, position: int, weights: Dict[str, torch.Tensor],
                k_lat: torch.Tensor, v_lat: torch.Tensor,
                attention_mask: Optional[torch.Tensor] = None,
                theta: float = 10000.0, max_pos: Optional[int] = None, latent_bits: int = 16,
                latent_group_size: int = 0):
    """One-token decode of the low-rank attention module (batch 1), fp16 tensors on CPU.

    hidden [hidden] fp16; weights: wq [H*D,hidden], vt_k [G*Rk,hidden], vt_v [G*Rv,hidden],
    b [H,Rk,D], wo [hidden,H*Rv]; k_lat [G,L,Rk], v_lat [G,L,Rv] are the caches *before*
    the append.  Returns (attn_output [hidden] fp16, attn_weights [H,L+1] fp16,
    k_lat_new [G,L+1,Rk], v_lat_new [G,L+1,Rv]).

    Follows kernel/palu_attention.py:147-263, decode branch :207-219:
      q/latent projections :164-174, cache append :193, q RoPE :214-215 (HF 4.37.2
      rotary: fp32 cos/sin table cast to fp16, q*cos + rot(q)*sin in fp16), abx / sqrt(D)
      :219, mask :229-234, softmax fp32 -> fp16 :238, fused latent P.V :248-251, o_proj :257.
    """
    wq, vt_k, vt_v, b, wo = (weights[k] for k in ("wq", "vt_k", "vt_v", "b", "wo"))
    H, Rk, D = b.shape
    G = k_lat.shape[0]
    gs = H // G
    Rv = v_lat.shape[-1]
    h2 = hidden.reshape(1, -1)
    q = torch.nn.functional.linear(h2, wq).reshape(H, 1, D)
    k_new = torch.nn.functional.linear(h2, vt_k).reshape(G, 1, Rk)
    v_new = torch.nn.functional.linear(h2, vt_v).reshape(G, 1, Rv)
    if latent_bits < 16:
        # accuracy-path semantics: project -> fake-quantise each (token, group) latent row -> attend
        # (palu/model/modules/svd_linear.py:84-90,124-139); the caches passed in are already fake-quantised
        # (latent_group_size: the Quantizer's group_size, quant.py:60-79 / --lt_group_size of utils.py:105)
        k_new = quantize_rows(k_new.reshape(G, Rk), latent_bits, latent_group_size)[0].reshape(G, 1, Rk)
        v_new = quantize_rows(v_new.reshape(G, Rv), latent_bits, latent_group_size)[0].reshape(G, 1, Rv)
    k_all = torch.cat((k_lat, k_new), dim=1)
    v_all = torch.cat((v_lat, v_new), dim=1)
    L = k_all.shape[1]

    cos, sin = rope_cos_sin(position + 1, D, theta, start=position)      # one row
    cos, sin = cos.to(q.dtype), sin.to(q.dtype)
    q = q * cos + torch.cat((-q[..., D // 2:], q[..., :D // 2]), dim=-1) * sin

    scores = abx_scores(q, b, k_all, theta) / math.sqrt(D)               # [H,1,L] fp16
    if attention_mask is not None:
        scores = scores + attention_mask.reshape(1, 1, L)
    probs = torch.softmax(scores, dim=-1, dtype=torch.float32).to(q.dtype)
    ctx = torch.matmul(probs.reshape(G, gs, L), v_all)                   # [G,gs,Rv]
    out = torch.nn.functional.linear(ctx.reshape(1, H * Rv), wo).reshape(-1)
    return out, probs.reshape(H, L), k_all, v_all


# ------------------------------------------------------------------ quantisation
def quantize_rows(w: torch.Tensor, n_bits: int, group_size: int = 0, sym: bool = False,
                  clip_ratio: float = 1.0):
    """Row-wise fake-quant with the integer codes exposed.

    w: [N, R] (arithmetic runs in w.dtype -- fp16 in practice).  Returns
    (dequant [N,R] w.dtype, codes [N,R] int16, scale [N*,1] w.dtype, zero [N*,1] w.dtype)
    where N* = N (group_size 0) or N*R/group_size rows of ``group_size`` columns.

    Follows palu/model/modules/quant.py:5-41 operation by operation:
      asym: scale = clamp(max-min, 1e-5)/q_max, zero = clamp(round(-min/scale), 0, q_max),
            code = clamp(round(w/scale)+zero, 0, q_max); dequant = (code-zero)*scale
      sym : scale = clamp(amax|w|, 1e-5)/q_max (q_max = 2^(b-1)-1), zero = 0,
            code = clamp(round(w/scale), -2^(b-1), q_max).
    ``torch.round`` is round-half-to-even; the 1e-5 floor is an fp16 subnormal for fp16 input.
    """
    assert w.dim() == 2 and n_bits < 16
    shape = w.shape
    if group_size > 0:
        assert shape[-1] % group_size == 0
        w = w.reshape(-1, group_size)
    if sym:
        q_max, q_min = 2 ** (n_bits - 1) - 1, -(2 ** (n_bits - 1))
        top = w.abs().amax(dim=-1, keepdim=True).clamp(min=1e-5)
        if clip_ratio < 1.0:
            top = top * clip_ratio
        scale = top / q_max
        zero = torch.zeros_like(scale)
    else:
        q_max, q_min = 2 ** n_bits - 1, 0
        hi = w.amax(dim=-1, keepdim=True)
        lo = w.amin(dim=-1, keepdim=True)
        if clip_ratio < 1.0:
            hi = hi * clip_ratio
            lo = lo * clip_ratio
        scale = (hi - lo).clamp(min=1e-5) / q_max
        zero = torch.round(-lo / scale).clamp(min=q_min, max=q_max)
    codes = torch.clamp(torch.round(w / scale) + zero, q_min, q_max)
    deq = (codes - zero) * scale
    return deq.reshape(shape), codes.reshape(shape).to(torch.int16), scale, zero


def packed_row_bytes(R: int, n_bits: int) -> int:
    """Bytes of one packed row of R codes (R % 32 == 0 for 3-bit, R % 8 == 0 for 4-bit)."""
    assert n_bits in (3, 4)
    assert R % (32 if n_bits == 3 else 8) == 0
    return R * n_bits // 8


def pack_codes(codes: np.ndarray, n_bits: int) -> np.ndarray:
    """Pack unsigned codes [..., R] (values < 2^n_bits) into bytes [..., R*n_bits/8].

    Layout (defined by this build; the reference has no packed format):
    a little-endian bit stream per row -- code j occupies bits [j*b, (j+1)*b) of the row.
    For b=4 byte k = code[2k] | code[2k+1] << 4; for b=3 every 8 codes fill 3 bytes and
    every 32 codes fill three little-endian uint32 words.
    """
    codes = np.asarray(codes)
    R = codes.shape[-1]
    nbytes = packed_row_bytes(R, n_bits)
    c = codes.astype(np.uint8).reshape(-1, R)
    assert int(c.max(initial=0)) < (1 << n_bits)
    bits = ((c[:, :, None] >> np.arange(n_bits, dtype=np.uint8)) & 1).reshape(c.shape[0], R * n_bits)
    out = np.packbits(bits, axis=1, bitorder="little")
    return out.reshape(codes.shape[:-1] + (nbytes,))


def unpack_codes(packed: np.ndarray, n_bits: int, R: int) -> np.ndarray:
    """Inverse of :func:`pack_codes` -> uint8 codes [..., R]."""
    packed = np.asarray(packed, dtype=np.uint8)
    nbytes = packed_row_bytes(R, n_bits)
    p = packed.reshape(-1, nbytes)
    bits = np.unpackbits(p, axis=1, bitorder="little").reshape(p.shape[0], R, n_bits)
    c = (bits.astype(np.uint8) << np.arange(n_bits, dtype=np.uint8)).sum(axis=2).astype(np.uint8)
    return c.reshape(packed.shape[:-1] + (R,))


def dequant_codes(codes: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor) -> torch.Tensor:
    """(code - zero) * scale in scale.dtype -- the last line of quant.py:39."""
    return (codes.to(scale.dtype) - zero) * scale


# --------------------------------------------------------------------- Hadamard
def fwht(x: torch.Tensor) -> torch.Tensor:
    """Unnormalised Walsh-Hadamard transform over the last dim (power of two), natural
    (Sylvester) order: y = x @ H_n.

    Equivalent of the butterfly loop of palu/model/modules/hadamard_utils.py:92-103
    (and of the external ``fast_hadamard_transform.hadamard_transform`` with scale 1).
    """
    n = x.shape[-1]
    assert n > 0 and (n & (n - 1)) == 0
    y = x.clone()
    h = 1
    while h < n:
        y = y.reshape(*x.shape[:-1], n // (2 * h), 2, h)
        lo, hi = y[..., 0, :], y[..., 1, :]
        y = torch.stack((lo + hi, lo - hi), dim=-2)
        h *= 2
    return y.reshape(x.shape)


def had12() -> torch.Tensor:
    """The 12x12 Hadamard matrix used for n = 12*2^m (hadamard_utils.py:196-211, data).

    Bordered circulant: row 0 = (+1, -1 x11); rows 1..11 = (+1, cyclic shifts of c).
    The golden test compares it element-wise with the reference's literal table.
    """
    c = torch.tensor([1, -1, 1, -1, -1, -1, 1, 1, 1, -1, 1], dtype=torch.float32)
    m = torch.empty(12, 12, dtype=torch.float32)
    m[:, 0] = 1.0
    m[0, 1:] = -1.0
    for r in range(11):
        m[r + 1, 1:] = torch.roll(c, r)
    return m


HAD_K_ORDER = (244, 180, 172, 156, 140, 108, 92, 84, 76, 68, 60, 52, 44, 36, 28, 40, 20, 12)


def hadK_for(n: int):
    """(hadK [K,K] fp32 or None, K): the table get_hadK (hadamard_utils.py:5-83) selects for a width n = K * 2^m --
    first K in its test order with n % K == 0.  K = 12 is the bordered circulant below; the other 17 factors are the
    reference's literal matrices, read as data from tests/golden/g8_hadk.npz (written by tests/golden/make_hadk.py)."""
    for K in HAD_K_ORDER:
        if n % K == 0:
            assert ((n // K) & (n // K - 1)) == 0, "n must be K * 2^m"
            if K == 12:
                return had12(), 12
            import os
            import numpy as np
            z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                     "g8_hadk_tables.npz"))
            bits = np.unpackbits(z[f"had{K}"])[:K * K].reshape(K, K)
            return torch.from_numpy(bits.astype(np.float32) * 2.0 - 1.0), K
    assert (n & (n - 1)) == 0, "n must be 2^m or K * 2^m"
    return None, 1


def apply_hadamard(x: torch.Tensor) -> torch.Tensor:
    """x -> x . Had_n / sqrt(n) over the last dim, n = 2^m or K * 2^m for the K of get_hadK.

    Follows hadamard_utils.py:85-90 + :138-147 (matmul_hadU_cuda): for n = K*2^m the row is
    viewed as [K, n/K], Sylvester-transformed over the inner axis, then mixed by hadK over
    the outer axis; one 1/sqrt(n) scale.
    """
    n = x.shape[-1]
    dt = x.dtype
    hk, K = hadK_for(n)
    if K == 1:
        return (fwht(x.float()) / math.sqrt(n)).to(dt)
    y = x.float().reshape(*x.shape[:-1], K, n // K)
    y = (fwht(y) if n // K > 1 else y) / math.sqrt(n)
    y = torch.matmul(hk, y)
    return y.reshape(x.shape).to(dt)


def fuse_hadamard_into_weights(vt: torch.Tensor, u_weights):
    """Offline rotation of one HeadwiseLowRankModule: VT_g <- (had(VT_g^T))^T, U_g <- had(U_g).

    Follows palu/model/modules/svd_linear.py:156-168.  vt [sum R, hidden]; u_weights list of
    [gs*D, R_g].  Returns (vt', [u'_g]); U'.VT' == U.VT up to rounding.
    """
    vt = vt.clone()
    new_u = []
    r0 = 0
    for u in u_weights:
        R = u.shape[1]
        vt[r0:r0 + R] = apply_hadamard(vt[r0:r0 + R].t().contiguous()).t()
        new_u.append(apply_hadamard(u.contiguous()))
        r0 += R
    return vt, new_u
