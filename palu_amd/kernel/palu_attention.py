"""Host-side mirror of the reference's `kernel/palu_attention.py` for the decode path.

Same class names, constructor arguments, method names, return tuples and error behaviour as
`HeadwiseLowRankModule` (kernel/palu_attention.py:16-122) and `LlamaPaluAttention` (:124-308), so
`run_latency_attention.py` and the reference's tests read unchanged -- but:

  * the decode branch (q_len == 1, :207-257) is ONE call into the HIP library
    (`palu_decode_step_f16`: qkv GEMV + RoPE + in-place cache append -> MFMA abx -> split-L
    softmax.PV -> o_proj GEMV); there is no PyTorch fallback for it;
  * the cache is a pre-allocated latent cache (`LatentCache`) that implements the 4.37.2 protocol
    the reference uses (`get_usable_length`, `update`) with an in-place row append instead of
    `torch.cat` (which re-copied the whole cache every step, :193);
  * the module does not inherit from `transformers.LlamaAttention` (the reference pins 4.37.2; the
    attributes it relied on are gone in 5.x), it only reads the config fields the reference reads.

The prefill branch (q_len > 1, :196-206; row N1 of SURVEY.md 8(f)) runs on the flash-style HIP kernel
`palu_prefill_attn_f16` (`_prefill_flash`); a torch-op composition of the same branch remains only for what that
kernel does not take (output_attentions, arbitrary masks, non-standard position ids, un-fused o_proj).
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import _lib
from .abx_rope import abx as recompute_k_gemv  # same alias as kernel/palu_attention.py:13
from .abx_rope import invalidate_b, prepare_b, rope_cs_table, rope_inv_freq, shared_b

__all__ = ["HeadwiseLowRankModule", "LlamaPaluAttention", "LatentCache", "QuantLatentCache", "DynamicCache",
           "build_b", "fuse_wo"]


# ------------------------------------------------------------------------------------ cache
class LatentCache:
    """Pre-allocated latent KV cache, one (k, v) pair of [1, G, capacity, R] fp16 buffers per layer.

    Implements what the reference asks of HF-4.37.2's DynamicCache (kernel/palu_attention.py:185,193):
    `get_usable_length(new_len, layer_idx)` and `update(k, v, layer_idx) -> (K_all, V_all)` where the
    returned tensors are views `[1, G, L, R]` of the buffers.  Rows are appended in place; capacity
    grows geometrically (one copy) only when exhausted, so a decode step never re-copies the cache.
    """

    def __init__(self, capacity: int = 0, headroom: int = 256):
        self._k: List[Optional[torch.Tensor]] = []
        self._v: List[Optional[torch.Tensor]] = []
        self._len: List[int] = []
        self._min_capacity = int(capacity)
        self._headroom = int(headroom)

    def __len__(self):
        return len(self._k)

    def _ensure_layer(self, layer_idx: int):
        while len(self._k) <= layer_idx:
            self._k.append(None)
            self._v.append(None)
            self._len.append(0)

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self._len[layer_idx] if layer_idx < len(self._len) else 0

    def get_usable_length(self, new_seq_length: int, layer_idx: int = 0) -> int:
        return self.get_seq_length(layer_idx)

    def capacity(self, layer_idx: int = 0) -> int:
        return 0 if layer_idx >= len(self._k) or self._k[layer_idx] is None else self._k[layer_idx].shape[2]

    def reserve(self, layer_idx: int, rows: int, like_k: torch.Tensor, like_v: torch.Tensor):
        """Make room for `rows` rows in total (keeps the valid prefix)."""
        self._ensure_layer(layer_idx)
        cur = self.capacity(layer_idx)
        if cur >= rows:
            return
        new_cap = max(rows, self._min_capacity, 2 * cur)
        new_cap = (new_cap + 63) // 64 * 64
        n = self._len[layer_idx]
        for store, like in ((self._k, like_k), (self._v, like_v)):
            buf = torch.empty((1, like.shape[1], new_cap, like.shape[3]), dtype=like.dtype, device=like.device)
            if n:
                buf[:, :, :n].copy_(store[layer_idx][:, :, :n])
            store[layer_idx] = buf

    def buffers(self, layer_idx: int = 0):
        return self._k[layer_idx], self._v[layer_idx]

    def advance(self, layer_idx: int, rows: int = 1):
        """Account for rows a kernel wrote in place (decode path)."""
        self._len[layer_idx] += rows

    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int, cache_kwargs=None):
        if key_states.dim() != 4 or value_states.dim() != 4:
            raise ValueError("LatentCache.update expects [bsz, groups, seq, rank] tensors")
        self._ensure_layer(layer_idx)
        t = key_states.shape[2]
        n = self._len[layer_idx]
        self.reserve(layer_idx, n + t + self._headroom, key_states, value_states)
        self._k[layer_idx][:, :, n:n + t].copy_(key_states)
        self._v[layer_idx][:, :, n:n + t].copy_(value_states)
        self._len[layer_idx] = n + t
        return self._k[layer_idx][:, :, :n + t], self._v[layer_idx][:, :, :n + t]


class QuantLatentCache:
    """Packed 3/4-bit latent KV cache (DESIGN.md "packed latent format"): per layer
    k_codes/v_codes uint8 [1, G, capacity, R*bits/8] and k_meta/v_meta fp16 [1, G, capacity, 2] = (scale, zero),
    quantised per (token, head-group) row like the reference's defaults (svd_linear.py:124-139, quant.py:60-79).
    Same protocol as LatentCache; `update` quantises the incoming fp16 latents on the GPU and returns the
    DEQUANTISED [1, G, L, R] tensors (the fake-quant values the reference's accuracy path attends over)."""

    def __init__(self, n_bits: int, capacity: int = 0, headroom: int = 256, group_size: int = 0):
        if n_bits not in (3, 4):
            raise ValueError("QuantLatentCache: n_bits must be 3 or 4")
        if group_size < 0 or group_size % 32:
            raise ValueError("QuantLatentCache: group_size must be 0 (one (scale, zero) pair per row) or a multiple of 32")
        self.n_bits = n_bits
        # quantize_tensor's group_size (quant.py:11-13, --lt_group_size): every `group_size` consecutive columns of a row
        # carry their own (scale, zero); the meta tensors are then [1, G, capacity, 2 * R / group_size] (pair q at
        # [..., 2q : 2q + 2]); the packed codes do not change
        self.group_size = int(group_size)
        self._store = []          # per layer: dict(kc, km, vc, vm, Rk, Rv)
        self._len: List[int] = []
        self._min_capacity, self._headroom = int(capacity), int(headroom)

    def __len__(self):
        return len(self._store)

    def _ensure_layer(self, layer_idx):
        while len(self._store) <= layer_idx:
            self._store.append(None)
            self._len.append(0)

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self._len[layer_idx] if layer_idx < len(self._len) else 0

    def get_usable_length(self, new_seq_length: int, layer_idx: int = 0) -> int:
        return self.get_seq_length(layer_idx)

    def capacity(self, layer_idx: int = 0) -> int:
        st = self._store[layer_idx] if layer_idx < len(self._store) else None
        return 0 if st is None else st["kc"].shape[2]

    def reserve(self, layer_idx: int, rows: int, G: int, Rk: int, Rv: int, device):
        from .quant import packed_row_bytes
        self._ensure_layer(layer_idx)
        cur = self.capacity(layer_idx)
        if cur >= rows:
            return
        cap = (max(rows, self._min_capacity, 2 * cur) + 63) // 64 * 64
        n = self._len[layer_idx]
        old = self._store[layer_idx]
        gsz = self.group_size
        if gsz and (Rk % gsz or Rv % gsz):
            raise ValueError(f"QuantLatentCache: group_size {gsz} must divide the group ranks ({Rk}, {Rv})")
        new = {"Rk": Rk, "Rv": Rv,
               "kc": torch.zeros((1, G, cap, packed_row_bytes(Rk, self.n_bits)), dtype=torch.uint8, device=device),
               "vc": torch.zeros((1, G, cap, packed_row_bytes(Rv, self.n_bits)), dtype=torch.uint8, device=device),
               "km": torch.zeros((1, G, cap, 2 * (Rk // gsz if gsz else 1)), dtype=torch.float16, device=device),
               "vm": torch.zeros((1, G, cap, 2 * (Rv // gsz if gsz else 1)), dtype=torch.float16, device=device)}
        if old is not None and n:
            for k in ("kc", "vc", "km", "vm"):
                new[k][:, :, :n].copy_(old[k][:, :, :n])
        self._store[layer_idx] = new

    def buffers(self, layer_idx: int = 0):
        return self._store[layer_idx]

    def advance(self, layer_idx: int, rows: int = 1):
        self._len[layer_idx] += rows

    def dequantized(self, layer_idx: int = 0):
        from .quant import unpack_dequant
        st, n = self._store[layer_idx], self._len[layer_idx]
        gsz = self.group_size

        def deq(codes, meta, R):
            codes, meta = codes[:, :, :n].contiguous(), meta[:, :, :n].contiguous()
            if not gsz:
                return unpack_dequant(codes, meta, self.n_bits, R)
            ng = R // gsz                       # every column group is a row of width group_size to the unpacker
            out = unpack_dequant(codes.reshape(*codes.shape[:-1], ng, -1), meta.reshape(*meta.shape[:-1], ng, 2), self.n_bits, gsz)
            return out.reshape(*codes.shape[:-1], R)
        return deq(st["kc"], st["km"], st["Rk"]), deq(st["vc"], st["vm"], st["Rv"])

    def _quantize(self, x: torch.Tensor):
        """[..., R] fp16 -> (codes [..., R*bits/8], meta [..., 2] or [..., 2 R / group_size]) with quantize_tensor's
        semantics (quant.py:5-41: whole rows, or rows cut into groups of group_size columns)."""
        from .quant import quantize_pack
        x = x.half().contiguous()
        gsz = self.group_size
        if not gsz:
            return quantize_pack(x, self.n_bits)
        R = x.shape[-1]
        c, m = quantize_pack(x.reshape(*x.shape[:-1], R // gsz, gsz), self.n_bits)
        return c.reshape(*x.shape[:-1], -1), m.reshape(*x.shape[:-1], -1)

    def append_rows(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int) -> None:
        """quantise + pack `[1, G, t, R]` latent rows behind the cached ones (what `update` does, without handing back a
        dequantised copy of the whole cache)."""
        if key_states.dim() != 4 or value_states.dim() != 4:
            raise ValueError("QuantLatentCache expects [bsz, groups, seq, rank] tensors")
        _, G, t, Rk = key_states.shape
        Rv = value_states.shape[3]
        self._ensure_layer(layer_idx)
        n = self._len[layer_idx]
        self.reserve(layer_idx, n + t + self._headroom, G, Rk, Rv, key_states.device)
        st = self._store[layer_idx]
        kc, km = self._quantize(key_states)
        vc, vm = self._quantize(value_states)
        st["kc"][:, :, n:n + t].copy_(kc)
        st["km"][:, :, n:n + t].copy_(km)
        st["vc"][:, :, n:n + t].copy_(vc)
        st["vm"][:, :, n:n + t].copy_(vm)
        self._len[layer_idx] = n + t

    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int, cache_kwargs=None):
        from .quant import quantize_pack
        if key_states.dim() != 4 or value_states.dim() != 4:
            raise ValueError("QuantLatentCache.update expects [bsz, groups, seq, rank] tensors")
        _, G, t, Rk = key_states.shape
        Rv = value_states.shape[3]
        self._ensure_layer(layer_idx)
        n = self._len[layer_idx]
        self.reserve(layer_idx, n + t + self._headroom, G, Rk, Rv, key_states.device)
        st = self._store[layer_idx]
        kc, km = self._quantize(key_states)
        vc, vm = self._quantize(value_states)
        st["kc"][:, :, n:n + t].copy_(kc)
        st["km"][:, :, n:n + t].copy_(km)
        st["vc"][:, :, n:n + t].copy_(vc)
        st["vm"][:, :, n:n + t].copy_(vm)
        self._len[layer_idx] = n + t
        return self.dequantized(layer_idx)


CACHE_FORMAT = "palu-latent-cache"
CACHE_FORMAT_VERSION = "1"


def save_cache(cache, path: str) -> None:
    """Write a LatentCache / QuantLatentCache to one safetensors file -- the stable on-disk form of the latent cache
    (SURVEY 8(f) N2).  Only the valid rows are stored, in the device layout of DESIGN.md section 3:
      fp16 cache : `layer{i}.k` [G, n, Rk], `layer{i}.v` [G, n, Rv]                         fp16
      packed     : `layer{i}.k_codes` [G, n, Rk*bits/8] uint8 (little-endian bit stream per row, code j at bits
                   [j*bits, (j+1)*bits)), `layer{i}.k_meta` [G, n, 2] fp16 = (scale, zero); same for v
    metadata: format, version, kind (fp16|packed), bits, layers, rank_k / rank_v per layer."""
    from safetensors.torch import save_file
    tensors, meta = {}, {"format": CACHE_FORMAT, "version": CACHE_FORMAT_VERSION, "layers": str(len(cache))}
    if isinstance(cache, QuantLatentCache):
        meta.update(kind="packed", bits=str(cache.n_bits), group_size=str(cache.group_size))
        for i in range(len(cache)):
            st, n = cache.buffers(i), cache.get_seq_length(i)
            if st is None:
                continue
            meta[f"layer{i}.rank_k"], meta[f"layer{i}.rank_v"] = str(st["Rk"]), str(st["Rv"])
            for name, key in (("k_codes", "kc"), ("k_meta", "km"), ("v_codes", "vc"), ("v_meta", "vm")):
                tensors[f"layer{i}.{name}"] = st[key][0, :, :n].contiguous().cpu()
    elif isinstance(cache, LatentCache):
        meta.update(kind="fp16", bits="16")
        for i in range(len(cache)):
            k, v = cache.buffers(i)
            n = cache.get_seq_length(i)
            if k is None:
                continue
            tensors[f"layer{i}.k"] = k[0, :, :n].contiguous().cpu()
            tensors[f"layer{i}.v"] = v[0, :, :n].contiguous().cpu()
    else:
        raise TypeError("save_cache expects a LatentCache or QuantLatentCache")
    save_file(tensors, path, metadata=meta)


def load_cache(path: str, device="cpu", capacity: int = 0, headroom: int = 256):
    """Inverse of save_cache: rebuilds the cache object (pre-allocated to max(capacity, rows + headroom)) on `device`."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != CACHE_FORMAT:
            raise ValueError(f"{path}: not a {CACHE_FORMAT} file")
        if meta.get("version") != CACHE_FORMAT_VERSION:
            raise ValueError(f"{path}: unsupported {CACHE_FORMAT} version {meta.get('version')}")
        layers = int(meta["layers"])
        names = set(f.keys())
        if meta["kind"] == "packed":
            cache = QuantLatentCache(int(meta["bits"]), capacity, headroom, int(meta.get("group_size", "0")))
            for i in range(layers):
                if f"layer{i}.k_codes" not in names:
                    cache._ensure_layer(i)
                    continue
                kc = f.get_tensor(f"layer{i}.k_codes")
                G, n = kc.shape[0], kc.shape[1]
                cache.reserve(i, n + headroom, G, int(meta[f"layer{i}.rank_k"]), int(meta[f"layer{i}.rank_v"]), device)
                st = cache.buffers(i)
                for name, key in (("k_codes", "kc"), ("k_meta", "km"), ("v_codes", "vc"), ("v_meta", "vm")):
                    st[key][0, :, :n].copy_(f.get_tensor(f"layer{i}.{name}"))
                cache.advance(i, n)
        elif meta["kind"] == "fp16":
            cache = LatentCache(capacity, headroom)
            for i in range(layers):
                if f"layer{i}.k" not in names:
                    cache._ensure_layer(i)
                    continue
                k, v = f.get_tensor(f"layer{i}.k"), f.get_tensor(f"layer{i}.v")
                cache.update(k.unsqueeze(0).to(device), v.unsqueeze(0).to(device), i)
        else:
            raise ValueError(f"{path}: unknown cache kind {meta['kind']}")
    return cache


DynamicCache = LatentCache  # name used by run_latency_attention.py:62-65 and the reference tests



# ------------------------------------------------------------------------- weight re-layouts
def fold_u_per_head(u_weights, head_dim: int) -> torch.Tensor:
    """[G x (gs*D, R)] -> [H, D, R]: head h = g*gs + j owns rows j*D:(j+1)*D of U_g."""
    stacked = torch.stack(list(u_weights))                      # [G, gs*D, R]
    G, gsD, R = stacked.shape
    return stacked.reshape(G * (gsD // head_dim), head_dim, R)


def build_b(u_weights, group_size: int, head_dim: int, n_rep: int = 1) -> torch.Tensor:
    """abx operand B[h] = U_{h//gs}.weight[(h%gs)*D:(h%gs+1)*D, :]^T -> [H, R, D]
    (kernel/palu_attention.py:108-114).  n_rep > 1 (GQA: `n_rep` query heads per KV head, the KV heads being what the
    U blocks reconstruct -- palu/model/svd_mistral/modeling_palu_mistral.py:37-59): query head h takes the block of KV
    head h // n_rep, so the heads of one KV head carry identical B (the shared-B score kernel detects that)."""
    b = fold_u_per_head(u_weights, head_dim).transpose(1, 2)
    if n_rep > 1:
        b = b.repeat_interleave(n_rep, dim=0)
    return b.contiguous()


def fuse_wo(wo: torch.Tensor, uv_weights, head_dim: int, n_rep: int = 1) -> torch.Tensor:
    """W_o'[:, h*Rv:(h+1)*Rv] = W_o[:, h*D:(h+1)*D] @ U_v[h//gs][(h%gs)*D:(h%gs+1)*D, :]
    (kernel/palu_attention.py:285-306) -> [hidden, H*Rv], fp32.  n_rep > 1: query head h uses the U_v block of KV head
    h // n_rep."""
    uv = fold_u_per_head([u.float() for u in uv_weights], head_dim)        # [KV heads, D, Rv]
    if n_rep > 1:
        uv = uv.repeat_interleave(n_rep, dim=0)                            # [H, D, Rv]
    H = uv.shape[0]
    w = wo.float().reshape(-1, H, head_dim)                                # [hidden, H, D]
    return torch.einsum("ohd,hdr->ohr", w, uv).reshape(w.shape[0], -1)


# ------------------------------------------------------------------------- low-rank projection
class HeadwiseLowRankModule(nn.Module):
    """Head-group-wise low-rank linear: y = cat_g U_g (VT x)[ranks of g]  (kernel/palu_attention.py:16-77).

    `VT`: Linear(in_features -> sum(ranks)); `U_list[g]`: Linear(ranks[g] -> out_features/len(ranks));
    `B` (set by `from_linear(..., attn_module=...)`): [H, R, D] per-head reconstruction used by abx.
    """

    def __init__(self, ranks, in_features, out_features, bias):
        super().__init__()
        self.ranks = ranks
        self.num_groups = len(ranks)
        self.in_features = in_features
        self.out_features = out_features
        self.group_dim = out_features // self.num_groups
        if self.group_dim * self.num_groups != self.out_features:
            raise ValueError(
                f"out_features must be divisible by num_groups (got `out_features`: {self.out_features}"
                f" and `num_groups`: {self.num_groups}).")
        self.VT = nn.Linear(in_features, sum(ranks), bias=False)
        ups = []
        for r in ranks:
            lin = nn.Linear(r, self.group_dim, bias=bias)
            nn.init.normal_(lin.weight)
            ups.append(lin)
        self.U_list = nn.ModuleList(ups)

    def _load_from_state_dict(self, *args, **kwargs):
        # load_state_dict copies into B.data in place: neither B._version nor data_ptr moves, so the cached MFMA
        # fragments of the old weight must be dropped explicitly
        super()._load_from_state_dict(*args, **kwargs)
        if hasattr(self, "B"):
            invalidate_b(self.B)

    @staticmethod
    def _check3(x):
        assert x.dim() == 3, f"hidden_states should have 3 dimensions, got {x.dim()}"

    def project_to_latent(self, hidden_states: torch.Tensor):
        """[bsz, seq, in_features] -> [bsz, seq, sum(ranks)]  (:59-65)"""
        self._check3(hidden_states)
        return self.VT(hidden_states)

    def reconstruct(self, hidden_states: torch.Tensor):
        """[bsz, seq, sum(ranks)] -> [bsz, seq, out_features]  (:67-77)"""
        self._check3(hidden_states)
        pieces = torch.split(hidden_states, list(self.ranks), dim=-1)
        return torch.cat([u(z) for u, z in zip(self.U_list, pieces)], dim=-1)

    def forward(self, hidden_states: torch.Tensor):
        self._check3(hidden_states)
        return self.reconstruct(self.VT(hidden_states))

    @staticmethod
    def from_linear(old_module: nn.Linear, ranks: list, attn_module=None):
        """Per-group truncated SVD of a dense projection (:79-122): W_g = (U S)[:, :r] . Vt[:r].
        With `attn_module` the kernel operand B[h] = U_{h//gs}[(h%gs)D:(h%gs+1)D, :]^T is built (:108-114)."""
        new = HeadwiseLowRankModule(ranks, old_module.in_features, old_module.out_features,
                                    bias=old_module.bias is not None)
        G = len(ranks)
        w = old_module.weight.data.reshape(G, -1, old_module.in_features).float()
        vt_rows = []
        for g, r in enumerate(ranks):
            u, s, vh = torch.linalg.svd(w[g], full_matrices=False)
            left = (u[:, :r] * s[:r]).contiguous()
            if new.U_list[g].weight.data.shape != left.shape:
                raise ValueError(f"{new.U_list[g].weight.data.shape} != {left.shape}")
            new.U_list[g].weight.data = left
            vt_rows.append(vh[:r, :])
        if attn_module is not None:
            new.B = nn.Parameter(build_b([u.weight.data for u in new.U_list], attn_module.group_size,
                                         attn_module.head_dim, getattr(attn_module, "n_rep", 1)))
        vt = torch.cat(vt_rows, dim=0).contiguous()
        assert new.VT.weight.data.shape == vt.shape
        new.VT.weight.data = vt
        return new


    @staticmethod
    def from_linear_whiten(old_module: nn.Linear, ranks: list):
        """Per-group truncated SVD of the WHITENED projection (palu/model/modules/svd_linear.py:6-34,170-204): with the
        activation scaling matrix S the calibration pass left on the layer (`old_module.scaling_diag_matrix`,
        palu/decomposition.py:21-80), W_g S = U Sigma Vt, L = U sqrt(Sigma)[:, :r], R = sqrt(Sigma) (Vt S^-1)[:r] -- sqrt(Sigma)
        on both factors, unlike from_linear.  The bias (if any) stays on the U side, split by group."""
        try:
            scaling = old_module.scaling_diag_matrix
        except AttributeError:
            raise FileExistsError("Cache may not be loaded correctly") from None          # (the reference's error, :9-11)
        new = HeadwiseLowRankModule(ranks, old_module.in_features, old_module.out_features,
                                    bias=old_module.bias is not None)
        G = len(ranks)
        dt = old_module.weight.dtype
        w = old_module.weight.data.reshape(G, -1, old_module.in_features)
        s32 = scaling.to(device=w.device, dtype=torch.float32)
        s_inv = torch.linalg.inv(s32)
        rows = []
        for g, r in enumerate(ranks):
            u, sig, vh = torch.linalg.svd(torch.matmul(w[g].float(), s32), full_matrices=False)
            root = torch.sqrt(sig[:r])
            left = (u[:, :r] * root).to(dt).contiguous()
            if new.U_list[g].weight.data.shape != left.shape:
                raise ValueError(f"{new.U_list[g].weight.data.shape} != {left.shape}")
            new.U_list[g].weight.data = left
            rows.append((root[:, None] * torch.matmul(vh, s_inv)[:r, :]).to(dt))
            if old_module.bias is not None:
                new.U_list[g].bias.data = old_module.bias.data.reshape(G, -1)[g].clone()
        vt = torch.cat(rows, dim=0).contiguous()
        assert new.VT.weight.data.shape == vt.shape
        new.VT.weight.data = vt
        return new


# --------------------------------------------------------------------------------- attention
def _q_scratch_ok(mod, Rk: int) -> bool:
    """palu_decode_step_q parks the new (unquantised) latent rows in its workspace: G * Rk <= 4096 halves."""
    return mod.num_groups * Rk <= 4096


def additive_mask(mask: torch.Tensor, dtype) -> torch.Tensor:
    """Additive form of an attention mask (kernel/palu_attention.py:229-234 adds it to the scores): boolean masks
    (True = attend, what newer transformers releases build) become 0 / finfo(dtype).min; others are cast to `dtype`."""
    if mask.dtype == torch.bool:
        return torch.zeros(mask.shape, dtype=dtype, device=mask.device).masked_fill_(~mask, torch.finfo(dtype).min)
    return mask if mask.dtype == dtype else mask.to(dtype)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class LlamaPaluAttention(nn.Module):
    """Llama attention with a low-rank latent KV cache (kernel/palu_attention.py:124-308).

    Reads from `config`: hidden_size, num_attention_heads, attention_bias, group_size, num_groups,
    total_rank_k, total_rank_v (+ optional head_dim, rope_theta, attention_dropout,
    max_position_embeddings) -- the same fields as :129-145.
    """

    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or self.hidden_size // self.num_heads
        # The reference's kernel-path module is MHA-only (:143, :201).  Here `num_key_value_heads < num_attention_heads`
        # (GQA) is taken too, with the reference's own grouping rule for such models
        # (palu/model/svd_mistral/modeling_palu_mistral.py:37-59, get_kv_info): config.group_size counts KV heads per
        # low-rank group and num_groups = num_key_value_heads // group_size.  A latent group then serves
        # group_size * n_rep QUERY heads -- that product is what the kernels see as the group size.
        kv = getattr(config, "num_key_value_heads", None) or self.num_heads
        if self.num_heads % kv:
            raise ValueError(f"num_attention_heads ({self.num_heads}) must be divisible by num_key_value_heads ({kv})")
        self.num_key_value_heads = kv
        self.n_rep = self.num_heads // kv
        self.attention_dropout = getattr(config, "attention_dropout", 0.0)
        self.rope_theta = float(getattr(config, "rope_theta", None) or 10000.0)
        bias = getattr(config, "attention_bias", False)

        self.kv_group_size = config.group_size                 # KV heads per latent group (== heads per group for MHA)
        self.num_groups = config.num_groups
        if self.num_groups * self.kv_group_size != kv:
            raise ValueError(f"num_groups ({self.num_groups}) * group_size ({self.kv_group_size}) must equal the number of "
                             f"key/value heads ({kv})")
        self.group_size = self.kv_group_size * self.n_rep      # QUERY heads per latent group
        self.total_rank_k = config.total_rank_k
        self.total_rank_v = config.total_rank_v
        self.group_rank_k = self.total_rank_k // self.num_groups
        self.group_rank_v = self.total_rank_v // self.num_groups
        self.fused_hidden_dim_o = self.group_rank_v * self.num_heads
        self.rank_k_list = [self.group_rank_k] * self.num_groups
        self.rank_v_list = [self.group_rank_v] * self.num_groups

        out = self.num_heads * self.head_dim
        out_kv = self.num_key_value_heads * self.head_dim
        self.q_proj = nn.Linear(self.hidden_size, out, bias=bias)
        self.k_proj = HeadwiseLowRankModule(self.rank_k_list, self.hidden_size, out_kv, bias=bias)
        self.v_proj = HeadwiseLowRankModule(self.rank_v_list, self.hidden_size, out_kv, bias=bias)
        self.o_proj = nn.Linear(self.fused_hidden_dim_o, self.hidden_size, bias=bias)
        self._ws = None          # HIP workspace (grows with the cache capacity)
        self._ws_cap = 0

    # -- helpers -----------------------------------------------------------------------------
    def _rope_tables(self, positions: torch.Tensor, dtype):
        """cos/sin rows for `positions` like HF-4.37.2's rotary_emb + apply_rotary_pos_emb:
        fp32 table from fl32(pos)*inv_freq, cast to the activation dtype (:214-215)."""
        inv = rope_inv_freq(positions.device, self.head_dim, self.rope_theta)
        ang = torch.outer(positions.to(inv.dtype), inv)
        ang = torch.cat((ang, ang), dim=-1)
        return ang.cos().to(dtype), ang.sin().to(dtype)

    def _workspace(self, device, capacity: int):
        if self._ws is None or self._ws_cap < capacity or self._ws.device != device:
            nbytes = _lib.lib.palu_decode_workspace_bytes(self.num_heads, self.num_groups, self.head_dim,
                                                          capacity, self.group_rank_v)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws_cap = capacity
        return self._ws

    def _decode_fused(self, hidden_states, attention_mask, pos, cache: "LatentCache", output_attentions):
        """q_len == 1, U_v folded into o_proj: the whole step in the HIP library."""
        if self.q_proj.bias is not None or self.o_proj.bias is not None:
            return self._decode_biased(hidden_states, attention_mask, pos, cache, output_attentions)
        dev = hidden_states.device
        li = self.layer_idx
        n = cache.get_seq_length(li)
        H, G, D = self.num_heads, self.num_groups, self.head_dim
        if cache.capacity(li) < n + 1:
            like_k = torch.empty((1, G, 0, self.group_rank_k), dtype=hidden_states.dtype, device=dev)
            like_v = torch.empty((1, G, 0, self.group_rank_v), dtype=hidden_states.dtype, device=dev)
            cache.reserve(li, n + 1 + cache._headroom, like_k, like_v)
        kbuf, vbuf = cache.buffers(li)
        cap = kbuf.shape[2]
        ws = self._workspace(dev, cap + 8)
        # heads that share B inside a group (true GQA) take the shared-B score kernel (keys reconstructed once per group)
        bg = shared_b(self.k_proj.B, G)
        frag = prepare_b(self.k_proj.B if bg is None else bg, G)
        inv = rope_inv_freq(dev, D, self.rope_theta)
        out = torch.empty((1, 1, self.hidden_size), dtype=hidden_states.dtype, device=dev)
        probs = torch.empty((1, H, 1, n + 1), dtype=hidden_states.dtype, device=dev) if output_attentions else None
        mask_ptr = 0
        if attention_mask is not None:
            attention_mask = additive_mask(attention_mask, hidden_states.dtype).reshape(-1).contiguous()
            mask_ptr = attention_mask.data_ptr()
        wq, vtk, vtv, wo = self.q_proj.weight, self.k_proj.VT.weight, self.v_proj.VT.weight, self.o_proj.weight
        x = hidden_states.reshape(-1)
        if not x.is_contiguous():
            x = x.contiguous()
        if probs is None:
            # through the dispatcher (torch.ops.palu.decode_step, palu_amd/ops.py): torch.compile / export see one node
            from .. import ops as _ops  # noqa: F401  (registers the ops on first use)
            o = torch.ops.palu.decode_step(x, wq, vtk, vtv, frag, wo, kbuf[0], vbuf[0], inv, ws, self._ws_cap, H, n, int(pos),
                                           attention_mask, bg is not None)
            cache.advance(li, 1)
            return o.view(1, 1, self.hidden_size), None
        step_fn = _lib.lib.palu_decode_step_f16 if bg is None else _lib.lib.palu_decode_step_sharedb_f16
        _lib.check(step_fn(
            x.data_ptr(), wq.data_ptr(), wq.stride(0), vtk.data_ptr(), vtk.stride(0), vtv.data_ptr(), vtv.stride(0),
            frag.data_ptr(), wo.data_ptr(), wo.stride(0),
            kbuf.data_ptr(), kbuf.stride(1), kbuf.stride(2), vbuf.data_ptr(), vbuf.stride(1), vbuf.stride(2),
            mask_ptr, inv.data_ptr(), out.data_ptr(),
            0 if probs is None else probs.data_ptr(), 0 if probs is None else probs.stride(1),
            ws.data_ptr(), self._ws_cap, H, G, D, self.hidden_size, self.group_rank_k, self.group_rank_v,
            n, int(pos), _lib.current_stream()), "palu_decode_step_f16")
        cache.advance(li, 1)
        return out, probs

    def _decode_biased(self, hidden_states, attention_mask, pos, cache, output_attentions):
        """config.attention_bias (kernel/palu_attention.py:142-145): the same HIP kernels, launched one by one, with
        q_proj.bias inside the qkv kernel (before the rotation) and o_proj.bias inside the last GEMV.  The latent
        projections have no bias (:33) and the decode branch never applies U's (:207-219)."""
        dev, dt = hidden_states.device, hidden_states.dtype
        li = self.layer_idx
        n = cache.get_seq_length(li)
        H, G, D, Rk, Rv = self.num_heads, self.num_groups, self.head_dim, self.group_rank_k, self.group_rank_v
        packed = isinstance(cache, QuantLatentCache)
        L = n + 1
        S = _lib.current_stream
        inv = rope_inv_freq(dev, D, self.rope_theta)
        frag = prepare_b(self.k_proj.B, G)
        x = hidden_states.reshape(-1).contiguous()
        wq, vtk, vtv, wo = self.q_proj.weight, self.k_proj.VT.weight, self.v_proj.VT.weight, self.o_proj.weight
        qb = 0 if self.q_proj.bias is None else self.q_proj.bias.data_ptr()
        ob = 0 if self.o_proj.bias is None else self.o_proj.bias.data_ptr()
        q = torch.empty(H * D, dtype=dt, device=dev)
        scores = torch.empty((H, (L + 15) // 8 * 8), dtype=dt, device=dev)
        ctx = torch.empty(H * Rv, dtype=dt, device=dev)
        pvws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=dev)
        out = torch.empty((1, 1, self.hidden_size), dtype=dt, device=dev)
        probs = torch.empty((1, H, 1, L), dtype=dt, device=dev) if output_attentions else None
        mask_ptr = 0
        if attention_mask is not None:
            attention_mask = additive_mask(attention_mask, dt).reshape(-1).contiguous()
            mask_ptr = attention_mask.data_ptr()
        if not packed:
            if cache.capacity(li) < L:
                cache.reserve(li, L + cache._headroom, torch.empty((1, G, 0, Rk), dtype=dt, device=dev),
                              torch.empty((1, G, 0, Rv), dtype=dt, device=dev))
            kbuf, vbuf = cache.buffers(li)
            _lib.check(_lib.lib.palu_decode_qkv_bias_f16(
                wq.data_ptr(), wq.stride(0), qb, vtk.data_ptr(), vtk.stride(0), vtv.data_ptr(), vtv.stride(0), x.data_ptr(),
                q.data_ptr(), kbuf.data_ptr(), kbuf.stride(1), kbuf.stride(2), vbuf.data_ptr(), vbuf.stride(1), vbuf.stride(2),
                inv.data_ptr(), H, D, self.hidden_size, G, Rk, Rv, int(pos), n, S()), "palu_decode_qkv_bias_f16")
            cache.advance(li, 1)
            nscr = _lib.lib.palu_abx_scratch_bytes(H, G, L, Rk)
            scr = torch.empty(nscr, dtype=torch.uint8, device=dev) if nscr else None
            _lib.check(_lib.lib.palu_abx_rope_ws_f16(q.data_ptr(), D, 1, frag.data_ptr(), kbuf.data_ptr(), kbuf.stride(1),
                                                     kbuf.stride(2), scores.data_ptr(), scores.stride(0), H, G, L, Rk, D,
                                                     inv.data_ptr(), 0, 0 if scr is None else scr.data_ptr(), S()),
                       "palu_abx_rope_ws_f16")
            _lib.check(_lib.lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), mask_ptr, vbuf.data_ptr(),
                                                    vbuf.stride(1), vbuf.stride(2), ctx.data_ptr(),
                                                    0 if probs is None else probs.data_ptr(),
                                                    0 if probs is None else probs.stride(1), pvws.data_ptr(), H, G, L, Rv,
                                                    math.sqrt(D), S()), "palu_softmax_pv_f16")
        else:
            knew = torch.empty((1, G, 1, Rk), dtype=dt, device=dev)
            vnew = torch.empty((1, G, 1, Rv), dtype=dt, device=dev)
            _lib.check(_lib.lib.palu_decode_qkv_bias_f16(
                wq.data_ptr(), wq.stride(0), qb, vtk.data_ptr(), vtk.stride(0), vtv.data_ptr(), vtv.stride(0), x.data_ptr(),
                q.data_ptr(), knew.data_ptr(), Rk, 0, vnew.data_ptr(), Rv, 0, inv.data_ptr(), H, D, self.hidden_size, G, Rk, Rv,
                int(pos), 0, S()), "palu_decode_qkv_bias_f16")
            cache.append_rows(knew, vnew, li)                     # quantise + pack the new rows in place
            st = cache.buffers(li)
            kc, km, vc, vm = st["kc"], st["km"], st["vc"], st["vm"]
            nscr = _lib.lib.palu_abx_scratch_bytes(H, G, L, Rk)
            scr = torch.empty(nscr, dtype=torch.uint8, device=dev) if nscr else None
            _lib.check(_lib.lib.palu_abx_rope_qg(q.data_ptr(), D, 1, frag.data_ptr(), kc.data_ptr(), kc.stride(1), kc.stride(2),
                                                 km.data_ptr(), km.stride(1), km.stride(2), scores.data_ptr(), scores.stride(0),
                                                 H, G, L, Rk, D, cache.n_bits, cache.group_size, inv.data_ptr(), 0,
                                                 0 if scr is None else scr.data_ptr(), S()), "palu_abx_rope_qg")
            _lib.check(_lib.lib.palu_softmax_pv_qg(scores.data_ptr(), scores.stride(0), mask_ptr, vc.data_ptr(), vc.stride(1),
                                                   vc.stride(2), vm.data_ptr(), vm.stride(1), vm.stride(2), ctx.data_ptr(),
                                                   0 if probs is None else probs.data_ptr(),
                                                   0 if probs is None else probs.stride(1), pvws.data_ptr(), H, G, L, Rv,
                                                   cache.n_bits, cache.group_size, math.sqrt(D), S()), "palu_softmax_pv_qg")
        _lib.check(_lib.lib.palu_gemv_bias_f16(wo.data_ptr(), wo.stride(0), ctx.data_ptr(), ob, out.data_ptr(),
                                               self.hidden_size, H * Rv, S()), "palu_gemv_bias_f16")
        return out, probs

    def _decode_fused_q(self, hidden_states, attention_mask, pos, cache: "QuantLatentCache", output_attentions):
        """q_len == 1 on a packed 3/4-bit cache: palu_decode_step_q."""
        if self.q_proj.bias is not None or self.o_proj.bias is not None:
            return self._decode_biased(hidden_states, attention_mask, pos, cache, output_attentions)
        dev = hidden_states.device
        li = self.layer_idx
        n = cache.get_seq_length(li)
        H, G, D = self.num_heads, self.num_groups, self.head_dim
        if cache.capacity(li) < n + 1:
            cache.reserve(li, n + 1 + cache._headroom, G, self.group_rank_k, self.group_rank_v, dev)
        st = cache.buffers(li)
        cap = st["kc"].shape[2]
        ws = self._workspace(dev, cap + 8)
        frag = prepare_b(self.k_proj.B, G)
        inv = rope_inv_freq(dev, D, self.rope_theta)
        out = torch.empty((1, 1, self.hidden_size), dtype=hidden_states.dtype, device=dev)
        probs = torch.empty((1, H, 1, n + 1), dtype=hidden_states.dtype, device=dev) if output_attentions else None
        mask_ptr = 0
        if attention_mask is not None:
            attention_mask = additive_mask(attention_mask, hidden_states.dtype).reshape(-1).contiguous()
            mask_ptr = attention_mask.data_ptr()
        wq, vtk, vtv, wo = self.q_proj.weight, self.k_proj.VT.weight, self.v_proj.VT.weight, self.o_proj.weight
        x = hidden_states.reshape(-1).contiguous()
        kc, km, vc, vm = st["kc"], st["km"], st["vc"], st["vm"]
        if probs is None:
            from .. import ops as _ops  # noqa: F401
            o = torch.ops.palu.decode_step_q(x, wq, vtk, vtv, frag, wo, kc[0], km[0], vc[0], vm[0], inv, ws, self._ws_cap, H,
                                             self.group_rank_k, self.group_rank_v, cache.n_bits, n, int(pos), attention_mask,
                                             cache.group_size)
            cache.advance(li, 1)
            return o.view(1, 1, self.hidden_size), None
        _lib.check(_lib.lib.palu_decode_step_qg(
            x.data_ptr(), wq.data_ptr(), wq.stride(0), vtk.data_ptr(), vtk.stride(0), vtv.data_ptr(), vtv.stride(0),
            frag.data_ptr(), wo.data_ptr(), wo.stride(0),
            kc.data_ptr(), kc.stride(1), kc.stride(2), km.data_ptr(), km.stride(1), km.stride(2),
            vc.data_ptr(), vc.stride(1), vc.stride(2), vm.data_ptr(), vm.stride(1), vm.stride(2),
            mask_ptr, inv.data_ptr(), out.data_ptr(),
            0 if probs is None else probs.data_ptr(), 0 if probs is None else probs.stride(1),
            ws.data_ptr(), self._ws_cap, H, G, D, self.hidden_size, self.group_rank_k, self.group_rank_v,
            cache.n_bits, cache.group_size, n, int(pos), _lib.current_stream()), "palu_decode_step_qg")
        cache.advance(li, 1)
        return out, probs

    def _project_into_cache(self, hidden_states, cache: "LatentCache"):
        """latents = X.VT^T for the whole prompt via palu_lowrank_project_gemm, appended in place (:167-168,193)."""
        li, G = self.layer_idx, self.num_groups
        q_len = hidden_states.shape[1]
        n = cache.get_seq_length(li)
        dev, dt = hidden_states.device, hidden_states.dtype
        like_k = torch.empty((1, G, 0, self.group_rank_k), dtype=dt, device=dev)
        like_v = torch.empty((1, G, 0, self.group_rank_v), dtype=dt, device=dev)
        cache.reserve(li, n + q_len + cache._headroom, like_k, like_v)
        kbuf, vbuf = cache.buffers(li)
        x = hidden_states.reshape(q_len, -1)
        if x.stride(1) != 1 or x.stride(0) % 8:
            x = x.contiguous()
        for w, buf, R in ((self.k_proj.VT.weight, kbuf, self.group_rank_k), (self.v_proj.VT.weight, vbuf, self.group_rank_v)):
            _lib.check(_lib.lib.palu_lowrank_project_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                                          buf.data_ptr(), buf.stride(1), buf.stride(2), q_len,
                                                          w.shape[0], w.shape[1], R, n, _lib.current_stream()),
                       "palu_lowrank_project_gemm")
        cache.advance(li, q_len)
        return kbuf[:, :, :n + q_len], vbuf[:, :, :n + q_len]

    def _project_into_packed_cache(self, hs, cache) -> bool:
        """The chunk's latent rows straight into a packed cache: the projection GEMM with the quantise + pack epilogue
        (palu_lowrank_project_gemm_q) -- no fp16 latents in memory.  False when the fused tile does not take the shape (short
        chunks, group ranks that do not tile, per-column-group metas): the caller projects, then appends."""
        if getattr(cache, "group_size", 0) or self.k_proj.VT.bias is not None or self.v_proj.VT.bias is not None:
            return False
        x = hs.reshape(-1, hs.shape[-1])
        t = x.shape[0]
        if x.stride(1) != 1 or x.dtype != torch.float16:
            return False
        G, Rk, Rv = self.num_groups, self.group_rank_k, self.group_rank_v
        lib = _lib.lib
        for w, R in ((self.k_proj.VT.weight, Rk), (self.v_proj.VT.weight, Rv)):
            if (w.dtype != torch.float16 or w.stride(1) != 1
                    or not lib.palu_lowrank_project_gemm_q_supported(t, w.shape[0], w.shape[1], R, cache.n_bits)):
                return False
        li = self.layer_idx
        cache._ensure_layer(li)
        n = cache.get_seq_length(li)
        cache.reserve(li, n + t + cache._headroom, G, Rk, Rv, x.device)
        st = cache.buffers(li)
        for w, codes, meta, R in ((self.k_proj.VT.weight, st["kc"], st["km"], Rk), (self.v_proj.VT.weight, st["vc"], st["vm"], Rv)):
            _lib.check(lib.palu_lowrank_project_gemm_q(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), codes.data_ptr(),
                                                       codes.stride(1), codes.stride(2), meta.data_ptr(), meta.stride(1),
                                                       meta.stride(2), t, w.shape[0], w.shape[1], R, n, cache.n_bits,
                                                       _lib.current_stream()), "palu_lowrank_project_gemm_q")
        cache.advance(li, t)
        return True

    def _mask_is_causal(self, attention_mask, q_len, past):
        """True iff the additive mask is the standard causal one (0 on/below the diagonal shifted by `past`,
        <= -1e4 above it) -- the only mask the flash prefill kernel understands."""
        m = attention_mask.reshape(q_len, past + q_len)
        j = torch.arange(past + q_len, device=m.device).unsqueeze(0)
        i = torch.arange(q_len, device=m.device).unsqueeze(1) + past
        above = j > i
        return bool(((m <= -1e4) == above).all()) and bool((m.masked_fill(above, 0) == 0).all())

    # bytes of transient prefill workspace above which the prompt pass runs in query chunks x latent groups.  Below it: one
    # kernel launch for all heads and queries, the fastest form (a 64k-token prompt at the config-2 ranks needs 2.9 GiB of
    # transients that way -- 1 % of this GPU's HBM -- and runs 1.9x faster than in 128-workgroup slices).  Both are class
    # attributes: a deployment that is short of memory lowers the budget.
    PREFILL_WORKSPACE_BUDGET = 6 << 30
    PREFILL_QUERY_CHUNK = 8192
    # kv PANELS: with PREFILL_PANEL_ROWS > 0 the prompt pass never holds more than one panel of reconstructed keys / transposed
    # values: query chunks of PREFILL_PANEL_QUERY rows x PREFILL_PANEL_GROUPS latent groups x panels of PREFILL_PANEL_ROWS kv
    # positions, the online-softmax state carried between the panel launches in fp32 (palu_prefill_attn_panel_f16).  Transient
    # memory is independent of the prompt length -- per (chunk of t queries, ng groups, panel of C rows), in bytes:
    #   2 t H (D + Rv) [q, context rows]  +  4 ng gs t (Rv + 8) [state]  +  2 ng C (gs D + Rv) [K~, V^T]
    #   (+ 2 ng C (Rk + Rv) de-quantised rows of a packed cache)  ~ 55 MiB at the defaults and the config-2 ranks --
    # at the price of small launches (t / 128 x ng x gs workgroups) and of rebuilding K~ per (chunk, panel): a mode for
    # memory-starved deployments, off by default.
    PREFILL_PANEL_ROWS = 0
    PREFILL_PANEL_QUERY = 512
    PREFILL_PANEL_GROUPS = 4

    # The LATENT form of the prompt pass (round 6, csrc/prefill_lat.hip): the flash kernel rebuilds
    # K~ = RoPE(X_k . B_h) per 64-position kv tile itself and reads the latent values from the cache rows -- no [H, kv, D] key
    # workspace, no transposed value copy: the only transients are the rotated queries and context rows of ONE query chunk
    # (2 t H (D + Rv) bytes: 96 MiB at PREFILL_LATENT_QUERY_CHUNK = 3072 and the config-2 ranks), whatever the prompt length.  It costs
    # the rebuild's +25 % of matrix work (64k tokens: ~105 ms against ~82 ms for the one-launch workspace form with its 2.4 GiB of
    # transients), so it is selected when the workspace form would exceed PREFILL_LATENT_ABOVE bytes of transients; 0 = always,
    # None = never.  fp16, packed 4-bit and (rank_k / G = 128) packed 3-bit caches at head_dim 128 -- the shapes
    # palu_prefill_attn_lat_supported_bits lists.  PREFILL_LATENT_PROJECT_ROWS: the latent projections (straight into cache rows:
    # no transient) run ahead of the attention in blocks of that many rows.
    PREFILL_LATENT_ABOVE = 256 << 20
    PREFILL_LATENT_QUERY_CHUNK = 3072
    PREFILL_LATENT_PROJECT_ROWS = 16384

    def _bt_fragments(self, permuted: bool = False):
        """B^T [H, D, Rk] contiguous (row d of head h = the weights that rebuild K[., d]): cached like the abx fragments.
        permuted: the columns of every group of 8 in the order 0 4 1 5 2 6 3 7 -- the order the nibbles of a dword of 4-bit codes come
        out in the packed form of the latent prefill kernel."""
        b = self.k_proj.B
        key = (b._version, b.data_ptr(), bool(permuted))
        hit = getattr(self, "_bt_cache", None)
        if hit is not None and key in hit:
            return hit[key]
        bt = b.detach().transpose(1, 2)
        if permuted:
            H, Dd, R = bt.shape
            idx = torch.tensor([0, 4, 1, 5, 2, 6, 3, 7], device=bt.device)
            bt = bt.reshape(H, Dd, R // 8, 8)[..., idx].reshape(H, Dd, R)
        bt = bt.contiguous()
        if hit is None or next(iter(hit))[:2] != key[:2]:
            hit = {}
        hit[key] = bt
        self._bt_cache = hit
        return bt

    def _prefill_latent(self, hidden_states, pos, cache, causal: bool):
        """Prompt branch (:196-257) in query chunks on palu_prefill_attn_lat_f16: projections of the chunk (latents straight
        into the cache rows), rotation of q, attention over the cache rows 0 .. kv - 1, o_proj of the chunk."""
        H, G, D = self.num_heads, self.num_groups, self.head_dim
        Rk, Rv = self.group_rank_k, self.group_rank_v
        q_len = hidden_states.shape[1]
        li = self.layer_idx
        past = cache.get_seq_length(li)
        dev, dt = hidden_states.device, hidden_states.dtype
        kv_all = past + q_len
        qc = max(128, int(self.PREFILL_LATENT_QUERY_CHUNK))
        inv = rope_inv_freq(dev, D, self.rope_theta)
        cs = rope_cs_table(dev, D, self.rope_theta, kv_all)
        packed = not isinstance(cache, LatentCache)
        bt = self._bt_fragments(permuted=packed)
        stream = _lib.current_stream()
        p0 = int(pos.reshape(-1)[0])

        def project(r0, r1):
            """Latent rows r0 .. r1 - 1 of this pass into the cache; False if that block size would need a temporary (retry smaller)."""
            hs = hidden_states[:, r0:r1]
            if not packed:
                self._project_into_cache(hs, cache)                                       # latents straight into the rows
            elif not self._project_into_packed_cache(hs, cache):
                if r1 - r0 > qc:
                    return False
                kh = self.k_proj.project_to_latent(hs).view(1, r1 - r0, G, Rk).transpose(1, 2)
                vh = self.v_proj.project_to_latent(hs).view(1, r1 - r0, G, Rv).transpose(1, 2)
                cache.append_rows(kh, vh, li)                                             # quantise + pack this chunk's rows
            return True
        # The projections write straight into cache rows and need no transient, so they run AHEAD of the attention in blocks of
        # PREFILL_LATENT_PROJECT_ROWS rows (a 3072-row block of the rank-128 projection is 96 workgroups on 256 CUs; cache rows
        # beyond a chunk's kv are never read).  No mask: every query attends every key of the pass -- all rows first.
        pb = q_len if not causal else max(qc, int(self.PREFILL_LATENT_PROJECT_ROWS))
        done = 0
        out = None
        for c0 in range(0, q_len, qc):
            c1 = min(q_len, c0 + qc)
            t = c1 - c0
            hs = hidden_states[:, c0:c1]
            q = self.q_proj(hs).view(t, H, D).transpose(0, 1)                             # [H,t,D] view of [t, H*D]
            while done < (c1 if causal else q_len):
                blk = min(q_len, done + pb)
                if not project(done, blk):
                    pb = qc                                                               # (no fused quantise tile: chunk by chunk)
                    continue
                done = blk
            kv = past + c1 if causal else kv_all
            _lib.check(_lib.lib.palu_rope_f16(q.data_ptr(), q.stride(0), q.stride(1), H, t, D, p0 + c0, inv.data_ptr(), stream),
                       "palu_rope_f16")
            ctx = torch.empty((t, H * Rv), dtype=dt, device=dev)
            if not packed:
                kbuf, vbuf = cache.buffers(li)
                _lib.check(_lib.lib.palu_prefill_attn_lat_f16(q.data_ptr(), q.stride(0), q.stride(1), kbuf.data_ptr(), kbuf.stride(1),
                                                              kbuf.stride(2), vbuf.data_ptr(), vbuf.stride(1), vbuf.stride(2),
                                                              bt.data_ptr(), cs.data_ptr(), ctx.data_ptr(), ctx.stride(0), H, G, D, t, kv,
                                                              Rk, Rv, past + c0, 1 if causal else 0, 1.0 / math.sqrt(D), stream),
                           "palu_prefill_attn_lat_f16")
            else:
                st = cache.buffers(li)
                kc, km, vc, vm = st["kc"], st["km"], st["vc"], st["vm"]
                _lib.check(_lib.lib.palu_prefill_attn_lat_q(q.data_ptr(), q.stride(0), q.stride(1), kc.data_ptr(), kc.stride(1),
                                                            kc.stride(2), km.data_ptr(), km.stride(1), km.stride(2), vc.data_ptr(),
                                                            vc.stride(1), vc.stride(2), vm.data_ptr(), vm.stride(1), vm.stride(2),
                                                            bt.data_ptr(), cs.data_ptr(), ctx.data_ptr(), ctx.stride(0), H, G, D, t, kv,
                                                            Rk, Rv, cache.n_bits, past + c0, 1 if causal else 0, 1.0 / math.sqrt(D),
                                                            stream), "palu_prefill_attn_lat_q")
            if out is None:
                out = torch.empty((q_len, self.o_proj.weight.shape[0]), dtype=dt, device=dev)
            if self.o_proj.bias is None:
                torch.matmul(ctx, self.o_proj.weight.t(), out=out[c0:c1])                 # (no chunk-sized o_proj output buffer)
            else:
                out[c0:c1].copy_(self.o_proj(ctx))
            del ctx, q
        return out.view(1, q_len, -1)

    def _prefill_flash(self, hidden_states, pos, cache, causal: bool):
        """Prompt branch (:196-257) on the flash-style HIP kernel: scores are never materialised.

        q and the latents are projected (latents straight into the cache rows, or quantised + packed chunk by chunk), the
        keys of ONE latent group at a time are rebuilt as K~ = RoPE(X_k . B) [gs, kv, D] and its latent values handed to the
        kernel transposed ([Rv, kv], zero-padded to 64); P.V stays in the latent space.  Long prompts run in query chunks:
        every transient -- K~ and V^T of one group, the context rows of one chunk, for packed caches the dequantised rows
        of one group -- is O(kv * gs * D) or O(chunk * H * Rv), never O(kv * H * D); rebuilding a group's K~ per chunk
        costs 1 MFLOP per cached position and chunk, a few percent of the attention itself."""
        H, G, D, gs = self.num_heads, self.num_groups, self.head_dim, self.group_size
        Rk, Rv = self.group_rank_k, self.group_rank_v
        q_len = hidden_states.shape[1]
        li = self.layer_idx
        past = cache.get_seq_length(li)
        dev, dt = hidden_states.device, hidden_states.dtype
        kv_all = past + q_len
        panel_rows = int(self.PREFILL_PANEL_ROWS)
        # what the one-launch form would allocate: K~ for every head, V^T of every group, the context rows
        one_launch_bytes = 2 * (H * kv_all * D + G * Rv * (kv_all + 63) + q_len * H * Rv)
        packed = not isinstance(cache, LatentCache)
        if packed:
            one_launch_bytes += 2 * kv_all * G * (Rk + Rv)          # the dequantised rows
        lat_above = self.PREFILL_LATENT_ABOVE
        pos_flat = pos.reshape(-1)
        nb = int(getattr(cache, "n_bits", 0)) if packed else 16
        lat_packed_ok = packed and not getattr(cache, "group_size", 0)
        if (lat_above is not None and (not packed or lat_packed_ok) and panel_rows == 0 and one_launch_bytes > lat_above and dt == torch.float16
                and _lib.lib.palu_prefill_attn_lat_supported_bits(H, G, D, Rk, Rv, nb)
                and bool((pos_flat == torch.arange(int(pos_flat[0]), int(pos_flat[0]) + q_len, device=pos_flat.device)).all())
                and self.q_proj.weight.dtype == dt):
            return self._prefill_latent(hidden_states, pos, cache, causal)
        grouped = one_launch_bytes > self.PREFILL_WORKSPACE_BUDGET or panel_rows > 0
        qc = self.PREFILL_QUERY_CHUNK if grouped else q_len
        if panel_rows > 0:
            panel_rows = max(64, panel_rows // 64 * 64)
            qc = max(128, int(self.PREFILL_PANEL_QUERY))
        inv = rope_inv_freq(dev, D, self.rope_theta)
        stream = _lib.current_stream()
        pos = pos.reshape(-1)
        p0 = int(pos[0])
        contiguous_pos = bool((pos == torch.arange(p0, p0 + q_len, device=pos.device)).all())
        u_ok = (self.n_rep == 1 and Rk % 64 == 0
                and all(u.weight.dtype == dt and u.weight.is_contiguous() and u.bias is None for u in self.k_proj.U_list))
        out = None

        def project_latents(hs):
            t_ = hs.shape[1]
            if not packed:
                self._project_into_cache(hs, cache)                                       # latents straight into the rows
            elif not self._project_into_packed_cache(hs, cache):
                # packed 3/4-bit cache: quantise + pack this chunk's rows; attention runs over the de-quantised values
                # (the accuracy path's fake-quant semantics, svd_linear.py:84-90,124-139)
                kh = self.k_proj.project_to_latent(hs).view(1, t_, G, Rk).transpose(1, 2)
                vh = self.v_proj.project_to_latent(hs).view(1, t_, G, Rv).transpose(1, 2)
                cache.append_rows(kh, vh, li)
                del kh, vh

        # Without a causal mask (the reference's no-mask prompt semantics, palu_attention.py:229-234 with attention_mask None)
        # every query attends EVERY key of the pass, also those of later chunks: all latents go into the cache before the
        # first chunk attends, and every chunk runs over kv_all.  (Causal chunks only need the keys up to their own end.)
        all_first = (not causal) and q_len > qc
        if all_first:
            for c0 in range(0, q_len, qc):
                project_latents(hidden_states[:, c0:min(q_len, c0 + qc)])
        for c0 in range(0, q_len, qc):
            c1 = min(q_len, c0 + qc)
            t = c1 - c0
            hs = hidden_states[:, c0:c1]
            q = self.q_proj(hs).view(t, H, D).transpose(0, 1)                             # [H,t,D] view of [t, H*D]
            if not all_first:
                project_latents(hs)
            kv = past + c1 if causal else kv_all
            if contiguous_pos and q.stride(2) == 1:
                _lib.check(_lib.lib.palu_rope_f16(q.data_ptr(), q.stride(0), q.stride(1), H, t, D, p0 + c0, inv.data_ptr(),
                                                  stream), "palu_rope_f16")
            else:
                cos, sin = self._rope_tables(pos[c0:c1], dt)
                q = (q * cos.view(1, t, D) + _rotate_half(q) * sin.view(1, t, D)).contiguous()
            kv_pad = (kv + 63) // 64 * 64
            ctx = torch.empty((t, H * Rv), dtype=dt, device=dev)
            if not grouped:
                groups = [list(range(G))]
            elif panel_rows > 0:
                ngl = max(1, min(G, int(self.PREFILL_PANEL_GROUPS)))
                groups = [list(range(g, min(G, g + ngl))) for g in range(0, G, ngl)]
            else:
                # enough latent groups per launch for >= 512 workgroups (2 per CU): one workgroup = 128 queries of one head
                per_group = gs * ((t + 127) // 128)
                ngl = max(1, min(G, (512 + per_group - 1) // per_group))
                groups = [list(range(g, min(G, g + ngl))) for g in range(0, G, ngl)]
            for gl in groups:
                ng = len(gl)
                g0 = gl[0]
                if panel_rows > 0:
                    self._prefill_panels(cache, q, ctx, g0, ng, kv, past + c0, panel_rows, causal, u_ok, inv, stream)
                    continue
                xk, xv = self._latent_rows(cache, g0, ng, kv)                             # [ng, kv, Rk], [ng, kv, Rv] fp16
                keys = torch.empty((ng * gs, kv, D), dtype=dt, device=dev)
                if u_ok:
                    # per group the reconstruct GEMM X_g . U_g^T (:67-77, :199-201) lands head-major, then the rotation
                    # runs in place
                    for j, g in enumerate(gl):
                        u = self.k_proj.U_list[g]
                        xg = xk[j]
                        _lib.check(_lib.lib.palu_lowrank_project_gemm(xg.data_ptr(), xg.stride(0), u.weight.data_ptr(),
                                                                      u.weight.stride(0), keys[j * gs].data_ptr(), keys.stride(0),
                                                                      keys.stride(1), kv, gs * D, Rk, D, 0, stream),
                                   "palu_lowrank_project_gemm")
                    _lib.check(_lib.lib.palu_rope_f16(keys.data_ptr(), keys.stride(0), keys.stride(1), ng * gs, kv, D, 0,
                                                      inv.data_ptr(), stream), "palu_rope_f16")
                else:
                    kc, ks = self._rope_tables(torch.arange(kv, device=dev), dt)
                    b = self.k_proj.B.view(G, gs, Rk, D)[g0:g0 + ng]
                    kk = torch.matmul(xk.unsqueeze(1), b).view(ng * gs, kv, D)            # K = X_k . B  (:199-201)
                    keys.copy_(kk * kc.view(1, kv, D) + _rotate_half(kk) * ks.view(1, kv, D))
                    del kk
                vt = torch.zeros((ng, Rv, kv_pad), dtype=dt, device=dev)
                vt[:, :, :kv].copy_(xv.transpose(1, 2))
                del xk, xv
                qg = q[g0 * gs:(g0 + ng) * gs]
                cg = ctx[:, g0 * gs * Rv:]
                _lib.check(_lib.lib.palu_prefill_attn_f16(qg.data_ptr(), qg.stride(0), qg.stride(1), keys.data_ptr(),
                                                          keys.stride(0), keys.stride(1), vt.data_ptr(), vt.stride(0),
                                                          vt.stride(1), cg.data_ptr(), ctx.stride(0), ng * gs, ng, D, t, kv,
                                                          Rv, past + c0, 1 if causal else 0, 1.0 / math.sqrt(D), stream),
                           "palu_prefill_attn_f16")
                del keys, vt
            o = self.o_proj(ctx)
            del ctx, q
            if c0 == 0 and c1 == q_len:
                out = o
            else:
                if out is None:
                    out = torch.empty((q_len, o.shape[-1]), dtype=o.dtype, device=dev)
                out[c0:c1].copy_(o)
            del o
        return out.view(1, q_len, -1)

    def _prefill_panels(self, cache, q, ctx, g0, ng, kv, qpos0, C, causal, u_ok, inv, stream):
        """The latent groups g0..g0+ng-1 of one query chunk (q [H, t, D] rotated, first row at absolute position qpos0) over
        the kv positions 0..kv-1 in panels of C rows: per panel K~ = RoPE(X_k . B) and V^T are built for the panel only and
        palu_prefill_attn_panel_f16 folds it into the carried online-softmax state; the last panel writes the context rows."""
        H, D, gs = self.num_heads, self.head_dim, self.group_size
        Rk, Rv = self.group_rank_k, self.group_rank_v
        dev, dt = q.device, q.dtype
        t = q.shape[1]
        nh = ng * gs
        st_o = torch.empty(_lib.lib.palu_prefill_state_bytes(nh, t, Rv, 0) // 4, dtype=torch.float32, device=dev)
        st_ml = torch.empty(_lib.lib.palu_prefill_state_bytes(nh, t, Rv, 1) // 4, dtype=torch.float32, device=dev)
        qg = q[g0 * gs:(g0 + ng) * gs]
        cg = ctx[:, g0 * gs * Rv:]
        starts = list(range(0, kv, C))
        cmax = min(C, kv)
        keys = torch.empty((nh, cmax, D), dtype=dt, device=dev)
        vt = torch.empty((ng, Rv, (cmax + 63) // 64 * 64), dtype=dt, device=dev)
        for pi, k0 in enumerate(starts):
            n = min(kv, k0 + C) - k0
            xk, xv = self._latent_rows(cache, g0, ng, k0 + n, k0)                          # [ng, n, Rk], [ng, n, Rv] fp16
            kp = keys[:, :n]
            if u_ok:
                for j in range(ng):
                    u = self.k_proj.U_list[g0 + j]
                    xg = xk[j]
                    _lib.check(_lib.lib.palu_lowrank_project_gemm(xg.data_ptr(), xg.stride(0), u.weight.data_ptr(),
                                                                  u.weight.stride(0), kp[j * gs].data_ptr(), kp.stride(0),
                                                                  kp.stride(1), n, gs * D, Rk, D, 0, stream),
                               "palu_lowrank_project_gemm")
                _lib.check(_lib.lib.palu_rope_f16(kp.data_ptr(), kp.stride(0), kp.stride(1), nh, n, D, k0, inv.data_ptr(), stream),
                           "palu_rope_f16")
            else:
                kc, ks = self._rope_tables(torch.arange(k0, k0 + n, device=dev), dt)
                b = self.k_proj.B.view(self.num_groups, gs, Rk, D)[g0:g0 + ng]
                kk = torch.matmul(xk.unsqueeze(1), b).view(nh, n, D)
                kp.copy_(kk * kc.view(1, n, D) + _rotate_half(kk) * ks.view(1, n, D))
                del kk
            n_pad = (n + 63) // 64 * 64
            vp = vt[:, :, :n_pad]
            if n_pad != n:
                vp[:, :, n:].zero_()
            vp[:, :, :n].copy_(xv.transpose(1, 2))
            del xk, xv
            _lib.check(_lib.lib.palu_prefill_attn_panel_f16(qg.data_ptr(), qg.stride(0), qg.stride(1), kp.data_ptr(), kp.stride(0),
                                                            kp.stride(1), vp.data_ptr(), vp.stride(0), vp.stride(1),
                                                            cg.data_ptr(), ctx.stride(0), nh, ng, D, t, n, Rv, qpos0 - k0,
                                                            1 if causal else 0, 1.0 / math.sqrt(D), st_o.data_ptr(),
                                                            st_ml.data_ptr(), 1 if pi == 0 else 0,
                                                            1 if pi == len(starts) - 1 else 0, stream),
                       "palu_prefill_attn_panel_f16")

    def _latent_rows(self, cache, g0: int, ng: int, kv: int, k0: int = 0):
        """fp16 latent rows [ng, kv - k0, R] (positions k0..kv-1) of groups g0..g0+ng-1: views of an fp16 cache, de-quantised
        copies (of these groups and rows only) of a packed one."""
        li = self.layer_idx
        if isinstance(cache, LatentCache):
            kbuf, vbuf = cache.buffers(li)
            return kbuf[0, g0:g0 + ng, k0:kv], vbuf[0, g0:g0 + ng, k0:kv]
        from .quant import unpack_dequant
        st = cache.buffers(li)
        gsz = cache.group_size
        kv = kv - k0

        def deq(codes, meta, R):
            c, m = codes[0, g0:g0 + ng, k0:k0 + kv], meta[0, g0:g0 + ng, k0:k0 + kv]
            if not gsz:
                return unpack_dequant(c.contiguous(), m.contiguous(), cache.n_bits, R)
            nb = gsz * cache.n_bits // 8
            x = unpack_dequant(c.reshape(ng, kv, R // gsz, nb).contiguous(), m.reshape(ng, kv, R // gsz, 2).contiguous(),
                               cache.n_bits, gsz)
            return x.reshape(ng, kv, R)
        return deq(st["kc"], st["km"], self.group_rank_k), deq(st["vc"], st["vm"], self.group_rank_v)

    @torch.no_grad()
    def fuse_hadamard(self):
        """Rotate the latent spaces by Hadamard matrices offline (the `--lt_hadamard` option:
        svd_linear.py:156-168 `fused_hadamard_matrix`, applied here to the kernel-flavoured module):
        VT_g <- (had(VT_g^T))^T, U_g <- had(U_g); consequently B[h] <- had(B[h]^T)^T and, with U_v folded into
        o_proj, W_o'[:, h-block] <- had(W_o'[:, h-block]).  Outputs are unchanged up to rounding; the cached
        latents are born rotated (flatter rows quantise better) and decode needs no online transform."""
        from .hadamard_utils import apply_hadamard, fuse_hadamard_into_weights
        for proj in (self.k_proj, self.v_proj):
            fuse_hadamard_into_weights(proj.VT.weight.data, [u.weight.data for u in proj.U_list])
        if hasattr(self.k_proj, "B"):
            b = self.k_proj.B.data
            self.k_proj.B = nn.Parameter(apply_hadamard(b.transpose(1, 2).contiguous()).transpose(1, 2).contiguous())
        if self.o_proj.in_features == self.fused_hidden_dim_o:
            w = self.o_proj.weight.data
            Rv = self.group_rank_v
            self.o_proj.weight.data = apply_hadamard(w.reshape(w.shape[0], self.num_heads, Rv).contiguous()).reshape(w.shape)
        return self

    def prepare_decode(self):
        """Build what the HIP decode step derives from the weights -- the MFMA fragments of B and the "do the heads of a
        group share B" decision (one device comparison = one host sync) -- NOW, i.e. outside any stream capture and off
        the first token's path.  Idempotent and cached per weight tensor; call it after the module sits on its GPU and
        again after its weights changed (`fuse_hadamard` and `load_state_dict` invalidate the cache themselves).  Without
        it the first decode step does the same lazily (which raises inside a hipGraph capture: warm up or call this first)."""
        if hasattr(self.k_proj, "B") and self.k_proj.B.is_cuda:
            with _lib.on_device(self.k_proj.B):
                bg = shared_b(self.k_proj.B, self.num_groups)
                prepare_b(self.k_proj.B if bg is None else bg, self.num_groups)
                rope_inv_freq(self.k_proj.B.device, self.head_dim, self.rope_theta)
        return self

    def _hip_step_shapes_ok(self, cache) -> bool:
        """What palu_decode_step_f16 / _q accept (everything else takes the general path BEFORE anything is written to
        the cache): gs in {1,2,3,4,8} query heads per latent group; fp16 -- ranks multiples of 8; packed cache --
        Rk % 32 == 0 at 3 bit / Rk % 8 == 0 at 4 bit (palu_abx_rope_q: fast kernels for (4, 32|64|128) and (3, 128), the
        chunked one for every other rank, e.g. the 96 / 160 / 224 / 256 of the rank search) and Rv % 32 == 0
        (palu_softmax_pv_q)."""
        gs, Rk, Rv = self.group_size, self.group_rank_k, self.group_rank_v
        if isinstance(cache, QuantLatentCache):
            bits = cache.n_bits
            gsz = cache.group_size
            return (gs in (1, 2, 3, 4, 8) and Rv % 32 == 0 and Rv // 16 <= 256
                    and ((bits == 4 and Rk % 8 == 0) or (bits == 3 and Rk % 32 == 0))
                    and (gsz == 0 or (Rk % gsz == 0 and Rv % gsz == 0))
                    and _q_scratch_ok(self, Rk) and self.num_groups * Rv <= 16384)
        return gs in (1, 2, 3, 4, 8) and Rk % 8 == 0 and Rv % 8 == 0 and Rv // 8 <= 256

    # -- forward -----------------------------------------------------------------------------
    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_value=None,
                output_attentions: bool = False, golden_kernel: bool = False, **kwargs
                ) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[object]]:
        if hidden_states.is_cuda:
            # the C ABI launches on the process's current device: make it the tensors' GPU for the whole call
            with _lib.on_device(hidden_states):
                return self._forward(hidden_states, attention_mask, position_ids, past_key_value, output_attentions,
                                     golden_kernel, **kwargs)
        return self._forward(hidden_states, attention_mask, position_ids, past_key_value, output_attentions,
                             golden_kernel, **kwargs)

    def _forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                 output_attentions=False, golden_kernel=False, **kwargs):
        if "padding_mask" in kwargs:
            warnings.warn("Passing `padding_mask` is deprecated; use `attention_mask` instead.")
        bsz, q_len, _ = hidden_states.size()
        H, G, D = self.num_heads, self.num_groups, self.head_dim
        if past_key_value is not None and self.layer_idx is None:
            raise ValueError(
                f"The cache structure has changed since version v4.36. If you are using {self.__class__.__name__} "
                "for auto-regressive decoding with k/v caching, please make sure to initialize the attention class "
                "with a layer index.")
        past = 0 if past_key_value is None else past_key_value.get_usable_length(q_len, self.layer_idx)
        kv_seq_len = q_len + past
        if attention_mask is not None and attention_mask.dtype == torch.bool:
            attention_mask = additive_mask(attention_mask, hidden_states.dtype)      # True = attend -> 0 / -max
        if attention_mask is not None and attention_mask.size() != (bsz, 1, q_len, kv_seq_len):
            raise ValueError(
                f"Attention mask should be of size {(bsz, 1, q_len, kv_seq_len)}, but is {attention_mask.size()}")

        fused_o = self.o_proj.in_features == self.fused_hidden_dim_o
        hip_step_ok = self.head_dim == 128 and self._hip_step_shapes_ok(past_key_value)      # (biases: _decode_biased)
        if (q_len == 1 and bsz == 1 and isinstance(past_key_value, (LatentCache, QuantLatentCache)) and fused_o
                and hip_step_ok and hidden_states.is_cuda and hidden_states.dtype == torch.float16
                and hasattr(self.k_proj, "B")):
            pos = kv_seq_len - 1 if position_ids is None else int(position_ids.reshape(-1)[-1])
            step = self._decode_fused_q if isinstance(past_key_value, QuantLatentCache) else self._decode_fused
            out, probs = step(hidden_states, attention_mask, pos, past_key_value, output_attentions)
            return out, probs, past_key_value

        # ---- prompt pass on the flash-style HIP kernel (no [q, kv] score matrix) -----------------
        if (q_len > 1 and bsz == 1 and isinstance(past_key_value, (LatentCache, QuantLatentCache)) and fused_o
                and not output_attentions
                and hidden_states.is_cuda and hidden_states.dtype == torch.float16 and self.head_dim == 128
                and self.group_rank_v % 32 == 0 and self.k_proj.VT.bias is None and hasattr(self.k_proj, "B")
                and self.o_proj.in_features == self.fused_hidden_dim_o):
            is_causal = kwargs.get("is_causal", None)
            if attention_mask is None:
                causal = bool(is_causal)            # the reference applies no mask at all when none is passed (:229)
                ok = True
            else:
                ok = bool(is_causal) if is_causal is not None else self._mask_is_causal(attention_mask, q_len, past)
                causal = True
            if ok:
                if position_ids is None:
                    position_ids = torch.arange(past, kv_seq_len, device=hidden_states.device).unsqueeze(0)
                pos = position_ids.to(hidden_states.device).reshape(-1, q_len)
                # the flash path rotates key row j at position j: only valid for the standard ids past..past+q_len-1
                # (offset / left-padded ids take the general path, which rotates the new keys with position_ids)
                std = torch.arange(past, kv_seq_len, device=pos.device)
                if pos.shape[0] == 1 and bool((pos[0] == std).all()):
                    out = self._prefill_flash(hidden_states, pos, past_key_value, causal)
                    return out, None, past_key_value

        # ---- general path (no_fusion, foreign cache objects, arbitrary masks, output_attentions): torch composition
        query_states = self.q_proj(hidden_states).view(bsz, q_len, H, D).transpose(1, 2)
        if (isinstance(past_key_value, LatentCache) and bsz == 1 and hidden_states.is_cuda
                and hidden_states.dtype == torch.float16 and self.k_proj.VT.bias is None):
            # prefill down-projection on the MFMA GEMM, written straight into the latent cache rows
            key_h, val_h = self._project_into_cache(hidden_states, past_key_value)
        else:
            key_h = self.k_proj.project_to_latent(hidden_states).view(bsz, q_len, G, self.group_rank_k).transpose(1, 2)
            val_h = self.v_proj.project_to_latent(hidden_states).view(bsz, q_len, G, self.group_rank_v).transpose(1, 2)
            if past_key_value is not None:
                key_h, val_h = past_key_value.update(key_h, val_h, self.layer_idx)
        if position_ids is None:
            position_ids = torch.arange(past, kv_seq_len, device=hidden_states.device).unsqueeze(0)
        pos = position_ids.to(hidden_states.device).reshape(-1, q_len)
        cos, sin = self._rope_tables(pos.reshape(-1), query_states.dtype)
        cos = cos.view(pos.shape[0], 1, q_len, D)
        sin = sin.view(pos.shape[0], 1, q_len, D)
        query_states = query_states * cos + _rotate_half(query_states) * sin
        if q_len > 1:
            if bsz != 1:
                raise ValueError("LlamaPaluAttention supports batch size 1 only (kernel/palu_attention.py:248,251)")
            lat = key_h.transpose(1, 2).reshape(bsz, kv_seq_len, self.total_rank_k)
            key_states = self.k_proj.reconstruct(lat).view(bsz, kv_seq_len, self.num_key_value_heads, D).transpose(1, 2)
            if self.n_rep > 1:
                key_states = key_states.repeat_interleave(self.n_rep, dim=1)        # repeat_kv: head h <- KV head h // n_rep
            kpos = torch.arange(kv_seq_len, device=hidden_states.device) if past else pos.reshape(-1)
            kc, ks = self._rope_tables(kpos, key_states.dtype)
            if past == 0:
                key_states = key_states * cos + _rotate_half(key_states) * sin
            else:
                key_states = key_states * kc.view(1, 1, kv_seq_len, D) + _rotate_half(key_states) * ks.view(1, 1, kv_seq_len, D)
            attn_weights = torch.matmul(query_states, key_states.transpose(2, 3)) / math.sqrt(D)
        else:
            attn_weights = recompute_k_gemv(query_states.squeeze(0), self.k_proj.B, key_h.squeeze(0),
                                            theta=self.rope_theta).unsqueeze(0) / math.sqrt(D)
        if attn_weights.size() != (bsz, H, q_len, kv_seq_len):
            raise ValueError(
                f"Attention weights should be of size {(bsz, H, q_len, kv_seq_len)}, but is {attn_weights.size()}")
        if attention_mask is not None:
            attn_weights = attn_weights + attention_mask
        attn_weights = nn.functional.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)
        attn_weights = nn.functional.dropout(attn_weights, p=self.attention_dropout, training=self.training)
        # latent-space P.V (:246-251): heads of a group share the group's V latents
        ctx = torch.matmul(attn_weights.reshape(1, G, q_len * self.group_size, kv_seq_len), val_h)
        ctx = ctx.reshape(1, H, q_len, self.group_rank_v)
        if fused_o:
            attn_output = ctx.transpose(1, 2).contiguous().reshape(bsz, q_len, -1)
        else:                                  # no_fusion: reconstruct V per head, dense o_proj
            uv = fold_u_per_head([u.weight for u in self.v_proj.U_list], D)             # [KV heads, D, Rv]
            if self.n_rep > 1:
                uv = uv.repeat_interleave(self.n_rep, dim=0)                            # [H, D, Rv]
            full = torch.einsum("hqr,hdr->hqd", ctx[0], uv.to(ctx.dtype))
            attn_output = full.transpose(0, 1).reshape(bsz, q_len, H * D)
        attn_output = self.o_proj(attn_output)
        if not output_attentions:
            attn_weights = None
        return attn_output, attn_weights, past_key_value

    # -- construction from a dense attention module ------------------------------------------
    @staticmethod
    def from_attention(module, config, no_fusion: bool = False):
        """Decompose k/v projections per head group and (unless `no_fusion`) fold U_v into o_proj:
        W_o'[:, h*Rv:(h+1)*Rv] = W_o[:, h*D:(h+1)*D] @ U_v[h//gs][(h%gs)*D:(h%gs+1)*D, :]  (:265-308)."""
        new = LlamaPaluAttention(config, getattr(module, "layer_idx", None))
        new.q_proj = module.q_proj
        new.k_proj = HeadwiseLowRankModule.from_linear(module.k_proj, new.rank_k_list, new)
        new.v_proj = HeadwiseLowRankModule.from_linear(module.v_proj, new.rank_v_list)
        if no_fusion:
            new.o_proj = module.o_proj
            return new
        fused = fuse_wo(module.o_proj.weight.data, [u.weight.data for u in new.v_proj.U_list], new.head_dim, new.n_rep)
        with torch.no_grad():
            new.o_proj.weight.copy_(fused)
        return new
