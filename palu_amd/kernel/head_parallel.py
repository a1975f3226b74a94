"""Head-group parallel decode across the GPUs of one node (BASELINE config 5).

The reference has no multi-GPU path; the decode step shards naturally by head GROUP: scores, softmax
and the latent P.V of a group never touch another group (kernel/palu_attention.py:216-251).  Rank r of
N owns groups [r*G/N, (r+1)*G/N): their latent caches (never moved), the B slice, the W_q rows of its
gs*D*G/N query dims and the VT_k / VT_v rows of its ranks; the token's hidden state is replicated.
The ONE exchange per step is an all-gather of the per-rank context slice [H/N * Rv] fp16 (3 KiB at
N=8, C2) -- RCCL over xGMI through torch.distributed -- followed by the replicated o_proj GEMV.

This file holds the rank-independent bookkeeping (testable on CPU with gloo) and the per-rank HIP
step; the collective is a single `all_gather_into_tensor` on the caller's process group.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch


@dataclass(frozen=True)
class ShardPlan:
    world: int
    rank: int
    num_heads: int
    num_groups: int
    head_dim: int
    rank_k: int          # per-group key rank
    rank_v: int          # per-group value rank

    @property
    def groups_local(self) -> int:
        return self.num_groups // self.world

    @property
    def group_size(self) -> int:
        return self.num_heads // self.num_groups

    @property
    def heads_local(self) -> int:
        return self.groups_local * self.group_size

    @property
    def group0(self) -> int:
        return self.rank * self.groups_local

    @property
    def head0(self) -> int:
        return self.group0 * self.group_size

    @property
    def ctx_local(self) -> int:
        return self.heads_local * self.rank_v


def make_plan(world: int, rank: int, num_heads: int, num_groups: int, head_dim: int, rank_k: int, rank_v: int) -> ShardPlan:
    if num_groups % world != 0:
        raise ValueError(f"head-group parallelism needs num_groups ({num_groups}) divisible by world size ({world}); "
                         "use split-L for fewer groups than GPUs (not implemented)")
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return ShardPlan(world, rank, num_heads, num_groups, head_dim, rank_k, rank_v)


def shard_weights(plan: ShardPlan, w: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Slice the full weights {wq [H*D,hid], vt_k [G*Rk,hid], vt_v [G*Rv,hid], b [H,Rk,D], wo [hid,H*Rv]} to
    what rank `plan.rank` owns.  `wo` stays whole (o_proj is replicated after the all-gather)."""
    D, gs = plan.head_dim, plan.group_size
    h0, h1 = plan.head0, plan.head0 + plan.heads_local
    g0, g1 = plan.group0, plan.group0 + plan.groups_local
    return {
        "wq": w["wq"][h0 * D:h1 * D],
        "vt_k": w["vt_k"][g0 * plan.rank_k:g1 * plan.rank_k],
        "vt_v": w["vt_v"][g0 * plan.rank_v:g1 * plan.rank_v],
        "b": w["b"][h0:h1],
        "wo": w["wo"],
    }


def shard_cache(plan: ShardPlan, k_lat: torch.Tensor, v_lat: torch.Tensor):
    """[G, L, R] -> this rank's groups."""
    g0, g1 = plan.group0, plan.group0 + plan.groups_local
    return k_lat[g0:g1], v_lat[g0:g1]


def gather_context(ctx_local: torch.Tensor, plan: ShardPlan, group=None) -> torch.Tensor:
    """All-gather of the context slices in head order: rank r contributes heads [r*H/N, (r+1)*H/N), so the
    rank-major concatenation IS the [H*Rv] o_proj input (kernel/palu_attention.py:251-255)."""
    import torch.distributed as dist
    ctx_local = ctx_local.reshape(-1).contiguous()
    if plan.world == 1:
        return ctx_local
    full = torch.empty(plan.world * ctx_local.numel(), dtype=ctx_local.dtype, device=ctx_local.device)
    dist.all_gather_into_tensor(full, ctx_local, group=group)
    return full


class HeadParallelDecoder:
    """Per-rank state + HIP launches of the sharded decode step (fp16).  `weights`/caches are this rank's
    shard already on its GPU; caches are [G_loc, Lcap, R] buffers holding `cache_len` valid rows."""

    def __init__(self, plan: ShardPlan, weights: Dict[str, torch.Tensor], k_cache: torch.Tensor,
                 v_cache: torch.Tensor, hidden_size: int, theta: float = 10000.0, group=None):
        from .. import _lib
        from .abx_rope import prepare_b, rope_inv_freq
        self._lib = _lib
        self.plan, self.w, self.k, self.v, self.hidden, self.group = plan, weights, k_cache, v_cache, hidden_size, group
        dev = k_cache.device
        self.frag = prepare_b(weights["b"], plan.groups_local)
        self.inv = rope_inv_freq(dev, plan.head_dim, theta)
        Hl, Gl, cap = plan.heads_local, plan.groups_local, k_cache.shape[1]
        self.q = torch.empty(Hl * plan.head_dim, dtype=torch.float16, device=dev)
        self.scores = torch.empty((Hl, (cap + 8) // 8 * 8), dtype=torch.float16, device=dev)
        self.ctx = torch.empty(Hl * plan.rank_v, dtype=torch.float16, device=dev)
        self.ctx_full = torch.empty(plan.num_heads * plan.rank_v, dtype=torch.float16, device=dev)
        self.pvws = torch.empty(_lib.lib.palu_pv_workspace_bytes(Hl, Gl, cap, plan.rank_v), dtype=torch.uint8, device=dev)
        self.out = torch.empty(hidden_size, dtype=torch.float16, device=dev)

    def local_step(self, hidden: torch.Tensor, cache_len: int, pos: int):
        """qkv + RoPE + append -> abx -> softmax.PV for this rank's groups; returns the context slice."""
        import math
        lib, p, w = self._lib, self.plan, self.w
        s = lib.current_stream()
        Hl, Gl, D = p.heads_local, p.groups_local, p.head_dim
        L = cache_len + 1
        lib.check(lib.lib.palu_decode_qkv_f16(
            w["wq"].data_ptr(), w["wq"].stride(0), w["vt_k"].data_ptr(), w["vt_k"].stride(0),
            w["vt_v"].data_ptr(), w["vt_v"].stride(0), hidden.data_ptr(), self.q.data_ptr(),
            self.k.data_ptr(), self.k.stride(0), self.k.stride(1), self.v.data_ptr(), self.v.stride(0), self.v.stride(1),
            self.inv.data_ptr(), Hl, D, self.hidden, Gl, p.rank_k, p.rank_v, pos, cache_len, s), "decode_qkv")
        lib.check(lib.lib.palu_abx_rope_f16(self.q.data_ptr(), D, 1, self.frag.data_ptr(), self.k.data_ptr(),
                                            self.k.stride(0), self.k.stride(1), self.scores.data_ptr(),
                                            self.scores.stride(0), Hl, Gl, L, p.rank_k, D, self.inv.data_ptr(), 0, s), "abx")
        lib.check(lib.lib.palu_softmax_pv_f16(self.scores.data_ptr(), self.scores.stride(0), 0, self.v.data_ptr(),
                                              self.v.stride(0), self.v.stride(1), self.ctx.data_ptr(), 0, 0,
                                              self.pvws.data_ptr(), Hl, Gl, L, p.rank_v, math.sqrt(D), s), "softmax_pv")
        return self.ctx

    def step(self, hidden: torch.Tensor, cache_len: int, pos: int) -> torch.Tensor:
        import torch.distributed as dist
        lib, p = self._lib, self.plan
        ctx = self.local_step(hidden, cache_len, pos)
        if p.world > 1:
            dist.all_gather_into_tensor(self.ctx_full, ctx, group=self.group)
            full = self.ctx_full
        else:
            full = ctx
        wo = self.w["wo"]
        lib.check(lib.lib.palu_gemv_f16(wo.data_ptr(), wo.stride(0), full.data_ptr(), self.out.data_ptr(),
                                        self.hidden, p.num_heads * p.rank_v, lib.current_stream()), "o_proj")
        return self.out
