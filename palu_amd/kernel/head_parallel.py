"""Head-group parallel decode across the GPUs of one node (BASELINE config 5).

The reference has no multi-GPU path; the decode step shards naturally by head GROUP: scores, softmax
and the latent P.V of a group never touch another group (kernel/palu_attention.py:216-251).  Rank r of
N owns groups [r*G/N, (r+1)*G/N): their latent caches (never moved), the B slice, the W_q rows of its
gs*D*G/N query dims and the VT_k / VT_v rows of its ranks; the token's hidden state is replicated.
ONE exchange per step, two variants (shard_weights(..., oproj=)):
  * "sharded" (default of bench.py / run_latency_attention.py): the rank multiplies its own [H/N * Rv] context
    slice by its column block of W_o' (1/N of the bytes) and the ranks ALL-REDUCE the [hidden] fp32 partial outputs
    (16 KiB), rounded once to fp16;
  * "replicated": ALL-GATHER of the per-rank context slices [H/N * Rv] fp16 (3 KiB at N = 8, C2) and the full o_proj
    GEMV on every rank.
The collective goes through torch.distributed (backend "nccl" = RCCL over xGMI) behind a two-method `exchange` object,
so that (a) `HeadParallelDecoder.capture()` records the whole step INCLUDING the collective into one hipGraph -- a
replay issues no Python-side launch -- and (b) tests can run several ranks on one GPU with an in-graph stand-in.

This file holds the rank-independent bookkeeping (testable on CPU with gloo) and the per-rank HIP step.
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
                         "use SplitLDecoder (split-L) for fewer groups than GPUs")
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return ShardPlan(world, rank, num_heads, num_groups, head_dim, rank_k, rank_v)


def shard_weights(plan: ShardPlan, w: Dict[str, torch.Tensor], oproj: str = "replicated") -> Dict[str, torch.Tensor]:
    """Slice the full weights {wq [H*D,hid], vt_k [G*Rk,hid], vt_v [G*Rv,hid], b [H,Rk,D], wo [hid,H*Rv]} to
    what rank `plan.rank` owns.  oproj = "replicated": `wo` stays whole (o_proj runs on the all-gathered context on every
    rank); "sharded": the rank keeps only its column block wo[:, h0*Rv:h1*Rv] (1/N of the bytes: 12 MiB instead of
    96 MiB at C2, N=8) and the ranks all-reduce the [hidden] fp32 partial outputs."""
    D, gs = plan.head_dim, plan.group_size
    h0, h1 = plan.head0, plan.head0 + plan.heads_local
    g0, g1 = plan.group0, plan.group0 + plan.groups_local
    if oproj not in ("replicated", "sharded"):
        raise ValueError("oproj must be 'replicated' or 'sharded'")
    wo = w["wo"] if oproj == "replicated" else w["wo"][:, h0 * plan.rank_v:h1 * plan.rank_v]
    return {
        "wq": w["wq"][h0 * D:h1 * D],
        "vt_k": w["vt_k"][g0 * plan.rank_k:g1 * plan.rank_k],
        "vt_v": w["vt_v"][g0 * plan.rank_v:g1 * plan.rank_v],
        "b": w["b"][h0:h1],
        "wo": wo,
    }


def reduce_partial_outputs(partial: torch.Tensor, plan: ShardPlan, group=None) -> torch.Tensor:
    """Sum of the per-rank o_proj partials ([hidden] fp32, one all-reduce of 16 KiB), rounded once to fp16: the same
    value the un-sharded GEMV produces up to the order of the fp32 additions."""
    import torch.distributed as dist
    if plan.world > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial.to(torch.float16)


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


class DistExchange:
    """The step's one collective through torch.distributed on `group` (nccl = RCCL on the GPUs, gloo in the CPU tests).
    Both calls are recorded into a capturing stream by the NCCL backend, i.e. they can be part of a hipGraph."""

    def __init__(self, group=None):
        self.group = group

    def all_reduce_sum_(self, t: torch.Tensor) -> None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> None:
        import torch.distributed as dist
        dist.all_gather_into_tensor(out, t, group=self.group)


class IpcExchange:
    """The step's collective as ONE kernel per rank writing straight into every peer's buffer (csrc/exchange.hip,
    include/palu_hip.h: palu_exchange_*): no ring, no host involvement, capturable.  For the 3 KiB / 16 KiB messages of the
    head-parallel step a collective library's small-message latency is the cost; this is the one-shot alternative over
    the same xGMI links.  Setup exchanges hipIpc handles through `group` (any torch.distributed backend: only
    `all_gather_object` is used) -- or takes the peers' pointers directly when all ranks live in one process (`peers=`).

    slot_bytes: the largest message (bytes) this exchange will carry, rounded up to 16."""

    def __init__(self, rank: int, world: int, slot_bytes: int, device, group=None, peers=None):
        import ctypes as C
        from .. import _lib
        self._lib, self.rank, self.world = _lib, rank, world
        self.slot = (int(slot_bytes) + 15) // 16 * 16
        self.device = torch.device(device)
        self._imported = []
        with torch.cuda.device(self.device):
            nbytes = _lib.lib.palu_exchange_bytes(world, self.slot)
            if not nbytes:
                raise ValueError(f"IpcExchange: unsupported world {world} / slot {self.slot}")
            ptr = C.c_void_p()
            _lib.check(_lib.lib.palu_exchange_alloc(nbytes, C.byref(ptr)), "palu_exchange_alloc")
            self.buffer = ptr.value
            if peers is None:
                import torch.distributed as dist
                hb = _lib.lib.palu_exchange_handle_bytes()
                h = C.create_string_buffer(hb)
                _lib.check(_lib.lib.palu_exchange_export(self.buffer, h), "palu_exchange_export")
                handles = [None] * world
                dist.all_gather_object(handles, bytes(h.raw), group=group)
                peers = []
                for r in range(world):
                    if r == rank:
                        peers.append(self.buffer)
                        continue
                    q = C.c_void_p()
                    _lib.check(_lib.lib.palu_exchange_import(C.create_string_buffer(handles[r], hb), C.byref(q)),
                               "palu_exchange_import")
                    self._imported.append(q.value)
                    peers.append(q.value)
            else:
                peers = list(peers)
                peers[rank] = self.buffer
            self.peers = torch.tensor(peers, dtype=torch.int64, device=self.device)

    def set_peers(self, peers) -> None:
        """single-process use: fill in the other ranks' buffer pointers once every rank has allocated"""
        self.peers = torch.tensor(list(peers), dtype=torch.int64, device=self.device)

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> None:
        nb = t.numel() * t.element_size()
        assert out.numel() * out.element_size() == nb * self.world and t.is_contiguous() and out.is_contiguous()
        self._lib.check(self._lib.lib.palu_exchange_allgather(t.data_ptr(), nb, self.peers.data_ptr(), self.rank, self.world,
                                                             self.slot, out.data_ptr(), self._lib.current_stream(self.device)),
                        "palu_exchange_allgather")

    def all_reduce_sum_(self, t: torch.Tensor) -> None:
        assert t.dtype == torch.float32 and t.is_contiguous()
        nb = t.numel() * 4
        self._lib.check(self._lib.lib.palu_exchange_allreduce_f32(t.data_ptr(), nb, self.peers.data_ptr(), self.rank,
                                                                 self.world, self.slot, t.data_ptr(),
                                                                 self._lib.current_stream(self.device)),
                        "palu_exchange_allreduce_f32")

    def status(self):
        """(exchanges done, epoch of a timed-out wait or 0) -- synchronises"""
        import ctypes as C
        e, err = C.c_uint(), C.c_uint()
        torch.cuda.synchronize(self.device)
        self._lib.check(self._lib.lib.palu_exchange_status(self.buffer, C.byref(e), C.byref(err)), "palu_exchange_status")
        return e.value, err.value

    def close(self) -> None:
        for q in self._imported:
            self._lib.lib.palu_exchange_close(q)
        self._imported = []
        if self.buffer:
            self._lib.lib.palu_exchange_free(self.buffer)
            self.buffer = None


class LocalExchange:
    """In-process stand-in for the collective: several ranks' decoders living on ONE device and stepped on one stream
    (tests, tools/hp_two_ranks_one_gpu.py).  Plain device ops, so a captured graph contains the exchange.  Protocol: every
    rank finishes `step_local` before the first rank calls `step_finish` (one stream: program order)."""

    def __init__(self, world: int):
        self.world = world
        self._src = {}        # kind -> {rank: tensor}
        self._calls = {"sum": 0, "gather": 0}
        self._total = None

    def register(self, rank: int, partial: torch.Tensor = None, ctx: torch.Tensor = None):
        if partial is not None:
            self._src.setdefault("sum", {})[rank] = partial
        if ctx is not None:
            self._src.setdefault("gather", {})[rank] = ctx
        return self

    def all_reduce_sum_(self, t: torch.Tensor) -> None:
        if self._calls["sum"] % self.world == 0:      # first finisher of a round: sum every rank's (still local) partial
            parts = [self._src["sum"][r] for r in range(self.world)]
            self._total = torch.stack(parts).sum(0)
        self._calls["sum"] += 1
        t.copy_(self._total)

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> None:
        self._calls["gather"] += 1
        torch.cat([self._src["gather"][r].reshape(-1) for r in range(self.world)], out=out)


class HeadParallelDecoder:
    """Per-rank state + HIP launches of the sharded decode step (fp16).  `weights`/caches are this rank's
    shard already on its GPU; caches are [G_loc, Lcap, R] buffers holding `cache_len` valid rows.
    `exchange`: the collective (default: DistExchange(group))."""

    def __init__(self, plan: ShardPlan, weights: Dict[str, torch.Tensor], k_cache: torch.Tensor,
                 v_cache: torch.Tensor, hidden_size: int, theta: float = 10000.0, group=None, exchange=None):
        from .. import _lib
        from .abx_rope import prepare_b, rope_inv_freq
        self._lib = _lib
        self.plan, self.w, self.k, self.v, self.hidden, self.group = plan, weights, k_cache, v_cache, hidden_size, group
        self.exchange = exchange if exchange is not None else DistExchange(group)
        self._graph = None
        # o_proj variant from the shape of the weight the rank was given (shard_weights(..., oproj=...))
        self.oproj_sharded = plan.world > 1 and weights["wo"].shape[1] == plan.heads_local * plan.rank_v
        self.partial = torch.empty(hidden_size, dtype=torch.float32, device=k_cache.device)
        self.t_collective = None      # optional (start, end) CUDA events around the collective of the last step
        dev = k_cache.device
        self.frag = prepare_b(weights["b"], plan.groups_local)
        self.inv = rope_inv_freq(dev, plan.head_dim, theta)
        Hl, Gl, cap = plan.heads_local, plan.groups_local, k_cache.shape[1]
        self.cap = cap
        self.ctx = torch.empty(Hl * plan.rank_v, dtype=torch.float16, device=dev)
        self.ctx_full = torch.empty(plan.num_heads * plan.rank_v, dtype=torch.float16, device=dev)
        self.ws = torch.empty(_lib.lib.palu_decode_workspace_bytes(Hl, Gl, plan.head_dim, cap, plan.rank_v),
                              dtype=torch.uint8, device=dev)
        self.out = torch.empty(hidden_size, dtype=torch.float16, device=dev)

    def local_step(self, hidden: torch.Tensor, cache_len: int, pos: int):
        """qkv + RoPE + append -> abx -> softmax.PV for this rank's groups in ONE native call (three launches, no
        host work in between); returns the context slice."""
        lib, p, w = self._lib, self.plan, self.w
        lib.check(lib.lib.palu_decode_attend_f16(
            hidden.data_ptr(), w["wq"].data_ptr(), w["wq"].stride(0), w["vt_k"].data_ptr(), w["vt_k"].stride(0),
            w["vt_v"].data_ptr(), w["vt_v"].stride(0), self.frag.data_ptr(),
            self.k.data_ptr(), self.k.stride(0), self.k.stride(1), self.v.data_ptr(), self.v.stride(0), self.v.stride(1),
            0, self.inv.data_ptr(), self.ctx.data_ptr(), self.ws.data_ptr(), self.cap, p.heads_local, p.groups_local,
            p.head_dim, self.hidden, p.rank_k, p.rank_v, cache_len, pos, lib.current_stream()), "decode_attend")
        return self.ctx

    def step_local(self, hidden: torch.Tensor, cache_len: int, pos: int) -> None:
        """Everything of a step that needs no other rank: attention core of the rank's groups and, with the sharded
        o_proj, the fp32 partial output of its column block."""
        lib, p = self._lib, self.plan
        ctx = self.local_step(hidden, cache_len, pos)
        if self.oproj_sharded:
            wo = self.w["wo"]
            lib.check(lib.lib.palu_gemv_f16_acc32(wo.data_ptr(), wo.stride(0), ctx.data_ptr(), self.partial.data_ptr(),
                                                  self.hidden, p.heads_local * p.rank_v, lib.current_stream()), "o_proj")

    def step_finish(self, time_collective: bool = False) -> torch.Tensor:
        """The collective and what follows it (one rounding, or the replicated o_proj)."""
        lib, p = self._lib, self.plan
        wo = self.w["wo"]
        ev = None
        if time_collective and p.world > 1:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if self.oproj_sharded:
            if ev:
                ev[0].record()
            self.exchange.all_reduce_sum_(self.partial)
            if ev:
                ev[1].record()
            self.out.copy_(self.partial)            # fp32 -> fp16, the single rounding of the output
        else:
            if p.world > 1:
                if ev:
                    ev[0].record()
                self.exchange.all_gather_into(self.ctx_full, self.ctx)
                if ev:
                    ev[1].record()
                full = self.ctx_full
            else:
                full = self.ctx
            lib.check(lib.lib.palu_gemv_f16(wo.data_ptr(), wo.stride(0), full.data_ptr(), self.out.data_ptr(),
                                            self.hidden, p.num_heads * p.rank_v, lib.current_stream()), "o_proj")
        self.t_collective = ev
        return self.out

    def step(self, hidden: torch.Tensor, cache_len: int, pos: int, time_collective: bool = False) -> torch.Tensor:
        """One decode step.  Replicated o_proj: all-gather of the [H/N*Rv] fp16 context slices, then the full GEMV on
        every rank.  Sharded o_proj: every rank multiplies its own slice by its column block (fp32 partial [hidden]),
        one all-reduce of 16 KiB, one rounding.  time_collective records CUDA events around the collective
        (self.t_collective) for the collective-only latency bench.py reports."""
        self.step_local(hidden, cache_len, pos)
        return self.step_finish(time_collective)

    def capture(self, hidden: torch.Tensor, cache_len: int, pos: int, warm: int = 3):
        """hipGraph of one whole step -- kernels AND the collective (RCCL records into the capturing stream) -- for fixed
        (cache_len, pos), as the reference harness captures its forward (run_latency_attention.py:81-90).  The
        communicator must exist before the capture: `warm` eager steps on a side stream first (every rank calls
        capture() at the same point, so those collectives match up).  Returns the replay callable; the result is in
        `self.out`.  A replay issues no Python-side launch."""
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warm):
                self.step(hidden, cache_len, pos)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.step(hidden, cache_len, pos)
        self._graph = g
        return g.replay


# ------------------------------------------------------------------------------------------------------------
# Split-L: fewer head groups than GPUs ("when groups < GPUs"; 8(f) N3).  Every rank holds all
# weights and a contiguous range of cache rows of EVERY group; a step exchanges, per head, the locally normalised
# context and the softmax statistics (max, sum) and LSE-merges them -- the same merge the single-GPU kernel does
# across its own splits (pv_combine), lifted to the process group.
def split_ranges(L: int, world: int, align: int = 128):
    """Contiguous [l0, l1) row ranges, `align`-row granularity (the abx tile), last rank takes the tail (and the
    rows appended while decoding)."""
    per = -(-L // world)
    per = -(-per // align) * align
    return [(min(r * per, L), min((r + 1) * per, L)) for r in range(world)]


def merge_partials(ctx: torch.Tensor, m: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """ctx [N, H, Rv] locally normalised contexts, m / s [N, H] local max / sum of exp(x - max) -> [H, Rv] fp32:
    softmax over the union = sum_r w_r ctx_r / sum_r w_r with w_r = s_r * exp(m_r - max_r m_r); ranks without rows
    carry m = -inf, s = 0."""
    M = m.max(dim=0, keepdim=True).values
    w = s.float() * torch.exp(m.float() - M.float())
    w = torch.where(torch.isfinite(m), w, torch.zeros_like(w))
    return (w.unsqueeze(-1) * ctx.float()).sum(0) / w.sum(0).unsqueeze(-1)


def gather_and_merge(ctx_local: torch.Tensor, stats_local: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """ONE all-gather of [H*Rv + 2H] fp32 per rank (context + (max, sum) per head), then the LSE merge."""
    import torch.distributed as dist
    H = stats_local.shape[0]
    pack = torch.cat((ctx_local.reshape(-1).float(), stats_local.reshape(-1).float()))
    if world == 1:
        allp = pack.unsqueeze(0)
    else:
        flat = torch.empty(world * pack.numel(), dtype=torch.float32, device=pack.device)
        dist.all_gather_into_tensor(flat, pack, group=group)
        allp = flat.view(world, pack.numel())
    n_ctx = ctx_local.numel()
    ctx = allp[:, :n_ctx].reshape(world, H, -1)
    st = allp[:, n_ctx:].reshape(world, H, 2)
    return merge_partials(ctx, st[..., 0], st[..., 1])


class SplitLDecoder:
    """Per-rank HIP step of the split-L decode: replicated weights {wq, vt_k, vt_v, b, wo}, this rank's rows
    [l0, l1) of every group in `k_cache` / `v_cache` ([G, cap, R], `rows` valid).  The rank that `owns_tail`
    appends the new token's latent row; the others compute it too (the projection is fused with q) but park it in
    the spare row behind their range."""

    def __init__(self, world: int, rank: int, num_heads: int, num_groups: int, head_dim: int, weights, k_cache,
                 v_cache, rows: int, row0: int, owns_tail: bool, hidden_size: int, theta: float = 10000.0, group=None):
        from .. import _lib
        from .abx_rope import prepare_b, rope_inv_freq
        self._lib = _lib
        self.world, self.rank, self.H, self.G, self.D = world, rank, num_heads, num_groups, head_dim
        self.w, self.k, self.v, self.rows, self.row0, self.owns_tail = weights, k_cache, v_cache, rows, row0, owns_tail
        self.hidden, self.group = hidden_size, group
        dev = k_cache.device
        self.Rk, self.Rv = k_cache.shape[2], v_cache.shape[2]
        cap = k_cache.shape[1]
        self.frag = prepare_b(weights["b"], num_groups)
        self.inv = rope_inv_freq(dev, head_dim, theta)
        self.q = torch.empty(num_heads * head_dim, dtype=torch.float16, device=dev)
        self.scores = torch.empty((num_heads, (cap + 8) // 8 * 8), dtype=torch.float16, device=dev)
        self.ctx = torch.zeros((num_heads, self.Rv), dtype=torch.float16, device=dev)
        self.ws_bytes = _lib.lib.palu_pv_workspace_bytes(num_heads, num_groups, cap, self.Rv)
        self.pvws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.out = torch.empty(hidden_size, dtype=torch.float16, device=dev)
        self.empty_stats = torch.tensor([[float("-inf"), 0.0]] * num_heads, dtype=torch.float32, device=dev)

    def local_step(self, hidden: torch.Tensor, pos: int):
        """-> (ctx [H, Rv] fp16 normalised over the local rows, stats [H, 2] fp32 = (max, sum))."""
        import math
        lib, w = self._lib, self.w
        s = lib.current_stream()
        H, G, D = self.H, self.G, self.D
        lib.check(lib.lib.palu_decode_qkv_f16(
            w["wq"].data_ptr(), w["wq"].stride(0), w["vt_k"].data_ptr(), w["vt_k"].stride(0),
            w["vt_v"].data_ptr(), w["vt_v"].stride(0), hidden.data_ptr(), self.q.data_ptr(),
            self.k.data_ptr(), self.k.stride(0), self.k.stride(1), self.v.data_ptr(), self.v.stride(0), self.v.stride(1),
            self.inv.data_ptr(), H, D, self.hidden, G, self.Rk, self.Rv, pos, self.rows, s), "decode_qkv")
        L = self.rows + (1 if self.owns_tail else 0)
        if L == 0:
            return self.ctx, self.empty_stats
        lib.check(lib.lib.palu_abx_rope_f16(self.q.data_ptr(), D, 1, self.frag.data_ptr(), self.k.data_ptr(),
                                            self.k.stride(0), self.k.stride(1), self.scores.data_ptr(),
                                            self.scores.stride(0), H, G, L, self.Rk, D, self.inv.data_ptr(), self.row0, s), "abx")
        lib.check(lib.lib.palu_softmax_pv_f16(self.scores.data_ptr(), self.scores.stride(0), 0, self.v.data_ptr(),
                                              self.v.stride(0), self.v.stride(1), self.ctx.data_ptr(), 0, 0,
                                              self.pvws.data_ptr(), H, G, L, self.Rv, math.sqrt(D), s), "softmax_pv")
        off = lib.lib.palu_pv_stats_offset(H, G, L, self.Rv)
        stats = self.pvws[off:off + H * 8].view(torch.float32).view(H, 2)
        if self.owns_tail:
            self.rows += 1
        return self.ctx, stats

    def step(self, hidden: torch.Tensor, pos: int) -> torch.Tensor:
        lib = self._lib
        ctx, stats = self.local_step(hidden, pos)
        full = gather_and_merge(ctx, stats, self.world, self.group).to(torch.float16).reshape(-1).contiguous()
        wo = self.w["wo"]
        lib.check(lib.lib.palu_gemv_f16(wo.data_ptr(), wo.stride(0), full.data_ptr(), self.out.data_ptr(),
                                        self.hidden, self.H * self.Rv, lib.current_stream()), "o_proj")
        return self.out
