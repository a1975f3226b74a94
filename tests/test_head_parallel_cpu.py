"""Head-group parallel decode on CPU with gloo, world_size 2: the sharding plan, the weight /
cache slices and the all-gather layout are the product code (palu_amd.kernel.head_parallel); the per-rank
compute is stood in for by the CPU oracle (the HIP step cannot run here).  Sharded == unsharded."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from palu_amd.kernel import head_parallel as hp
from tests.golden import inputs as gi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_context(plan, w, k_lat, v_lat, tok, pos):
    """per-rank: projections + append + scores + softmax + latent P.V for this rank's groups (oracle math)"""
    import math
    H, G, D = plan.heads_local, plan.groups_local, plan.head_dim
    wd = dict(w)
    wd["wo"] = torch.zeros(1, H * plan.rank_v, dtype=torch.float16)          # o_proj happens after the gather
    # reuse oracle.decode_step up to the context: recompute the pieces explicitly
    h2 = tok.reshape(1, -1)
    q = torch.nn.functional.linear(h2, w["wq"]).reshape(H, 1, D)
    k_new = torch.nn.functional.linear(h2, w["vt_k"]).reshape(G, 1, plan.rank_k)
    v_new = torch.nn.functional.linear(h2, w["vt_v"]).reshape(G, 1, plan.rank_v)
    k_all, v_all = torch.cat((k_lat, k_new), 1), torch.cat((v_lat, v_new), 1)
    cos, sin = oracle.rope_cos_sin(pos + 1, D, start=pos)
    cos, sin = cos.half(), sin.half()
    q = q * cos + torch.cat((-q[..., D // 2:], q[..., :D // 2]), dim=-1) * sin
    s = oracle.abx_scores(q, w["b"], k_all) / math.sqrt(D)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).half()
    ctx = torch.matmul(p.reshape(G, plan.group_size, -1), v_all)
    return ctx.reshape(-1)


def _worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, _ = case
    G = H // gs
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)
    full = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
            "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    plan = hp.make_plan(world, rank, H, G, D, rank_k // G, rank_v // G)
    wl = hp.shard_weights(plan, full)
    kl, vl = hp.shard_cache(plan, k_lat, v_lat)
    assert wl["wq"].shape == (H // world * D, hidden) and wl["b"].shape[0] == H // world
    assert kl.shape[0] == G // world and wl["wo"].shape == full["wo"].shape
    ctx_local = _oracle_local_context(plan, wl, kl, vl, tok, L)
    assert ctx_local.numel() == plan.ctx_local
    ctx = hp.gather_context(ctx_local, plan)                                  # the one collective of the step
    out = torch.nn.functional.linear(ctx.reshape(1, -1), full["wo"]).reshape(-1)
    ref, _, _, _ = oracle.decode_step(tok, L, full, k_lat, v_lat)
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-3)
    # every rank holds the same full context after the gather
    allc = [torch.empty_like(ctx) for _ in range(world)]
    dist.all_gather(allc, ctx)
    for c in allc:
        assert torch.equal(c, ctx)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("case", [gi.STEP_CASES[0]], ids=["small_gs2"])
def test_head_group_parallel_world2(tmp_path, case):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _worker_sharded(rank, world, port, shape, out_dir):
    """column-sharded o_proj: every rank multiplies ITS context slice by ITS column block of W_o', the ranks all-reduce
    the [hidden] fp32 partials (hp.reduce_partial_outputs) -- against the un-sharded oracle step AND the all-gather
    variant, for the shard plan of `world` ranks (world 8: one latent group per rank, the BASELINE config-5 layout)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    seed, hidden, H, D, gs, rank_k, rank_v, L = shape
    G = H // gs
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)
    full = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
            "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    plan = hp.make_plan(world, rank, H, G, D, rank_k // G, rank_v // G)
    ws = hp.shard_weights(plan, full, oproj="sharded")
    wr = hp.shard_weights(plan, full, oproj="replicated")
    Rv = rank_v // G
    assert ws["wo"].shape == (hidden, H // world * Rv) and wr["wo"].shape == (hidden, H * Rv)
    assert torch.equal(ws["wo"], full["wo"][:, plan.head0 * Rv:(plan.head0 + plan.heads_local) * Rv])
    kl, vl = hp.shard_cache(plan, k_lat, v_lat)
    assert kl.shape[0] == G // world == plan.groups_local
    ctx_local = _oracle_local_context(plan, ws, kl, vl, tok, L)
    # sharded: fp32 partial of this rank's column block, all-reduce, one rounding
    partial = torch.nn.functional.linear(ctx_local.float().reshape(1, -1), ws["wo"].float()).reshape(-1)
    out_s = hp.reduce_partial_outputs(partial.clone(), plan)
    # replicated: all-gather, full GEMV
    ctx = hp.gather_context(ctx_local, plan)
    out_r = torch.nn.functional.linear(ctx.reshape(1, -1), full["wo"]).reshape(-1)
    ref, _, _, _ = oracle.decode_step(tok, L, full, k_lat, v_lat)
    torch.testing.assert_close(out_s, ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out_r, ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out_s, out_r, rtol=1e-3, atol=2e-4)
    # every rank ends with the same output vector
    allo = [torch.empty_like(out_s) for _ in range(world)]
    dist.all_gather(allo, out_s)
    for o in allo:
        assert torch.equal(o, out_s)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("world,shape", [(2, (10, 512, 4, 128, 2, 64, 128, 96)),
                                         (8, (21, 256, 16, 128, 2, 8 * 16, 8 * 32, 40))], ids=["world2", "world8"])
def test_head_group_parallel_sharded_oproj(tmp_path, world, shape):
    port = _free_port()
    mp.spawn(_worker_sharded, args=(world, port, shape, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_plan_validation_and_layout():
    plan = hp.make_plan(4, 2, 32, 8, 128, 128, 384)
    assert (plan.groups_local, plan.heads_local, plan.group0, plan.head0, plan.ctx_local) == (2, 8, 4, 16, 8 * 384)
    with pytest.raises(ValueError):
        hp.make_plan(3, 0, 32, 8, 128, 128, 384)       # groups not divisible -> split-L would be needed
    with pytest.raises(ValueError):
        hp.make_plan(2, 2, 32, 8, 128, 128, 384)
    # rank-major concatenation of slices == head-major [H*Rv] o_proj input
    H, Rv = 8, 4
    full = torch.arange(H * Rv).reshape(H, Rv)
    parts = [full[hp.make_plan(4, r, H, 4, 128, 32, Rv).head0:][:2].reshape(-1) for r in range(4)]
    assert torch.equal(torch.cat(parts), full.reshape(-1))


# ---------------------------------------------------------------------------------------- split-L (G < N)
def _oracle_split_partial(full, k_all, v_all, q_rot, l0, l1, G, gs):
    """oracle math for one rank's row range: locally normalised context + (max, sum) per head"""
    import math
    H = G * gs
    D = q_rot.shape[-1]
    s = (oracle.abx_scores(q_rot, full["b"], k_all) / math.sqrt(D)).reshape(H, -1)[:, l0:l1].float()
    if l1 == l0:
        return torch.zeros(H, v_all.shape[-1]), torch.full((H,), float("-inf")), torch.zeros(H)
    m = s.max(dim=1).values
    e = torch.exp(s - m.unsqueeze(1))
    ssum = e.sum(1)
    p = (e / ssum.unsqueeze(1)).half()
    ctx = torch.matmul(p.reshape(G, gs, -1), v_all[:, l0:l1]).reshape(H, -1)
    return ctx, m, ssum


def _worker_split(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, _ = case
    G = H // gs
    w, k_lat, v_lat, tok, _ = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, False)
    full = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
            "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    ref, _, k_all, v_all = oracle.decode_step(tok, L, full, k_lat, v_lat)
    ranges = hp.split_ranges(L, world, align=32)
    l0, l1 = ranges[rank]
    if rank == world - 1:
        l1 = L + 1                                                  # the owner of the tail sees the new row
    h2 = tok.reshape(1, -1)
    q = torch.nn.functional.linear(h2, full["wq"]).reshape(H, 1, D)
    cos, sin = oracle.rope_cos_sin(L + 1, D, start=L)
    q = q * cos.half() + torch.cat((-q[..., D // 2:], q[..., :D // 2]), dim=-1) * sin.half()
    ctx, m, ssum = _oracle_split_partial(full, k_all, v_all, q, l0, l1, G, gs)
    merged = hp.gather_and_merge(ctx.half(), torch.stack((m, ssum), dim=1), world)      # the one collective
    out = torch.nn.functional.linear(merged.half().reshape(1, -1), full["wo"]).reshape(-1)
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=1e-3)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("case", [gi.STEP_CASES[0]], ids=["small_gs2"])
def test_split_l_world2(tmp_path, case):
    world = 2
    port = _free_port()
    mp.spawn(_worker_split, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_split_ranges_and_merge():
    assert hp.split_ranges(65536, 8) == [(i * 8192, (i + 1) * 8192) for i in range(8)]
    assert hp.split_ranges(300, 2) == [(0, 256), (256, 300)]
    assert hp.split_ranges(100, 4) == [(0, 100), (100, 100), (100, 100), (100, 100)]
    # merge of per-range softmax pieces == softmax over the union (incl. an empty range)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 50, generator=g) * 5
    v = torch.randn(50, 7, generator=g)
    ref = torch.softmax(x, dim=-1) @ v
    ctx, m, s = [], [], []
    for a, b in ((0, 20), (20, 20), (20, 50)):
        if a == b:
            ctx.append(torch.zeros(3, 7)); m.append(torch.full((3,), float("-inf"))); s.append(torch.zeros(3))
            continue
        mm = x[:, a:b].max(1).values
        e = torch.exp(x[:, a:b] - mm.unsqueeze(1))
        ctx.append((e / e.sum(1, keepdim=True)) @ v[a:b]); m.append(mm); s.append(e.sum(1))
    out = hp.merge_partials(torch.stack(ctx), torch.stack(m), torch.stack(s))
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_bench_dry_plan_world2_self_spawn():
    """`python bench.py --gpus 2 --dry_plan` started WITHOUT torch.distributed.run spawns its two ranks itself (gloo, CPU):
    sharding plan, weight shards of both o_proj forms, the step's collectives at their real message sizes, and ONE JSON
    line from rank 0 with the contract's keys (the scaling run must be a one-command certainty)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry_plan"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in rec
    assert rec["n_gpus"] == 2 and rec["dry_plan"] is True and rec["value"] is None
    assert rec["plan"]["groups_local"] == 4 and rec["plan"]["heads_local"] == 16
    assert rec["plan"]["shards"]["sharded"]["wo"][1] == 16 * 384 and rec["plan"]["shards"]["replicated"]["wo"][1] == 32 * 384
    ck = rec["checks"]
    assert ck["all_gather_ok"] and ck["all_reduce_ok"] and ck["sharded_oproj_sums_to_replicated"]
    assert ck["all_gather_message_bytes"] == 16 * 384 * 2 and ck["all_reduce_message_bytes"] == 4096 * 4
