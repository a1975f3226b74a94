#!/usr/bin/env python3
"""Exercise HeadParallelDecoder.step with world_size 2 on ONE GPU (both ranks on cuda:0, gloo carrying the
all-gather through the host) and compare with the unsharded step -- a functional check of the N > 1 code path
that bench.py --gpus N uses, for boxes with a single GPU (RCCL refuses two ranks on one device)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

H, G, D, HIDDEN, RK, RV, LP = 32, 8, 128, 4096, 1024, 3072, 4096


OPROJ = "replicated"


def build(world, rank, dev):
    from palu_amd.kernel import head_parallel as hp
    torch.manual_seed(1234)
    Rk, Rv = RK // G, RV // G
    full = {"wq": (torch.randn(H * D, HIDDEN, device=dev) / 64).half(),
            "vt_k": (torch.randn(RK, HIDDEN, device=dev) / 64).half(),
            "vt_v": (torch.randn(RV, HIDDEN, device=dev) / 64).half(),
            "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": (torch.randn(HIDDEN, H * Rv, device=dev) * 0.01).half()}
    cap = LP + 64
    k_all = torch.randn(G, cap, Rk, device=dev, dtype=torch.float16)
    v_all = torch.randn(G, cap, Rv, device=dev, dtype=torch.float16)
    hidden = torch.randn(HIDDEN, device=dev, dtype=torch.float16)
    plan = hp.make_plan(world, rank, H, G, D, Rk, Rv)
    w = {k: v.contiguous() if k != "wo" else v for k, v in hp.shard_weights(plan, full, oproj=OPROJ).items()}
    kc, vc = hp.shard_cache(plan, k_all, v_all)
    return hp.HeadParallelDecoder(plan, w, kc.contiguous(), vc.contiguous(), HIDDEN), hidden


def worker(rank, world, port, q, oproj, p2p=False):
    global OPROJ
    OPROJ = oproj
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dec, hidden = build(world, rank, dev)
    if p2p:
        # the step's collective as the one-shot peer-to-peer exchange kernel (csrc/exchange.hip) between the two PROCESSES:
        # eager step, then the same step captured into one hipGraph (kernels + exchange) and replayed
        from palu_amd.kernel import head_parallel as hp
        ex = hp.IpcExchange(rank, world, max(HIDDEN * 4, dec.plan.heads_local * dec.plan.rank_v * 2), dev)
        dec.exchange = ex
        out = dec.step(hidden, LP, LP).float().cpu()
        torch.cuda.synchronize()
        dist.barrier()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dec.step(hidden, LP, LP)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o2 = dec.step(hidden, LP, LP)
        for _ in range(3):
            g.replay()
            torch.cuda.synchronize()
            dist.barrier()
        rep = o2.float().cpu()
        done, err = ex.status()
        if rank == 0:
            q.put((out, rep, err))
        dist.barrier()
        ex.close()
        dist.destroy_process_group()
        return
    # gloo moves host tensors: wrap the one collective
    orig = dist.all_gather_into_tensor

    def via_host(out, inp, group=None):
        o, i = out.cpu(), inp.cpu()
        orig(o, i, group=group)
        out.copy_(o)
    dist.all_gather_into_tensor = via_host
    orig_ar = dist.all_reduce

    def ar_via_host(t, op=dist.ReduceOp.SUM, group=None):
        h = t.cpu()
        orig_ar(h, op=op, group=group)
        t.copy_(h)
    dist.all_reduce = ar_via_host
    assert dec.oproj_sharded == (oproj == "sharded")
    out = dec.step(hidden, LP, LP).float().cpu()
    torch.cuda.synchronize()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def run_world2(oproj, p2p=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q, oproj, p2p)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
    return sharded


def main():
    ok = True
    dec, hidden = build(1, 0, torch.device("cuda", 0))
    ref = dec.step(hidden, LP, LP).float().cpu()
    for oproj in ("replicated", "sharded"):
        sharded = run_world2(oproj)
        err = (sharded - ref).abs().max().item()
        print(f"world 2 on one GPU ({oproj} o_proj) vs unsharded: max|diff| = {err:.3e} (scale {ref.abs().max().item():.3e})")
        ok &= err <= 1e-3 * max(1.0, ref.abs().max().item())
        eager, replay, terr = run_world2(oproj, p2p=True)
        e1, e2 = (eager - ref).abs().max().item(), (replay - ref).abs().max().item()
        print(f"world 2 on one GPU ({oproj} o_proj), one-shot P2P exchange between the processes: eager max|diff| = {e1:.3e}, "
              f"graph replay {e2:.3e}, timeout word {terr}")
        ok &= max(e1, e2) <= 1e-3 * max(1.0, ref.abs().max().item()) and terr == 0
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
