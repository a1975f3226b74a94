#!/usr/bin/env python3
"""The one-shot peer-to-peer exchange (csrc/exchange.hip, palu_amd.kernel.head_parallel.IpcExchange) between N PROCESSES
that share ONE GPU: hipIpc handles, direct stores into the peer's buffer, flags, bounded waits -- the protocol of an N-GPU
node exercised where only one GPU exists (there the "peer" memory is local, the processes and their caches are not).

    PYTHONPATH=. python tools/exchange_two_procs.py [--world 2] [--rounds 200]

Every rank checks every round of an all-gather (3 KiB slices) and of an fp32 all-reduce (16 KiB) against the values the
other ranks are known to send, then times the exchange alone (and inside a replayed hipGraph)."""
import argparse
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, rounds, initfile, out):
    from palu_amd.kernel.head_parallel import IpcExchange
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ex = IpcExchange(rank, world, 16384, dev)
    n16, n32 = 1536, 4096
    ok = True
    gath = torch.empty(world * n16, dtype=torch.float16, device=dev)
    for r in range(rounds):
        mine = torch.full((n16,), float(rank + 1) + (r % 64) / 64.0, dtype=torch.float16, device=dev)
        ex.all_gather_into(gath, mine)
        want = torch.cat([torch.full((n16,), float(q + 1) + (r % 64) / 64.0, dtype=torch.float16) for q in range(world)])
        if not torch.equal(gath.cpu(), want):
            ok = False
            print(f"rank {rank}: all-gather round {r} wrong", flush=True)
            break
        part = torch.full((n32,), (rank + 1) * 0.25 + r, dtype=torch.float32, device=dev)
        ex.all_reduce_sum_(part)
        want = sum((q + 1) * 0.25 + r for q in range(world))
        if not torch.equal(part.cpu(), torch.full((n32,), want, dtype=torch.float32)):
            ok = False
            print(f"rank {rank}: all-reduce round {r} wrong: {part[:4].tolist()} vs {want}", flush=True)
            break
    done, err = ex.status()
    ok = ok and err == 0 and done == 2 * rounds
    # timing: back-to-back exchanges (every rank runs the same count)
    dist.barrier()
    mine = torch.ones(n16, dtype=torch.float16, device=dev)
    for _ in range(20):
        ex.all_gather_into(gath, mine)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ex.all_gather_into(gath, mine)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    # inside a captured graph: the epoch lives in the buffer, a replay is a new exchange
    g_ok = True
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ex.all_gather_into(gath, mine)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ex.all_gather_into(gath, mine)
        for i in range(5):
            mine.fill_(float(i + 2))
            gath.zero_()
            graph.replay()
            torch.cuda.synchronize()
            g_ok = g_ok and bool((gath == float(i + 2)).all())
            dist.barrier()
    except Exception as e:                       # noqa: BLE001
        g_ok = False
        print(f"rank {rank}: graph capture failed: {e!r}", flush=True)
    done, err = ex.status()
    print(f"rank {rank}/{world}: checks {'OK' if ok else 'FAILED'}, {done} exchanges, timeout word {err}, "
          f"all-gather {us:.1f} us per exchange back to back, graph replay {'OK' if g_ok else 'FAILED'}", flush=True)
    dist.barrier()
    ex.close()
    dist.destroy_process_group()
    out.put((rank, ok and g_ok and err == 0, us))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=200)
    a = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.set_start_method("spawn", force=True)
    q = mp.Queue()
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        procs = [mp.Process(target=worker, args=(r, a.world, a.rounds, initfile, q)) for r in range(a.world)]
        for p in procs:
            p.start()
        t0 = time.time()
        res = []
        while len(res) < a.world and time.time() - t0 < 240:
            try:
                res.append(q.get(timeout=5))
            except Exception:                    # noqa: BLE001
                if not any(p.is_alive() for p in procs):
                    break
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    good = len(res) == a.world and all(r[1] for r in res)
    print("EXCHANGE", "OK" if good else "FAILED", sorted(res))
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
