#!/usr/bin/env python3
"""Kernel micro-benchmark for the fused abx_rope score kernel -- same CLI as the reference's
run_latency_kernel.py (:5-13) / kernel/abx_rope.py::run_benchmark (:173-228), without Triton:
warm-up 25, 100 timed reps, median/p20/p80 in microseconds (abx_rope.py:198-223), randn fp16
inputs (:200-204).  Providers, as in the reference (:180-181, :208-221): `WX` (uncompressed q.K^T, torch.matmul),
`torch` (the reference's own PyTorch path `torch_abx`, :152-171 -- reconstruct K with a batched matmul through
hipBLASLt, fp32 RoPE, fp16 q.K^T -- restated on the GPU) and `ours` (HIP abx).  Results are printed and, like
`bench_low_rank.run(save_path='results/')` (:227-228), written to results/low-rank-rank-<R>-group-<G>.csv.
Adds achieved HBM GB/s and MFMA TFLOP/s from the algorithmic bytes/flops of SURVEY.md 8(d).
"""
from __future__ import annotations

import argparse
import csv
import json
import os
# hipGraph replay: ROCm 7.2's graph "packet capture" path (on by default) costs ~3.5 us per replay of this 5-kernel step
# (157.3 us against 153.5 with it off, direct launches 152.7: profiles/r05_graph_replay.txt); it is read when the HIP runtime
# loads, so it must be set before torch is imported.  An explicit setting in the environment wins.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch


def do_bench(fn, warmup=25, rep=100, flush_mb=0):
    """hipEvent timing of `fn` on the current stream; returns (median, p20, p80) in us.

    Like triton.testing.do_bench (kernel/abx_rope.py:198-223) each rep is bracketed by its own event
    pair; the reps are enqueued back to back (no host sync in between) so that the events measure
    device time, not the Python launch latency of an idle stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    flush = torch.empty(flush_mb * 1024 * 1024, dtype=torch.uint8, device="cuda") if flush_mb else None
    evs = []
    for _ in range(rep):
        if flush is not None:
            flush.zero_()                     # evict L2 / Infinity Cache between reps
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    t = torch.tensor([s.elapsed_time(e) * 1e3 for s, e in evs])
    return t.median().item(), t.quantile(0.2).item(), t.quantile(0.8).item()


def do_bench_total(fn, warmup=25, rep=100):
    """One event pair around `rep` back-to-back calls (run_latency_attention.py:97-106 style): us/call."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rep):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / rep


def torch_provider(a, b, x, theta=10000.0):
    """The reference's PyTorch path for the same scores (kernel/abx_rope.py:152-171 `torch_abx`, with the rotary tables of
    kernel/pytorch_reference.py:3-21) as stock torch ops on the GPU: K = x @ B per group (fp16 batched matmul), RoPE in
    fp32 at positions 0..L-1, one rounding to fp16, q @ K^T."""
    H, R, D = b.shape
    G, L, _ = x.shape
    keys = torch.matmul(x[:, None], b.reshape(G, H // G, R, D)).reshape(H, L, D)
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64, device=x.device).float() / D))
    ang = torch.outer(torch.arange(L, device=x.device, dtype=torch.int64).float(), inv)
    ang = torch.cat((ang, ang), dim=-1)
    cos, sin = ang.cos(), ang.sin()
    rot = torch.cat((-keys[..., D // 2:], keys[..., :D // 2]), dim=-1)
    keys = (keys * cos + rot * sin).to(torch.float16)
    return torch.matmul(a, keys.transpose(-1, -2))


def abx_algorithmic(H, G, L, R, D=128):
    nbytes = 2 * G * L * R + 2 * H * R * D + 2 * H * D + 2 * H * L
    flops = 2 * H * L * R * D + 5 * H * L * D
    return nbytes, flops


def main():
    ap = argparse.ArgumentParser(description="abx_rope kernel latency (MI355X)")
    ap.add_argument("--total_rank", type=int, default=2048)
    ap.add_argument("--num_heads", type=int, default=32)
    ap.add_argument("--head_dim", type=int, default=128)
    ap.add_argument("--group_size", type=int, default=4)
    ap.add_argument("--target_seq_lens", nargs="+", type=int, default=[4096, 16384, 65536, 262144])
    ap.add_argument("--flush_mb", type=int, default=0, help="write this many MiB between reps to defeat the 256 MiB Infinity Cache")
    ap.add_argument("--no_fold", action="store_true", help="keep q in fp32 instead of folding it into B (slower, exact)")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--save_path", default="results/", help="directory of the CSV (abx_rope.py:227-228); '' = no file")
    ap.add_argument("--no_torch_provider", action="store_true", help="skip the PyTorch provider (6 GB of temporaries at 262144)")
    args = ap.parse_args()

    from palu_amd.kernel.abx_rope import abx, set_fold
    set_fold(not args.no_fold)
    torch.manual_seed(0)
    H, D = args.num_heads, args.head_dim
    G = H // args.group_size
    R = args.total_rank // G
    dev, dt = "cuda", torch.float16
    print(f"Total Rank: {args.total_rank}  Heads: {H}  Head dim: {D}  Group size: {args.group_size}  "
          f"Groups: {G}  Rank per group: {R}")
    rows = []
    for L in args.target_seq_lens:
        A = torch.randn(H, 1, D, dtype=dt, device=dev)
        B = torch.randn(H, R, D, dtype=dt, device=dev)
        X = torch.randn(G, L, R, dtype=dt, device=dev)
        org_A = torch.randn(H, 1, D, dtype=dt, device=dev)
        org_X = torch.randn(H, L, D, dtype=dt, device=dev)
        abx(A, B, X)                                                    # builds the B fragments once
        out_buf = torch.empty(H, 1, L, dtype=dt, device=dev)
        ours = do_bench(lambda: abx(A, B, X, out=out_buf), flush_mb=args.flush_mb)
        ours_total = do_bench_total(lambda: abx(A, B, X, out=out_buf))
        wx = do_bench(lambda: torch.matmul(org_A, org_X.transpose(-1, -2)), flush_mb=args.flush_mb)
        tch = (float("nan"),) * 3
        if not args.no_torch_provider:
            tch = do_bench(lambda: torch_provider(A, B, X), warmup=5, rep=20, flush_mb=args.flush_mb)
        nbytes, flops = abx_algorithmic(H, G, L, R, D)
        row = {"seq_len": L, "ours_us": ours[0], "ours_p20": ours[1], "ours_p80": ours[2], "WX_us": wx[0], "torch_us": tch[0],
               "hbm_GBps": nbytes / ours[0] * 1e-3, "hbm_frac": nbytes / ours[0] * 1e-3 / 8000.0,
               "mfma_TFLOPs": flops / ours[0] * 1e-6, "mfma_frac": flops / ours[0] * 1e-6 / 2500.0}
        rows.append(row)
        row["ours_back_to_back_us"] = ours_total
        print(f"L={L:7d}  ours {ours[0]:9.1f} us (p20 {ours[1]:.1f}, p80 {ours[2]:.1f}; {ours_total:.1f} us/call back-to-back)   WX {wx[0]:9.1f} us   "
              f"torch {tch[0]:9.1f} us   "
              f"{row['hbm_GBps']:7.0f} GB/s ({100 * row['hbm_frac']:.1f}% of 8 TB/s)   "
              f"{row['mfma_TFLOPs']:6.0f} TF ({100 * row['mfma_frac']:.1f}% of 2.5 PF)")
        del X, org_X
    if args.save_path:
        os.makedirs(args.save_path, exist_ok=True)
        fn = os.path.join(args.save_path, f"low-rank-rank-{args.total_rank}-group-{G}.csv")
        with open(fn, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["seq_len", "WX", "Torch", "Ours", "Ours_p20", "Ours_p80", "hbm_GBps", "mfma_TFLOPs"])
            for r in rows:
                wr.writerow([r["seq_len"], f"{r['WX_us']:.3f}", f"{r['torch_us']:.3f}", f"{r['ours_us']:.3f}",
                             f"{r['ours_p20']:.3f}", f"{r['ours_p80']:.3f}", f"{r['hbm_GBps']:.1f}", f"{r['mfma_TFLOPs']:.1f}"])
        print("saved", fn)
    if args.json:
        print(json.dumps(rows))


if __name__ == "__main__":
    main()
