#!/usr/bin/env python3
"""Headline benchmark: one low-rank-KV attention decode step on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload ("step" = one pass of the hot path over one token): Llama-2-7B attention geometry as built by
the reference harness (run_latency_attention.py:40-55: hidden 4096, 32 heads x 128, theta 1e4), Palu
rank_k=1024, rank_v=3072, group_size=4 (G=8 latent groups), fp16 latent KV cache of prompt_len=65536
positions, batch 1; synthetic randn latents / token and random-init weights (no checkpoints offline),
all resident in HBM before the timed region.  One step = qkv GEMV + q-RoPE + in-place cache append ->
fused reconstruct-K/RoPE/q.K^T (abx) -> softmax + latent P.V -> o_proj GEMV, i.e. the decode branch of
kernel/palu_attention.py:147-263 through the C ABI (palu_decode_step_f16).

N > 1: head-group parallel (strong scaling of the same step): rank r owns G/N groups; one RCCL collective per
step -- `--oproj replicated` (default, BASELINE.json config 5's wording): all-gather of the [H/N*Rv] fp16 context
slices and the full o_proj on every rank; `--oproj sharded`: every rank multiplies its context slice by its 1/N
column block of W_o' and the ranks all-reduce the [hidden] fp32 partials (16 KiB).  The other form is timed beside
the headline one (`oproj_alt`).  `collective_us` is the collective alone (CUDA events, max over ranks).
`python bench.py --gpus N` without torch.distributed.run spawns its N ranks itself (one JSON line from rank 0).
`--dry_plan`: the sharding plan, the weight shards and the step's collective at the real message sizes over gloo on
the CPU -- no kernels (a multi-process plumbing check that runs without GPUs).

Prints ONE JSON line (rank 0).  `value` = decode-step microseconds (lower is better); `roofline` is the
dominant kernel of the step by measured time, `roofline_abx` the fused score kernel the north star
targets (both HBM fraction from algorithmic bytes and MFMA fraction from algorithmic flops);
`cpu_baseline` = the CPU oracle (a port of the reference's PyTorch path) timed on this host's cores.
"""
from __future__ import annotations

import argparse
import json
import math
import os
# hipGraph replay: ROCm 7.2's graph "packet capture" path (on by default) costs ~3.5 us per replay of this 5-kernel step
# (157.3 us against 153.5 with it off, direct launches 152.7: profiles/r05_graph_replay.txt); it is read when the HIP runtime
# loads, so it must be set before torch is imported.  An explicit setting in the environment wins.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA (no sparsity)
MFMA_WALL_TFLOPS = 1630.0     # measured: MFMA-only random-operand fp16 stream, 1.69 GHz at 92 % MFMA busy (profiles/r02_ubench_issue_pmc_clock.txt)

H, D, GS, HIDDEN = 32, 128, 4, 4096
G = H // GS


def algorithmic(rank_k, rank_v, L):
    """SURVEY.md 8(d) per-launch algorithmic bytes / flops (L = cached positions incl. the new token)."""
    Rk, Rv = rank_k // G, rank_v // G
    abx_b = 2 * G * L * Rk + 2 * H * Rk * D + 2 * H * D + 2 * H * L
    abx_f = 2 * H * L * Rk * D + 5 * H * L * D
    pv_b = 2 * G * L * Rv + 2 * H * L + 2 * H * Rv
    pv_f = 2 * H * L * Rv
    qkv_b = 2 * HIDDEN * (H * D + rank_k + rank_v) + 2 * HIDDEN
    o_b = 2 * HIDDEN * H * Rv + 2 * H * Rv
    return {"abx": (abx_b, abx_f), "softmax_pv": (pv_b, pv_f), "qkv": (qkv_b, 2 * HIDDEN * (H * D + rank_k + rank_v)),
            "o_proj": (o_b, 2 * HIDDEN * H * Rv), "step": (abx_b + pv_b + qkv_b + o_b, abx_f + pv_f)}


def baseline_metric(rank_k, Lp):
    """BASELINE.json's metric string verbatim when this run is the configuration it is quoted on (rank_k 1024, 64k),
    otherwise the same wording with the actual numbers.  The geometry is what run_latency_attention.py builds
    (LlamaConfig() defaults: 32 heads, MHA, hidden 4096 -- SURVEY.md F8), spelled out in config.workload."""
    try:
        m = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        if rank_k == 1024 and Lp == 65536:
            return m
    except Exception:                        # noqa: BLE001 -- the file is absent on a bare checkout
        pass
    return "decode-step us + achieved HBM GB/s, rank_k=%d prompt_len=%dk" % (rank_k, Lp // 1024)


REPS = 7                      # repetitions of the K-step timed loop; the headline is their median


def _gate():
    """~100 us of device-side spin in front of a timed loop: the host queues the loop's launches while the GPU is still
    busy, so the device-event time between the two records is back-to-back execution from a full queue -- not the host's
    first-launch latency (3-16 us per forward, MI355X_MICROARCH.md `graph-replay-floor`) amortised over K steps."""
    try:
        torch.cuda._sleep(200_000)
    except Exception:                      # noqa: BLE001 -- private torch API: without it the events see the first launch latency
        pass


def time_reps(fn, steps, warmup, sync, reps=REPS, settle_ms=20.0):
    """`warmup` untimed calls, an internal settle phase to a steady clock (>= settle_ms of further calls), then `reps`
    repetitions, each timing EXACTLY `steps` calls twice: (a) host wall clock between two sync()s, (b) device events on
    the launch stream between two sync()s, recorded behind a short device-side gate (see _gate).  Returns the
    per-repetition lists (host wall ms, device-event ms) over `steps` calls."""
    for _ in range(warmup):
        fn()
    sync()
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < settle_ms and n < 4096:
        fn()
        n += 1
        if n % 64 == 0:
            torch.cuda.synchronize()
    walls, evs = [], []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        walls.append((time.perf_counter() - t0) * 1e3)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _gate()
        ev0.record()
        for _ in range(steps):
            fn()
        ev1.record()
        sync()
        evs.append(ev0.elapsed_time(ev1))
    return walls, evs


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def time_loop(fn, steps, warmup, sync, reps=3):
    """(median host wall ms, median device-event ms) over `steps` calls -- the sub-record / per-kernel form"""
    walls, evs = time_reps(fn, steps, warmup, sync, reps=reps, settle_ms=5.0)
    return _median(walls), _median(evs)


def cpu_baseline(rank_k, rank_v, L, reps=3):
    """The CPU oracle's decode step (port of the reference's PyTorch path) at the SAME shapes on this host: 1 warm-up +
    `reps` timed runs.  Returns (min us, median us, inputs, output) -- the inputs/output feed the GPU parity check."""
    import oracle
    torch.manual_seed(0)
    Rk, Rv = rank_k // G, rank_v // G
    w = {"wq": torch.randn(H * D, HIDDEN).mul_(1 / 64).half(), "vt_k": torch.randn(rank_k, HIDDEN).mul_(1 / 64).half(),
         "vt_v": torch.randn(rank_v, HIDDEN).mul_(1 / 64).half(), "b": torch.randn(H, Rk, D).mul_(Rk ** -0.5).half(),
         "wo": torch.randn(HIDDEN, H * Rv).mul_(0.01).half()}
    k = torch.randn(G, L - 1, Rk).half()
    v = torch.randn(G, L - 1, Rv).half()
    tok = torch.randn(HIDDEN).half()
    times = []
    out = None
    with torch.no_grad():
        for i in range(reps + 1):
            t0 = time.perf_counter()
            out = oracle.decode_step(tok, L - 1, w, k, v)[0]
            times.append(time.perf_counter() - t0)
    ts = sorted(times[1:])
    return ts[0] * 1e6, ts[len(ts) // 2] * 1e6, (w, k, v, tok), out


def gpu_step_parity(inputs, ref_out, L):
    """The HIP decode step on the SAME inputs the CPU oracle just ran: max |gpu - oracle| of the attention output."""
    from palu_amd.kernel import head_parallel as hp
    w, k, v, tok = inputs
    dev = torch.device("cuda", torch.cuda.current_device())
    Rk, Rv = k.shape[2], v.shape[2]
    plan = hp.make_plan(1, 0, H, G, D, Rk, Rv)
    wd = {n: t.to(dev) for n, t in w.items()}
    cap = (L + 64 + 63) // 64 * 64
    kc = torch.zeros(G, cap, Rk, dtype=torch.float16, device=dev)
    vc = torch.zeros(G, cap, Rv, dtype=torch.float16, device=dev)
    kc[:, :L - 1] = k.to(dev)
    vc[:, :L - 1] = v.to(dev)
    dec = hp.HeadParallelDecoder(plan, wd, kc, vc, HIDDEN)
    out = dec.step(tok.to(dev), L - 1, L - 1)
    torch.cuda.synchronize()
    d = (out.float().cpu() - ref_out.float()).abs()
    return float(d.max()), float(ref_out.float().abs().max())


def best_form(fn, steps, warmup=10):
    """(us per call, form) of the faster of: direct launches, replay of a captured hipGraph of `fn` -- device-event medians.
    On this stack direct launches from a full queue beat the replay of the same kernels by 4-5 us per step
    (profiles/r03_launch_ab.txt); both are the same kernels on the same stream."""
    _, d_ms = time_loop(fn, steps, warmup, torch.cuda.synchronize)
    best = (d_ms * 1e3 / steps, "direct")
    try:
        replay = graph_of(fn)
        _, g_ms = time_loop(replay, steps, warmup, torch.cuda.synchronize)
        if g_ms < d_ms:
            best = (g_ms * 1e3 / steps, "graph_replay")
    except Exception:                                        # noqa: BLE001
        torch.cuda.synchronize()
    return best


def graph_of(fn, warm=3):
    """Capture one call of `fn` (kernel launches on the current stream, no allocation) into a hipGraph; returns replay."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def alg_bytes_q(rank_k, rank_v, L, bits):
    """SURVEY.md 8(d) with packed latents: codes b/8 bytes per value + 4 bytes (scale, zero) per (row, group)."""
    Rk, Rv = rank_k // G, rank_v // G
    kb = G * L * Rk * bits // 8 + 4 * G * L
    vb = G * L * Rv * bits // 8 + 4 * G * L
    abx_b = kb + 2 * H * Rk * D + 2 * H * D + 2 * H * L
    pv_b = vb + 2 * H * L + 2 * H * Rv
    qkv_b = 2 * HIDDEN * (H * D + rank_k + rank_v) + 2 * HIDDEN
    o_b = 2 * HIDDEN * H * Rv + 2 * H * Rv
    return {"abx": abx_b, "softmax_pv": pv_b, "step": abx_b + pv_b + qkv_b + o_b}


def bench_quant_config(name, rank_k, rank_v, Lp, bits, steps, dev, parity=True):
    """BASELINE configs 3 / 4: the decode step on a packed 3/4-bit latent cache (palu_decode_step_q): step us (graph
    replay) + the two attention kernels on their own, and -- `parity` -- the step's output against
    oracle.decode_step(latent_bits=bits) on the SAME inputs at the full size (the oracle fake-quantises the cached rows
    with quantize_rows, the GPU packs the same fp16 rows with palu_quantize_pack: bit-identical codes).  Hadamard
    (config 3) is folded into the weights offline (LlamaPaluAttention.fuse_hadamard): with random-init weights it
    changes no shape and adds no kernel -- the random weights stand for the rotated ones on both sides."""
    from palu_amd import _lib
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq
    from palu_amd.kernel import quant as q
    lib = _lib.lib
    Rk, Rv = rank_k // G, rank_v // G
    L = Lp + 1
    cap = (Lp + 64 + 63) // 64 * 64
    torch.manual_seed(4321)
    wc = {"wq": (torch.randn(H * D, HIDDEN) / 64).half(), "vt_k": (torch.randn(rank_k, HIDDEN) / 64).half(),
          "vt_v": (torch.randn(rank_v, HIDDEN) / 64).half(), "b": (torch.randn(H, Rk, D) * Rk ** -0.5).half(),
          "wo": (torch.randn(HIDDEN, H * Rv) * 0.01).half()}
    k_cpu = torch.randn(G, Lp, Rk).half()
    v_cpu = torch.randn(G, Lp, Rv).half()
    tok = torch.randn(HIDDEN).half()
    ref = None
    if parity:
        import oracle
        t0 = time.perf_counter()
        with torch.no_grad():
            kq = oracle.quantize_rows(k_cpu.reshape(-1, Rk), bits)[0].reshape(G, Lp, Rk)
            vq = oracle.quantize_rows(v_cpu.reshape(-1, Rv), bits)[0].reshape(G, Lp, Rv)
            ref = oracle.decode_step(tok, Lp, wc, kq, vq, latent_bits=bits)[0]
        del kq, vq
        oracle_s = time.perf_counter() - t0
    wq, vtk, vtv, b, wo = (wc[n].to(dev) for n in ("wq", "vt_k", "vt_v", "b", "wo"))
    kf = torch.zeros(G, cap, Rk, device=dev, dtype=torch.float16)
    vf = torch.zeros(G, cap, Rv, device=dev, dtype=torch.float16)
    kf[:, :Lp] = k_cpu.to(dev)
    vf[:, :Lp] = v_cpu.to(dev)
    kc, km = q.quantize_pack(kf, bits)
    vc, vm = q.quantize_pack(vf, bits)
    del kf, vf, k_cpu, v_cpu
    hidden = tok.to(dev)
    frag = prepare_b(b, G)
    inv = rope_inv_freq(dev)
    ws = torch.zeros(lib.palu_decode_workspace_bytes(H, G, D, cap + 8, Rv), dtype=torch.uint8, device=dev)
    out = torch.empty(HIDDEN, dtype=torch.float16, device=dev)
    scores = torch.empty(H, (L + 7) // 8 * 8, dtype=torch.float16, device=dev)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=dev)
    pvws = torch.empty(lib.palu_pv_workspace_bytes(H, G, cap, Rv), dtype=torch.uint8, device=dev)
    q_buf = torch.randn(H * D, device=dev).half()
    s = _lib.current_stream

    def step():
        _lib.check(lib.palu_decode_step_q(
            hidden.data_ptr(), wq.data_ptr(), wq.stride(0), vtk.data_ptr(), vtk.stride(0), vtv.data_ptr(), vtv.stride(0),
            frag.data_ptr(), wo.data_ptr(), wo.stride(0),
            kc.data_ptr(), kc.stride(0), kc.stride(1), km.data_ptr(), km.stride(0), km.stride(1),
            vc.data_ptr(), vc.stride(0), vc.stride(1), vm.data_ptr(), vm.stride(0), vm.stride(1),
            0, inv.data_ptr(), out.data_ptr(), 0, 0, ws.data_ptr(), cap + 8, H, G, D, HIDDEN, Rk, Rv, bits, Lp, Lp,
            s()), "decode_step_q")

    def k_abx():
        _lib.check(lib.palu_abx_rope_q(q_buf.data_ptr(), D, 1, frag.data_ptr(), kc.data_ptr(), kc.stride(0), kc.stride(1),
                                       km.data_ptr(), km.stride(0), km.stride(1), scores.data_ptr(), scores.stride(0),
                                       H, G, L, Rk, D, bits, inv.data_ptr(), 0, s()), "abx_q")

    def k_pv():
        _lib.check(lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, vc.data_ptr(), vc.stride(0), vc.stride(1),
                                         vm.data_ptr(), vm.stride(0), vm.stride(1), ctx.data_ptr(), 0, 0, pvws.data_ptr(),
                                         H, G, L, Rv, bits, math.sqrt(D), s()), "pv_q")
    par = None
    if ref is not None:
        step()
        torch.cuda.synchronize()
        d = float((out.float().cpu() - ref.float()).abs().max())
        sc = float(ref.float().abs().max())
        par = {"max_abs_diff_vs_oracle": round(d, 6), "oracle_out_max_abs": round(sc, 5), "positions": L,
               "oracle": "oracle.decode_step(latent_bits=%d) on quantize_rows() of the same fp16 latents" % bits,
               "tolerance": "rtol=atol=1e-3 (test_palu_attention.py:183-195)", "ok": bool(d <= 1e-3 + 1e-3 * sc),
               "oracle_seconds": round(oracle_s, 1)}
    us, form = best_form(step, steps)
    alg = alg_bytes_q(rank_k, rank_v, L, bits)
    rec = {"workload": "%s: rank_k=%d rank_v=%d prompt_len=%d %d-bit packed latents (asym, per (token, group) row)"
                       % (name, rank_k, rank_v, Lp, bits),
           "step_us": round(us, 2), "launch": form, "step_algorithmic_bytes": alg["step"],
           "step_hbm_frac": round(alg["step"] / us * 1e-3 / HBM_PEAK_GBPS, 4), "kernels": {}, "parity": par}
    for kn, fn in (("abx", k_abx), ("softmax_pv", k_pv)):
        _, kms = time_loop(fn, 50, 5, torch.cuda.synchronize)
        kus = kms * 1e3 / 50
        rec["kernels"][kn] = {"us": round(kus, 2), "algorithmic_bytes": alg[kn],
                              "hbm_frac": round(alg[kn] / kus * 1e-3 / HBM_PEAK_GBPS, 4)}
    return rec


def bench_shared_b(rank_k, rank_v, Lp, steps, dev):
    """Config 2 with the B factor TIED inside every latent group (true-GQA checkpoints: the query heads of a group share
    one KV head, palu/model/svd_mistral/modeling_palu_mistral.py:37-59): palu_decode_step_sharedb_f16 reconstructs the
    keys once per group.  The one configuration in which the contract's HBM target for the score kernel is reachable."""
    from palu_amd import _lib
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq
    lib = _lib.lib
    Rk, Rv = rank_k // G, rank_v // G
    L = Lp + 1
    cap = (Lp + 64 + 63) // 64 * 64
    torch.manual_seed(777)
    wq = (torch.randn(H * D, HIDDEN, device=dev) / 64).half()
    vtk = (torch.randn(rank_k, HIDDEN, device=dev) / 64).half()
    vtv = (torch.randn(rank_v, HIDDEN, device=dev) / 64).half()
    bg = (torch.randn(G, Rk, D, device=dev) * Rk ** -0.5).half()          # one factor per group
    wo = (torch.randn(HIDDEN, H * Rv, device=dev) * 0.01).half()
    kc = torch.randn(G, cap, Rk, device=dev, dtype=torch.float16)
    vc = torch.randn(G, cap, Rv, device=dev, dtype=torch.float16)
    hidden = torch.randn(HIDDEN, device=dev, dtype=torch.float16)
    frag = prepare_b(bg, G)                                               # fragments of the [G, R, D] shared factor
    inv = rope_inv_freq(dev)
    ws = torch.zeros(lib.palu_decode_workspace_bytes(H, G, D, cap + 8, Rv), dtype=torch.uint8, device=dev)
    out = torch.empty(HIDDEN, dtype=torch.float16, device=dev)
    scores = torch.empty(H, (L + 7) // 8 * 8, dtype=torch.float16, device=dev)
    q_buf = torch.randn(H * D, device=dev).half()
    s = _lib.current_stream

    def step():
        _lib.check(lib.palu_decode_step_sharedb_f16(
            hidden.data_ptr(), wq.data_ptr(), wq.stride(0), vtk.data_ptr(), vtk.stride(0), vtv.data_ptr(), vtv.stride(0),
            frag.data_ptr(), wo.data_ptr(), wo.stride(0), kc.data_ptr(), kc.stride(0), kc.stride(1),
            vc.data_ptr(), vc.stride(0), vc.stride(1), 0, inv.data_ptr(), out.data_ptr(), 0, 0, ws.data_ptr(), cap + 8,
            H, G, D, HIDDEN, Rk, Rv, Lp, Lp, s()), "decode_step_sharedb")

    def k_abx():
        _lib.check(lib.palu_abx_rope_shared_f16(q_buf.data_ptr(), D, 1, frag.data_ptr(), kc.data_ptr(), kc.stride(0),
                                                kc.stride(1), scores.data_ptr(), scores.stride(0), H, G, L, Rk, D,
                                                inv.data_ptr(), 0, s()), "abx_shared")
    us, form = best_form(step, steps)
    _, kms = time_loop(k_abx, 50, 5, torch.cuda.synchronize)
    kus = kms * 1e3 / 50
    ab_b = 2 * G * L * Rk + 2 * G * Rk * D + 2 * H * D + 2 * H * L          # one B factor per group
    alg = algorithmic(rank_k, rank_v, L)
    step_b = alg["step"][0] - alg["abx"][0] + ab_b
    return {"workload": "C2 shapes with B shared by the %d heads of a group (true-GQA weights): rank_k=%d rank_v=%d "
                        "prompt_len=%d fp16" % (GS, rank_k, rank_v, Lp),
            "step_us": round(us, 2), "launch": form, "step_algorithmic_bytes": step_b,
            "step_hbm_frac": round(step_b / us * 1e-3 / HBM_PEAK_GBPS, 4),
            "kernels": {"abx_shared": {"us": round(kus, 2), "algorithmic_bytes": ab_b,
                                       "hbm_GBps": round(ab_b / kus * 1e-3, 1),
                                       "hbm_frac": round(ab_b / kus * 1e-3 / HBM_PEAK_GBPS, 4)}}}


def bench_prefill(rank_k, rank_v, T, dev):
    """the prompt pass of ONE attention module (LlamaPaluAttention.forward, q_len = T, empty cache) through
    the flash-style prefill kernel -- ms and peak transient memory (above weights, input, output and the cache) for the
    one-launch form (K~ of every head, V^T of every group, every context row at once: the default up to 6 GiB of transients),
    the bounded-workspace form (query chunks x latent groups), fp16 and packed 4-bit caches, and the kv-panel form (carried
    softmax state: transients independent of the prompt length)."""
    from torch import nn
    from palu_amd.kernel.palu_attention import LatentCache, LlamaPaluAttention, QuantLatentCache, build_b

    class Cfg:
        pass
    cfg = Cfg()
    cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = HIDDEN, H, False
    cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = GS, G, rank_k, rank_v
    torch.manual_seed(0)
    with torch.device(dev):
        m = LlamaPaluAttention(cfg, 0).half()
        with torch.no_grad():
            for lin in (m.q_proj, m.k_proj.VT, m.v_proj.VT, m.o_proj):
                lin.weight.normal_(0.0, 0.02)
            for u in m.k_proj.U_list:
                u.weight.normal_(0.0, (rank_k // G) ** -0.5)
        m.k_proj.B = nn.Parameter(build_b([u.weight for u in m.k_proj.U_list], GS, D))
    m = m.eval().prepare_decode()
    x = torch.randn(1, T, HIDDEN, device=dev, dtype=torch.float16)
    rec = {"workload": "prompt pass of one attention module, %d tokens, rank_k=%d rank_v=%d gs=%d, causal" % (T, rank_k, rank_v, GS)}
    # "fp16_cache" = the module's default: from 256 MiB of workspace-form transients on, the LATENT form (csrc/prefill_lat.hip: keys
    # rebuilt per kv tile inside the flash kernel, V from the cache rows; transients = one 2048-query chunk); "..._workspace_form" =
    # the one-launch form with its [H, kv, D] key workspace and transposed value copy (PREFILL_LATENT_ABOVE = None)
    for tag, bits, budget in (("fp16_cache", 16, None), ("fp16_cache_workspace_form", 16, "ws"), ("fp16_cache_bounded_workspace", 16, 0),
                              ("packed_4bit_cache", 4, None), ("packed_3bit_cache", 3, None), ("packed_4bit_cache_bounded_workspace", 4, 0),
                              ("packed_4bit_cache_kv_panels", 4, "panels")):
        mk = (lambda: LatentCache(capacity=T + 512)) if bits >= 16 else (lambda: QuantLatentCache(bits, capacity=T + 512))
        panels = budget == "panels"         # kv panels with carried softmax state: transients independent of T (<= 64 MiB)
        if panels:
            budget = None
            m.PREFILL_PANEL_ROWS = 2048
        nolat = budget == "ws" or budget == 0
        if budget == "ws":
            budget = None
        if nolat:
            m.PREFILL_LATENT_ABOVE = None
        if budget is not None:
            m.PREFILL_WORKSPACE_BUDGET = budget
        try:
            with torch.no_grad():
                def fresh():
                    c = mk()
                    if bits >= 16:
                        c.reserve(0, T + 512, torch.empty((1, G, 0, rank_k // G), dtype=torch.float16, device=dev),
                                  torch.empty((1, G, 0, rank_v // G), dtype=torch.float16, device=dev))
                    else:
                        c.reserve(0, T + 512, G, rank_k // G, rank_v // G, x.device)
                    return c
                c = fresh()
                out, _, _ = m(x, past_key_value=c, is_causal=True)      # untimed: the caching allocator gets its blocks
                del out, c
                ts, extra = [], 0
                for _ in range(2):
                    c = fresh()
                    torch.cuda.synchronize()
                    torch.cuda.reset_peak_memory_stats()
                    base = torch.cuda.memory_allocated()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    out, _, _ = m(x, past_key_value=c, is_causal=True)
                    e1.record()
                    torch.cuda.synchronize()
                    extra = torch.cuda.max_memory_allocated() - base - out.numel() * 2
                    ts.append(e0.elapsed_time(e1))
                    del out, c
                ms = min(ts)
                rec[tag] = {"ms": round(ms, 2), "transient_MiB": round(extra / 2 ** 20, 1),
                            "causal_TFLOPs": round((2.0 * H * T * T / 2 * (D + rank_v // G)) / (ms * 1e-3) * 1e-12, 1)}
        finally:
            if budget is not None:
                del m.PREFILL_WORKSPACE_BUDGET
            if panels:
                del m.PREFILL_PANEL_ROWS
            if nolat:
                del m.PREFILL_LATENT_ABOVE
        torch.cuda.empty_cache()
    return rec


def bench_c5_slice(steps, dev):
    """BASELINE config 5, what ONE of the 8 GPUs runs per step: its latent group (G=1, H=4) of the rank 1024/3072 model at
    prompt_len 256k -- qkv for its heads, attention core (the fused single-kernel path is selected for G=1), no o_proj
    (that follows the all-gather).  The RCCL leg needs 8 GPUs and is measured by `bench.py --gpus 8`."""
    from palu_amd import _lib
    from palu_amd.kernel import head_parallel as hp
    Lp, Rk, Rv = 262144, 128, 384
    plan = hp.make_plan(8, 0, H, G, D, Rk, Rv)
    torch.manual_seed(99)
    full = {"wq": (torch.randn(H * D, HIDDEN, device=dev) / 64).half(),
            "vt_k": (torch.randn(G * Rk, HIDDEN, device=dev) / 64).half(),
            "vt_v": (torch.randn(G * Rv, HIDDEN, device=dev) / 64).half(),
            "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": torch.zeros(1, 1, device=dev).half()}
    w = {k_: v_.contiguous() for k_, v_ in hp.shard_weights(plan, full).items()}
    cap = Lp + 64
    kc = torch.randn(1, cap, Rk, device=dev, dtype=torch.float16)
    vc = torch.randn(1, cap, Rv, device=dev, dtype=torch.float16)
    hidden = torch.randn(HIDDEN, device=dev, dtype=torch.float16)
    dec = hp.HeadParallelDecoder(plan, w, kc, vc, HIDDEN)
    us, form = best_form(lambda: dec.local_step(hidden, Lp, Lp), steps)
    L = Lp + 1
    byt = 2 * L * (Rk + Rv) + 2 * 4 * Rk * D + 2 * HIDDEN * (4 * D + Rk + Rv)
    return {"workload": "config-5 per-GPU slice: G=1 H=4 rank_k=1024/8 rank_v=3072/8 prompt_len=262144 fp16 (qkv + attention "
                        "core of one rank, before the all-gather)",
            "attend_us": round(us, 2), "launch": form, "algorithmic_bytes": byt, "hbm_frac": round(byt / us * 1e-3 / HBM_PEAK_GBPS, 4),
            "fused_attention_core": bool(_lib.lib.palu_decode_attn_preferred(4, 1, L, Rk, Rv, D))}


def executed_two_band_flops(Rk, L):
    """Matrix flops the two-band score kernels EXECUTE per launch: per 128-position tile and latent group
    36 NKS v_mfma_f32_32x32x16_f16 (8 high-band M-blocks + the low band's stage 2, NKS = R / 16 k-steps, 4 blocks of 32
    positions) and 8 NKS v_mfma_f32_16x16x32_f16 (stage 1)."""
    nks = Rk // 16
    tiles = (L + 127) // 128
    return G * tiles * (36 * nks * 2 * 32 * 32 * 16 + 8 * nks * 2 * 16 * 16 * 32)


def dry_plan(args, world, rank):
    """`--dry_plan`: everything of an N-rank run that needs no GPU -- the sharding plan, the weight shards of both o_proj
    forms (shapes and that the shards tile the full weights), and the step's collective at the real message sizes over
    gloo -- then rank 0 prints ONE JSON line with the contract's keys (value = null: nothing was timed)."""
    import torch.distributed as dist
    from palu_amd.kernel import head_parallel as hp
    rank_k, rank_v, Lp = args.rank_k, args.rank_v, args.prompt_len
    Rk, Rv = rank_k // G, rank_v // G
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    plan = hp.make_plan(world, rank, H, G, D, Rk, Rv)
    torch.manual_seed(1234)
    hid = 256                                           # (the plan does not depend on the hidden size: small weights)
    full = {"wq": torch.randn(H * D, hid).half(), "vt_k": torch.randn(rank_k, hid).half(), "vt_v": torch.randn(rank_v, hid).half(),
            "b": torch.randn(H, Rk, D).half(), "wo": torch.randn(hid, H * Rv).half()}
    shapes = {}
    for form in ("replicated", "sharded"):
        w = hp.shard_weights(plan, full, oproj=form)
        shapes[form] = {k: list(v.shape) for k, v in w.items()}
        assert w["wq"].shape[0] == plan.heads_local * D and w["b"].shape[0] == plan.heads_local
        assert w["vt_k"].shape[0] == plan.groups_local * Rk and w["vt_v"].shape[0] == plan.groups_local * Rv
        assert w["wo"].shape[1] == (H * Rv if form == "replicated" else plan.heads_local * Rv)
    checks = {"world": world}
    if world > 1:
        # the collective of each form at the step's real message size
        ctx = torch.full((plan.heads_local * Rv,), float(rank + 1), dtype=torch.float16)
        allc = torch.empty(H * Rv, dtype=torch.float16)
        hp.DistExchange(None).all_gather_into(allc, ctx)
        ok_g = all(float(allc[r * plan.heads_local * Rv]) == r + 1 for r in range(world))
        part = torch.full((HIDDEN,), float(rank + 1), dtype=torch.float32)
        hp.DistExchange(None).all_reduce_sum_(part)
        ok_r = float(part[0]) == world * (world + 1) / 2
        # the shards tile the full weight: the sum over ranks of the sharded o_proj partials equals the replicated product
        c_full = torch.randn(H * Rv, generator=torch.Generator().manual_seed(7)).half()
        mine = (hp.shard_weights(plan, full, "sharded")["wo"].float()
                @ c_full[plan.head0 * Rv:(plan.head0 + plan.heads_local) * Rv].float())
        dist.all_reduce(mine)
        ok_w = bool(torch.allclose(mine, full["wo"].float() @ c_full.float(), rtol=1e-4, atol=1e-2))
        checks.update({"all_gather_ok": ok_g, "all_reduce_ok": ok_r, "sharded_oproj_sums_to_replicated": ok_w,
                       "all_gather_message_bytes": plan.heads_local * Rv * 2, "all_reduce_message_bytes": HIDDEN * 4})
        dist.barrier()
    if rank == 0:
        print(json.dumps({
            "metric": baseline_metric(rank_k, Lp), "value": None, "unit": "us", "n_gpus": world, "steps": 0, "warmup": 0,
            "ms_per_step": None, "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic", "dry_plan": True,
            "config": {"workload": "dry plan (no kernels): head-group sharding of rank_k=%d rank_v=%d prompt_len=%d over %d ranks"
                                   % (rank_k, rank_v, Lp, world),
                       "parallelism": "head-group x%d, oproj=%s" % (world, args.oproj)},
            "plan": {"groups_local": plan.groups_local, "heads_local": plan.heads_local, "ctx_local": plan.ctx_local,
                     "cache_rows_per_rank": plan.groups_local * (Lp + 1), "shards": shapes},
            "checks": checks}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    bad = [k for k, v in checks.items() if v is False]
    if bad:
        raise SystemExit("dry plan failed: %s" % bad)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--rank_k", type=int, default=1024)
    ap.add_argument("--rank_v", type=int, default=3072)
    ap.add_argument("--prompt_len", type=int, default=65536)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--oproj", choices=("replicated", "sharded"), default="replicated",
                    help="N > 1: all-gather + replicated o_proj, or column-sharded o_proj + all-reduce of the [hidden] partials")
    ap.add_argument("--exchange", choices=("rccl", "p2p"), default="rccl",
                    help="N > 1: the step's one collective through RCCL (torch.distributed) or through the one-shot "
                         "peer-to-peer exchange kernel over hipIpc buffers (csrc/exchange.hip); the other one is timed alone "
                         "and reported beside it")
    ap.add_argument("--no_graph", action="store_true", help="launch the step directly instead of replaying a captured hipGraph")
    ap.add_argument("--no_extra_configs", action="store_true", help="skip the config 3/4/5 sub-records")
    ap.add_argument("--no_abx_sweep", action="store_true",
                    help="skip the score-kernel records at other shapes / on the other kernels (profiling runs: one shape per kernel name)")
    ap.add_argument("--no_model32", action="store_true", help="skip the whole-model (32-layer) decode sub-record")
    ap.add_argument("--cpu_sample_len", type=int, default=0, help="positions used for the CPU baseline (0 = full)")
    ap.add_argument("--dry_plan", action="store_true",
                    help="no kernels: sharding plan + weight shards + the step's collective over gloo on the CPU, one JSON line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: spawn the N ranks (one per GPU) and hand their output through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_plan:
        return dry_plan(args, world, rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from palu_amd import _lib
    from palu_amd.kernel import head_parallel as hp
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

    rank_k, rank_v, Lp = args.rank_k, args.rank_v, args.prompt_len
    Rk, Rv = rank_k // G, rank_v // G
    L = Lp + 1
    cap = (Lp + 64 + 63) // 64 * 64
    torch.manual_seed(1234)
    plan = hp.make_plan(world, rank, H, G, D, Rk, Rv)
    full = {"wq": (torch.randn(H * D, HIDDEN, device=dev) / 64).half(),
            "vt_k": (torch.randn(rank_k, HIDDEN, device=dev) / 64).half(),
            "vt_v": (torch.randn(rank_v, HIDDEN, device=dev) / 64).half(),
            "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": (torch.randn(HIDDEN, H * Rv, device=dev) * 0.01).half()}
    w = {k: (v.contiguous() if k != "wo" else v) for k, v in hp.shard_weights(plan, full, oproj=args.oproj).items()}
    alt_oproj = "sharded" if args.oproj == "replicated" else "replicated"
    w_alt = None
    if world > 1:            # the step's other collective form, timed beside the headline one
        w_alt = dict(w)
        w_alt["wo"] = hp.shard_weights(plan, full, oproj=alt_oproj)["wo"]
    del full
    Gl = plan.groups_local
    k_cache = torch.randn(Gl, cap, Rk, device=dev, dtype=torch.float16)       # run_latency_attention.py:62-63
    v_cache = torch.randn(Gl, cap, Rv, device=dev, dtype=torch.float16)
    hidden = torch.randn(HIDDEN, device=dev, dtype=torch.float16)             # :70
    p2p = None
    if world > 1:
        try:
            p2p = hp.IpcExchange(rank, world, max(HIDDEN * 4, plan.heads_local * Rv * 2), dev)
        except Exception as e:                                  # noqa: BLE001 -- e.g. IPC not available between these devices
            print("bench.py[rank %d]: peer-to-peer exchange unavailable (%s)" % (rank, repr(e)[:200]), file=sys.stderr)
        if args.exchange == "p2p":
            okt = torch.tensor([1 if p2p is not None else 0], device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if not int(okt.item()):
                raise SystemExit("--exchange p2p: the exchange could not be set up on every rank")
    dec = hp.HeadParallelDecoder(plan, w, k_cache, v_cache, HIDDEN, exchange=p2p if args.exchange == "p2p" else None)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        dec.step(hidden, Lp, Lp)

    sync()
    # A/B of the two launch forms of the SAME five kernels: direct launches (two C-ABI calls per step) and the replay of
    # a captured hipGraph of the step (what the reference harness times with --cache_graph, run_latency_attention.py:81-90).
    # Every form: W warm-up steps + settle, then REPS repetitions of exactly K steps; medians; the headline is the faster
    # form's DEVICE-EVENT time (max over ranks), its host wall time is reported beside it.
    forms = {"direct": step}
    if not args.no_graph:
        # N > 1: the step's RCCL collective is captured with it (torch.distributed's NCCL backend records collectives
        # into the capturing stream), so a replay issues no Python-side launch at all.  Every rank must take the same
        # decision, or the timed loops would issue different numbers of collectives: agree on min(success) first.
        replay, ok = None, 1
        try:
            replay = graph_of(step)
        except Exception as e:                              # noqa: BLE001 -- e.g. a collective that cannot be captured
            ok = 0
            print("bench.py[rank %d]: graph capture of the step failed (%s)" % (rank, repr(e)[:200]), file=sys.stderr)
        if dist is not None:
            t = torch.tensor([ok], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if ok:
            forms["graph_replay"] = replay
    ab = {}
    for name, fn in forms.items():
        walls, evs = time_reps(fn, args.steps, args.warmup, sync)
        t = torch.tensor([walls, evs], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        walls, evs = t[0].tolist(), t[1].tolist()
        sev = sorted(evs)
        ab[name] = {"device_us": round(_median(evs) * 1e3 / args.steps, 2), "device_us_min": round(min(evs) * 1e3 / args.steps, 2),
                    # p20 / p80 over the repetitions (the quantiles kernel/abx_rope.py:198-223 reports)
                    "device_us_p20": round(sev[int(0.2 * (len(sev) - 1) + 0.5)] * 1e3 / args.steps, 2),
                    "device_us_p80": round(sev[int(0.8 * (len(sev) - 1) + 0.5)] * 1e3 / args.steps, 2),
                    "host_wall_us": round(_median(walls) * 1e3 / args.steps, 2),
                    "host_wall_us_min": round(min(walls) * 1e3 / args.steps, 2)}
    best = min(ab, key=lambda k_: ab[k_]["device_us"])
    us_step = ab[best]["device_us"]
    coll_us = None
    alt_rec = None
    if world > 1 and w_alt is not None:
        dec_alt = hp.HeadParallelDecoder(plan, w_alt, k_cache, v_cache, HIDDEN, exchange=p2p if args.exchange == "p2p" else None)
        walls, evs = time_reps(lambda: dec_alt.step(hidden, Lp, Lp), args.steps, args.warmup, sync, reps=3)
        t = torch.tensor([walls, evs], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        alt_rec = {"oproj": alt_oproj, "launch": "direct",
                   "message": ("all-reduce of [hidden] fp32 (16 KiB)" if dec_alt.oproj_sharded else
                               "all-gather of [H/N * Rv] fp16 slices (%d B per rank)" % (plan.heads_local * Rv * 2)),
                   "device_us": round(_median(t[1].tolist()) * 1e3 / args.steps, 2),
                   "host_wall_us": round(_median(t[0].tolist()) * 1e3 / args.steps, 2)}
        del dec_alt
    if world > 1:
        # collective-only latency: CUDA events around the step's one collective, median over 50 steps, max over ranks
        ts = []
        for _ in range(50):
            dec.step(hidden, Lp, Lp, time_collective=True)
            torch.cuda.synchronize()
            e0, e1 = dec.t_collective
            ts.append(e0.elapsed_time(e1) * 1e3)
        tc = torch.tensor([sorted(ts)[len(ts) // 2]], device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        coll_us = float(tc.item())
        # the other exchange on the step's message, alone: 200 back-to-back collectives between events (max over ranks)
        other_us, other_name = None, ("p2p" if args.exchange == "rccl" else "rccl")
        other = p2p if args.exchange == "rccl" else hp.DistExchange(None)
        okt = torch.tensor([1 if other is not None else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()):
            if dec.oproj_sharded:
                buf = torch.zeros(HIDDEN, dtype=torch.float32, device=dev)
                coll = lambda: other.all_reduce_sum_(buf)
            else:
                src = torch.zeros(plan.heads_local * Rv, dtype=torch.float16, device=dev)
                dst = torch.empty(H * Rv, dtype=torch.float16, device=dev)
                coll = lambda: other.all_gather_into(dst, src)
            coll()                                           # one exchange first: a peer-to-peer exchange whose peers never
            sync()                                           # show up reports a timed-out wait (bounded spin) instead of hanging
            healthy = 1
            if other is p2p and p2p.status()[1] != 0:
                healthy = 0
            ht = torch.tensor([healthy], device=dev)
            dist.all_reduce(ht, op=dist.ReduceOp.MIN)
            if int(ht.item()):
                for _ in range(20):
                    coll()
                sync()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    coll()
                e1.record()
                torch.cuda.synchronize()
                to = torch.tensor([e0.elapsed_time(e1) * 1e3 / 200], device=dev)
                dist.all_reduce(to, op=dist.ReduceOp.MAX)
                other_us = float(to.item())
            else:
                other_name += " (timed out: not measured)"

    rec = None
    if rank == 0:
        alg = algorithmic(rank_k, rank_v, L)
        kern = {}
        if world == 1:
            # per-kernel durations, each kernel in its own back-to-back loop between HIP events on the
            # stream the kernels are launched on (torch's current stream)
            lib, s = _lib.lib, _lib.current_stream
            Hl = H
            inv = dec.inv
            # scratch of the per-kernel loops (the step itself runs through palu_decode_attend_f16's own workspace)
            q_buf = torch.empty(H * D, dtype=torch.float16, device=dev)
            sc_buf = torch.empty((H, (cap + 8) // 8 * 8), dtype=torch.float16, device=dev)
            pv_buf = torch.empty(lib.palu_pv_workspace_bytes(H, G, cap, Rv), dtype=torch.uint8, device=dev)

            # the step's own launches: when its score launch takes the position-split kernel, the projection kernel folds the
            # query into the B fragments in the tail of its q waves and the score kernel reads the folded fragments
            # (csrc/abx_fold.h; palu_decode_attend_f16 makes the same choice)
            nfold = lib.palu_abx_fold_bytes(H, G, Rk)
            prefold = bool(nfold and lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, L, Rk, 0))
            qf_buf = torch.zeros(max(nfold, 16), dtype=torch.uint8, device=dev)

            def k_qkv():
                _lib.check(lib.palu_decode_qkv_fold_f16(w["wq"].data_ptr(), w["wq"].stride(0), 0, w["vt_k"].data_ptr(),
                                                        w["vt_k"].stride(0), w["vt_v"].data_ptr(), w["vt_v"].stride(0),
                                                        hidden.data_ptr(), q_buf.data_ptr(), k_cache.data_ptr(),
                                                        k_cache.stride(0), k_cache.stride(1), v_cache.data_ptr(),
                                                        v_cache.stride(0), v_cache.stride(1), inv.data_ptr(), Hl, D, HIDDEN,
                                                        G, Rk, Rv, Lp, Lp, dec.frag.data_ptr() if prefold else 0,
                                                        qf_buf.data_ptr() if prefold else 0, s()), "qkv")

            def k_abx_in_kernel_fold():            # round 5's form: every workgroup folds the query in its prologue
                _lib.check(lib.palu_abx_rope_f16(q_buf.data_ptr(), D, 1, dec.frag.data_ptr(), k_cache.data_ptr(),
                                                 k_cache.stride(0), k_cache.stride(1), sc_buf.data_ptr(),
                                                 sc_buf.stride(0), Hl, G, L, Rk, D, inv.data_ptr(), 0, s()), "abx")

            def k_abx_fold_launch():               # the stand-alone op: fold kernel + score kernel (palu_abx_rope_ws_f16 with scratch)
                _lib.check(lib.palu_abx_rope_ws_f16(q_buf.data_ptr(), D, 1, dec.frag.data_ptr(), k_cache.data_ptr(),
                                                    k_cache.stride(0), k_cache.stride(1), sc_buf.data_ptr(),
                                                    sc_buf.stride(0), Hl, G, L, Rk, D, inv.data_ptr(), 0, qf_buf.data_ptr(), s()), "abx")

            def k_abx():
                if not prefold:
                    return k_abx_in_kernel_fold()
                _lib.check(lib.palu_abx_rope_pf_f16(qf_buf.data_ptr(), k_cache.data_ptr(), k_cache.stride(0), k_cache.stride(1),
                                                    sc_buf.data_ptr(), sc_buf.stride(0), Hl, G, L, Rk, D, inv.data_ptr(), 0,
                                                    s()), "abx_pf")

            def k_pv():
                _lib.check(lib.palu_softmax_pv_f16(sc_buf.data_ptr(), sc_buf.stride(0), 0, v_cache.data_ptr(),
                                                   v_cache.stride(0), v_cache.stride(1), dec.ctx.data_ptr(), 0, 0,
                                                   pv_buf.data_ptr(), Hl, G, L, Rv, math.sqrt(D), s()), "pv")

            def k_o():
                _lib.check(lib.palu_gemv_f16(w["wo"].data_ptr(), w["wo"].stride(0), dec.ctx.data_ptr(),
                                             dec.out.data_ptr(), HIDDEN, H * Rv, s()), "o")
            n = max(20, min(args.steps, 200))
            for name, fn in (("qkv", k_qkv), ("abx", k_abx), ("softmax_pv", k_pv), ("o_proj", k_o)):
                _, kms = time_loop(fn, n, 10, torch.cuda.synchronize, reps=5)
                us = kms * 1e3 / n
                b, f = alg[name]
                kern[name] = {"us": round(us, 2), "algorithmic_bytes": b, "hbm_GBps": round(b / us * 1e-3, 1),
                              "hbm_frac": round(b / us * 1e-3 / HBM_PEAK_GBPS, 4),
                              "tflops": round(f / us * 1e-6, 1)}
        step_b, _ = alg["step"]
        rec = {
            "metric": baseline_metric(rank_k, Lp),
            "value": round(us_step, 2), "unit": "us", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(us_step * 1e-3, 5), "higher_is_better": False,
            "scaling": "strong",   # one decode step of the same problem: total work is fixed as N grows
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "Palu low-rank KV attention decode step (kernel/palu_attention.py decode branch), attention "
                                   "geometry as built by run_latency_attention.py (LlamaConfig() defaults): "
                                   "H=32 D=128 hidden=4096 gs=4 G=8 rank_k=%d rank_v=%d prompt_len=%d fp16 latents batch=1"
                                   % (rank_k, rank_v, Lp),
                       "parallelism": ("head-group x%d + %s %s" % (world, "RCCL" if args.exchange == "rccl" else "one-shot P2P exchange", "all-gather, replicated o_proj" if args.oproj == "replicated"
                                                                       else "column-sharded o_proj + all-reduce of [hidden] fp32"))
                       if world > 1 else "single GPU",
                       "kernels_per_step": 5 if world == 1 else 6,   # qkv, abx, softmax.PV partials, merge, o_proj (+ the collective)
                       "launch": ("hipGraph replay of the captured step" if best == "graph_replay" else "direct launches")
                                 + " (the faster of the forms in launch_ab)",
                       "timing": "value = median over %d repetitions of the device-event time of exactly --steps steps "
                                 "(events on the launch stream behind a device-side gate, sync + barrier on both sides, "
                                 "max over ranks) after --warmup steps and a settle phase; host wall of the same loops in "
                                 "launch_ab" % REPS},
            "launch_ab": ab,
            "host_wall_us": ab[best]["host_wall_us"],
            "p20_us": ab[best]["device_us_p20"], "p80_us": ab[best]["device_us_p80"],
            "collective_us": None if coll_us is None else round(coll_us, 2),
            "oproj_alt": alt_rec,
            "collective": None if world == 1 else {
                "in_step": args.exchange, "in_step_us": round(coll_us, 2), "alone_back_to_back": other_name,
                "alone_back_to_back_us": None if other_us is None else round(other_us, 2),
                "message": ("all-reduce of [hidden] fp32 (16 KiB)" if dec.oproj_sharded else
                            "all-gather of [H/N * Rv] fp16 slices (%d B per rank)" % (plan.heads_local * Rv * 2))},
            "step_algorithmic_bytes": step_b,
            "step_hbm_GBps": round(step_b / us_step * 1e-3, 1),
            "step_hbm_frac": round(step_b / us_step * 1e-3 / HBM_PEAK_GBPS, 4),
        }
        if kern:
            dom = max(("abx", "softmax_pv"), key=lambda k_: kern[k_]["us"])
            b, f = alg[dom]
            traffic = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")        # PMC-derived HBM bytes per launch (committed)
            if os.path.exists(tf):
                traffic = json.load(open(tf)).get(dom)
            rec["kernels"] = kern
            tsrc = ("profiles/traffic.json: (FETCH_SIZE x 2 + WRITE_SIZE) per launch from separate rocprofv3 --pmc passes "
                    "of this command, committed with the profile it came from; not re-collected by this run")
            rec["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["hbm_GBps"], "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": kern[dom]["hbm_frac"], "traffic": traffic,
                               "traffic_source": tsrc if traffic is not None else None}
            ab_, af = alg["abx"]
            # the same launch on the other score kernels, same box, same loop: the one-band kernel (coefficient tables out
            # of the registry) and the pair-split form of the two-band kernel
            from palu_amd.kernel.abx_rope import one_band, pair_split
            two_band = bool(lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, Rk, 0))
            split = bool(lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, L, Rk, 0))
            kms1 = kms2 = kms3 = kms4 = float("nan")
            if not args.no_abx_sweep:
                with one_band():
                    _, kms1 = time_loop(k_abx_in_kernel_fold, n, 10, torch.cuda.synchronize, reps=5)
                with pair_split():
                    _, kms2 = time_loop(k_abx_in_kernel_fold, n, 10, torch.cuda.synchronize, reps=5)
                if prefold:
                    _, kms3 = time_loop(k_abx_in_kernel_fold, n, 10, torch.cuda.synchronize, reps=5)
                    _, kms4 = time_loop(k_abx_fold_launch, n, 10, torch.cuda.synchronize, reps=5)
            ex_f = executed_two_band_flops(Rk, L) if two_band else af
            us_abx = kern["abx"]["us"]
            pmc = {}
            if os.path.exists(tf):
                pmc = json.load(open(tf))
            rec["roofline_abx"] = {
                "kernel": ("abx_rope3_kernel (two-band, position-split" + (", fragments folded by the projection kernel's q waves)"
                                                                          if prefold else ")") if split else
                           "abx_rope2_kernel (two-band, pair-split)" if two_band else "abx_rope_kernel (one-band)"),
                "one_band_kernel_us": None if kms1 != kms1 else round(kms1 * 1e3 / n, 2),
                "pair_split_kernel_us": None if kms2 != kms2 else round(kms2 * 1e3 / n, 2),
                # the same kernel with the fold in every workgroup's prologue (round 5's form), and the stand-alone op
                # (fold kernel + score kernel, two launches): bit-identical scores (tests/test_fold_gpu.py)
                "in_kernel_fold_us": None if kms3 != kms3 else round(kms3 * 1e3 / n, 2),
                "fold_launch_plus_kernel_us": None if kms4 != kms4 else round(kms4 * 1e3 / n, 2),
                "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS,
                # contract fields: ALGORITHMIC flops (2*H*L*R*D + 5*H*L*D) / time
                "achieved": kern["abx"]["tflops"], "frac": round(kern["abx"]["tflops"] / MFMA_PEAK_TFLOPS, 4),
                # what the matrix pipe actually does: the two-band kernels execute 36 NKS big + 8 NKS small MFMAs per tile
                # and group (320 of every 512 MFMA-equivalents of the algorithmic count)
                "executed_flops": ex_f, "executed_tflops": round(ex_f / us_abx * 1e-6, 1),
                "mfma_frac_executed": round(ex_f / us_abx * 1e-6 / MFMA_PEAK_TFLOPS, 4),
                "measured_power_wall_TFLOPs": MFMA_WALL_TFLOPS,
                "frac_of_power_wall": round(ex_f / us_abx * 1e-6 / MFMA_WALL_TFLOPS, 4),
                "mfma_busy": pmc.get("abx_mfma_busy"), "clock_GHz": pmc.get("abx_clock_GHz"),
                "socket_power_W": pmc.get("abx_socket_power_W"),
                "pmc_source": pmc.get("abx_pmc_source"),
                "hbm_achieved_GBps": kern["abx"]["hbm_GBps"], "hbm_frac": kern["abx"]["hbm_frac"],
                "traffic": pmc.get("abx"),
                "note": "arithmetic intensity gs*D = 512 flop/B > ridge: matrix-pipe work (SURVEY F4).  frac_of_power_wall and "
                        "mfma_frac_executed count EXECUTED flops; the power wall is what an MFMA-only random-operand fp16 "
                        "stream sustains on this part (profiles/r02_ubench_issue_pmc_clock.txt: 1.69 GHz at 92 % MFMA busy); "
                        "mfma_busy / clock / socket power come from committed profiler and rocm-smi runs of this command "
                        "(pmc_source), not from this run"}
            # the reference's own bench points (run_latency_kernel.py:11-12: 4k / 16k / 64k / 256k cached positions), fp16,
            # with the kernel each length selects
            sweep = []
            for Ls in (() if args.no_abx_sweep else (4096, 16384, 65536, 262144)):
                try:
                    xs = torch.randn(G, Ls, Rk, device=dev, dtype=torch.float16)
                    so = torch.empty(H, Ls, dtype=torch.float16, device=dev)

                    sp = bool(lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, Ls, Rk, 0))
                    if sp:                                  # (the query's fold is the projection kernel's, as in a step)
                        _lib.check(lib.palu_abx_fold_f16(q_buf.data_ptr(), D, 1, dec.frag.data_ptr(), qf_buf.data_ptr(), H, G, Rk,
                                                         s()), "fold")

                    def k_s():
                        if sp:
                            _lib.check(lib.palu_abx_rope_pf_f16(qf_buf.data_ptr(), xs.data_ptr(), xs.stride(0), xs.stride(1),
                                                                so.data_ptr(), so.stride(0), H, G, Ls, Rk, D, inv.data_ptr(), 0,
                                                                s()), "abx_pf")
                            return
                        _lib.check(lib.palu_abx_rope_f16(q_buf.data_ptr(), D, 1, dec.frag.data_ptr(), xs.data_ptr(), xs.stride(0),
                                                         xs.stride(1), so.data_ptr(), so.stride(0), H, G, Ls, Rk, D,
                                                         inv.data_ptr(), 0, s()), "abx")
                    _, kms = time_loop(k_s, 30, 5, torch.cuda.synchronize, reps=3)
                    tb = bool(lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, Ls, Rk, 0))
                    bs = 2 * G * Ls * Rk + 2 * H * Rk * D + 2 * H * D + 2 * H * Ls
                    sweep.append({"L": Ls, "us": round(kms * 1e3 / 30, 2),
                                  "kernel": "two-band position-split" if sp else "two-band pair-split" if tb else "one-band",
                                  "hbm_frac": round(bs / (kms * 1e3 / 30) * 1e-3 / HBM_PEAK_GBPS, 4)})
                    del xs, so
                except Exception as e:                      # noqa: BLE001
                    sweep.append({"L": Ls, "error": repr(e)[:120]})
            rec["abx_sweep"] = sweep
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            cl = args.cpu_sample_len or L
            t0 = time.perf_counter()
            cpu_min, cpu_med, cpu_in, cpu_out = cpu_baseline(rank_k, rank_v, cl)
            rec["cpu_baseline"] = {"value": round(cpu_min, 1), "median": round(cpu_med, 1), "unit": "us",
                                   "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "oracle.decode_step (CPU port of the reference's PyTorch decode branch) at "
                                             "the same shapes, %d cached positions, min / median of 3 timed runs after 1 "
                                             "warm-up; host %s" % (cl, _cpu_model()),
                                   "seconds_spent": round(time.perf_counter() - t0, 1)}
            # BASELINE.md section 3 / configs[0]: the reference's own CPU-runnable case (rank_k 256, rank_v 768, gs 4, 2048 cached
            # positions), the same port, timed beside the headline configuration's
            c1_min, c1_med, _, _ = cpu_baseline(256, 768, 2048 + 1)
            rec["cpu_baseline_c1"] = {"value": round(c1_min, 1), "median": round(c1_med, 1), "unit": "us",
                                      "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": "oracle.decode_step at BASELINE configs[0] (rank_k=256 rank_v=768 gs=4 "
                                                "prompt_len=2048 fp16), min / median of 3 timed runs after 1 warm-up"}
            # parity of the HIP step against the oracle output just computed, same inputs, full size
            pmax, pscale = gpu_step_parity(cpu_in, cpu_out, cl)
            rec["parity_max_abs"] = round(pmax, 6)
            rec["parity"] = {"max_abs_diff_vs_oracle": round(pmax, 6), "oracle_out_max_abs": round(pscale, 5),
                             "positions": cl, "tolerance": "rtol=atol=1e-3 (test_palu_attention.py:183-195)",
                             "ok": bool(pmax <= 1e-3 + 1e-3 * pscale)}
        if world == 1 and not args.no_extra_configs:
            sub = {}
            for nm, rk_, rv_, lp_, bits_ in (("C3", 1024, 3072, 65536, 3), ("C4", 512, 1536, 131072, 4)):
                try:
                    sub[nm] = bench_quant_config(nm, rk_, rv_, lp_, bits_, max(50, args.steps // 2), dev,
                                                 parity=not args.no_cpu_baseline)
                except Exception as e:                      # noqa: BLE001 -- a sub-record must not kill the headline line
                    sub[nm] = {"error": repr(e)[:200]}
                torch.cuda.empty_cache()
            try:
                sub["C2_sharedB"] = bench_shared_b(rank_k, rank_v, Lp, max(50, args.steps // 2), dev)
            except Exception as e:                          # noqa: BLE001
                sub["C2_sharedB"] = {"error": repr(e)[:200]}
            try:
                sub["C5_per_gpu_slice"] = bench_c5_slice(max(50, args.steps // 2), dev)
            except Exception as e:                          # noqa: BLE001
                sub["C5_per_gpu_slice"] = {"error": repr(e)[:200]}
            try:
                sub["prefill_64k"] = bench_prefill(rank_k, rank_v, Lp, dev)
            except Exception as e:                          # noqa: BLE001
                sub["prefill_64k"] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
            if not args.no_model32:
                # the whole 32-layer model decoding through the latent caches (tools/bench_model.py)
                try:
                    torch.cuda.empty_cache()
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import bench_model
                    m32 = bench_model.run(32, rank_k, rank_v, 4, Lp, 16, reps=15, dev=str(dev))
                    m32["attention_share"] = ("32 x the single-layer decode step above = %.2f ms of the %.2f ms per token "
                                              "(the rest: RMSNorm / gated MLP / lm_head one-token kernels, residual adds and the embedding through torch)"
                                              % (32 * us_step * 1e-3, m32.get("graph_ms_per_token", m32["eager_ms_per_token"])))
                    sub["model32"] = m32
                except Exception as e:                      # noqa: BLE001
                    sub["model32"] = {"error": repr(e)[:200]}
                torch.cuda.empty_cache()
            rec["configs"] = sub
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " x%d logical" % (os.cpu_count() or 0)
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
