#!/usr/bin/env python3
"""TPOT harness for one low-rank attention module -- same CLI and output line as the reference's
run_latency_attention.py (:155-190, result line :106), running on the MI355X HIP decode path.

    python run_latency_attention.py --palu --rank_k 1024 --rank_v 3072 --group_size 4 --prompt_len 65536

Differences from the reference, all additive: the module comes from palu_amd (HIP kernels behind the
reference's Python API), the cache is the pre-allocated latent cache (no torch.cat per step),
`--json` prints a machine-readable record with per-step algorithmic bytes / achieved HBM GB/s, and
`--fast_init` skips the 64 per-group SVDs by drawing random low-rank factors directly (timings do
not depend on the weight values).  Without `--palu` the DENSE attention baseline of the reference
(`build_attention`, :29-38: a stock HF LlamaAttention on full-rank K/V caches) runs instead -- here a
torch-op module with the HF-4.37.2 eager semantics on rocBLAS/stock kernels, which is what the
reference's own dense leg is (nothing custom): the Palu-vs-dense comparison of the reference CLI.
`--gpus N` (under `python -m torch.distributed.run --nproc-per-node N`) runs the head-group-parallel
decode step of palu_amd.kernel.head_parallel: rank r owns G/N latent groups, one RCCL
collective per step; rank 0 prints the same result line.
"""
from __future__ import annotations

import argparse
import json
import logging
import math
import os
# hipGraph replay: ROCm 7.2's graph "packet capture" path (on by default) costs ~3.5 us per replay of this 5-kernel step
# (157.3 us against 153.5 with it off, direct launches 152.7: profiles/r05_graph_replay.txt); it is read when the HIP runtime
# loads, so it must be set before torch is imported.  An explicit setting in the environment wins.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys

import torch
from torch import nn

from palu_amd.kernel.palu_attention import DynamicCache, LlamaPaluAttention, QuantLatentCache, build_b


class LlamaLikeConfig:
    """The LlamaConfig() defaults the reference harness relies on (:45-50): Llama-2-7B geometry."""

    def __init__(self):
        self.hidden_size = 4096
        self.num_attention_heads = 32
        self.num_key_value_heads = 32
        self.attention_bias = False
        self.attention_dropout = 0.0
        self.rope_theta = 10000.0
        self.max_position_embeddings = 300000


class _DenseAttention(nn.Module):
    """Dense Llama attention with the eager semantics of transformers-4.37.2's LlamaAttention (what the reference's
    build_attention instantiates, run_latency_attention.py:29-38): q/k/v projections, HF rotary at `position_ids`,
    K/V cache of full head_dim rows, q.k^T/sqrt(D), softmax fp32 -> fp16, P.V, o_proj.  Stock torch ops only: this is
    the BASELINE leg of the harness (and the donor of LlamaPaluAttention.from_attention), not a product path."""

    def __init__(self, config, layer_idx=0):
        super().__init__()
        d = config.hidden_size
        self.config = config
        self.layer_idx = layer_idx
        self.num_heads = config.num_attention_heads
        self.head_dim = d // config.num_attention_heads
        self.rope_theta = float(getattr(config, "rope_theta", 10000.0))
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.o_proj = nn.Linear(d, d, bias=False)
        self._inv_freq = {}          # per device: the fp32 inverse frequencies, built once (no host work inside a step)

    def _inv(self, device):
        t = self._inv_freq.get(device)
        if t is None:
            D = self.head_dim
            t = (1.0 / (self.rope_theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))).to(device)
            self._inv_freq[device] = t
        return t

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, **kwargs):
        bsz, q_len, _ = hidden_states.shape
        H, D = self.num_heads, self.head_dim
        q = self.q_proj(hidden_states).view(bsz, q_len, H, D).transpose(1, 2)
        k = self.k_proj(hidden_states).view(bsz, q_len, H, D).transpose(1, 2)
        v = self.v_proj(hidden_states).view(bsz, q_len, H, D).transpose(1, 2)
        past = 0 if past_key_value is None else past_key_value.get_usable_length(q_len, self.layer_idx)
        if position_ids is None:
            position_ids = torch.arange(past, past + q_len, device=hidden_states.device)
        # angles on the device from a cached table: no CPU arithmetic and no blocking pageable copy per step (ADVICE r3;
        # such a copy is also illegal inside a graph capture).  Callers hand over device-resident position_ids.
        ang = torch.outer(position_ids.reshape(-1).to(hidden_states.device, non_blocking=True).float(),
                          self._inv(hidden_states.device))
        emb = torch.cat((ang, ang), dim=-1)
        cos, sin = emb.cos().to(q.dtype), emb.sin().to(q.dtype)

        def rot(x):
            return x * cos + torch.cat((-x[..., D // 2:], x[..., :D // 2]), dim=-1) * sin
        q, k = rot(q), rot(k)
        if past_key_value is not None:
            k, v = past_key_value.update(k, v, self.layer_idx)
        w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D)
        if attention_mask is not None:
            w = w + attention_mask
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(w, v).transpose(1, 2).reshape(bsz, q_len, H * D)
        return self.o_proj(o), (w if output_attentions else None), past_key_value


def build_attention(args, device="cuda:0", dtype=torch.float16):
    """run_latency_attention.py:29-38: the dense baseline."""
    logging.info(f"Creating Attention, dtype: {dtype}, device: {device}")
    config = LlamaLikeConfig()
    return _DenseAttention(config, layer_idx=0).to(device, dtype), config


def build_attention_palu(args, device="cuda:0", dtype=torch.float16):
    logging.info(f"Creating Attention_Palu, dtype: {dtype}, device: {device}")
    config = LlamaLikeConfig()
    config.group_size = args.group_size
    config.num_groups = config.num_attention_heads // args.group_size
    config.total_rank_k = args.rank_k
    config.total_rank_v = args.rank_v
    logging.info(f"rank_k: {config.total_rank_k}, rank_v: {config.total_rank_v}, "
                 f"group_size: {config.group_size}, num_groups: {config.num_groups}")
    if args.fast_init:
        attn = LlamaPaluAttention(config, layer_idx=0)
        D = attn.head_dim
        with torch.no_grad():
            for u in attn.k_proj.U_list:
                u.weight.mul_(1.0 / u.weight.shape[1] ** 0.5)
        attn.k_proj.B = nn.Parameter(build_b([u.weight.data for u in attn.k_proj.U_list], config.group_size, D))
    else:
        attn = LlamaPaluAttention.from_attention(_DenseAttention(config, 0), config)
    return attn.to(device, dtype), config


def step_algorithmic_bytes(config, prompt_len, bits=16):
    """latents once + weights once + scores write/read (fp16)."""
    H, G = config.num_attention_heads, config.num_groups
    rk, rv, hid = config.total_rank_k, config.total_rank_v, config.hidden_size
    L = prompt_len + 1
    D = hid // H
    latents = 2 * L * (rk + rv) if bits >= 16 else L * (rk + rv) * bits // 8 + 2 * 4 * G * L
    weights = 2 * (hid * hid + rk * hid + rv * hid + hid * H * (rv // G))
    b = 2 * H * (rk // G) * D
    scores = 2 * 2 * H * L
    return latents + weights + b + scores


class _DecodeStep:
    """One decode token through `model` on a fixed token buffer, cache and position -- eager, or as the replay of a captured
    hipGraph (the reference harness's --cache_graph, run_latency_attention.py:81-90 there).  The cache grows by one row per
    eager call; a replay re-runs the captured launches (same row, same position: what the reference's replay does)."""

    def __init__(self, model, cache, position_ids, token):
        self.model, self.cache, self.position_ids, self.token = model, cache, position_ids, token
        self.graph, self.out = None, None

    @torch.no_grad()
    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
            return self.out
        return self.model(self.token, past_key_value=self.cache, position_ids=self.position_ids)

    def run(self, n):
        for _ in range(n):
            self()
        torch.cuda.synchronize()

    @torch.no_grad()
    def capture(self):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.out = self.model(self.token, past_key_value=self.cache, position_ids=self.position_ids)
        self.graph = g


def _device_ms(fn, reps):
    """Device time of `reps` back-to-back calls between two events on the current stream (milliseconds, total)."""
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1)


def profile_tpot(model, cache_size_k, cache_size_v, cache_type=torch.float16, batch_size=1, prompt_len=1024,
                 repeats=100, cache_graph=False, torch_profile=False, outfile="", bits=16):
    logging.info(">>> Profiling TPOT (generation stage)")
    device = next(iter(model.parameters())).device
    cache_k = torch.randn(cache_size_k, dtype=cache_type, device=device)
    cache_v = torch.randn(cache_size_v, dtype=cache_type, device=device)
    hidden_dim = model.config.hidden_size
    warm, reps = 25, repeats
    cap = prompt_len + 2 * (warm + reps) + 64
    past_key_value = DynamicCache(capacity=cap) if bits >= 16 else QuantLatentCache(bits, capacity=cap)
    if bits >= 16:
        past_key_value.update(cache_k, cache_v, 0)
    else:                                        # quantise the synthetic prompt latents in slabs (bounded temporaries)
        for s0 in range(0, prompt_len, 8192):
            past_key_value.reserve(0, cap, cache_k.shape[1], cache_k.shape[3], cache_v.shape[3], device)
            from palu_amd.kernel.quant import quantize_pack
            st, n = past_key_value.buffers(0), past_key_value.get_seq_length(0)
            t = min(8192, prompt_len - s0)
            for src, cdst, mdst in ((cache_k, "kc", "km"), (cache_v, "vc", "vm")):
                c, m = quantize_pack(src[:, :, s0:s0 + t].contiguous(), bits)
                st[cdst][:, :, n:n + t].copy_(c)
                st[mdst][:, :, n:n + t].copy_(m)
            past_key_value.advance(0, t)
    del cache_k, cache_v
    position_ids = torch.arange(prompt_len, prompt_len + 1)
    if isinstance(model, _DenseAttention):
        # the dense leg builds its rotary angles on the device: hand the position over once, outside the timed / captured
        # region (the Palu module reads it on the host: a CPU tensor costs it nothing)
        position_ids = position_ids.to(device)
    input_token = torch.randn((batch_size, 1, hidden_dim), dtype=torch.float16, device=device)

    step = _DecodeStep(model, past_key_value, position_ids, input_token)
    step.run(warm)                                           # warm-up: kernels loaded, B fragments and RoPE tables cached
    if cache_graph:
        step.capture()                                       # one hipGraph of the module's launches, replayed per token
    step.token.copy_(torch.randn_like(step.token))           # a fresh token, as the generation loop would feed
    dur = _device_ms(step, reps)
    logging.info(f"Finished, prompt_len: {prompt_len}, latency: {dur / reps:.2f} milliseconds (cache_graph={cache_graph})")
    if torch_profile:
        from torch.profiler import ProfilerActivity, profile, schedule
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA],
                     schedule=schedule(wait=1, warmup=5, active=6, repeat=1), record_shapes=True) as prof:
            for _ in range(12):
                step()
                prof.step()
        prof.export_chrome_trace(f"{outfile or 'tpot_palu_fp16'}.json.gz")
    return dur / reps


def profile_ttft(model, prompt_len, repeats=5, bits=16):
    """Prompt pass (q_len = prompt_len, empty cache) through the flash-style prefill kernel: time to first token
    of ONE attention module.  Additive to the reference harness, which only profiles TPOT."""
    device = next(iter(model.parameters())).device
    x = torch.randn((1, prompt_len, model.config.hidden_size), dtype=torch.float16, device=device)
    with torch.no_grad():
        def new_cache():
            cap = prompt_len + 64
            return DynamicCache(capacity=cap) if bits >= 16 else QuantLatentCache(bits, capacity=cap)
        for _ in range(2):
            model(x, past_key_value=new_cache(), is_causal=True)
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        caches = [new_cache() for _ in range(repeats)]
        start.record()
        for c in caches:
            model(x, past_key_value=c, is_causal=True)
        end.record()
        torch.cuda.synchronize()
    ms = start.elapsed_time(end) / repeats
    logging.info(f"Finished, prompt_len: {prompt_len}, prefill latency: {ms:.2f} milliseconds")
    return ms


def profile_tpot_head_parallel(args):
    """`--gpus N`: the head-group-parallel decode step (palu_amd.kernel.head_parallel), one process per GPU under
    torch.distributed.run.  Same synthetic set-up as profile_tpot (randn latent caches of prompt_len rows, randn token,
    random low-rank factors as with --fast_init: timings do not depend on the values); rank 0 logs the result line."""
    import torch.distributed as dist
    from palu_amd.kernel import head_parallel as hp
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus N must be launched as: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 run_latency_attention.py --palu --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "LOCAL_RANK" in os.environ:
        dist.init_process_group("nccl", device_id=dev)
    cfg = LlamaLikeConfig()
    H, hid = cfg.num_attention_heads, cfg.hidden_size
    D, gs = hid // H, args.group_size
    G = H // gs
    Rk, Rv = args.rank_k // G, args.rank_v // G
    plan = hp.make_plan(world, rank, H, G, D, Rk, Rv)
    torch.manual_seed(1234)                                  # every rank draws the same full weights, keeps its shard
    full = {"wq": (torch.randn(H * D, hid, device=dev) / 64).half(), "vt_k": (torch.randn(args.rank_k, hid, device=dev) / 64).half(),
            "vt_v": (torch.randn(args.rank_v, hid, device=dev) / 64).half(), "b": (torch.randn(H, Rk, D, device=dev) * Rk ** -0.5).half(),
            "wo": (torch.randn(hid, H * Rv, device=dev) * 0.01).half()}
    w = {k: (v.contiguous() if k != "wo" else v) for k, v in hp.shard_weights(plan, full, oproj=args.oproj).items()}
    del full
    warm, reps = 25, args.repeats
    cap = (args.prompt_len + 64 + 63) // 64 * 64
    kc = torch.randn(plan.groups_local, cap, Rk, device=dev, dtype=torch.float16)
    vc = torch.randn(plan.groups_local, cap, Rv, device=dev, dtype=torch.float16)
    tok = torch.randn(hid, device=dev, dtype=torch.float16)
    dec = hp.HeadParallelDecoder(plan, w, kc, vc, hid)
    n = args.prompt_len

    def generate():
        return dec.step(tok, n, n)
    for _ in range(warm):
        generate()
    torch.cuda.synchronize()
    if args.cache_graph:
        generate = dec.capture(tok, n, n)                    # the collective is captured with the step
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        generate()
    end.record()
    torch.cuda.synchronize()
    t = torch.tensor([start.elapsed_time(end) / reps], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item())
    if rank == 0:
        logging.info(f"Finished, prompt_len: {n}, latency: {ms:.2f} milliseconds (cache_graph={args.cache_graph}, "
                     f"gpus={world}, oproj={args.oproj})")
        if args.json:
            print(json.dumps({"latency_us": ms * 1e3, "prompt_len": n, "rank_k": args.rank_k, "rank_v": args.rank_v,
                              "group_size": gs, "gpus": world, "oproj": args.oproj, "cache_graph": args.cache_graph}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return ms


def main(args):
    bs = 1
    if args.gpus > 1 or "LOCAL_RANK" in os.environ:      # launched by torch.distributed.run (also with one rank)
        if not args.palu or args.bits < 16 or args.ttft:
            raise SystemExit("--gpus N: the head-group-parallel step is the fp16 Palu decode step (--palu, --bits 16, no --ttft)")
        profile_tpot_head_parallel(args)
        return
    if not args.palu:
        # the reference's dense leg (:172-177): full-rank K/V caches [bs, heads, prompt_len, head_dim]
        if args.bits < 16 or args.hadamard or args.ttft:
            raise SystemExit("--bits / --hadamard / --ttft belong to the --palu module")
        attention, config = build_attention(args)
        attention.eval()
        H, D = config.num_attention_heads, config.hidden_size // config.num_attention_heads
        ms = profile_tpot(attention, (bs, H, args.prompt_len, D), (bs, H, args.prompt_len, D), torch.float16, bs,
                          args.prompt_len, args.repeats, args.cache_graph, args.torch_profile, "tpot_fp16")
        if args.json:
            nbytes = 2 * 2 * H * (args.prompt_len + 1) * D + 2 * 4 * config.hidden_size ** 2
            print(json.dumps({"latency_us": ms * 1e3, "prompt_len": args.prompt_len, "attention": "dense",
                              "cache_graph": args.cache_graph, "algorithmic_bytes": nbytes,
                              "hbm_GBps": nbytes / (ms * 1e-3) * 1e-9}))
        return
    attention, config = build_attention_palu(args)
    attention.eval()
    num_groups = config.num_groups
    group_dim_k = config.total_rank_k // num_groups
    group_dim_v = config.total_rank_v // num_groups
    cache_size_k = (bs, num_groups, args.prompt_len, group_dim_k)
    cache_size_v = (bs, num_groups, args.prompt_len, group_dim_v)
    if args.hadamard:
        attention.fuse_hadamard()
    if args.ttft:
        ms = profile_ttft(attention, args.prompt_len, max(1, min(args.repeats, 5)), bits=args.bits)
        if args.json:
            H, D, Rv = config.num_attention_heads, config.hidden_size // config.num_attention_heads, group_dim_v
            flops = H * (args.prompt_len ** 2 / 2) * (D + Rv) * 2
            print(json.dumps({"prefill_ms": ms, "prompt_len": args.prompt_len, "rank_k": args.rank_k, "rank_v": args.rank_v,
                              "attention_TFLOPs_causal": flops / (ms * 1e-3) * 1e-12}))
        return
    ms = profile_tpot(attention, cache_size_k, cache_size_v, torch.float16, bs, args.prompt_len, args.repeats,
                      args.cache_graph, args.torch_profile, "tpot_palu_fp16" if args.bits >= 16 else f"tpot_palu_int{args.bits}",
                      bits=args.bits)
    if args.json:
        nbytes = step_algorithmic_bytes(config, args.prompt_len, args.bits)
        print(json.dumps({"latency_us": ms * 1e3, "prompt_len": args.prompt_len, "rank_k": args.rank_k,
                          "rank_v": args.rank_v, "group_size": args.group_size, "cache_graph": args.cache_graph, "bits": args.bits, "hadamard": args.hadamard,
                          "algorithmic_bytes": nbytes, "hbm_GBps": nbytes / (ms * 1e-3) * 1e-9,
                          "hbm_frac_of_8TBps": nbytes / (ms * 1e-3) * 1e-9 / 8000.0}))


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--palu", action="store_true", help="Whether to use PALU attention.")
    parser.add_argument("--rank_k", type=int, default=1024, help="The rank of key matrix for PALU attention.")
    parser.add_argument("--rank_v", type=int, default=2048, help="The rank of value matrix for PALU attention.")
    parser.add_argument("--group_size", type=int, default=4, help="The group size for PALU attention.")
    parser.add_argument("--repeats", type=int, default=100, help="The number of profiling to repeat (default: 100)")
    parser.add_argument("--prompt_len", type=int, default=1024, help="The number of input tokens to model. (default: 1024)")
    parser.add_argument("--cache_graph", action="store_true", default=False,
                        help="To enable graph capture of the decode step (HIP graph via torch.cuda.CUDAGraph)")
    parser.add_argument("--torch_profile", action="store_true", help="Whether to launch the pytorch profiler.")
    parser.add_argument("--fast_init", action="store_true", help="random low-rank factors instead of 64 SVDs")
    parser.add_argument("--bits", type=int, default=16, choices=[16, 4, 3],
                        help="latent KV precision: 16 = fp16 cache, 4/3 = packed codes (--lt_bits of the reference's eval scripts)")
    parser.add_argument("--hadamard", action="store_true", help="fuse Hadamard rotations into the weights (--lt_hadamard)")
    parser.add_argument("--ttft", action="store_true", help="profile the prompt pass (prefill) instead of the decode step")
    parser.add_argument("--json", action="store_true", help="also print a JSON record with achieved HBM GB/s")
    parser.add_argument("--gpus", type=int, default=1,
                        help="head-group-parallel decode over N GPUs of one node (launch with torch.distributed.run)")
    parser.add_argument("--oproj", choices=("sharded", "replicated"), default="sharded",
                        help="--gpus N: column-sharded o_proj + all-reduce of [hidden] fp32, or all-gather + replicated o_proj")
    args = parser.parse_args()
    logging.basicConfig(level=logging.INFO,
                        format="[%(asctime)s] %(levelname)s [%(filename)s:%(lineno)3d] %(message)s",
                        datefmt="%d/%b/%Y %H:%M:%S", stream=sys.stdout)
    main(args)
