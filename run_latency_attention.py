#!/usr/bin/env python3
"""TPOT harness for one low-rank attention module -- same CLI and output line as the reference's
run_latency_attention.py (:155-190, result line :106), running on the MI355X HIP decode path.

    python run_latency_attention.py --palu --rank_k 1024 --rank_v 3072 --group_size 4 --prompt_len 65536

Differences from the reference, all additive: the module comes from palu_amd (HIP kernels behind the
reference's Python API), the cache is the pre-allocated latent cache (no torch.cat per step),
`--json` prints a machine-readable record with per-step algorithmic bytes / achieved HBM GB/s, and
`--fast_init` skips the 64 per-group SVDs by drawing random low-rank factors directly (timings do
not depend on the weight values).  `--palu` is required: the dense-attention baseline of the
reference (`build_attention`, :29-38) is not part of this build.
"""
from __future__ import annotations

import argparse
import json
import logging
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
    def __init__(self, config, layer_idx=0):
        super().__init__()
        d = config.hidden_size
        self.layer_idx = layer_idx
        self.head_dim = d // config.num_attention_heads
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.o_proj = nn.Linear(d, d, bias=False)


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
    input_token = torch.randn((batch_size, 1, hidden_dim), dtype=torch.float16, device=device)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.no_grad():
        with torch.cuda.stream(s):
            for _ in range(warm):
                _ = model(input_token, past_key_value=past_key_value, position_ids=position_ids)
    torch.cuda.current_stream().wait_stream(s)

    if cache_graph:
        with torch.no_grad():
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model(input_token, past_key_value=past_key_value, position_ids=position_ids)

        def generate(new_input_token, past_key_value, position_ids):
            input_token.copy_(new_input_token)
            graph.replay()
            return out
    else:
        def generate(new_input_token, past_key_value, position_ids):
            return model(new_input_token, past_key_value=past_key_value, position_ids=position_ids)

    new_input_token = torch.randn((batch_size, 1, hidden_dim), dtype=torch.float16, device=device)
    with torch.no_grad():
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            generate(new_input_token, past_key_value=past_key_value, position_ids=position_ids)
        end.record()
        torch.cuda.synchronize()
    dur = start.elapsed_time(end)
    logging.info(f"Finished, prompt_len: {prompt_len}, latency: {dur / reps:.2f} milliseconds (cache_graph={cache_graph})")

    if torch_profile:
        from torch.profiler import ProfilerActivity, profile, schedule
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA],
                     schedule=schedule(wait=1, warmup=5, active=6, repeat=1), record_shapes=True) as prof:
            with torch.no_grad():
                for _ in range(12):
                    generate(new_input_token, past_key_value, position_ids=position_ids)
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


def main(args):
    if not args.palu:
        raise SystemExit("only --palu is implemented in this build (the dense baseline is out of scope)")
    bs = 1
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
    args = parser.parse_args()
    logging.basicConfig(level=logging.INFO,
                        format="[%(asctime)s] %(levelname)s [%(filename)s:%(lineno)3d] %(message)s",
                        datefmt="%d/%b/%Y %H:%M:%S", stream=sys.stdout)
    main(args)
