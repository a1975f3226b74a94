#!/usr/bin/env python3
"""End-to-end use of one low-rank attention module on an MI355X: prompt pass (flash-style prefill kernel, latents
written straight into the cache), then token-by-token decode (one native call per step), with an fp16 or a packed
3/4-bit latent cache; optionally the cache is saved to / restored from a safetensors file in between.

    python examples/prefill_then_decode.py --prompt_len 16384 --new_tokens 64 [--bits 4] [--hadamard] [--cache_file f]
"""
import argparse
import time

import torch

from palu_amd.kernel.palu_attention import (LatentCache, LlamaPaluAttention, QuantLatentCache, build_b, load_cache,
                                             save_cache)


class Cfg:
    hidden_size, num_attention_heads, attention_bias = 4096, 32, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt_len", type=int, default=16384)
    ap.add_argument("--new_tokens", type=int, default=64)
    ap.add_argument("--rank_k", type=int, default=1024)
    ap.add_argument("--rank_v", type=int, default=3072)
    ap.add_argument("--group_size", type=int, default=4)
    ap.add_argument("--bits", type=int, default=16, choices=[16, 4, 3])
    ap.add_argument("--hadamard", action="store_true")
    ap.add_argument("--cache_file", default="")
    a = ap.parse_args()
    cfg = Cfg()
    cfg.group_size, cfg.num_groups = a.group_size, cfg.num_attention_heads // a.group_size
    cfg.total_rank_k, cfg.total_rank_v = a.rank_k, a.rank_v
    torch.manual_seed(0)
    m = LlamaPaluAttention(cfg, layer_idx=0)
    with torch.no_grad():                                   # random low-rank factors (timings do not depend on values)
        for u in m.k_proj.U_list:
            u.weight.mul_(u.weight.shape[1] ** -0.5)
    m.k_proj.B = torch.nn.Parameter(build_b([u.weight.data for u in m.k_proj.U_list], a.group_size, 128))
    m = m.to("cuda", torch.float16).eval()
    if a.hadamard:
        m.fuse_hadamard()
    cap = a.prompt_len + a.new_tokens + 64
    cache = LatentCache(capacity=cap) if a.bits == 16 else QuantLatentCache(a.bits, capacity=cap)
    x = torch.randn(1, a.prompt_len, cfg.hidden_size, dtype=torch.float16, device="cuda")
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _, _ = m(x, past_key_value=cache, is_causal=True)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        if a.cache_file:
            save_cache(cache, a.cache_file)
            cache = load_cache(a.cache_file, device="cuda", capacity=cap)
        tok = out[:, -1:, :].contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_first = 0.0
        for i in range(a.new_tokens):
            tok, _, _ = m(tok, past_key_value=cache, position_ids=torch.tensor([[a.prompt_len + i]]))
            if i == 0:                                      # the first step prepares fragments / workspace once
                torch.cuda.synchronize()
                t_first = time.perf_counter() - t0
                t0 = time.perf_counter()
        torch.cuda.synchronize()
        t_decode = (time.perf_counter() - t0) / max(a.new_tokens - 1, 1)
    assert cache.get_seq_length(0) == a.prompt_len + a.new_tokens and torch.isfinite(tok.float()).all()
    kind = "fp16" if a.bits == 16 else f"{a.bits}-bit" + (" + Hadamard" if a.hadamard else "")
    print(f"{kind} latent cache, rank {a.rank_k}/{a.rank_v}: prompt {a.prompt_len} tokens in {t_prefill * 1e3:.1f} ms (first "
          f"call, incl. one-off setup), first decode step {t_first * 1e3:.1f} ms (one-off setup), then {t_decode * 1e6:.0f} us per decoded "
          f"token ({a.new_tokens - 1} tokens, host loop)")


if __name__ == "__main__":
    main()
