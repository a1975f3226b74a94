#!/usr/bin/env python3
"""Whole-model decode throughput with latent caches (the reference's L4 models,
palu/model/svd_llama/modeling_palu_llama.py:7-35, reconstruct full K/V and never reach the kernel path).

A transformers-5 `LlamaForCausalLM` of Llama-2-7B geometry (32 layers, hidden 4096, 32 heads x 128, MLP 11008, vocab 32000;
random weights -- there is no network for checkpoints) whose attention modules are `LlamaPaluAttention` behind
`palu_amd.hf.PaluAttentionHF` (random low-rank factors of the named ranks: a throughput measurement needs no SVD), one
`PaluCacheHF` with `prompt_len` cached positions per layer (synthetic latents, fp16 or packed).  One decode step =
`model(token, past_key_values=cache)`: 32 HIP decode steps + the model's RMSNorm / MLP / lm_head (HIP one-token kernels,
`palu_amd.hf.use_hip_decode_linears`, or torch with --torch_mlp).

Reports ms per token and tokens/s for eager launches and for the replay of ONE captured hipGraph of the whole forward, the
host microseconds per layer that the graph removes, and 32 x the single-layer attention step beside it.

    PYTHONPATH=. python tools/bench_model.py [--layers 32] [--prompt_len 65536] [--bits 16] [--json]
"""
from __future__ import annotations

import argparse
import json
import time

import torch


def build_model(layers, rank_k, rank_v, group_size, dev):
    from transformers import LlamaConfig, LlamaForCausalLM
    from torch import nn
    from palu_amd.hf import PaluAttentionHF, palu_config_from
    from palu_amd.kernel.palu_attention import LlamaPaluAttention, build_b
    cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=layers,
                      num_attention_heads=32, num_key_value_heads=32, head_dim=128, max_position_embeddings=300000,
                      rope_theta=10000.0, attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"      # decode mask = None (the eager mask builder does an H2D copy: not capturable)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        with torch.device(dev):
            model = LlamaForCausalLM(cfg)                      # random init directly on the GPU
            pcfg = palu_config_from(cfg, rank_k, rank_v, group_size)
            for i, layer in enumerate(model.model.layers):
                inner = LlamaPaluAttention(pcfg, i)
                G, D = pcfg.num_groups, 128
                with torch.no_grad():
                    for lin in (inner.q_proj, inner.k_proj.VT, inner.v_proj.VT, inner.o_proj):
                        lin.weight.normal_(0.0, 0.02)
                    for u in inner.k_proj.U_list:
                        u.weight.normal_(0.0, (rank_k // G) ** -0.5)
                    inner.k_proj.B = nn.Parameter(build_b([u.weight for u in inner.k_proj.U_list], group_size, D))
                layer.self_attn = PaluAttentionHF(inner.eval().prepare_decode())
    finally:
        torch.set_default_dtype(old)
    return model.eval(), cfg


def fill_cache(cache, layers, G, Rk, Rv, L, dev, bits):
    g = torch.Generator(device=dev).manual_seed(0)
    for li in range(layers):
        k = torch.randn((1, G, L, Rk), generator=g, device=dev, dtype=torch.float16)
        v = torch.randn((1, G, L, Rv), generator=g, device=dev, dtype=torch.float16)
        if bits >= 16:
            cache.latent.update(k, v, li)
        else:
            cache.latent.append_rows(k, v, li)
        del k, v


def run(layers=32, rank_k=1024, rank_v=3072, group_size=4, prompt_len=65536, bits=16, reps=20, dev="cuda:0", hip_mlp=True):
    from palu_amd.hf import PaluCacheHF, use_hip_decode_linears
    torch.manual_seed(0)
    t0 = time.perf_counter()
    model, cfg = build_model(layers, rank_k, rank_v, group_size, dev)
    if hip_mlp:
        use_hip_decode_linears(model)      # MLP and lm_head GEMVs of the one-token step on the HIP kernels too
    G = 32 // group_size
    cache = PaluCacheHF(bits=bits, capacity=prompt_len + 64)
    fill_cache(cache, layers, G, rank_k // G, rank_v // G, prompt_len, dev, bits)
    cache.assume_standard_positions()      # (filled without a prompt pass: the decode position IS the cache length)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    tok = torch.randint(0, 32000, (1, 1), device=dev)

    def rewind():
        # every timed step decodes at the same position (like run_latency_attention.py:101-102 replays one step)
        lat = cache.latent
        for li in range(layers):
            lat._len[li] = prompt_len

    def step():
        rewind()
        with torch.no_grad():
            return model(tok, past_key_values=cache, use_cache=True).logits

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    walls = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t) * 1e3)
    walls.sort()
    eager_ms = walls[len(walls) // 2]
    rec = {"workload": "Llama-2-7B geometry (%d layers, random weights), rank_k=%d rank_v=%d gs=%d, %d cached positions per layer, "
                       "%s latent caches, batch 1: one decode step of the WHOLE model through palu_amd.hf" %
                       (layers, rank_k, rank_v, group_size, prompt_len, "fp16" if bits >= 16 else "%d-bit packed" % bits),
           "layers": layers, "mlp_and_lm_head": "HIP GEMVs (palu_amd.hf.use_hip_decode_linears)" if hip_mlp else "torch",
           "eager_ms_per_token": round(eager_ms, 3), "eager_tokens_per_s": round(1e3 / eager_ms, 1),
           "setup_s": round(setup_s, 1)}
    # one hipGraph for the whole forward
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        rewind()
        with torch.cuda.graph(graph):
            with torch.no_grad():
                out = model(tok, past_key_values=cache, use_cache=True).logits
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            evs.append(e0.elapsed_time(e1))
        evs.sort()
        g_ms = evs[len(evs) // 2]
        rec.update({"graph_ms_per_token": round(g_ms, 3), "graph_tokens_per_s": round(1e3 / g_ms, 1),
                    "host_us_per_layer_removed_by_the_graph": round((eager_ms - g_ms) * 1e3 / layers, 1),
                    "logits_finite": bool(torch.isfinite(out.float()).all())})
    except Exception as e:                                  # noqa: BLE001 -- capture of third-party model code may not be possible
        import traceback
        rec["graph_error"] = repr(e)[:200] + " | " + " <- ".join(l.strip() for l in traceback.format_exc().splitlines() if l.strip().startswith("File"))[-1500:]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--rank_k", type=int, default=1024)
    ap.add_argument("--rank_v", type=int, default=3072)
    ap.add_argument("--group_size", type=int, default=4)
    ap.add_argument("--prompt_len", type=int, default=65536)
    ap.add_argument("--bits", type=int, default=16)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--torch_mlp", action="store_true", help="leave the MLPs and lm_head on torch (hipBLASLt GEMVs)")
    a = ap.parse_args()
    rec = run(a.layers, a.rank_k, a.rank_v, a.group_size, a.prompt_len, a.bits, hip_mlp=not a.torch_mlp)
    print(json.dumps(rec) if a.json else "\n".join(f"{k}: {v}" for k, v in rec.items()))


if __name__ == "__main__":
    main()
