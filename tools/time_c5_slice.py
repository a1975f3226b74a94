#!/usr/bin/env python3
"""BASELINE config 5's per-GPU slice (G = 1, H = 4, 262 144 cached positions): qkv + attention core through the fused
single-kernel core and through the two kernels (PALU_FUSED_ATTN=1 / 0 in the environment: run once per setting)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

r = bench.bench_c5_slice(50, torch.device("cuda", 0))
print("PALU_FUSED_ATTN=%s attend_us %.2f fused_core %s" % (os.environ.get("PALU_FUSED_ATTN", "auto"), r["attend_us"], r["fused_attention_core"]))
