#!/usr/bin/env python3
"""Per-step cycle stamps of the fused decode attention kernel (debug entry palu_decode_attn_f16_timed).

    [PALU_FUSED_EXP=flags] python tools/time_fused.py [L]
"""
import ctypes as C
import math
import sys

import numpy as np
import torch

from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

L = int(sys.argv[1]) if len(sys.argv) > 1 else 65537
fn = _lib.lib.palu_decode_attn_f16_timed
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
               C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float,
               C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
torch.manual_seed(0)
H, G, Rk, Rv, D = 32, 8, 128, 384, 128
a = torch.randn(H, D, dtype=torch.float16, device="cuda")
b = (torch.randn(H, Rk, D, device="cuda") * Rk ** -0.5).half()
k = torch.randn(G, L + 64, Rk, dtype=torch.float16, device="cuda")
v = torch.randn(G, L + 64, Rv, dtype=torch.float16, device="cuda")
frag = prepare_b(b, G)
inv = rope_inv_freq(k.device)
ws = torch.zeros(_lib.lib.palu_pv_workspace_bytes(H, G, L + 64, Rv), dtype=torch.uint8, device="cuda")
dbg = torch.zeros(256 * 8 * 64, dtype=torch.int64, device="cuda")
nwg = C.c_int(0)


def call():
    _lib.check(fn(a.data_ptr(), a.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0), k.stride(1), v.data_ptr(),
                  v.stride(0), v.stride(1), ws.data_ptr(), H, G, L, Rk, Rv, inv.data_ptr(), 0, math.sqrt(D),
                  dbg.data_ptr(), C.byref(nwg), torch.cuda.current_stream().cuda_stream), "timed")


for it in range(3):
    dbg.zero_()
    call()
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8, 64)[:nwg.value].astype(np.int64)
nst = int((d[0, 0] != 0).sum())
rel = d - d[:, :1, :1]
print("nwg", nwg.value, "stamps", nst)
names = ["start", "B issued", "rope init", "fold", "dma landed"]
for i, nm in enumerate(names):
    print(f"{nm:12s} per-wave mean " + " ".join(f"{rel[:, w, i].mean():7.0f}" for w in range(8)))
P0 = 5                                  # stamps 5..: steps 8..19, each (arrive, leave, side start, side end)
nstep = (nst - P0 - 2) // 4
blk = rel[:, :, P0:P0 + 4 * nstep].reshape(rel.shape[0], 8, nstep, 4)
arr, lea, ss, se = blk[..., 0], blk[..., 1], blk[..., 2], blk[..., 3]
per = np.diff(lea, axis=2)
print("step period (leave->leave) steps 8..19:", np.round(per.mean(axis=(0, 1))).astype(int), " mean", round(per.mean()))
for w in range(8):
    print(f"   wave {w}: barrier wait {(lea - arr)[:, w].mean():6.0f}  side work {(se - ss)[:, w].mean():6.0f}  "
          f"side start after leave {(ss - lea)[:, w].mean():6.0f}  busy {(arr[:, w, 1:] - lea[:, w, :-1]).mean():6.0f}")
print("prologue (first stamped step - 8 periods) approx:", lea[:, :, 0].mean() - 8 * per.mean(), " span", rel[:, :, nst - 1].max(axis=1).mean())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for it in range(20):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"wall {us:.1f} us/call (timing build)")
ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")


def call2():
    _lib.check(_lib.lib.palu_decode_attn_f16(a.data_ptr(), a.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                             k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), ctx.data_ptr(),
                                             ws.data_ptr(), H, G, L, Rk, Rv, D, inv.data_ptr(), 0, math.sqrt(D),
                                             torch.cuda.current_stream().cuda_stream), "fused")


for it in range(5):
    call2()
torch.cuda.synchronize()
e0.record()
for it in range(50):
    call2()
e1.record()
torch.cuda.synchronize()
print(f"wall {e0.elapsed_time(e1) * 1e3 / 50:.1f} us/call (product entry: fused kernel + split merge)")
