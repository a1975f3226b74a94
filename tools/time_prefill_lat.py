#!/usr/bin/env python3
"""Timeline of one workgroup of prefill_lat_kernel (csrc/prefill_lat.hip): needs a -DPL_TIMELINE build of the library
(tools/build_variant.sh pltl prefill_lat.hip "-DPL_TIMELINE"; PALU_HIP_LIB=...).  Workgroup 0 (the heaviest query tile of head 0)
stamps s_memtime at 8 points per tile for tiles 16..23: 0 phase alpha starts, 1 staging requested, 2 work done, 3 staging landed
(vmcnt(0)), 4 phase beta starts (behind the barrier), 5 softmax / P.V done, 6 rebuild done, 7 staging landed.  Waves 0-3 = S, 4-7 = O."""
import ctypes as C
import math
import numpy as np
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import rope_inv_freq

lib, S = _lib.lib, _lib.current_stream
fn = getattr(lib, "palu_prefill_lat_timeline_buffer")
fn.restype = None
fn.argtypes = [C.c_void_p]
H, G, D, Rk, Rv, T = 32, 8, 128, 128, 384, 8192
dev = torch.device("cuda:0")
inv = rope_inv_freq(dev)
torch.manual_seed(0)
q = torch.randn(H, T, D, device=dev, dtype=torch.float16)
xk = torch.randn(G, T, Rk, device=dev, dtype=torch.float16)
xv = torch.randn(G, T, Rv, device=dev, dtype=torch.float16)
bt = (torch.randn(H, D, Rk, device=dev) * Rk ** -0.5).half()
cs = torch.empty(lib.palu_rope_cs_table_bytes(T), dtype=torch.uint8, device=dev)
_lib.check(lib.palu_rope_cs_table_build(inv.data_ptr(), 0, T, cs.data_ptr(), S()), "cs")
out = torch.empty(T, H * Rv, dtype=torch.float16, device=dev)
dbg = torch.zeros(8 * 64, dtype=torch.int64, device=dev)


def run():
    _lib.check(lib.palu_prefill_attn_lat_f16(q.data_ptr(), q.stride(0), q.stride(1), xk.data_ptr(), xk.stride(0), xk.stride(1),
                                             xv.data_ptr(), xv.stride(0), xv.stride(1), bt.data_ptr(), cs.data_ptr(), out.data_ptr(),
                                             out.stride(0), H, G, D, T, T, Rk, Rv, 0, 1, 1.0 / math.sqrt(D), S()), "lat")


for _ in range(2):
    run()
fn(dbg.data_ptr())
run()
torch.cuda.synchronize()
fn(None)
d = dbg.cpu().numpy().reshape(8, 8, 8).astype(np.int64)          # [wave][tile 16..23][stamp]
t0 = d[:, 0, 0].min()
rel = d - t0
names = ["alpha start", "staging requested", "work done", "staging landed", "beta start", "softmax / P.V done", "rebuild done", "staging landed"]
print("per-tile deltas (ticks), mean over tiles 17..22; S = waves 0-3, O = waves 4-7")
for role, ws in (("S", range(0, 4)), ("O", range(4, 8))):
    seg = np.diff(rel[list(ws)][:, 1:7, :], axis=2).mean(axis=(0, 1))
    nxt = (rel[list(ws)][:, 2:8, 0] - rel[list(ws)][:, 1:7, 7]).mean()
    print(f"  {role}: " + "  ".join(f"{names[i]}->{names[i + 1]}: {seg[i]:.0f}" for i in range(7)) + f"  | barrier to next tile: {nxt:.0f}")
tile = (rel[:, 2:8, 0] - rel[:, 1:7, 0]).mean()
print(f"tile period: {tile:.0f} ticks")
print("wave 0 / wave 4, tile 18, absolute:", rel[0, 2].tolist(), rel[4, 2].tolist())
