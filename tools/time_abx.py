#!/usr/bin/env python3
"""Per-phase cycle stamps of the abx kernel (debug entry palu_abx_rope_f16_timed)."""
import ctypes as C
import sys
import numpy as np
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

L = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fn = _lib.lib.palu_abx_rope_f16_timed
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
torch.manual_seed(0)
H, G, R = 32, 8, 128
a = torch.randn(H, 1, 128, dtype=torch.float16, device="cuda")
b = torch.randn(H, R, 128, dtype=torch.float16, device="cuda")
x = torch.randn(G, L, R, dtype=torch.float16, device="cuda")
out = torch.empty(H, 1, L, dtype=torch.float16, device="cuda")
frag = prepare_b(b, G)
inv = rope_inv_freq(x.device)
dbg = torch.zeros(256 * 8 * 64, dtype=torch.int64, device="cuda")
nwg = C.c_int(0)
for it in range(3):
    dbg.zero_()
    _lib.check(fn(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1),
                  out.data_ptr(), out.stride(0), H, G, L, R, inv.data_ptr(), 0, dbg.data_ptr(), C.byref(nwg),
                  torch.cuda.current_stream().cuda_stream), "timed")
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8, 64)[:nwg.value]
nwv = int((d[0, :, 0] != 0).sum())          # 8 waves, or 4 for the one-wave-per-SIMD kernel (PALU_ABX_W4=1)
d = d[:, :nwv]
nst = int((d[0, 0] != 0).sum())
print("nwg", nwg.value, "stamps", nst)
# stamps: 0 start, 1 prologue done, 2 first tiles issued, then per tile (arrive, leave), last epilogue, end
for xcd in range(8):
    sub = d[xcd::8].astype(np.int64)
    t0 = sub[:, :, 0].min()
    st = sub[:, :, 0] - t0
    en = sub[:, :, nst - 1] - t0
    print(f"xcd {xcd}: start min {st.min()} max {st.max()}  end min {en.min()} max {en.max()}  wg-duration mean {(sub[:,:,nst-1]-sub[:,:,0]).mean():.0f}")
sub = d.astype(np.int64)
rel = sub - sub[:, :1, :1]
ntile = (nst - 7) // 2
names = ["start", "B issued", "rope init", "fold", "dma landed"]
for i, nm in enumerate(names):
    print(f"{nm:12s} " + "  ".join(f"w{k} {rel[:, k, i].mean():8.0f}" for k in sorted({0, 3, nwv // 2, nwv - 1})))
for i, nm in enumerate(names + ["arrive0", "leave0"]):
    print(f"{nm:12s} per-wave mean " + " ".join(f"{rel[:, k, i].mean():7.0f}" for k in range(nwv)) + f"   slowest wave of a WG: mean {rel[:, :, i].max(axis=1).mean():.0f} max {rel[:, :, i].max():.0f}")
L0 = 6   # first leave
print("first leave:", rel[:, :, L0].mean())
per = [(rel[:, :, L0 + 2 * t] - rel[:, :, L0 + 2 * (t - 1)]).mean() for t in range(1, ntile)]
print("tile period (leave->leave) mean over waves:", np.round(per))
wait = [(rel[:, :, L0 + 2 * t] - rel[:, :, L0 - 1 + 2 * t]) for t in range(1, ntile)]
print("barrier wait: first half of the waves mean", np.mean([w_[:, :nwv // 2].mean() for w_ in wait]), " second half mean", np.mean([w_[:, nwv // 2:].mean() for w_ in wait]))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(20):
    fn(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1),
       out.data_ptr(), out.stride(0), H, G, L, R, inv.data_ptr(), 0, dbg.data_ptr(), C.byref(nwg),
       torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"wall {dt*1e6:.1f} us/call (timing build, back-to-back) -> {rel[:, :, nst - 1].max(axis=1).mean() / dt * 1e-9:.2f} ticks/ns")
print("tail (last leave -> end):", (rel[:, :, nst - 1] - rel[:, :, L0 + 2 * (ntile - 1)]).mean(), " total", rel[:, :, nst - 1].max(axis=1).mean())
