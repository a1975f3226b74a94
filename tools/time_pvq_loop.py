#!/usr/bin/env python3
"""us per palu_softmax_pv_q call (partial kernel + merge), back to back:  time_pvq_loop.py BITS RV_PER_GROUP L [reps]
(C3: 3 384 65536; C4: 4 192 131072)"""
import math, sys
import torch
from palu_amd import _lib
lib = _lib.lib
H, G, D = 32, 8, 128
bits, Rv, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
torch.manual_seed(0)
vb = lib.palu_packed_row_bytes(Rv, bits)
vc = torch.randint(0, 256, (G, L, vb), device="cuda", dtype=torch.uint8)
vm = torch.rand(G, L, 2, device="cuda").half() * 0.1 + 0.05
scores = torch.randn(H, (L + 7) // 8 * 8, device="cuda", dtype=torch.float16)
ctx = torch.empty(H, Rv, device="cuda", dtype=torch.float16)
ws = torch.zeros(lib.palu_pv_workspace_bytes(H, G, L, Rv) + (4 << 20), dtype=torch.uint8, device="cuda")
def call():
    _lib.check(lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, vc.data_ptr(), vc.stride(0), vc.stride(1),
                                     vm.data_ptr(), vm.stride(0), vm.stride(1), ctx.data_ptr(), 0, 0, ws.data_ptr(),
                                     H, G, L, Rv, bits, math.sqrt(D), _lib.current_stream()), "pv_q")
for _ in range(5):
    call()
best = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) * 1e3 / reps)
best.sort()
print(f"bits={bits} Rv={Rv} L={L}: median {best[2]:.2f} us, min {best[0]:.2f} us; ctx checksum {float(ctx.float().abs().sum()):.4f}")
