#!/usr/bin/env python3
"""Timeline of pv_partial_qr_kernel (PALU_PVQ_TIMELINE_DUMP=<extra bytes behind the workspace> makes every wave dump 5
wall-clock stamps there)."""
import math, os, sys
os.environ["PALU_PVQ_TIMELINE_DUMP"] = str(4 << 20)
import numpy as np
import torch
from palu_amd import _lib
lib = _lib.lib
H, G, D = 32, 8, 128
bits, Rv, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
vb = lib.palu_packed_row_bytes(Rv, bits if bits != 16 else 4)
vc = torch.randint(0, 256, (G, L, vb), device="cuda", dtype=torch.uint8)
vm = torch.rand(G, L, 2, device="cuda").half() * 0.1 + 0.05
scores = torch.randn(H, (L + 7) // 8 * 8, device="cuda", dtype=torch.float16)
ctx = torch.empty(H, Rv, device="cuda", dtype=torch.float16)
ws = torch.zeros(lib.palu_pv_workspace_bytes(H, G, L, Rv) + (4 << 20), dtype=torch.uint8, device="cuda")
v16 = torch.randn(G, L, Rv, device="cuda", dtype=torch.float16) if bits == 16 else None
for it in range(3):
    if bits == 16:
        _lib.check(lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0, v16.data_ptr(), v16.stride(0), v16.stride(1),
                                           ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, math.sqrt(D), _lib.current_stream()), "pv")
    else:
        _lib.check(lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, vc.data_ptr(), vc.stride(0), vc.stride(1),
                                         vm.data_ptr(), vm.stride(0), vm.stride(1), ctx.data_ptr(), 0, 0, ws.data_ptr(),
                                         H, G, L, Rv, bits, math.sqrt(D), _lib.current_stream()), "pv_q")
    torch.cuda.synchronize()
# the dump sits right behind part | ml | stats of THIS call's split count: find it by scanning for plausible stamps
w64 = ws.view(torch.int64).cpu().numpy()
cand = np.nonzero((w64 > 1_000_000_000) & (w64 < (1 << 62)))[0]
def plausible(seg):
    return len(seg) == 5 and seg[0] > 1_000_000_000 and np.all(np.diff(seg) >= 0) and 50 < seg[4] - seg[0] < 10_000_000
base = None
for i in cand:
    if all(plausible(w64[i + 5 * r:i + 5 * r + 5]) for r in range(4)):      # four records in a row
        base = i
        break
t = w64[base:]
t = t[: (len(t) // 5) * 5].reshape(-1, 5)
ok = (t[:, 0] > 0) & (t[:, 4] >= t[:, 0]) & (t[:, 4] - t[:, 0] < 10_000_000)
idx = np.nonzero(ok)[0]
t = t[ok].astype(np.float64)
n = len(t)
t0 = t[:, 0].min()
t = (t - t0) * 0.01          # 100 MHz ticks -> us
print(f"{n} waves; kernel span {t[:, 4].max():.1f} us")
d = np.diff(t, axis=1)
print("per wave, mean (min..max) us:  stats %.2f (%.2f..%.2f)  loop %.2f (%.2f..%.2f)  to-barrier %.2f (%.2f..%.2f)  merge %.2f (%.2f..%.2f)" % (
    d[:, 0].mean(), d[:, 0].min(), d[:, 0].max(), d[:, 1].mean(), d[:, 1].min(), d[:, 1].max(),
    d[:, 2].mean(), d[:, 2].min(), d[:, 2].max(), d[:, 3].mean(), d[:, 3].min(), d[:, 3].max()))
print("wave start times (us), percentiles 0/10/50/90/100:", np.round(np.percentile(t[:, 0], [0, 10, 50, 90, 100]), 1))
print("wave end times (us), percentiles 0/10/50/90/100:  ", np.round(np.percentile(t[:, 4], [0, 10, 50, 90, 100]), 1))

# per latent group (= XCD: workgroup id % G) and per workgroup: where is the spread?
wg = idx // 8
grp = wg % G
loop_end = t[:, 2]
print("per group (XCD): mean / max time at which a wave leaves the unit loop (us)")
print("  " + "  ".join(f"g{g}: {loop_end[grp == g].mean():.1f}/{loop_end[grp == g].max():.1f}" for g in range(G)))
wg_end = np.array([loop_end[wg == w].max() for w in np.unique(wg)])
print("per workgroup: slowest wave leaves the loop at  min %.1f  median %.1f  max %.1f us" % (wg_end.min(), np.median(wg_end), wg_end.max()))
wi = idx % 8
print("per wave index: mean time of leaving the unit loop (us): " + "  ".join(f"w{w}: {loop_end[wi == w].mean():.1f}" for w in range(8)))
print("per wave index: mean stats-done time (us):               " + "  ".join(f"w{w}: {t[wi == w, 1].mean():.1f}" for w in range(8)))
