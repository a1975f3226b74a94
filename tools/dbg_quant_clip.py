import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests.golden import inputs as gi
from palu_amd.kernel import quant as q
R, bits = 32, 3
x = gi.quant_inputs(0, R)
deq, codes, scale, zero = oracle.quantize_rows(x.clone(), bits, 0, False, 0.9)
c, m, d = q.quantize_pack(x.cuda(), bits, want_dequant=True, sym=False, clip_ratio=0.9)
d = d.cpu(); m = m.cpu()
bad = (d.view(torch.int16) != deq.view(torch.int16))
rows = sorted(set(bad.nonzero()[:, 0].tolist()))
print("bad rows", rows)
for r in rows[:6]:
    print(r, "x max/min", x[r].max().item(), x[r].min().item(), "gpu scale/zero", m[r, 0].item(), m[r, 1].item(), "oracle", scale[r].item(), zero[r].item())
    j = bad[r].nonzero()[0].item()
    print("   elem", j, "x", x[r, j].item(), "gpu deq", d[r, j].item(), "oracle deq", deq[r, j].item(), "oracle code", codes[r, j].item(),
          "gpu code", q.unpack_codes(c, bits, R)[r, j].item())
