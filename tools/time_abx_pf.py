#!/usr/bin/env python3
"""us per score launch (fp16 latents, 32 heads, 8 groups), back to back, for the three forms of the position-split kernel's
query fold:  time_abx_pf.py R L [R L ...]
  in_kernel : palu_abx_rope_f16 -- every workgroup folds the query in its prologue (round 5)
  prefolded : palu_abx_rope_pf_f16 on fragments folded once (what a decode step launches after its projection kernel)
  fold+kernel: palu_abx_rope_ws_f16 with scratch -- the stand-alone op (fold kernel + score kernel)
PALU_HIP_LIB selects a variant build (e.g. -DABX3_WAIT0: vmcnt(0) in every block)."""
import sys
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq

lib, S = _lib.lib, _lib.current_stream
H, G, D = 32, 8, 128
torch.manual_seed(0)
args = [int(v) for v in sys.argv[1:]] or [128, 65536]
inv = rope_inv_freq(torch.device("cuda:0"))
for R, L in zip(args[0::2], args[1::2]):
    a = torch.randn(H, D, device="cuda", dtype=torch.float16)
    b = (torch.randn(H, R, D, device="cuda") * R ** -0.5).half()
    x = torch.randn(G, L, R, device="cuda", dtype=torch.float16)
    out = torch.empty(H, (L + 15) // 8 * 8, device="cuda", dtype=torch.float16)
    frag = prepare_b(b, G)
    qf = torch.zeros(max(16, lib.palu_abx_fold_bytes(H, G, R)), dtype=torch.uint8, device="cuda")
    sel = lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, L, R, 0)

    def in_kernel():
        _lib.check(lib.palu_abx_rope_f16(a.data_ptr(), D, 1, frag.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(),
                                         out.stride(0), H, G, L, R, D, inv.data_ptr(), 0, S()), "abx")

    def prefolded():
        _lib.check(lib.palu_abx_rope_pf_f16(qf.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(), out.stride(0),
                                            H, G, L, R, D, inv.data_ptr(), 0, S()), "abx_pf")

    def fold_kernel():
        _lib.check(lib.palu_abx_rope_ws_f16(a.data_ptr(), D, 1, frag.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(),
                                            out.stride(0), H, G, L, R, D, inv.data_ptr(), 0, qf.data_ptr(), S()), "abx_ws")

    forms = [("in_kernel", in_kernel)]
    if sel:
        _lib.check(lib.palu_abx_fold_f16(a.data_ptr(), D, 1, frag.data_ptr(), qf.data_ptr(), H, G, R, S()), "fold")
        forms += [("prefolded", prefolded), ("fold+kernel", fold_kernel)]
    res = {}
    sums = {}
    for rnd in range(3):                                   # interleaved rounds: box drift cancels
        for name, fn in forms:
            for _ in range(5):
                fn()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 30)
            res.setdefault(name, []).extend(ts)
            sums[name] = float(out[:, :L].float().abs().sum())
    line = f"R={R} L={L} position_split={sel}:"
    for name, _ in forms:
        ts = sorted(res[name])
        line += f"  {name} median {ts[len(ts) // 2]:.2f} min {ts[0]:.2f} us"
    line += "  checksums " + ("equal" if len(set(sums.values())) == 1 else repr(sums))
    print(line, flush=True)
