"""First launch vs second / third launch of one kernel in a FRESH process (profiles/r03_shared_b_cold_start.txt):
    python tools/diag_cold_start.py perhead | pairsplit | split64 | shared | q3 | q4 | fused_c5 | fused_c2 | pvq3 | pvq4
Run it many times (one process each): a kernel with a start-up race differs in its first launch only."""
import sys, math, numpy as np, torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, prepare_b, rope_inv_freq
from palu_amd.kernel import quant as q
kind = sys.argv[1]
rng = np.random.default_rng(7)
D = 128
def t16(*shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float16)).cuda()
S = lambda: torch.cuda.current_stream().cuda_stream
if kind in ("perhead", "shared", "pairsplit", "split64"):
    # perhead: the default fp16 score kernel at C2 (round 5: the position-split form of the two-band kernel); pairsplit: the
    # same launch on the pair-split form (the caller sets PALU_ABX_SPLIT=0); split64: position-split at R = 64, 8 tiles per wave
    H, gs, R, L = (32, 4, 64, 131073) if kind == "split64" else (32, 4, 128, 65537)
    G = H // gs
    a = t16(H, 1, D)
    if kind == "shared":
        b = t16(G, 1, R, D, scale=R ** -0.5).expand(G, gs, R, D).reshape(H, R, D).contiguous()
    else:
        b = t16(H, R, D, scale=R ** -0.5)
    x = t16(G, L, R)
    f = lambda: abx(a, b, x)
elif kind in ("q3", "q4"):
    bits = 3 if kind == "q3" else 4
    H, gs, R, L = (32, 4, 128, 65537) if bits == 3 else (32, 4, 64, 131073)
    G = H // gs
    a = t16(H, 1, D); b = t16(H, R, D, scale=R ** -0.5); x = t16(G, L, R)
    codes, meta = q.quantize_pack(x, bits)
    frag = prepare_b(b, G); inv = rope_inv_freq(x.device)
    def f():
        out = torch.empty(H, 1, L, dtype=torch.float16, device="cuda")
        _lib.check(_lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                            codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                            out.data_ptr(), out.stride(0), H, G, L, R, 128, bits, inv.data_ptr(), 0, S()), "abx_q")
        return out
elif kind in ("fused_c5", "fused_c2"):
    H, G, Rk, Rv, L = (4, 1, 128, 384, 262144) if kind == "fused_c5" else (32, 8, 128, 384, 30000)
    qv = t16(H, D); b = t16(H, Rk, D, scale=Rk ** -0.5); k = t16(G, L, Rk); v = t16(G, L, Rv)
    frag = prepare_b(b, G); inv = rope_inv_freq(k.device)
    ws = torch.zeros(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
    def f():
        ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
        _lib.check(_lib.lib.palu_decode_attn_f16(qv.data_ptr(), qv.stride(0), qv.stride(1), frag.data_ptr(), k.data_ptr(),
                                                 k.stride(0), k.stride(1), v.data_ptr(), v.stride(0), v.stride(1),
                                                 ctx.data_ptr(), ws.data_ptr(), H, G, L, Rk, Rv, D, inv.data_ptr(), 0,
                                                 math.sqrt(D), S()), "decode_attn")
        return ctx
elif kind in ("pvq3", "pvq4"):
    # quantised softmax.PV (pv_partial_qr_kernel: the one kernel that raises a wave priority, statically for its whole loop)
    bits = 3 if kind == "pvq3" else 4
    H, G, Rv, L = (32, 8, 384, 65537) if bits == 3 else (32, 8, 192, 131073)
    scores = t16(H, L, scale=20.0)
    v = t16(G, L, Rv)
    codes, meta = q.quantize_pack(v, bits)
    ws = torch.zeros(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
    def f():
        ctx = torch.empty(H, Rv, dtype=torch.float16, device="cuda")
        _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, codes.data_ptr(), codes.stride(0),
                                              codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1), ctx.data_ptr(),
                                              0, 0, ws.data_ptr(), H, G, L, Rv, bits, math.sqrt(D), S()), "pv_q")
        return ctx
first = f(); torch.cuda.synchronize()
second = f(); torch.cuda.synchronize()
third = f(); torch.cuda.synchronize()
n12 = int((first != second).sum().item()); n23 = int((second != third).sum().item())
print(kind, "first!=second:", n12, " second!=third:", n23)
