#!/usr/bin/env python3
"""Debug: compare the two-band kernel's first W image / folded low fragments / coefficient fragments (workgroup 0) with a
numpy restatement.  PYTHONPATH=. python tools/diag_two_band_dump.py"""
import ctypes, math
import numpy as np
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, rope_inv_freq

dev = torch.device("cuda:0")
H, G, R, L, D = 32, 8, 128, 384, 128
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.randn(H, 1, D, generator=g).half()
b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half()
x = torch.randn(G, L, R, generator=g).half()
inv = rope_inv_freq(dev)
dbg = torch.zeros(512 * 16 * 11, dtype=torch.uint8, device=dev)
fn = _lib.lib.palu_abx2_debug_buffer
fn.restype = None
fn.argtypes = [ctypes.c_void_p]
fn(dbg.data_ptr())
y = abx(a.to(dev), b.to(dev), x.to(dev))
torch.cuda.synchronize()
fn(None)
raw = dbg.cpu().numpy()
wimg = raw[:8192].view(np.float16).reshape(8, 2, 32, 8)           # [ks][hiA][m][e]
lowf = raw[8192:8192 + 8 * 8192].view(np.float16).reshape(4, 2, 8, 64, 8)   # [h][cs][wave][lane][8]
cf = raw[8192 * 9:8192 * 11].view(np.float16).reshape(2, 8, 64, 8)          # [cs][wave][lane][8]

invh = inv.cpu().numpy().astype(np.float64)
q = a[:4, 0].float().numpy()            # group 0 heads
Bf = b[:4].float().numpy()              # [4, R, D]
qi, qj = q[:, :64], q[:, 64:]
P = (qi[:, None, :] * Bf[:, :, :64] + qj[:, None, :] * Bf[:, :, 64:]).astype(np.float16)
Q = (qj[:, None, :] * Bf[:, :, :64] - qi[:, None, :] * Bf[:, :, 64:]).astype(np.float16)
# expected folded low fragments: wave rb, lane (m16, q): [(P,Q) of pairs 32 + 16cs + 4q + e4] of row 16 rb + m16
bad = 0
for h in range(4):
    for cs in range(2):
        for rb in range(8):
            for lane in range(64):
                m16, qq = lane & 15, lane >> 4
                r = 16 * rb + m16
                exp = np.empty(8, np.float16)
                for e4 in range(4):
                    i = 32 + 16 * cs + 4 * qq + e4
                    exp[2 * e4], exp[2 * e4 + 1] = P[h, r, i], Q[h, r, i]
                if not np.array_equal(exp, lowf[h, cs, rb, lane]):
                    bad += 1
                    if bad < 4:
                        print("lowf mismatch h", h, "cs", cs, "rb", rb, "lane", lane, exp, lowf[h, cs, rb, lane])
print("lowf mismatching fragments:", bad)
# expected coefficient fragments of tile 0
psimax = 64.0 * invh[32]
bad = 0
for cs in range(2):
    for lane in range(64):
        k, qq = lane & 15, lane >> 4
        exp = np.zeros(8, np.float16)
        for e4 in range(4):
            i = 32 + 16 * cs + 4 * qq + e4
            phi = 63.5 * invh[i]
            rel = (64.0 * invh[i] / psimax) ** k
            if k < 8:
                exp[2 * e4] = np.float16(rel * math.cos(phi + k * math.pi / 2))
                exp[2 * e4 + 1] = np.float16(rel * math.sin(phi + k * math.pi / 2))
        got = cf[cs, 0, lane]
        if np.abs(exp.astype(np.float32) - got.astype(np.float32)).max() > 2e-3:
            bad += 1
            if bad < 4:
                print("coef mismatch cs", cs, "lane", lane, exp, got)
print("coef mismatching fragments:", bad)
# expected W image
al = np.zeros((8, 32)); be = np.zeros((8, 32))
for k in range(8):
    for ii in range(32):
        i = 32 + ii
        phi = 63.5 * invh[i]
        rel = (64.0 * invh[i] / psimax) ** k
        al[k, ii] = np.float16(rel * math.cos(phi + k * math.pi / 2))
        be[k, ii] = np.float16(rel * math.sin(phi + k * math.pi / 2))
W = np.einsum("ki,hri->hkr", al, P[:, :, 32:].astype(np.float64)) + np.einsum("ki,hri->hkr", be, Q[:, :, 32:].astype(np.float64))
got = np.zeros((4, 8, 128))
for ks in range(8):
    for hiA in range(2):
        for m in range(32):
            for e in range(8):
                got[m >> 3, m & 7, 16 * ks + 8 * hiA + e] = wimg[ks, hiA, m, e]
err = np.abs(got - W)
print("W image: max |err|", err.max(), "max |W|", np.abs(W).max())
if err.max() > 0.05:
    for h in range(4):
        for k in range(8):
            print("h", h, "k", k, "err", err[h, k].max(), "got[:6]", got[h, k, :6], "exp[:6]", W[h, k, :6])
