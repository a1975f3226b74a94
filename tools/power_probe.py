#!/usr/bin/env python3
"""Socket power and shader clock while one score kernel runs back to back for a few seconds (rocm-smi sampled from a thread).

    python tools/power_probe.py [seconds per arm]

Arms: the position-split kernel, the pair-split kernel, both again on all-zero latents (the same instruction stream at low
switching power), softmax.PV (HBM-bound) and an idle gap.  Shows whether the score kernels run against the power limit
(clock below its maximum at the limit's wattage) and what a launch costs in energy (power x time per launch)."""
import re
import subprocess
import sys
import threading
import time
import torch
from palu_amd import _lib
from palu_amd.kernel.abx_rope import abx, rope_inv_freq, pair_split

dev = torch.device("cuda:0")
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
H, G, R, L, D = 32, 8, 128, 65536, 128
g = torch.Generator().manual_seed(1)
a = torch.randn(H, 1, D, generator=g).half().to(dev)
b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half().to(dev)
x = torch.randn(G, L, R, generator=g).half().to(dev)
xz = torch.zeros_like(x)
out = torch.empty(H, 1, L, device=dev, dtype=torch.float16)
rope_inv_freq(dev)
_lib.lib.palu_abx_set_position_split(-1)

samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            o = str(e)
        pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([0-9.]+)", o) or re.findall(r"Power \(W\):\s*([0-9.]+)", o)
        sclk = re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", o)
        samples.append((time.time(), float(pw[0]) if pw else -1.0, int(sclk[0]) if sclk else -1))
        time.sleep(0.2)


def arm(name, fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(200):
            fn()
        n += 200
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    us = e0.elapsed_time(e1) * 1e3 / n
    ss = [s for s in samples if t0 + 0.7 <= s[0] <= t1]
    pw = [s[1] for s in ss if s[1] > 0]
    ck = [s[2] for s in ss if s[2] > 0]
    p = sum(pw) / len(pw) if pw else float("nan")
    print(f"{name:40s} {us:8.2f} us/launch   power {p:7.1f} W ({len(pw)} samples, min {min(pw) if pw else 0:.0f} max {max(pw) if pw else 0:.0f})   "
          f"sclk {sum(ck) / len(ck) if ck else float('nan'):6.0f} MHz   energy/launch {p * us * 1e-3:7.2f} mJ", flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
print("idle:", samples[-1] if samples else None, flush=True)
# the position-split kernel on fragments folded once (what a decode step launches behind its projection kernel), and with the
# fold in every workgroup's prologue (round 5's form)
from palu_amd.kernel.abx_rope import prepare_b, in_kernel_fold
lib, S = _lib.lib, _lib.current_stream
frag = prepare_b(b, G)
qf = torch.zeros(lib.palu_abx_fold_bytes(H, G, R), dtype=torch.uint8, device=dev)
inv = rope_inv_freq(dev)
_lib.check(lib.palu_abx_fold_f16(a.data_ptr(), a.stride(0), 1, frag.data_ptr(), qf.data_ptr(), H, G, R, S()), "fold")
out2 = out.view(H, L)


def prefolded(xx):
    _lib.check(lib.palu_abx_rope_pf_f16(qf.data_ptr(), xx.data_ptr(), xx.stride(0), xx.stride(1), out2.data_ptr(), out2.stride(0),
                                        H, G, L, R, D, inv.data_ptr(), 0, S()), "abx_pf")


arm("position-split, prefolded, randn", lambda: prefolded(x))
with in_kernel_fold():
    arm("position-split, in-kernel fold, randn", lambda: abx(a, b, x, out=out))
arm("position-split, prefolded, zero latents", lambda: prefolded(xz))
with pair_split():
    arm("pair-split, randn latents", lambda: abx(a, b, x, out=out))
with pair_split():
    arm("pair-split, zero latents", lambda: abx(a, b, xz, out=out))
big = torch.randn(1 << 28, device=dev, dtype=torch.float16)
arm("copy 512 MB (HBM-bound)", lambda: big[: 1 << 27].copy_(big[1 << 27:]))
stop = True
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout[-1500:])
except Exception as e:  # noqa: BLE001
    print("rocm-smi:", e)
