"""Build the C-ABI shared library with hipcc for gfx950 (cross-compiles without a GPU).

    python -m palu_amd.build [--force]

Output: palu_amd/lib/libpalu_hip.so (git-ignored; travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libpalu_hip.so")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
# per-file flags: the position-split score kernel schedules its VALU work by hand (no packed fp32 forms)
FILE_FLAGS = {"abx_rope3.hip": ["-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage"]}
# kernels that must not touch scratch: the position-split score kernel sits at 256 VGPRs + 256 AGPRs with hand-placed
# asm MFMAs -- a spill there is slow at best and has produced wrong scores.  Validated with the hipcc of ROCm 7.2.0;
# after a compiler change re-run tools/check_fused.py and tools/stress_tail_cold.py on a GPU.  (name fragment, excluded fragment)
NO_SCRATCH = {"abx_rope3.hip": ("abx_rope3_kernelILi", "ELb1ELb")}     # (the TIMING instantiations of experiment builds may spill)


def _check_no_scratch(src: str, out: str) -> str:
    """Parse -Rpass-analysis=kernel-resource-usage remarks; raise when a guarded kernel needs scratch.  Returns the
    compiler output without the remarks."""
    want, skip = NO_SCRATCH[os.path.basename(src)]
    keep, name, seen = [], None, 0
    for line in out.splitlines():
        if "remark:" not in line:
            keep.append(line)
            continue
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize" in line and name and want in name and skip not in name:
            seen += 1
            nbytes = int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])
            if nbytes:
                raise RuntimeError(f"{os.path.basename(src)}: {name} spills {nbytes} bytes/lane to scratch "
                                   f"(it must fit 256 VGPRs + 256 AGPRs; see palu_amd/build.py NO_SCRATCH)")
    if not seen:
        raise RuntimeError(f"{os.path.basename(src)}: no resource-usage remark for {want}* (compiler output format changed?)")
    # (the remarks come with source excerpts; keep the rest only when it carries a diagnostic of its own)
    return "\n".join(keep) if any("warning:" in l or "error:" in l for l in keep) else ""


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(PKG), "include", "*.h"))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objs, procs = [], []
    for src in sources():                      # one hipcc per translation unit, in parallel
        obj = os.path.join(LIBDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, *CFLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *os.environ.get("PALU_EXTRA_CFLAGS", "").split(), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if os.path.basename(src) in NO_SCRATCH:
            out = _check_no_scratch(src, out)
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
