"""palu_amd -- MI355X-native (gfx950) implementation of Palu's low-rank-KV attention decode path.

Only the hot path named in BASELINE.json lives here:
  palu_amd/csrc/     hand-written HIP kernels + the C ABI declared in include/palu_hip.h
  palu_amd/_lib.py   ctypes binding of that C ABI (fails loudly when the library is missing)
  palu_amd/kernel/   host-side mirror of the reference's Python interface for this path
                     (kernel/abx_rope.py::abx, kernel/palu_attention.py::LlamaPaluAttention)
There is no CPU fallback: every op raises if the HIP library cannot be loaded.
"""
__version__ = "0.4.0"
