"""`abx(a, b, x)` -- drop-in for the reference's `kernel.abx_rope.abx` (kernel/abx_rope.py:114-150,
imported as `recompute_k_gemv` at kernel/palu_attention.py:13 and called at :219).

Same signature, shapes and conventions: a [H,1,D], b [H,R,D], x [G,L,R] fp16 on one device ->
[H,1,L] fp16 freshly allocated; no 1/sqrt(D); RoPE theta 1e4; key position = row index of x.
Extensions (keyword-only): `theta`, `pos_offset`, `out`.  Arbitrary L (masked tail), arbitrary
H/G, strided a/b; x rows must be contiguous (else one contiguous copy is made, like the cat in
HF's cache did).  Runs on the current stream, allocation only through torch's caching allocator:
safe under torch.cuda.graph capture once the B fragments are cached (first call outside capture).
"""
from __future__ import annotations

import contextlib
import weakref

import torch

from .. import _lib

_inv_freq_cache = {}
_bfrag_cache = {}


def rope_inv_freq(device, head_dim: int = 128, theta: float = 10000.0) -> torch.Tensor:
    """fp32 table 1/theta^(2i/D), computed with the same torch expression as
    kernel/pytorch_reference.py:4 so the angles are bit-identical to the oracle's."""
    key = (str(device), head_dim, float(theta))
    t = _inv_freq_cache.get(key)
    if t is None:
        host = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
        t = host.to(device)
        _inv_freq_cache[key] = t
        if head_dim == 128 and torch.device(device).type == "cuda":
            _register_rope_table(t, host)
    return t


_cs_tables = {}


def rope_cs_table(device, head_dim: int, theta: float, npos: int) -> torch.Tensor:
    """The rotary cache of the key positions 0 .. npos - 1 as the latent prefill kernel reads it (csrc/prefill_lat.hip):
    [pos][cos 64 | sin 64] fp16 with HF's arithmetic (fp32 angle fl32(pos) * inv_freq, fp32 cos / sin, cast to fp16: the
    `cos_cached` / `sin_cached` of LlamaRotaryEmbedding, kernel/palu_attention.py:204).  One per (device, theta), grown in steps of
    4096 positions and kept (16 MiB at 64k positions) -- persistent like HF's cache, not a transient of a prompt pass."""
    key = (str(device), head_dim, float(theta))
    t = _cs_tables.get(key)
    if t is None or t.shape[0] < npos:
        n = (npos + 4095) // 4096 * 4096
        inv = rope_inv_freq(device, head_dim, theta)
        with _lib.on_device(inv):
            t = torch.empty((n, head_dim), dtype=torch.float16, device=inv.device)
            _lib.check(_lib.lib.palu_rope_cs_table_build(inv.data_ptr(), 0, n, t.data_ptr(), _lib.current_stream()),
                       "palu_rope_cs_table_build")
        _cs_tables[key] = t
    return t


ROPE_TABLE_POSITIONS = (1 << 18) + 4096      # what the two-band score kernels cover (csrc/abx_rope2.hip): a 256k prompt + 4096 generated tokens; 2.6 MB per (device, theta)
_rope_tables = {}


def _register_rope_table(inv: torch.Tensor, host: torch.Tensor) -> None:
    """Low-band RoPE coefficients of the two-band score kernel for positions [0, 2^18): a function of the frequencies
    only, built once per (device, theta) next to the inverse-frequency table and registered under its device pointer
    (include/palu_hip.h: palu_rope_table_build / _register).  Launches that pass this `inv` then select that kernel
    when their shape and positions allow."""
    with _lib.on_device(inv):
        tab = torch.empty(_lib.lib.palu_rope_table_bytes(ROPE_TABLE_POSITIONS), dtype=torch.uint8, device=inv.device)
        _lib.check(_lib.lib.palu_rope_table_build(inv.data_ptr(), 0, ROPE_TABLE_POSITIONS, tab.data_ptr(),
                                                  _lib.current_stream()), "palu_rope_table_build")
        _lib.check(_lib.lib.palu_rope_table_register(inv.data_ptr(), tab.data_ptr(), 0, ROPE_TABLE_POSITIONS,
                                                     float(host[32])), "palu_rope_table_register")
    _rope_tables[inv.data_ptr()] = (tab, float(host[32]))   # keeps the table alive as long as the cached frequencies


@contextlib.contextmanager
def one_band():
    """Run the enclosed launches on the one-band score kernel (csrc/abx_rope_kernel.h) by taking the coefficient tables out
    of the registry for the duration -- for A/B measurements and for tests that compare scores bit for bit with the fused
    attention core, which carries the one-band pipeline.  Not thread-safe (process-wide registry)."""
    for ptr in _rope_tables:
        _lib.lib.palu_rope_table_unregister(ptr)
    try:
        yield
    finally:
        for ptr, (tab, f32_) in _rope_tables.items():
            _lib.check(_lib.lib.palu_rope_table_register(ptr, tab.data_ptr(), 0, ROPE_TABLE_POSITIONS, f32_),
                       "palu_rope_table_register")


@contextlib.contextmanager
def pair_split():
    """Run the enclosed launches on the pair-split form of the two-band kernel (csrc/abx_rope2_kernel.h) instead of the
    position-split one (csrc/abx_rope3_kernel.h) -- A/B measurements and cross-checks.  Process-wide."""
    old = _lib.lib.palu_abx_set_position_split(0)
    try:
        yield
    finally:
        _lib.lib.palu_abx_set_position_split(old)


@contextlib.contextmanager
def position_split(min_tiles: int = 0):
    """Run the enclosed launches on the position-split kernel: on every shape it takes (min_tiles = 0, the default of this
    context manager) or from `min_tiles` tiles of 128 positions per wave on (the library's own rule: from 1)."""
    old = _lib.lib.palu_abx_set_position_split(int(min_tiles) if min_tiles > 0 else -1)
    try:
        yield
    finally:
        _lib.lib.palu_abx_set_position_split(old)


_fold_once_per_launch = True


@contextlib.contextmanager
def in_kernel_fold():
    """Run the enclosed `abx` calls without the scratch that lets the position-split kernel take fragments folded ONCE per
    launch (csrc/abx_fold.h): the kernel then folds the query in every workgroup's prologue, as in round 5 -- same arithmetic,
    bit-identical scores (tests/test_fold_gpu.py), for A/B measurements."""
    global _fold_once_per_launch
    old, _fold_once_per_launch = _fold_once_per_launch, False
    try:
        yield
    finally:
        _fold_once_per_launch = old


def set_fold(enable: bool) -> bool:
    """Numerics switch of the fast path (palu_abx_set_fold): True (default) folds q into the B
    fragments (one extra fp16 operand rounding, fewer VALU ops); False keeps q in fp32."""
    return bool(_lib.lib.palu_abx_set_fold(1 if enable else 0))


def invalidate_b(b: torch.Tensor | None = None) -> None:
    """Drop the cached fragments of `b` (or of every tensor).  The cache key is (object, _version, data_ptr); writes
    through `.data` (`b.data.copy_(...)`, the idiom of weight loading and of fuse_hadamard) change none of them, so code
    that mutates B that way calls this afterwards (LlamaPaluAttention.fuse_hadamard / load_state_dict hooks do)."""
    if b is None:
        _bfrag_cache.clear()
        _shared_cache.clear()
    else:
        _bfrag_cache.pop(id(b), None)
        _shared_cache.pop(id(b), None)


def prepare_b(b: torch.Tensor, num_groups: int) -> torch.Tensor:
    """Lay the weight B [H,R,D] out as MFMA A-operand fragments (palu_abx_prepare_b).  Cached per
    tensor object + version, because B is a weight (nn.Parameter at kernel/palu_attention.py:114); see invalidate_b
    for in-place updates through `.data`."""
    key = id(b)
    hit = _bfrag_cache.get(key)
    if hit is not None:
        ref, ver, ptr, G, frag = hit
        if ref() is b and ver == b._version and ptr == b.data_ptr() and G == num_groups:
            return frag
    H, R, D = b.shape
    nbytes = _lib.lib.palu_abx_bfrag_bytes(H, num_groups, R)
    if nbytes == 0:
        raise ValueError(f"abx: unsupported shape H={H} G={num_groups} R={R} (need H%G==0, R%8==0)")
    frag = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    _lib.check(_lib.lib.palu_abx_prepare_b(b.data_ptr(), b.stride(0), b.stride(1), b.stride(2),
                                           H, num_groups, R, D, frag.data_ptr(), _lib.current_stream()),
               "palu_abx_prepare_b")
    if len(_bfrag_cache) > 256:
        for k in [k for k, v in list(_bfrag_cache.items()) if v[0]() is None]:
            del _bfrag_cache[k]
    _bfrag_cache[key] = (weakref.ref(b), b._version, b.data_ptr(), num_groups, frag)
    return frag


_shared_cache: dict = {}


def shared_b(b: torch.Tensor, num_groups: int):
    """If B [H,R,D] is identical for the heads of every latent group (true-GQA: the query heads of a group share one KV
    head), return the [G,R,D] shared factor, else None.  Cached per tensor like prepare_b (one device comparison per
    weight); shapes the shared kernel does not cover return None."""
    H, R, D = b.shape
    gs = H // num_groups
    if gs < 2 or gs > 4 or R not in (32, 64, 128) or H % num_groups:
        return None
    key = id(b)
    hit = _shared_cache.get(key)
    if hit is not None:
        ref, ver, ptr, G, res = hit
        if ref() is b and ver == b._version and ptr == b.data_ptr() and G == num_groups:
            return res
    bg = b.view(num_groups, gs, R, D)
    res = bg[:, 0].contiguous() if bool((bg == bg[:, :1]).all()) else None
    if len(_shared_cache) > 256:
        for k in [k for k, v in list(_shared_cache.items()) if v[0]() is None]:
            del _shared_cache[k]
    _shared_cache[key] = (weakref.ref(b), b._version, b.data_ptr(), num_groups, res)
    return res


def abx(a: torch.Tensor, b: torch.Tensor, x: torch.Tensor, *, theta: float = 10000.0,
        pos_offset: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    assert a.dim() == 3
    assert b.dim() == 3
    assert x.dim() == 3
    if not (a.is_cuda and b.is_cuda and x.is_cuda):
        raise RuntimeError("abx: tensors must live on a ROCm device (no CPU fallback)")
    if not (a.dtype == b.dtype == x.dtype == torch.float16):
        raise TypeError("abx: fp16 only (kernel/abx_rope.py:170 casts to float16)")
    H, one, D = a.shape
    Hb, R, Db = b.shape
    G, L, Rx = x.shape
    if one != 1 or Hb != H or Db != D or Rx != R or H % G != 0:
        raise ValueError(f"abx: inconsistent shapes a{tuple(a.shape)} b{tuple(b.shape)} x{tuple(x.shape)}")
    if x.stride(2) != 1 or x.stride(1) % 8 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    with _lib.on_device(x):                 # launches go to the current device: make it the tensors' GPU
        if out is None:
            out = torch.empty((H, 1, L), dtype=x.dtype, device=x.device)
        else:
            assert out.shape == (H, 1, L) and out.dtype == torch.float16 and out.stride(2) == 1
        inv = rope_inv_freq(x.device, D, theta)
        bg = shared_b(b, G)
        if bg is not None:
            # every head of a group uses the same B: reconstruct the keys once per group (a quarter of the MFMA work)
            frag = prepare_b(bg, G)
            _lib.check(_lib.lib.palu_abx_rope_shared_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(),
                                                         x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(),
                                                         out.stride(0), H, G, L, R, D, inv.data_ptr(), int(pos_offset),
                                                         _lib.current_stream()), "palu_abx_rope_shared_f16")
            return out
        frag = prepare_b(b, G)
        # ranks above 128: fp32 scratch for the multi-pass form of the fast kernel; 4 heads per group at R in {32, 64, 128}: the
        # folded fragments of the position-split kernel (16 R KB per group); 0 bytes otherwise
        nscr = _lib.lib.palu_abx_scratch_bytes(H, G, L, R)
        if R <= 128 and not _fold_once_per_launch:
            nscr = 0
        scratch = torch.empty(nscr, dtype=torch.uint8, device=x.device) if nscr else None
        _lib.check(_lib.lib.palu_abx_rope_ws_f16(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(),
                                                 x.data_ptr(), x.stride(0), x.stride(1),
                                                 out.data_ptr(), out.stride(0), H, G, L, R, D,
                                                 inv.data_ptr(), int(pos_offset), 0 if scratch is None else scratch.data_ptr(),
                                                 _lib.current_stream()),
                   "palu_abx_rope_ws_f16")
    return out
