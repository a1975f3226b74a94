"""The C-ABI library loads on a GPU-less host and exports exactly what include/palu_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "palu_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(palu_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("palu_abx_rope_f16", "palu_abx_prepare_b", "palu_softmax_pv_f16", "palu_decode_step_f16",
                 "palu_gemv_f16", "palu_decode_qkv_f16", "palu_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from palu_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in palu_hip.h but not exported by libpalu_hip.so"


def test_binding_table_matches_header():
    from palu_amd import _lib
    declared = set(_declared())
    bound = set(_lib.SIGNATURES)
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_extern_c_no_cxx_symbols_leak():
    from palu_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    palu = [s for s in exported if s.startswith("palu_")]
    assert set(_declared()) <= set(palu)


def test_host_only_entry_points_work_without_gpu():
    from palu_amd import _lib
    import numpy as np
    buf = (ctypes.c_float * 64)()
    assert _lib.lib.palu_rope_inv_freq_host(10000.0, 128, buf) == 0
    ref = 1.0 / (10000.0 ** (np.arange(0, 128, 2, dtype=np.float32) / 128))
    np.testing.assert_allclose(np.array(buf[:]), ref, rtol=2e-7)
    assert _lib.lib.palu_rope_inv_freq_host(10000.0, 127, buf) == -1          # PALU_ERR_ARG
    assert b"rope_inv_freq" in _lib.lib.palu_last_error()
    # 4 heads per group at R in {32, 64, 128}: the fragments of both score kernels (one-band | two-band)
    assert _lib.lib.palu_abx_bfrag_bytes(32, 8, 128) == 2 * 32 * 128 * 128 * 2
    assert _lib.lib.palu_abx_bfrag_bytes(16, 8, 128) == 16 * 128 * 128 * 2       # other group sizes: one-band only
    # 1 KB of low-band coefficients + 256 B of exact-angle start values per 128-position tile, 33 x 256 B of in-block offsets
    assert _lib.lib.palu_rope_table_bytes(1 << 18) == 2048 * (1024 + 256) + 33 * 256
    assert _lib.lib.palu_abx_bfrag_bytes(32, 8, 512) == 2 * 32 * 512 * 128 * 2   # + one two-band set per 128-column window
    assert _lib.lib.palu_abx_bfrag_bytes(32, 5, 128) == 0                       # H % G != 0
    assert _lib.lib.palu_pv_workspace_bytes(32, 8, 2048, 96) > 0
    assert _lib.lib.palu_decode_workspace_bytes(32, 8, 128, 4096, 96) > 0
    assert _lib.lib.palu_version() >= 100
    # round 4 host arithmetic (no GPU): column windows of the two-band kernel (one fragment set per window: 96 -> one padded
    # 128-wide window; 160 -> 128 + 32; 224 -> 128 + padded 128), the fused-quantise GEMM's tiling rule, the prefill state
    one_band = lambda R: 32 * ((R + 127) // 128 * 128) * 128 * 2            # chunked layout: whole 128-column chunks
    two_band = lambda cols: 32 * cols * 128 * 2
    assert _lib.lib.palu_abx_bfrag_bytes(32, 8, 96) == one_band(96) + two_band(128)
    assert _lib.lib.palu_abx_bfrag_bytes(32, 8, 160) == one_band(160) + two_band(128 + 32)
    assert _lib.lib.palu_abx_bfrag_bytes(32, 8, 224) == one_band(224) + two_band(128 + 128)
    sup = _lib.lib.palu_lowrank_project_gemm_q_supported
    assert sup(4096, 1024, 4096, 128, 4) == 1 and sup(4096, 3072, 4096, 384, 3) == 1 and sup(1000, 1536, 2048, 192, 4) == 1
    assert sup(256, 1024, 4096, 128, 4) == 0 and sup(4096, 1280, 4096, 160, 4) == 0 and sup(4096, 1024, 4096, 128, 8) == 0
    assert _lib.lib.palu_prefill_state_bytes(32, 512, 384, 0) == 32 * 512 * 384 * 4
    assert _lib.lib.palu_prefill_state_bytes(32, 512, 384, 1) == 32 * 512 * 8 * 4 * (384 // 32)     # one slice per 32-column block
    # quantised P.V on the matrix cores: 32-code chunks, or 24-code chunks for 4-bit rows they fill better (192 = 8 x 24)
    assert _lib.lib.palu_pv_direct_nsplit(8, 65536, 384, 3) == 32 and _lib.lib.palu_pv_direct_nsplit(8, 131072, 192, 4) == 32
    assert _lib.lib.palu_pv_direct_nsplit(8, 1000, 40, 4) == 0


def test_position_split_switch_round_trips_without_gpu():
    """palu_abx_set_position_split (include/palu_hip.h): 0 = never, n >= 1 = from n tiles per wave, n < 0 = every shape the kernel
    takes; the call returns the previous setting in the same encoding (what the Python context managers restore)."""
    from palu_amd import _lib
    f = _lib.lib.palu_abx_set_position_split
    first = f(3)
    try:
        assert first != 0                       # the library starts with the position-split form enabled ...
        assert f(-1) == 3 and f(0) == -1 and f(2) == 0 and f(first) == 2
        assert f(first) == first                # ... from one tile per wave on (unless PALU_ABX_SPLIT=0 was set)
    finally:
        f(first)


def test_pv_workspace_covers_every_fill_level():
    """the split count is not monotone in L, so a workspace sized for a cache capacity must hold the
    partials of EVERY L <= capacity -- for the split-L P.V kernel and for the fused decode kernel (host arithmetic only;
    without a GPU the library assumes 256 CUs)."""
    from palu_amd import _lib
    lib = _lib.lib
    Rv = 384
    for G, H in ((8, 32), (1, 4), (2, 8), (4, 16), (3, 12)):
        for cap in list(range(64, 4200, 64)) + [8192, 16392, 16512, 65536 + 64, 131072 + 64, 300032]:
            nbytes = lib.palu_pv_workspace_bytes(H, G, cap, Rv)
            worst = 0
            step = 1 if cap <= 4200 else 61
            for L in list(range(1, cap + 1, step)) + [cap]:
                ns = max(lib.palu_pv_nsplit(G, L), lib.palu_decode_attn_nsplit(G, L))
                worst = max(worst, ns)
                need = (H * ns * (Rv + 2) + H * 2) * 4
                assert need <= nbytes, (G, cap, L, ns, need, nbytes)
                assert lib.palu_pv_stats_offset(H, G, L, Rv) + H * 2 * 4 <= nbytes
                assert lib.palu_decode_attn_stats_offset(H, G, L, Rv) + H * 2 * 4 <= nbytes
            assert worst >= 1
    # the register-direct quantised / fp16 P.V kernel plans its own ranges (column slices for wide rows): same bound
    for G, H in ((8, 32), (1, 4), (4, 16)):
        for Rv in (64, 128, 192, 384, 512, 1024, 4096):
            for bits in (3, 4, 16):
                for cap in (64, 1000, 4096, 16512, 65600, 300032):
                    nbytes = lib.palu_pv_workspace_bytes(H, G, cap, Rv)
                    for L in sorted({1, 31, 33, 64, 65, cap // 3, cap // 2 + 1, cap - 1, cap}):
                        if L < 1 or L > cap:
                            continue
                        ns = lib.palu_pv_direct_nsplit(G, L, Rv, bits)
                        if ns == 0:                       # (more than 128 chunks per row: the older kernels run)
                            assert bits == 16 and Rv > 1024
                            continue
                        assert (64 + H * ns * (Rv + 2)) * 4 <= nbytes, (G, Rv, bits, cap, L, ns, nbytes)
    # the round-1 failure: capacity 16512 was sized for 122 splits while L <= 16384 uses up to 128
    assert lib.palu_pv_nsplit(8, 16384) == 128
    assert lib.palu_pv_workspace_bytes(32, 8, 16512, 384) >= (32 * 128 * (384 + 2) + 64) * 4


def test_fused_attention_shape_policy():
    from palu_amd import _lib
    lib = _lib.lib
    assert lib.palu_decode_attn_supported(32, 8, 128, 384, 128) == 1
    assert lib.palu_decode_attn_supported(32, 8, 64, 192, 128) == 1
    assert lib.palu_decode_attn_supported(32, 8, 32, 96, 128) == 0       # C1 ranks: two-kernel path
    assert lib.palu_decode_attn_supported(32, 8, 128, 384, 64) == 0
    assert lib.palu_decode_attn_supported(32, 16, 128, 384, 128) == 0    # gs = 2
    if "PALU_FUSED_ATTN" not in os.environ:
        assert lib.palu_decode_attn_preferred(4, 1, 262145, 128, 384, 128) == 0    # (r5) one group, 256k: the two kernels with the two-band score kernel win again
        assert lib.palu_decode_attn_preferred(4, 1, 131073, 128, 384, 128) == 1    # one group per GPU up to ~192k positions: measured win
        assert lib.palu_decode_attn_preferred(4, 1, 65537, 128, 384, 128) == 1     # 8-GPU shard of config 2
        assert lib.palu_decode_attn_preferred(8, 2, 65537, 128, 384, 128) == 0     # 4-GPU shard of config 2: two kernels (r4)
        assert lib.palu_decode_attn_preferred(32, 8, 65537, 128, 384, 128) == 0    # config 2 on one GPU: two kernels
        assert lib.palu_decode_attn_preferred(32, 8, 2049, 128, 384, 128) == 1     # short cache
