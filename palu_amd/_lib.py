"""ctypes binding of include/palu_hip.h.  No torch types cross this boundary: tensors are passed
as (data_ptr, strides, sizes) and the current HIP stream handle.

There is deliberately NO fallback: if libpalu_hip.so is missing the import raises, so a GPU run
can never silently pass on a PyTorch/CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

LIB_PATH = os.environ.get("PALU_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpalu_hip.so")
# (PALU_HIP_LIB: an alternative build of the same library, for kernel experiments -- never a different implementation)


class PaluError(RuntimeError):
    pass


def _load():
    # torch bundles its own libamdhip64: import it FIRST so that libpalu_hip.so binds to the HIP runtime
    # torch already initialised (two runtimes in one process do not see each other's devices/streams)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m palu_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); lists every symbol declared in include/palu_hip.h
# (tests/test_cabi.py cross-checks this table against the header and the .so exports)
SIGNATURES = {
    "palu_last_error": (C.c_char_p, []),
    "palu_version": (i32, []),
    "palu_rope_inv_freq_host": (i32, [f32, i32, C.POINTER(C.c_float)]),
    "palu_abx_bfrag_bytes": (sz, [i32, i32, i32]),
    "palu_abx_set_fold": (i32, [i32]),
    "palu_abx_prepare_b": (i32, [vp, i64, i64, i64, i32, i32, i32, i32, vp, vp]),
    "palu_abx_rope_f16": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, vp, i32, vp]),
    "palu_rope_table_bytes": (sz, [i32]),
    "palu_rope_table_build": (i32, [vp, i32, i32, vp, vp]),
    "palu_rope_table_register": (i32, [vp, vp, i32, i32, f32]),
    "palu_rope_table_unregister": (i32, [vp]),
    "palu_abx_two_band_selected": (i32, [vp, i32, i32, i32, i32, i32]),
    "palu_abx_set_position_split": (i32, [i32]),
    "palu_abx_position_split_selected": (i32, [vp, i32, i32, i32, i32, i32]),
    "palu_abx_scratch_bytes": (sz, [i32, i32, i32, i32]),
    "palu_abx_fold_bytes": (sz, [i32, i32, i32]),
    "palu_abx_fold_f16": (i32, [vp, i64, i64, vp, vp, i32, i32, i32, vp]),
    "palu_abx_rope_pf_f16": (i32, [vp, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, vp, i32, vp]),
    "palu_abx_rope_ws_f16": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "palu_abx_rope_shared_f16": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, vp, i32, vp]),
    "palu_pv_nsplit": (i32, [i32, i32]),
    "palu_pv_direct_nsplit": (i32, [i32, i32, i32, i32]),
    "palu_pv_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "palu_pv_stats_offset": (sz, [i32, i32, i32, i32]),
    "palu_softmax_pv_f16": (i32, [vp, i64, vp, vp, i64, i64, vp, vp, i64, vp, i32, i32, i32, i32, f32, vp]),
    "palu_decode_attn_supported": (i32, [i32, i32, i32, i32, i32]),
    "palu_decode_attn_preferred": (i32, [i32, i32, i32, i32, i32, i32]),
    "palu_decode_attn_nsplit": (i32, [i32, i32]),
    "palu_decode_attn_stats_offset": (sz, [i32, i32, i32, i32]),
    "palu_decode_attn_f16": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, i32, i32, i32, i32, i32, i32,
                                   vp, i32, f32, vp]),
    "palu_decode_attn_mask_f16": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, vp, i32, i32, i32, i32, i32,
                                        i32, vp, i32, f32, vp]),
    "palu_gemv_f16": (i32, [vp, i64, vp, vp, i32, i32, vp]),
    "palu_gemv_bias_f16": (i32, [vp, i64, vp, vp, vp, i32, i32, vp]),
    "palu_gemv_silu_mul_f16": (i32, [vp, i64, vp, i64, vp, vp, i32, i32, vp]),
    "palu_rmsnorm_row_f16": (i32, [vp, vp, vp, i32, f32, vp]),
    "palu_decode_qkv_bias_f16": (i32, [vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, i64, i64, vp, i64, i64, vp,
                                       i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "palu_decode_qkv_fold_f16": (i32, [vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, i64, i64, vp, i64, i64, vp,
                                       i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "palu_gemv_f16_acc32": (i32, [vp, i64, vp, vp, i32, i32, vp]),
    "palu_decode_qkv_f16": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, i64, i64, vp, i64, i64, vp,
                                  i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "palu_abx_rope_q": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "palu_softmax_pv_q": (i32, [vp, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, i64, vp, i32, i32, i32, i32, i32, f32, vp]),
    "palu_decode_step_q": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64,
                                 vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64,
                                 vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "palu_abx_rope_qg": (i32, [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    "palu_softmax_pv_qg": (i32, [vp, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "palu_decode_step_qg": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64,
                                  vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64,
                                  vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "palu_lowrank_project_gemm": (i32, [vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, i32, vp]),
    "palu_lowrank_project_gemm_q": (i32, [vp, i64, vp, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, vp]),
    "palu_lowrank_project_gemm_q_supported": (i32, [i32, i32, i32, i32, i32]),
    "palu_rope_f16": (i32, [vp, i64, i64, i32, i32, i32, i32, vp, vp]),
    "palu_prefill_attn_f16": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, i32,
                                    i32, i32, f32, vp]),
    "palu_prefill_attn_panel_f16": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i32, i32, i32, i32, i32, i32,
                                          i32, i32, f32, vp, vp, i32, i32, vp]),
    "palu_prefill_state_bytes": (sz, [i32, i32, i32, i32]),
    "palu_rope_cs_table_bytes": (sz, [i32]),
    "palu_rope_cs_table_build": (i32, [vp, i32, i32, vp, vp]),
    "palu_prefill_attn_lat_supported": (i32, [i32, i32, i32, i32, i32]),
    "palu_prefill_attn_lat_supported_bits": (i32, [i32, i32, i32, i32, i32, i32]),
    "palu_prefill_attn_lat_q": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i64,
                                      i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "palu_prefill_attn_lat_f16": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32,
                                        i32, i32, i32, f32, vp]),
    "palu_packed_row_bytes": (sz, [i32, i32]),
    "palu_quantize_pack": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp]),
    "palu_quantize_pack_ex": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, f32, vp]),
    "palu_unpack_dequant": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp]),
    "palu_pack_codes": (i32, [vp, vp, i64, i32, vp]),
    "palu_unpack_codes": (i32, [vp, vp, i64, i32, vp]),
    "palu_hadamard_transform": (i32, [vp, vp, i64, i32, f32, i32, vp]),
    "palu_exchange_bytes": (sz, [i32, sz]),
    "palu_exchange_alloc": (i32, [sz, C.POINTER(C.c_void_p)]),
    "palu_exchange_free": (i32, [vp]),
    "palu_exchange_handle_bytes": (sz, []),
    "palu_exchange_export": (i32, [vp, vp]),
    "palu_exchange_import": (i32, [vp, C.POINTER(C.c_void_p)]),
    "palu_exchange_close": (i32, [vp]),
    "palu_exchange_allgather": (i32, [vp, sz, vp, i32, i32, sz, vp, vp]),
    "palu_exchange_allreduce_f32": (i32, [vp, sz, vp, i32, i32, sz, vp, vp]),
    "palu_exchange_status": (i32, [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "palu_decode_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "palu_decode_attend_f16": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, vp, vp,
                                     i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "palu_decode_step_f16": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, vp, i64, i64,
                                   vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "palu_decode_step_sharedb_f16": (i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, vp, i64, i64,
                                           vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib.palu_last_error().decode("utf-8", "replace")
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        raise PaluError(f"{what}: {msg} (code {code})")


def current_stream(device=None) -> int:
    """Stream handle the launches go to.  The library launches on the PROCESS's current device (hipGetDevice), so callers
    working on tensors of another GPU wrap their calls in `on_device(t)`; passing the device here only selects whose
    current stream is returned."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def on_device(t):
    """Context manager: make `t`'s GPU the current device for the enclosed library calls (the C ABI takes raw
    pointers and uses the current device for launches, CU counts and workspace sizing)."""
    import torch
    return torch.cuda.device(t.device)
