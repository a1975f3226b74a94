"""The query fold of the position-split score kernel as its own step (csrc/abx_fold.h, round 6): three forms of the same
arithmetic -- the fold inside the kernel's prologue (round 5), the stand-alone fold kernel, and the fold in the tail of the
projection kernel's q waves (the decode step's form) -- must give bit-identical folded fragments and scores.  The weight the
fold builds is the one the reference's `_abx_fwd` multiplies the rotated key with (kernel/abx_rope.py:79-111), with the query
moved onto B; the scores are checked against the oracle (`torch_abx`, kernel/abx_rope.py:152-171) as well."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle

D, HIDDEN = 128, 1024


def _mods():
    from palu_amd import _lib
    from palu_amd.kernel import abx_rope
    return _lib, abx_rope


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().cuda()


@pytest.mark.parametrize("R,L", [(128, 65537), (128, 4400), (128, 1000), (64, 4396), (64, 131), (32, 300), (32, 16400)])
def test_three_forms_of_the_fold_are_bit_identical(R, L):
    _lib, ar = _mods()
    lib, S = _lib.lib, _lib.current_stream
    H, G = 32, 8
    Rv = 64
    wq, vtk, vtv = _rand((H * D, HIDDEN), 1, 1 / 32), _rand((G * R, HIDDEN), 2, 1 / 32), _rand((G * Rv, HIDDEN), 3, 1 / 32)
    hidden = _rand((HIDDEN,), 4)
    b = _rand((H, R, D), 5, R ** -0.5)
    cap = L + 7
    kc, vc = _rand((G, cap, R), 6), _rand((G, cap, Rv), 7)
    kc2, vc2 = kc.clone(), vc.clone()
    inv = ar.rope_inv_freq(kc.device)
    frag = ar.prepare_b(b, G)
    nfold = lib.palu_abx_fold_bytes(H, G, R)
    assert nfold == G * 16 * (R // 16) * 1024 and lib.palu_abx_scratch_bytes(H, G, L, R) == nfold
    pos = L - 1

    # (1) plain projection kernel, then the score kernel folding in its own prologue
    q1 = torch.empty(H * D, dtype=torch.float16, device="cuda")
    _lib.check(lib.palu_decode_qkv_f16(wq.data_ptr(), HIDDEN, vtk.data_ptr(), HIDDEN, vtv.data_ptr(), HIDDEN, hidden.data_ptr(),
                                       q1.data_ptr(), kc.data_ptr(), kc.stride(0), kc.stride(1), vc.data_ptr(), vc.stride(0),
                                       vc.stride(1), inv.data_ptr(), H, D, HIDDEN, G, R, Rv, pos, pos, S()), "qkv")
    # (3) projection kernel with the fold in its tail
    q3 = torch.empty_like(q1)
    f3 = torch.zeros(nfold, dtype=torch.uint8, device="cuda")
    _lib.check(lib.palu_decode_qkv_fold_f16(wq.data_ptr(), HIDDEN, 0, vtk.data_ptr(), HIDDEN, vtv.data_ptr(), HIDDEN,
                                            hidden.data_ptr(), q3.data_ptr(), kc2.data_ptr(), kc2.stride(0), kc2.stride(1),
                                            vc2.data_ptr(), vc2.stride(0), vc2.stride(1), inv.data_ptr(), H, D, HIDDEN, G, R, Rv,
                                            pos, pos, frag.data_ptr(), f3.data_ptr(), S()), "qkv_fold")
    assert torch.equal(q1, q3) and torch.equal(kc, kc2) and torch.equal(vc, vc2)
    # (2) stand-alone fold of the same query
    f2 = torch.zeros(nfold, dtype=torch.uint8, device="cuda")
    _lib.check(lib.palu_abx_fold_f16(q1.data_ptr(), D, 1, frag.data_ptr(), f2.data_ptr(), H, G, R, S()), "fold")
    assert torch.equal(f2, f3)

    a = q1.view(H, 1, D)
    x = kc[:, :L]
    with ar.position_split(0):
        assert lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, L, R, 0) == 1
        with ar.in_kernel_fold():
            s1 = ar.abx(a, b, x)
        s2 = ar.abx(a, b, x)                                    # fold kernel + PREFOLD kernel (scratch from palu_abx_scratch_bytes)
        s3 = torch.empty((H, (L + 15) // 8 * 8), dtype=torch.float16, device="cuda")
        _lib.check(lib.palu_abx_rope_pf_f16(f3.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), s3.data_ptr(), s3.stride(0),
                                            H, G, L, R, D, inv.data_ptr(), 0, S()), "abx_pf")
    assert torch.equal(s1, s2)
    assert torch.equal(s1.view(H, L), s3[:, :L])
    # and they are the oracle's scores (P2)
    ref = oracle.abx_scores(a.cpu(), b.cpu(), x.cpu())
    scale = ref.float().abs().max().item()
    assert (s1.cpu().float() - ref.float()).abs().max().item() <= 1e-3 * scale


def test_prefolded_entry_refuses_launches_without_the_position_split_kernel():
    _lib, ar = _mods()
    lib, S = _lib.lib, _lib.current_stream
    H, G, R, L = 32, 8, 128, 300                     # far below one tile per wave: the library's own rule picks the pair-split form
    inv = ar.rope_inv_freq(torch.device("cuda:0"))
    x = _rand((G, L, R), 1)
    f = torch.zeros(lib.palu_abx_fold_bytes(H, G, R), dtype=torch.uint8, device="cuda")
    out = torch.empty((H, L + 8), dtype=torch.float16, device="cuda")
    assert lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, L, R, 0) == 0
    rc = lib.palu_abx_rope_pf_f16(f.data_ptr(), x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(), out.stride(0), H, G, L, R,
                                  D, inv.data_ptr(), 0, S())
    assert rc != 0 and b"position-split" in lib.palu_last_error()
    assert lib.palu_abx_fold_bytes(24, 8, 128) == 0 and lib.palu_abx_fold_bytes(32, 8, 96) == 0
    rc = lib.palu_abx_fold_f16(x.data_ptr(), D, 1, f.data_ptr(), f.data_ptr(), 24, 8, 128, S())
    assert rc != 0


def test_decode_step_takes_the_prefolded_kernel_and_matches_the_unfused_pieces():
    """palu_decode_step_f16 at a shape where the position-split kernel is selected (forced here): projection kernel with the fold
    in its tail + palu_abx_rope_pf_f16; its output must equal the step run with the in-kernel fold (palu_abx_set_position_split
    off -> pair-split kernel gives other roundings, so the comparison is against the pieces launched by hand)."""
    _lib, ar = _mods()
    lib, S = _lib.lib, _lib.current_stream
    H, G, R, Rv, L = 32, 8, 128, 384, 4500      # (G L > 24576: the two-kernel path, not the fused core)
    hid = 4096
    wq, vtk, vtv = _rand((H * D, hid), 1, 1 / 64), _rand((G * R, hid), 2, 1 / 64), _rand((G * Rv, hid), 3, 1 / 64)
    wo = _rand((hid, H * Rv), 8, 0.01)
    hidden = _rand((hid,), 4)
    b = _rand((H, R, D), 5, R ** -0.5)
    cap = L + 8
    kc, vc = _rand((G, cap, R), 6), _rand((G, cap, Rv), 7)
    kc2, vc2 = kc.clone(), vc.clone()
    inv = ar.rope_inv_freq(kc.device)
    frag = ar.prepare_b(b, G)
    ws = torch.empty(lib.palu_decode_workspace_bytes(H, G, D, cap, Rv), dtype=torch.uint8, device="cuda")
    out = torch.empty(hid, dtype=torch.float16, device="cuda")
    n = L - 1
    with ar.position_split(0):
        _lib.check(lib.palu_decode_step_f16(hidden.data_ptr(), wq.data_ptr(), hid, vtk.data_ptr(), hid, vtv.data_ptr(), hid,
                                            frag.data_ptr(), wo.data_ptr(), wo.stride(0), kc.data_ptr(), kc.stride(0), kc.stride(1),
                                            vc.data_ptr(), vc.stride(0), vc.stride(1), 0, inv.data_ptr(), out.data_ptr(), 0, 0,
                                            ws.data_ptr(), cap, H, G, D, hid, R, Rv, n, n, S()), "step")
        # the pieces by hand, the score kernel folding in its prologue
        q = torch.empty(H * D, dtype=torch.float16, device="cuda")
        _lib.check(lib.palu_decode_qkv_f16(wq.data_ptr(), hid, vtk.data_ptr(), hid, vtv.data_ptr(), hid, hidden.data_ptr(),
                                           q.data_ptr(), kc2.data_ptr(), kc2.stride(0), kc2.stride(1), vc2.data_ptr(), vc2.stride(0),
                                           vc2.stride(1), inv.data_ptr(), H, D, hid, G, R, Rv, n, n, S()), "qkv")
        sc = torch.empty((H, (L + 15) // 8 * 8), dtype=torch.float16, device="cuda")
        _lib.check(lib.palu_abx_rope_f16(q.data_ptr(), D, 1, frag.data_ptr(), kc2.data_ptr(), kc2.stride(0), kc2.stride(1),
                                         sc.data_ptr(), sc.stride(0), H, G, L, R, D, inv.data_ptr(), 0, S()), "abx")
    ctx = torch.empty(H * Rv, dtype=torch.float16, device="cuda")
    pv = torch.empty(lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device="cuda")
    _lib.check(lib.palu_softmax_pv_f16(sc.data_ptr(), sc.stride(0), 0, vc2.data_ptr(), vc2.stride(0), vc2.stride(1), ctx.data_ptr(),
                                       0, 0, pv.data_ptr(), H, G, L, Rv, float(np.sqrt(D)), S()), "pv")
    out2 = torch.empty_like(out)
    _lib.check(lib.palu_gemv_f16(wo.data_ptr(), wo.stride(0), ctx.data_ptr(), out2.data_ptr(), hid, H * Rv, S()), "o")
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
    assert torch.equal(out, out2)
