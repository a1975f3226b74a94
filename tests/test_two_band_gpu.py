"""GPU parity of the two-band score kernel (csrc/abx_rope2_kernel.h) -- the kernel `abx` runs for 4 heads per latent group
at R in {32, 64, 128} -- against the CPU oracle (`torch_abx`, kernel/abx_rope.py:152-171), against the one-band kernel on
the same inputs, band by band, and the rules that select it.  Criteria: SURVEY.md 8(c) P2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle

D = 128


def _mods():
    from palu_amd import _lib
    from palu_amd.kernel import abx_rope
    return _lib, abx_rope


def _inputs(H, G, R, L, seed, band="all", scale_b=None):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((H, 1, D)).astype(np.float16)
    if band == "high":                     # only the components of RoPE pairs 0..31 (d in 0..31 and 64..95)
        a[:, :, 32:64] = 0
        a[:, :, 96:128] = 0
    elif band == "low":                    # only pairs 32..63
        a[:, :, 0:32] = 0
        a[:, :, 64:96] = 0
    b = (rng.standard_normal((H, R, D)) * (R ** -0.5 if scale_b is None else scale_b)).astype(np.float16)
    x = rng.standard_normal((G, L, R)).astype(np.float16)
    return torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(x)


def _form(ar, form):
    """The two forms of the two-band kernel: "pair_split" (csrc/abx_rope2_kernel.h), "position_split" (csrc/abx_rope3_kernel.h)
    on EVERY shape it takes -- the library itself selects it from one tile per wave on, which small test shapes never reach."""
    return ar.pair_split() if form == "pair_split" else ar.position_split(0)


FORMS = ["pair_split", "position_split"]


def _p2(got, a, b, x, theta=10000.0):
    exact = oracle.abx_scores_f64(a, b, x, theta)
    ref16 = oracle.abx_scores(a, b, x, theta)
    scale = exact.abs().max().item()
    got = got.detach().cpu().double()
    err_vs_oracle = (got - ref16.double()).abs().max().item() / scale
    err_mine = (got - exact).abs().max().item() / scale
    err_oracle = (ref16.double() - exact).abs().max().item() / scale
    assert err_vs_oracle <= 1e-3, (err_vs_oracle, err_mine, err_oracle)
    assert err_mine <= max(1.5 * err_oracle, 2.0 ** -10), (err_mine, err_oracle)
    rms_mine = ((got - exact) ** 2).mean().sqrt().item()
    rms_oracle = ((ref16.double() - exact) ** 2).mean().sqrt().item()
    assert rms_mine <= 1.25 * rms_oracle, (rms_mine, rms_oracle)       # the kernel's error stays at the oracle's own level
    return err_mine, err_oracle


def test_selection_rules():
    _lib, ar = _mods()
    dev = torch.device("cuda:0")
    inv = ar.rope_inv_freq(dev)
    sel = lambda H, G, L, R, pos0: _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, pos0)
    assert sel(32, 8, 65537, 128, 0) == 1 and sel(32, 8, 131073, 64, 0) == 1 and sel(4, 1, 100, 32, 256) == 1
    assert sel(32, 8, 1000, 128, 64) == 0              # tiles must start at multiples of 128
    assert sel(16, 8, 1000, 128, 0) == 0               # 2 heads per group
    assert sel(32, 8, 1000, 96, 0) == 1 and sel(32, 8, 1000, 224, 0) == 1      # column windows (multiples of 32)
    assert sel(32, 8, 1000, 40, 0) == 0 and sel(32, 8, 1000, 136, 0) == 0      # other ranks: one-band kernels
    assert sel(32, 8, 262145, 128, 0) == 1             # (round 5: a 256k prompt + generated tokens: 2^18 + 4096 positions)
    assert sel(32, 8, 266241, 128, 0) == 0             # positions beyond the table
    assert sel(32, 8, 204800, 128, 0) == 1             # (round 5: the low band's angle bound is 2700 rad, measured at 2621)
    with ar.one_band():
        assert sel(32, 8, 65537, 128, 0) == 0
    assert sel(32, 8, 65537, 128, 0) == 1
    inv_8400 = ar.rope_inv_freq(dev, 128, 8400.0)                   # f_32 = 0.0109: psi_max = 0.698 passes, 262 145 positions = 2860 rad do not
    sel2 = lambda L: _lib.lib.palu_abx_two_band_selected(inv_8400.data_ptr(), 32, 8, L, 128, 0)
    assert sel2(200000) == 1 and sel2(262145) == 0
    inv_small_theta = ar.rope_inv_freq(dev, 128, 1000.0)            # psi_max = 64 * 1000^-0.5 = 2.0 rad: polynomial too short
    assert _lib.lib.palu_abx_two_band_selected(inv_small_theta.data_ptr(), 32, 8, 1000, 128, 0) == 0
    inv_llama3 = ar.rope_inv_freq(dev, 128, 500000.0)
    assert _lib.lib.palu_abx_two_band_selected(inv_llama3.data_ptr(), 32, 8, 65537, 128, 0) == 1


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("band", ["high", "low", "all"])
@pytest.mark.parametrize("R,L", [(128, 1000), (64, 517), (32, 300)])
def test_bands_in_isolation_vs_oracle(band, R, L, form):
    """A query with only high-band (or only low-band) components exercises one half of the kernel alone."""
    _lib, ar = _mods()
    a, b, x = _inputs(32, 8, R, L, seed=R + L, band=band)
    with _form(ar, form):
        got = ar.abx(a.cuda(), b.cuda(), x.cuda())
    _p2(got, a, b, x)


@pytest.mark.parametrize("H,G,R,L", [(32, 8, 128, 1), (32, 8, 128, 33), (32, 8, 128, 128), (32, 8, 128, 129), (32, 8, 128, 255),
                                     (32, 8, 128, 4096 + 97), (4, 1, 128, 8191), (32, 8, 64, 2113), (32, 8, 32, 2048),
                                     (8, 2, 64, 5000)])
@pytest.mark.parametrize("form", FORMS)
def test_two_band_and_one_band_agree(H, G, R, L, form):
    _lib, ar = _mods()
    a, b, x = _inputs(H, G, R, L, seed=7 * L + R)
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    inv = ar.rope_inv_freq(xc.device)
    assert _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, 0) == 1
    with _form(ar, form):
        if form == "position_split":
            assert _lib.lib.palu_abx_position_split_selected(inv.data_ptr(), H, G, L, R, 0) == 1
        two = ar.abx(ac, bc, xc)
        again = ar.abx(ac, bc, xc)
    assert torch.equal(two, again)                     # (deterministic: no atomics, no order-dependent reduction)
    with ar.one_band():
        one = ar.abx(ac, bc, xc)
    _p2(two, a, b, x)
    _p2(one, a, b, x)
    scale = float(one.float().abs().max())
    assert float((two.float() - one.float()).abs().max()) <= 1e-3 * scale


def test_randn_scale_inputs_full_size_c2():
    """randn-scale weights (abx_rope.py:200-202) at the full config-2 size: both kernels against an fp64 evaluation on the
    GPU (oracle.abx_scores_f64 restated with torch.cuda fp64), error no worse than 1.25x the one-band kernel's."""
    _lib, ar = _mods()
    H, G, R, L = 32, 8, 128, 65537
    g = torch.Generator().manual_seed(5)
    a = torch.randn(H, 1, D, generator=g).half().cuda()
    b = torch.randn(H, R, D, generator=g).half().cuda()
    x = torch.randn(G, L, R, generator=g).half().cuda()
    inv = ar.rope_inv_freq(x.device)
    two = ar.abx(a, b, x).reshape(H, L).double()
    with ar.one_band():
        one = ar.abx(a, b, x).reshape(H, L).double()
    e2 = e1 = 0.0
    mx = 0.0
    worst = 0.0
    for l0 in range(0, L, 8192):
        l1 = min(L, l0 + 8192)
        keys = torch.matmul(x[:, None, l0:l1].double(), b.double().reshape(G, 4, R, D))
        ang = torch.outer(torch.arange(l0, l1, device=x.device, dtype=torch.int64).to(torch.float32), inv).double()
        c, s = ang.cos(), ang.sin()
        k1, k2 = keys[..., :64], keys[..., 64:]
        rot = torch.cat((k1 * c - k2 * s, k2 * c + k1 * s), dim=-1)
        ref = torch.einsum("ghd,ghld->ghl", a.double().reshape(G, 4, D), rot).reshape(H, l1 - l0)
        e2 += float(((two[:, l0:l1] - ref) ** 2).sum())
        e1 += float(((one[:, l0:l1] - ref) ** 2).sum())
        mx = max(mx, float(ref.abs().max()))
        worst = max(worst, float((two[:, l0:l1] - ref).abs().max()))
    assert worst <= 1e-3 * mx, (worst, mx)
    assert e2 <= 1.25 ** 2 * e1, (e2, e1)


@pytest.mark.parametrize("form", ["pair_split", "position_split"])
@pytest.mark.parametrize("R", [128, 64])
def test_two_band_at_256k_positions(R, form):
    """the reference's longest bench point (run_latency_kernel.py:11-12, 262 144 cached positions) + the new
    token.  The low band evaluates the exact angle l f where the oracle rounds l f to fp32 first (<= 2^-13 rad at 2621 rad):
    P2 on the LAST 8192 positions, where that difference and the high band's angle residual are largest -- error against
    fp64 no worse than 1.5x the oracle's, rms no worse than 1.25x -- for both forms of the kernel."""
    _lib, ar = _mods()
    H, G, L, W = 4, 1, 262145, 8192
    g = torch.Generator().manual_seed(R)
    a = torch.randn(H, 1, D, generator=g).half()
    b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half()
    x = torch.randn(G, L, R, generator=g).half()
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    inv = ar.rope_inv_freq(xc.device)
    assert _lib.lib.palu_abx_two_band_selected(inv.data_ptr(), H, G, L, R, 0) == 1
    ctx = ar.pair_split() if form == "pair_split" else ar.position_split(0)
    with ctx:
        got = ar.abx(ac, bc, xc)[:, :, L - W:].cpu().double().reshape(H, W)
    l0 = L - W
    xw = x[:, l0:]
    cos, sin = oracle.rope_cos_sin(L, D, start=l0)
    keys = torch.matmul(xw[:, None], b.reshape(G, H // G, R, D)).reshape(H, W, D)
    ref16 = torch.matmul(a, oracle.rope_rotate(keys, cos, sin).to(torch.float16).transpose(-1, -2)).double().reshape(H, W)
    keys64 = torch.matmul(xw.double()[:, None], b.double().reshape(G, H // G, R, D)).reshape(H, W, D)
    ang = torch.outer(torch.arange(l0, L, dtype=torch.int64).to(torch.float32), oracle.rope_inv_freq(D)).double()
    ang = torch.cat((ang, ang), dim=-1)
    exact = torch.matmul(a.double(), oracle.rope_rotate(keys64, ang.cos(), ang.sin()).transpose(-1, -2)).reshape(H, W)
    scale = exact.abs().max().item()
    e_mine, e_or = (got - exact).abs().max().item() / scale, (ref16 - exact).abs().max().item() / scale
    assert (got - ref16).abs().max().item() / scale <= 1e-3
    assert e_mine <= max(1.5 * e_or, 2.0 ** -10), (e_mine, e_or)
    rms_mine, rms_or = ((got - exact) ** 2).mean().sqrt().item(), ((ref16 - exact) ** 2).mean().sqrt().item()
    assert rms_mine <= 1.25 * rms_or, (rms_mine, rms_or)


@pytest.mark.parametrize("bits,R,L", [(4, 128, 1000), (4, 64, 4097), (4, 32, 777), (3, 128, 4193), (3, 64, 2100), (3, 32, 1300)])
def test_packed_latents_score_like_their_dequantised_rows(bits, R, L):
    """3/4-bit latents through the two-band kernel: bit-identical scores to the fp16 two-band kernel on quantize_tensor(x)
    (same LDS tile image, same MFMA stream), and P2 against the oracle on those rows."""
    _lib, ar = _mods()
    from palu_amd.kernel import quant as pq
    H, G = 32, 8
    a, b, x = _inputs(H, G, R, L, seed=bits * 1000 + L)
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    codes, meta = pq.quantize_pack(xc, bits)
    xdq = pq.unpack_dequant(codes, meta, bits, R)
    inv = ar.rope_inv_freq(xc.device)
    frag = ar.prepare_b(bc, G)
    out = torch.empty(H, 1, L, dtype=torch.float16, device=xc.device)
    _lib.check(_lib.lib.palu_abx_rope_q(ac.data_ptr(), ac.stride(0), ac.stride(2), frag.data_ptr(), codes.data_ptr(),
                                        codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                        out.data_ptr(), out.stride(0), H, G, L, R, D, bits, inv.data_ptr(), 0,
                                        torch.cuda.current_stream().cuda_stream), "abx_q")
    ref = ar.abx(ac, bc, xdq)
    assert torch.equal(out, ref)
    _p2(out, a, b, xdq.cpu())


@pytest.mark.parametrize("R,L", [(96, 2100), (160, 3000), (192, 129), (224, 4097), (256, 2500), (352, 300), (384, 1000), (512, 777)])
def test_ranks_above_128_run_as_two_band_column_windows(R, L):
    """Rank 96 and the ranks the rank search emits above 128 (palu/rank_search.py:11-17) and the reference test's 512: column windows
    (128, ..., then 32 / 64 / a 128-wide window with 96 valid columns) of the two-band kernel, fp32 partial scores, one rounding.  P2 against the oracle, and agreement with
    the one-band window passes."""
    _lib, ar = _mods()
    H, G = 32, 8
    a, b, x = _inputs(H, G, R, L, seed=11 * R + L)
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    two = ar.abx(ac, bc, xc)
    with ar.one_band():
        one = ar.abx(ac, bc, xc)
    _p2(two, a, b, x)
    _p2(one, a, b, x)
    scale = float(one.float().abs().max())
    assert float((two.float() - one.float()).abs().max()) <= 1e-3 * scale
    assert not torch.equal(two, one)          # (the two kernels round differently: identical outputs = the hook did not run)


@pytest.mark.parametrize("bits,R,L", [(4, 96, 700), (3, 96, 1500), (4, 160, 2000), (4, 224, 515), (4, 256, 4100), (3, 256, 1300), (3, 224, 1000), (3, 160, 600)])
def test_packed_ranks_above_128_score_like_their_dequantised_rows(bits, R, L):
    """Packed latents at a windowed rank: the same window plan as the fp16 rows (3-bit rows of 32 / 64 codes are staged 12 bytes per lane),
    so bit-identical to `abx` on quantize_tensor(x)."""
    _lib, ar = _mods()
    from palu_amd.kernel import quant as pq
    H, G = 32, 8
    a, b, x = _inputs(H, G, R, L, seed=bits * 100 + R)
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    codes, meta = pq.quantize_pack(xc, bits)
    xdq = pq.unpack_dequant(codes, meta, bits, R)
    inv = ar.rope_inv_freq(xc.device)
    frag = ar.prepare_b(bc, G)
    out = torch.empty(H, 1, L, dtype=torch.float16, device=xc.device)
    scr = torch.empty(max(int(_lib.lib.palu_abx_scratch_bytes(H, G, L, R)), 16), dtype=torch.uint8, device=xc.device)
    _lib.check(_lib.lib.palu_abx_rope_qg(ac.data_ptr(), ac.stride(0), ac.stride(2), frag.data_ptr(), codes.data_ptr(),
                                         codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                         out.data_ptr(), out.stride(0), H, G, L, R, D, bits, 0, inv.data_ptr(), 0,
                                         scr.data_ptr(), torch.cuda.current_stream().cuda_stream), "abx_qg")
    _p2(out, a, b, xdq.cpu())
    assert torch.equal(out, ar.abx(ac, bc, xdq))


@pytest.mark.parametrize("form", FORMS)
def test_pos_offset_in_whole_tiles_and_other_theta(form):
    """pos_offset a multiple of 128 indexes the coefficient table by absolute tile (split-L ranks, chunked callers); a
    Llama-3 style theta = 5e5 builds its own table; a cache VIEW with a row stride above R (interleaved layouts)."""
    _lib, ar = _mods()
    H, G, R, L = 32, 8, 128, 1500
    a, b, x = _inputs(H, G, R, L, seed=3)
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    with _form(ar, form):
        full = ar.abx(ac, bc, xc)
        part = ar.abx(ac, bc, xc[:, 384:].contiguous(), pos_offset=384)
        scale = float(full.float().abs().max())
        assert float((part.float() - full[:, :, 384:].float()).abs().max()) <= 5e-4 * scale
        got = ar.abx(ac, bc, xc, theta=500000.0)
        _p2(got, a, b, x, theta=500000.0)
        wide = torch.zeros(G, L, 160, dtype=torch.float16, device="cuda")
        wide[:, :, :R] = xc
        view = ar.abx(ac, bc, wide[:, :, :R])
    assert torch.equal(view, full)


@pytest.mark.parametrize("two_band", [1, 0])
@pytest.mark.parametrize("kind", ["perhead", "pairsplit", "split64", "q3", "q4", "fused_c5", "fused_c2"])
def test_cold_start_first_launch_is_deterministic(kind, two_band):
    """the first launch of every score kernel of the family in a FRESH process equals its second and
    third (the shared-B kernel's cold-start flake is guarded in test_abx_gpu.py); 8 processes per kernel kind, with the
    two-band kernel (the default for these shapes) and with the one-band kernel (PALU_ABX_TWO_BAND=0)."""
    import os
    import subprocess
    import sys
    if kind.startswith("fused") and not two_band:
        pytest.skip("the fused core has one score pipeline")
    if kind in ("pairsplit", "split64") and not two_band:
        pytest.skip("forms of the two-band kernel")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, PALU_ABX_TWO_BAND=str(two_band))
    env.pop("PALU_ABX_SPLIT", None)
    if kind == "pairsplit":                      # (round 5: "perhead" is the position-split form at this shape)
        env["PALU_ABX_SPLIT"] = "0"
    env.pop("PALU_ABX_PRIO_MODE", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "diag_cold_start.py"), kind], env=env, cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(4)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "diag_cold_start.py"), kind], env=env, cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(4)]
    outs += [p.communicate(timeout=600)[0] for p in procs]
    for o in outs:
        assert "first!=second: 0  second!=third: 0" in o, o


@pytest.mark.parametrize("bits,R,gsz,L", [(4, 128, 32, 1500), (4, 128, 64, 700), (3, 128, 32, 2100), (4, 64, 32, 900), (3, 64, 32, 1300),
                                          (4, 256, 64, 1000), (3, 160, 32, 800), (4, 96, 32, 600), (4, 224, 32, 500)])
def test_rows_quantised_in_column_groups_run_the_two_band_kernel(bits, R, gsz, L):
    """quantize_tensor with group_size > 0 (quant.py:11-13, `--lt_group_size`): every `gsz` columns of a row carry their own
    (scale, zero).  The two-band kernel picks the pair per lane piece (single launch at R in {32, 64, 128}, column windows
    otherwise): bit-identical to the fp16 kernel on the dequantised rows, P2 against the oracle."""
    _lib, ar = _mods()
    from palu_amd.kernel import quant as pq
    H, G = 32, 8
    a, b, x = _inputs(H, G, R, L, seed=bits * 100 + R + gsz)
    ac, bc, xc = a.cuda(), b.cuda(), x.cuda()
    ng = R // gsz
    c, m = pq.quantize_pack(xc.reshape(G, L, ng, gsz).contiguous(), bits)
    codes = c.reshape(G, L, -1).contiguous()
    meta = m.reshape(G, L, 2 * ng).contiguous()
    xdq = pq.unpack_dequant(c, m, bits, gsz).reshape(G, L, R)
    inv = ar.rope_inv_freq(xc.device)
    frag = ar.prepare_b(bc, G)
    out = torch.empty(H, 1, L, dtype=torch.float16, device=xc.device)
    scr = torch.empty(max(int(_lib.lib.palu_abx_scratch_bytes(H, G, L, R)), 16), dtype=torch.uint8, device=xc.device)
    _lib.check(_lib.lib.palu_abx_rope_qg(ac.data_ptr(), ac.stride(0), ac.stride(2), frag.data_ptr(), codes.data_ptr(),
                                         codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                         out.data_ptr(), out.stride(0), H, G, L, R, D, bits, gsz, inv.data_ptr(), 0,
                                         scr.data_ptr(), torch.cuda.current_stream().cuda_stream), "abx_qg")
    _p2(out, a, b, xdq.cpu())
    assert torch.equal(out, ar.abx(ac, bc, xdq))

@pytest.mark.parametrize("form", FORMS)
def test_tail_blocks_on_cold_launches(form):
    """A wave's LAST block on launches that start cold (fresh inputs, L2 / MALL turned over in between): the position-split
    kernel's counted `s_waitcnt vmcnt(N)` relied on younger requests that, past the wave's range, are re-reads of its last
    block -- and such a re-read was seen retiring ahead of the block's own request: 4..16 stale rows at the end of a group in
    7 % of such launches at R = 32, rarely at R = 64 (round 5; fixed by waiting for every request in a wave's last tile).
    48 cold launches per form, ragged lengths, against an fp64 evaluation on the GPU."""
    _lib, ar = _mods()
    dev = torch.device("cuda:0")
    inv = ar.rope_inv_freq(dev)
    big = torch.empty(1 << 28, device=dev, dtype=torch.float16)

    def ref(a, b, x):
        H, R, _ = b.shape
        G, L, _ = x.shape
        keys = torch.matmul(x[:, None].double(), b.double().reshape(G, H // G, R, D))
        ang = torch.outer(torch.arange(L, device=dev).float(), inv).double()
        c, s = ang.cos(), ang.sin()
        k1, k2 = keys[..., :64], keys[..., 64:]
        rot = torch.cat((k1 * c - k2 * s, k2 * c + k1 * s), -1)
        return torch.einsum("ghd,ghld->ghl", a.double().reshape(G, H // G, D), rot).reshape(H, L)

    bad = []
    with _form(ar, form):
        for rep in range(12):
            for R, L in ((32, 4396), (32, 300), (64, 4396), (128, 4400)):
                g = torch.Generator().manual_seed(1000 * rep + L + R)
                a = torch.randn(32, 1, D, generator=g).half().to(dev)
                b = (torch.randn(32, R, D, generator=g) * R ** -0.5).half().to(dev)
                x = torch.randn(8, L, R, generator=g).half().to(dev)
                r = ref(a, b, x)
                big[: 1 << 27].copy_(big[1 << 27:])          # 512 MB through the caches: the launch below starts cold
                y = ar.abx(a, b, x).reshape(32, L).double()
                err = float((y - r).abs().max()) / float(r.abs().max())
                if not err <= 2e-3:
                    bad.append((rep, R, L, err))
    assert not bad, bad


@pytest.mark.parametrize("fold", ["once_per_launch", "in_kernel"])
def test_one_block_tails_behind_full_tiles_on_cold_launches(fold):
    """a wave that owns full tiles AND a one-block tail (L % 128 in 1..32 with L large enough that the last wave has
    both): block 3 of its penultimate tile requested the tail block a second time (clamped re-read) and then waited with a
    counted vmcnt for the first request -- the pattern that left stale rows in test_tail_blocks_on_cold_launches' shapes, which
    never reach this path.  Round 6 waits for every request in each tile's last block.  Cold launches, every row against fp64."""
    _lib, ar = _mods()
    import contextlib
    dev = torch.device("cuda:0")
    inv = ar.rope_inv_freq(dev)
    big = torch.empty(1 << 28, device=dev, dtype=torch.float16)

    def ref_rows(a, b, x, lo):
        H, R, _ = b.shape
        G, L, _ = x.shape
        xs = x[:, lo:]
        keys = torch.matmul(xs[:, None].double(), b.double().reshape(G, H // G, R, D))
        ang = torch.outer(torch.arange(lo, L, device=dev).float(), inv).double()
        c, s = ang.cos(), ang.sin()
        k1, k2 = keys[..., :64], keys[..., 64:]
        rot = torch.cat((k1 * c - k2 * s, k2 * c + k1 * s), -1)
        return torch.einsum("ghd,ghld->ghl", a.double().reshape(G, H // G, D), rot).reshape(H, L - lo)

    bad = []
    with contextlib.ExitStack() as st:
        st.enter_context(ar.position_split(0))
        if fold == "in_kernel":
            st.enter_context(ar.in_kernel_fold())
        for rep in range(6):
            for R, L in ((32, 16400), (64, 32790), (32, 65537), (64, 65537), (128, 65537), (32, 16416)):
                g = torch.Generator().manual_seed(77 * rep + L + R)
                a = torch.randn(32, 1, D, generator=g).half().to(dev)
                b = (torch.randn(32, R, D, generator=g) * R ** -0.5).half().to(dev)
                x = torch.randn(8, L, R, generator=g).half().to(dev)
                lo = max(0, L - 4096)                        # the last waves' rows: full tiles + the tail
                r = ref_rows(a, b, x, lo)
                big[: 1 << 27].copy_(big[1 << 27:])          # 512 MB through the caches: the launch below starts cold
                y = ar.abx(a, b, x).reshape(32, L)[:, lo:].double()
                err = float((y - r).abs().max()) / float(r.abs().max())
                if not err <= 2e-3:
                    bad.append((rep, R, L, err))
    assert not bad, bad
