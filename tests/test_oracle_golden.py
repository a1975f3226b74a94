"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from tests.golden import inputs as gi


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("case", gi.ABX_CASES, ids=[c[0] for c in gi.ABX_CASES])
def test_abx_matches_reference(golden_dir, case):
    tag, seed, H, D, gs, R, L, regime = case
    g = _load(golden_dir, "g1_abx")
    a, b, x = gi.abx_inputs(seed, H, D, gs, R, L, regime)
    assert gi.digest(a, b, x) == str(g[tag + "/digest"]), "input generation drifted"
    ref = torch.from_numpy(g[tag + "/out"])
    got = oracle.abx_scores(a, b, x)
    assert got.shape == (H, 1, L) and got.dtype == torch.float16
    # same torch build -> bit-identical; allow 2 fp16 ulp for other BLAS backends
    scale = ref.float().abs().max().item()
    assert (got.float() - ref.float()).abs().max().item() <= 2 * scale * 2 ** -10
    # the fp64 restatement is what the fp16 oracle approximates
    exact = oracle.abx_scores_f64(a, b, x)
    assert (ref.double() - exact).abs().max().item() <= 3e-3 * exact.abs().max().item()


def test_rope_tables_and_rotation(golden_dir):
    g = _load(golden_dir, "g2_rope")
    for k, p in enumerate(g["positions"].tolist()):
        cos, sin = oracle.rope_cos_sin(p + 1, 128, start=p)
        np.testing.assert_array_equal(cos[0].numpy(), g["cos"][k])
        np.testing.assert_array_equal(sin[0].numpy(), g["sin"][k])
    cos = torch.from_numpy(g["cos"])
    sin = torch.from_numpy(g["sin"])
    rot = oracle.rope_rotate(torch.from_numpy(g["x"]), cos, sin)
    np.testing.assert_array_equal(rot.numpy(), g["rotated"])


def test_b_layout_and_wo_fusion(golden_dir):
    g = _load(golden_dir, "g4_layout")
    gs, D = int(g["gs"]), int(g["D"])
    u_k = [torch.from_numpy(u) for u in g["u_k"]]
    u_v = [torch.from_numpy(u) for u in g["u_v"]]
    np.testing.assert_array_equal(oracle.build_b_from_u(u_k, gs, D).numpy(), g["b"])
    fused = oracle.fuse_uv_into_wo(torch.from_numpy(g["wo"]), u_v, gs, D)
    np.testing.assert_allclose(fused.numpy(), g["wo_fused"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", gi.STEP_CASES, ids=[c[0] for c in gi.STEP_CASES])
def test_decode_step_matches_reference(golden_dir, case):
    tag, seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask = case
    g = _load(golden_dir, "g3_decode_step")
    w, k_lat, v_lat, tok, mask = gi.step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask)
    flat = [w["wq"], w["vt_k"], w["vt_v"], w["wo"], *w["u_k"], k_lat, v_lat, tok]
    assert gi.digest(*flat) == str(g[tag + "/digest"])
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    out, probs, k_all, v_all = oracle.decode_step(tok, L, wd, k_lat, v_lat, mask)
    torch.testing.assert_close(out, torch.from_numpy(g[tag + "/attn_output"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(probs, torch.from_numpy(g[tag + "/attn_weights"]), rtol=1e-3, atol=1e-5)
    np.testing.assert_array_equal(k_all[:, L].numpy(), g[tag + "/k_new"])
    np.testing.assert_array_equal(v_all[:, L].numpy(), g[tag + "/v_new"])
    assert abs(probs.float().sum(-1) - 1).max() < 2e-2          # fp16 rows still sum to ~1


@pytest.mark.parametrize("case", gi.PREFILL_CASES, ids=[c[0] for c in gi.PREFILL_CASES])
def test_prefill_matches_reference(golden_dir, case):
    tag, seed, hidden, H, D, gs, rank_k, rank_v, T, causal = case
    g = _load(golden_dir, "g7_prefill")
    w, prompt, mask = gi.prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, causal)
    flat = [w["wq"], w["vt_k"], w["vt_v"], w["wo"], *w["u_k"], prompt]
    assert gi.digest(*flat) == str(g[tag + "/digest"])
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "u_k": [u.half() for u in w["u_k"]], "wo": w["wo"].half()}
    out, probs, k_lat, v_lat = oracle.prefill(prompt, wd, mask)
    torch.testing.assert_close(out, torch.from_numpy(g[tag + "/attn_output"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(probs, torch.from_numpy(g[tag + "/attn_weights"]), rtol=1e-3, atol=1e-5)
    np.testing.assert_array_equal(k_lat.numpy(), g[tag + "/k_lat"])
    np.testing.assert_array_equal(v_lat.numpy(), g[tag + "/v_lat"])


def test_quantizer_bit_exact(golden_dir):
    g = _load(golden_dir, "g5_quant")
    n = 0
    for R in gi.QUANT_R:
        x = gi.quant_inputs(0, R)
        assert gi.digest(x) == str(g[f"R{R}/digest"])
        for bits in (3, 4):
            for sym in (False, True):
                for gsz in (0, 32):
                    for clip in ((1.0, 0.9) if (gsz == 0 and not sym) else (1.0,)):
                        ref = g[f"R{R}/b{bits}_sym{int(sym)}_g{gsz}_c{clip}"]
                        deq, codes, sc, zp = oracle.quantize_rows(x.clone(), bits, gsz, sym, clip)
                        assert deq.dtype == torch.float16
                        np.testing.assert_array_equal(deq.numpy().view(np.uint16), ref.view(np.uint16))
                        # the exposed integer codes reproduce the dequantised values bit-exactly
                        rows = codes.reshape(sc.shape[0], -1)
                        back = oracle.dequant_codes(rows, sc, zp).reshape(deq.shape)
                        np.testing.assert_array_equal(back.numpy().view(np.uint16), ref.view(np.uint16))
                        lo, hi = (-(2 ** (bits - 1)), 2 ** (bits - 1) - 1) if sym else (0, 2 ** bits - 1)
                        assert int(codes.min()) >= lo and int(codes.max()) <= hi
                        n += 1
    assert n == 4 * 2 * (2 + 1 + 2 * 1)


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("R", [32, 64, 128, 192, 384])
def test_pack_unpack_roundtrip(bits, R):
    rng = np.random.default_rng(R * 10 + bits)
    codes = rng.integers(0, 1 << bits, size=(7, 5, R), dtype=np.uint8)
    codes[0, 0, :] = (1 << bits) - 1
    codes[0, 1, :] = 0
    packed = oracle.pack_codes(codes, bits)
    assert packed.shape == (7, 5, oracle.packed_row_bytes(R, bits)) and packed.dtype == np.uint8
    np.testing.assert_array_equal(oracle.unpack_codes(packed, bits, R), codes)
    # layout contract: little-endian bit stream, code j at bits [j*b, (j+1)*b)
    row = packed[3, 2].astype(np.uint64)
    stream = sum(int(v) << (8 * i) for i, v in enumerate(row.tolist()))
    for j in (0, 1, 7, 8, 31, R - 1):
        assert (stream >> (j * bits)) & ((1 << bits) - 1) == int(codes[3, 2, j])


def test_pack_exhaustive_small():
    # every code value at every position of a 32-code group (3-bit) / 8-code group (4-bit)
    for bits in (3, 4):
        R = 32
        for j in range(R):
            for v in range(1 << bits):
                c = np.zeros((1, R), dtype=np.uint8)
                c[0, j] = v
                np.testing.assert_array_equal(oracle.unpack_codes(oracle.pack_codes(c, bits), bits, R), c)


def test_hadamard_matches_reference(golden_dir):
    g = _load(golden_dir, "g6_hadamard")
    np.testing.assert_array_equal(oracle.had12().numpy(), g["had12"])
    h12 = oracle.had12()
    torch.testing.assert_close(h12 @ h12.t(), 12 * torch.eye(12))
    for n in (32, 64, 128, 256, 512, 192, 384):
        x = torch.from_numpy(g[f"n{n}/x"])
        y = oracle.apply_hadamard(x)
        np.testing.assert_allclose(y.numpy(), g[f"n{n}/hadU"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(y.numpy(), g[f"n{n}/apply"], rtol=0, atol=5e-6)
        torch.testing.assert_close(y.norm(dim=-1), x.norm(dim=-1), rtol=1e-5, atol=1e-5)  # orthogonal
    vt1, u1 = oracle.fuse_hadamard_into_weights(torch.from_numpy(g["fuse/vt0"]),
                                                [torch.from_numpy(u) for u in g["fuse/u0"]])
    np.testing.assert_allclose(vt1.numpy(), g["fuse/vt1"], rtol=0, atol=5e-6)
    for a, b in zip(u1, g["fuse/u1"]):
        np.testing.assert_allclose(a.numpy(), b, rtol=0, atol=5e-6)
    # invariance U'.VT' == U.VT (svd_linear.py:156-168)
    vt0 = torch.from_numpy(g["fuse/vt0"])
    for i, (un, uo) in enumerate(zip(u1, g["fuse/u0"])):
        torch.testing.assert_close(un @ vt1[32 * i:32 * i + 32], torch.from_numpy(uo) @ vt0[32 * i:32 * i + 32],
                                   rtol=1e-4, atol=1e-4)


def test_fwht_is_sylvester():
    import scipy.linalg
    for n in (2, 8, 64, 512):
        x = torch.randn(3, n, dtype=torch.float64)
        ref = x @ torch.from_numpy(scipy.linalg.hadamard(n).astype(np.float64))
        torch.testing.assert_close(oracle.fwht(x), ref)


def test_reftest_fixture_is_lossless(golden_dir):
    """Full-rank Palu == vanilla attention (kernel/test_palu_attention.py:158-195, rtol=atol=1e-3)."""
    g = _load(golden_dir, "g3b_reftest")
    torch.testing.assert_close(torch.from_numpy(g["decode_weights"]).float(),
                               torch.from_numpy(g["vanilla_weights"]), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(torch.from_numpy(g["decode_output"]).float(),
                               torch.from_numpy(g["vanilla_output"]), rtol=1e-3, atol=1e-3)


def test_oracle_hadamard_every_hadk_width(golden_dir):
    """G8: every width n = K * 2^m get_hadK (hadamard_utils.py:5-83) accepts, K in {12 ... 244}, incl. the widths the
    Fisher rank search produces (160, 224, 320, ...): oracle.apply_hadamard == the reference's matmul_hadU."""
    g = np.load(os.path.join(golden_dir, "g8_hadk.npz"))
    widths = sorted({int(k.split("/")[0][1:]) for k in g.files if k.startswith("n")})
    assert len(widths) >= 40 and {160, 224, 320, 96, 384}.issubset(widths)
    seen = set()
    for n in widths:
        x = torch.from_numpy(g[f"n{n}/x"])
        hk, K = oracle.hadK_for(n)
        assert K == int(g[f"n{n}/K"]), (n, K)
        seen.add(K)
        np.testing.assert_allclose(oracle.apply_hadamard(x).numpy(), g[f"n{n}/hadU"], rtol=0, atol=5e-6, err_msg=str(n))
        if hk is not None:
            m = hk.numpy().astype(np.int64)
            assert np.array_equal(m @ m.T, K * np.eye(K, dtype=np.int64))
    assert seen == set(oracle.HAD_K_ORDER)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_from_linear_whiten_matches_reference(golden_dir, tag):
    """palu/model/modules/svd_linear.py:6-34,170-204 (whitened group-wise SVD): the restatement against factors the reference
    produced (g9_whiten.npz).  Factors of an SVD are defined up to the signs of the singular vectors: the products U_g VT_g
    (what the module computes) are compared element-wise, the factors themselves up to a per-component sign."""
    g = _load(golden_dir, "g9_whiten")
    ranks = [int(r) for r in g[f"{tag}/ranks"]]
    w, sc = torch.from_numpy(g[f"{tag}/w"]), torch.from_numpy(g[f"{tag}/scaling"])
    bias = torch.from_numpy(g[f"{tag}/b"]) if f"{tag}/b" in g else None
    us, vt, bs = oracle.from_linear_whiten(w, bias, sc, ranks)
    vt_ref = torch.from_numpy(g[f"{tag}/vt"])
    assert vt.shape == vt_ref.shape
    r0 = 0
    for i, r in enumerate(ranks):
        u_ref = torch.from_numpy(g[f"{tag}/u{i}"])
        np.testing.assert_allclose((us[i] @ vt[r0:r0 + r]).numpy(), (u_ref @ vt_ref[r0:r0 + r]).numpy(), rtol=0, atol=2e-5)
        sign = torch.sign((us[i] * u_ref).sum(dim=0))
        np.testing.assert_allclose((us[i] * sign).numpy(), u_ref.numpy(), rtol=0, atol=2e-4)
        if bias is not None:
            np.testing.assert_array_equal(bs[i].numpy(), g[f"{tag}/bias{i}"])
        r0 += r
