"""GPU: bit-exact int pack/unpack and quantiser parity (criterion P3 of SURVEY.md 8(c)), Hadamard vs
the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle
from tests.golden import inputs as gi


@pytest.mark.parametrize("bits", [3, 4])
@pytest.mark.parametrize("R", [32, 64, 128, 192, 384])
def test_pack_unpack_bit_exact(bits, R):
    from palu_amd.kernel import quant as q
    rng = np.random.default_rng(bits * 1000 + R)
    codes = rng.integers(0, 1 << bits, size=(5, 33, R), dtype=np.uint8)
    codes[0, 0] = (1 << bits) - 1
    codes[0, 1] = 0
    packed = q.pack_codes(torch.from_numpy(codes).cuda(), bits)
    np.testing.assert_array_equal(packed.cpu().numpy(), oracle.pack_codes(codes, bits))      # layout contract
    np.testing.assert_array_equal(q.unpack_codes(packed, bits, R).cpu().numpy(), codes)       # inverse
    # unpack of oracle-packed bytes
    np.testing.assert_array_equal(q.unpack_codes(torch.from_numpy(oracle.pack_codes(codes, bits)).cuda(), bits, R).cpu().numpy(), codes)


def test_pack_every_value_every_position():
    from palu_amd.kernel import quant as q
    for bits in (3, 4):
        R = 32
        cases = np.zeros((R * (1 << bits), R), dtype=np.uint8)
        k = 0
        for j in range(R):
            for v in range(1 << bits):
                cases[k, j] = v
                k += 1
        packed = q.pack_codes(torch.from_numpy(cases).cuda(), bits)
        np.testing.assert_array_equal(q.unpack_codes(packed, bits, R).cpu().numpy(), cases)
        np.testing.assert_array_equal(packed.cpu().numpy(), oracle.pack_codes(cases, bits))


@pytest.mark.parametrize("R", [32, 64, 128, 384])
@pytest.mark.parametrize("bits", [3, 4])
def test_quantizer_bit_exact_vs_reference(golden_dir, R, bits):
    """codes / scale / zero equal the oracle's, dequantised values equal quantize_tensor's output bit for bit
    (golden G5, reference defaults asym / per-row / clip 1.0), including constant, zero, tie and subnormal rows."""
    from palu_amd.kernel import quant as q
    g = np.load(os.path.join(golden_dir, "g5_quant.npz"))
    x = gi.quant_inputs(0, R)
    assert gi.digest(x) == str(g[f"R{R}/digest"])
    ref = g[f"R{R}/b{bits}_sym0_g0_c1.0"]
    codes, meta, deq = q.quantize_pack(x.cuda(), bits, want_dequant=True)
    np.testing.assert_array_equal(deq.cpu().numpy().view(np.uint16), ref.view(np.uint16))
    _, ocodes, oscale, ozero = oracle.quantize_rows(x.clone(), bits)
    np.testing.assert_array_equal(q.unpack_codes(codes, bits, R).cpu().numpy(), ocodes.numpy().astype(np.uint8))
    np.testing.assert_array_equal(meta[:, 0].cpu().numpy().view(np.uint16), oscale.reshape(-1).numpy().view(np.uint16))
    np.testing.assert_array_equal(meta[:, 1].cpu().numpy(), ozero.reshape(-1).numpy())   # values (+0 == -0)
    np.testing.assert_array_equal(q.unpack_dequant(codes, meta, bits, R).cpu().numpy().view(np.uint16), ref.view(np.uint16))
    # the reference-named fake-quant entry points
    np.testing.assert_array_equal(q.quantize_tensor(x.cuda(), bits, 0, False).cpu().numpy().view(np.uint16), ref.view(np.uint16))
    if R % 32 == 0:
        refg = g[f"R{R}/b{bits}_sym0_g32_c1.0"]
        np.testing.assert_array_equal(q.quantize_tensor(x.cuda(), bits, 32, False).cpu().numpy().view(np.uint16), refg.view(np.uint16))
    qz = q.Quantizer(bits, 0, False, 1.0)
    np.testing.assert_array_equal(qz(x.cuda().reshape(2, -1, R)).cpu().numpy().reshape(-1, R).view(np.uint16), ref.view(np.uint16))
    assert q.Quantizer(16, 0, False, 1.0)(x) is x


def test_quantizer_random_rows_bit_exact_vs_oracle():
    from palu_amd.kernel import quant as q
    rng = np.random.default_rng(11)
    for R, bits in ((128, 3), (384, 3), (64, 4), (192, 4)):
        x = torch.from_numpy((rng.standard_normal((4096, R)) * rng.uniform(0.01, 30, (4096, 1))).astype(np.float16))
        deq, ocodes, osc, ozp = oracle.quantize_rows(x.clone(), bits)
        codes, meta, d = q.quantize_pack(x.cuda(), bits, want_dequant=True)
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint16), deq.numpy().view(np.uint16))
        np.testing.assert_array_equal(q.unpack_codes(codes, bits, R).cpu().numpy(), ocodes.numpy().astype(np.uint8))


@pytest.mark.parametrize("R", [32, 64, 128, 384])
@pytest.mark.parametrize("bits", [3, 4])
def test_quantizer_sym_and_clip_bit_exact_vs_reference(golden_dir, R, bits):
    """The non-default modes of quantize_tensor (quant.py:18-36: lt_sym, lt_clip_ratio < 1) against the reference's
    own outputs (golden G5): dequantised values bit for bit; codes / scale / zero against the oracle; symmetric rows are
    stored in offset binary and decode through the ordinary unpack_dequant."""
    from palu_amd.kernel import quant as q
    g = np.load(os.path.join(golden_dir, "g5_quant.npz"))
    x = gi.quant_inputs(0, R)
    assert gi.digest(x) == str(g[f"R{R}/digest"])
    for sym, gsz, clip in ((True, 0, 1.0), (False, 0, 0.9), (True, 32, 1.0)):
        key = f"R{R}/b{bits}_sym{int(sym)}_g{gsz}_c{clip}"
        if key not in g.files:
            continue
        ref = g[key]
        got = q.quantize_tensor(x.cuda(), bits, gsz, sym, clip)
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16), err_msg=key)
        if gsz == 0:
            codes, meta, deq = q.quantize_pack(x.cuda(), bits, want_dequant=True, sym=sym, clip_ratio=clip)
            _, ocodes, oscale, ozero = oracle.quantize_rows(x.clone(), bits, 0, sym, clip)
            offs = 2 ** (bits - 1) if sym else 0
            np.testing.assert_array_equal(q.unpack_codes(codes, bits, R).cpu().numpy().astype(np.int16) - offs, ocodes.numpy())
            np.testing.assert_array_equal(meta[:, 0].cpu().numpy().view(np.uint16), oscale.reshape(-1).numpy().view(np.uint16))
            np.testing.assert_array_equal(meta[:, 1].cpu().numpy() - offs, ozero.reshape(-1).numpy())
            np.testing.assert_array_equal(q.unpack_dequant(codes, meta, bits, R).cpu().numpy().view(np.uint16), ref.view(np.uint16))
    qz = q.Quantizer(bits, 0, True, 1.0)
    ref = g[f"R{R}/b{bits}_sym1_g0_c1.0"]
    np.testing.assert_array_equal(qz(x.cuda()).cpu().numpy().view(np.uint16), ref.view(np.uint16))


def test_quantizer_unsupported_modes_raise():
    from palu_amd.kernel import quant as q
    x = torch.zeros(4, 64, dtype=torch.float16, device="cuda")
    with pytest.raises(NotImplementedError):
        q.quantize_tensor(x, 8, 0, False)              # only the packed-cache bit widths
    with pytest.raises(ValueError):
        q.quantize_tensor(x, 4, 0, False, 1.5)
    with pytest.raises(ValueError):
        q.quantize_pack(x[:, :40], 3)


def test_hadamard_every_hadk_width_vs_reference(golden_dir):
    """N4: apply_hadamard accepts every width get_hadK accepts (K * 2^m, K = 12 ... 244), against the reference's own
    matmul_hadU outputs (golden G8), in both orientations; tables are the packed data file of the package."""
    from palu_amd.kernel import hadamard_utils as hu
    g = np.load(os.path.join(golden_dir, "g8_hadk.npz"))
    widths = sorted({int(k.split("/")[0][1:]) for k in g.files if k.startswith("n")})
    for n in widths:
        x = torch.from_numpy(g[f"n{n}/x"]).cuda()
        hk, K = hu.get_hadK(n)
        assert K == int(g[f"n{n}/K"])
        np.testing.assert_allclose(hu.apply_hadamard(x).cpu().numpy(), g[f"n{n}/hadU"], rtol=0, atol=1e-5, err_msg=str(n))
        np.testing.assert_allclose(hu.apply_hadamard(x, transpose=True).cpu().numpy(), g[f"n{n}/hadUt"], rtol=0, atol=1e-5)
    # orthonormal: applying the transform and its transpose returns the input
    x = torch.randn(5, 320, device="cuda")
    np.testing.assert_allclose(hu.apply_hadamard(hu.apply_hadamard(x), transpose=True).cpu().numpy(), x.cpu().numpy(), atol=2e-5)
    with pytest.raises(ValueError):
        hu.get_hadK(40 * 5)


def test_fuse_hadamard_at_rank_search_widths():
    """Ranks 160 / 224 per group (Fisher rank search widths, K = 40 and 28): fuse_hadamard() keeps the decode output."""
    from palu_amd.kernel.palu_attention import LatentCache
    from tests.test_decode_gpu import _module_from_palu_weights
    hidden, H, D, gs, L = 1024, 8, 128, 4, 200
    G = H // gs
    rank_k, rank_v = 160 * G, 224 * G
    w, k_lat, v_lat, tok, _ = gi.step_inputs(9, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    x = torch.from_numpy(np.random.default_rng(2).standard_normal((1, L, hidden)).astype(np.float16)).cuda()

    def run(mod):
        cache = LatentCache()
        with torch.no_grad():
            mod(x, past_key_value=cache, position_ids=torch.arange(L).unsqueeze(0))
            out, probs, _ = mod(tok.reshape(1, 1, hidden).cuda(), position_ids=torch.arange(L, L + 1),
                                past_key_value=cache, output_attentions=True)
        return out, probs
    o0, p0 = run(m)
    m.fuse_hadamard()
    o1, p1 = run(m)
    torch.testing.assert_close(p1, p0, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(o1, o0, rtol=2e-3, atol=2e-3)


def test_hadamard_vs_reference(golden_dir):
    from palu_amd.kernel import hadamard_utils as hu
    g = np.load(os.path.join(golden_dir, "g6_hadamard.npz"))
    np.testing.assert_array_equal(hu.get_had12().numpy(), g["had12"])
    for n in (32, 64, 128, 256, 512, 192, 384):
        x = torch.from_numpy(g[f"n{n}/x"]).cuda()
        y = hu.apply_hadamard(x)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"n{n}/hadU"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"n{n}/apply"], rtol=0, atol=5e-6)
        y16 = hu.apply_hadamard(x.half())
        np.testing.assert_allclose(y16.float().cpu().numpy(), g[f"n{n}/hadU"], rtol=0, atol=6e-3)
    # raw transform = x @ H (scale 1), large n through the same kernel
    x = torch.randn(3, 4096, device="cuda")
    y = hu.hadamard_transform(x, 1.0)
    ref = oracle.fwht(x.cpu().double())
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-3)
    vt0 = torch.from_numpy(g["fuse/vt0"]).cuda()
    u0 = [torch.from_numpy(u).cuda() for u in g["fuse/u0"]]
    vt1, u1 = hu.fuse_hadamard_into_weights(vt0.clone(), [u.clone() for u in u0])
    np.testing.assert_allclose(vt1.cpu().numpy(), g["fuse/vt1"], rtol=0, atol=5e-6)
    for a, b in zip(u1, g["fuse/u1"]):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=0, atol=5e-6)
    with pytest.raises(ValueError):
        hu.apply_hadamard(torch.zeros(2, 200, device="cuda"))      # 200 = 40 * 5: not K * 2^m (the reference asserts too)



@pytest.mark.parametrize("n", [2, 8, 32, 64, 128, 512, 1024, 2048, 4096])
def test_hadamard_transform_every_kernel_form(n):
    """palu_hadamard_transform over its kernel forms: several rows per wave (n < 64), one row per wave with 1 .. 32 elements per
    lane in registers (64 <= n <= 2048, cross-lane butterflies), one workgroup per row through LDS (above); ragged row counts;
    fp32 against the fp64 butterfly of the oracle, fp16 against the fp32 result rounded once."""
    from palu_amd.kernel import hadamard_utils as hu
    rows = 37 if n >= 64 else 3 * (64 // n) + 1
    g = torch.Generator().manual_seed(n)
    x = torch.randn(rows, n, generator=g)
    ref = oracle.fwht(x.double())
    y = hu.hadamard_transform(x.cuda(), 0.5).cpu()
    np.testing.assert_allclose(y.double().numpy(), 0.5 * ref.numpy(), rtol=0, atol=2e-6 * n ** 0.5 * 4)
    xh = x.half()
    yh = hu.hadamard_transform(xh.cuda(), 1.0).cpu()
    exp = hu.hadamard_transform(xh.float().cuda(), 1.0).cpu().half()        # same fp32 arithmetic, one rounding
    finite = torch.isfinite(exp)
    assert torch.equal(yh[finite], exp[finite])
