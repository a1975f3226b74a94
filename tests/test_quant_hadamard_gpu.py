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


def test_quantizer_unsupported_modes_raise():
    from palu_amd.kernel import quant as q
    x = torch.zeros(4, 64, dtype=torch.float16, device="cuda")
    with pytest.raises(NotImplementedError):
        q.quantize_tensor(x, 4, 0, True)
    with pytest.raises(NotImplementedError):
        q.quantize_tensor(x, 4, 0, False, 0.9)
    with pytest.raises(ValueError):
        q.quantize_pack(x[:, :40], 3)


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
    with pytest.raises(NotImplementedError):
        hu.apply_hadamard(torch.zeros(2, 160, device="cuda"))
