"""GPU parity of the HIP abx_rope kernel (through the C ABI) against the CPU oracle and the golden
vectors generated from the reference.  Criteria: SURVEY.md section 8(c) P1/P2."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle
from tests.golden import inputs as gi


def _abx():
    from palu_amd.kernel.abx_rope import abx
    return abx


def _check(got, a, b, x, ref16=None, tol=1e-3):
    """P2: normalised max error vs the fp16 oracle <= 1e-3 and error vs fp64 no worse than 1.5x
    the oracle's own error vs fp64."""
    exact = oracle.abx_scores_f64(a, b, x)
    if ref16 is None:
        ref16 = oracle.abx_scores(a, b, x)
    scale = exact.abs().max().item()
    got = got.detach().cpu()
    assert got.shape == ref16.shape and got.dtype == torch.float16
    assert torch.isfinite(got.float()).all()
    err_vs_oracle = (got.double() - ref16.double()).abs().max().item() / scale
    err_mine = (got.double() - exact).abs().max().item() / scale
    err_oracle = (ref16.double() - exact).abs().max().item() / scale
    assert err_vs_oracle <= tol, (err_vs_oracle, err_mine, err_oracle)
    assert err_mine <= max(1.5 * err_oracle, 2.0 ** -10), (err_mine, err_oracle)
    return err_mine, err_oracle


# "default": the library's own selection (at these lengths the pair-split form of the two-band kernel, or the one-band kernel for
# the shapes it does not take); the other forms are forced onto every case they take: the position-split
# kernel on fragments folded once per launch (round 6) and with the fold in its own prologue (round 5), the pair-split
# kernel, the one-band kernel -- each of them against the vectors the reference itself produced
@pytest.mark.parametrize("form", ["default", "position_split", "position_split_in_kernel_fold", "pair_split", "one_band"])
@pytest.mark.parametrize("case", gi.ABX_CASES, ids=[c[0] for c in gi.ABX_CASES])
def test_abx_golden(golden_dir, case, form):
    import contextlib
    from palu_amd import _lib
    from palu_amd.kernel import abx_rope as ar
    tag, seed, H, D, gs, R, L, regime = case
    g = np.load(os.path.join(golden_dir, "g1_abx.npz"))
    a, b, x = gi.abx_inputs(seed, H, D, gs, R, L, regime)
    assert gi.digest(a, b, x) == str(g[tag + "/digest"])
    with contextlib.ExitStack() as st:
        if form.startswith("position_split"):
            st.enter_context(ar.position_split(0))
            if form.endswith("in_kernel_fold"):
                st.enter_context(ar.in_kernel_fold())
            if gs == 4 and R in (32, 64, 128):
                inv = ar.rope_inv_freq(torch.device("cuda:0"), D)
                assert _lib.lib.palu_abx_position_split_selected(inv.data_ptr(), H, H // gs, L, R, 0) == 1
        elif form == "pair_split":
            st.enter_context(ar.pair_split())
        elif form == "one_band":
            st.enter_context(ar.one_band())
        got = _abx()(a.cuda(), b.cuda(), x.cuda())
    _check(got, a, b, x, ref16=torch.from_numpy(g[tag + "/out"]))


@pytest.mark.parametrize("H,gs,R,L", [
    (32, 4, 128, 1), (32, 4, 128, 31), (32, 4, 128, 127), (32, 4, 128, 128), (32, 4, 128, 129),
    (32, 4, 128, 4096 + 65), (32, 4, 64, 3000), (32, 4, 32, 5000), (32, 4, 96, 300),
    (32, 4, 256, 513), (16, 8, 128, 700), (12, 3, 64, 257), (6, 6, 32, 130), (4, 1, 128, 200),
    (8, 2, 160, 333),
])
def test_abx_shapes(H, gs, R, L):
    rng = np.random.default_rng(H * 1000 + R + L)
    G = H // gs
    a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16))
    b = torch.from_numpy((rng.standard_normal((H, R, 128)) / np.sqrt(R)).astype(np.float16))
    x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16))
    got = _abx()(a.cuda(), b.cuda(), x.cuda())
    _check(got, a, b, x)


def test_abx_strided_inputs_and_cache_view():
    """a from a transposed view (palu_attention.py:170,216), b non-contiguous, x a [:, :L] view of
    a pre-allocated cache (stride_g = Lmax*R)."""
    rng = np.random.default_rng(5)
    H, G, R, L, Lmax = 32, 8, 128, 777, 1024
    q = torch.from_numpy(rng.standard_normal((1, 1, H, 128)).astype(np.float16)).cuda()
    a = q.transpose(1, 2).squeeze(0)                       # [H,1,D] view
    bfull = torch.from_numpy((rng.standard_normal((H, R, 256)) / 11.3).astype(np.float16)).cuda()
    b = bfull[:, :, ::2]                                    # stride_d = 2
    cache = torch.from_numpy(rng.standard_normal((G, Lmax, R)).astype(np.float16)).cuda()
    x = cache[:, :L]
    got = _abx()(a, b, x)
    _check(got, a.cpu().contiguous(), b.cpu().contiguous(), x.cpu().contiguous())


def test_abx_pos_offset_equals_shifted_rows():
    rng = np.random.default_rng(6)
    H, G, R, L = 32, 8, 64, 384
    a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16)).cuda()
    b = torch.from_numpy((rng.standard_normal((H, R, 128)) / 8).astype(np.float16)).cuda()
    x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16)).cuda()
    full = _abx()(a, b, x)
    part = _abx()(a, b, x[:, 100:].contiguous(), pos_offset=100)
    assert (full[:, :, 100:].float() - part.float()).abs().max().item() <= 2e-3 * full.float().abs().max().item()


def test_abx_long_context_positions():
    """RoPE angles at positions up to 64k must follow the oracle's fp32-rounded l*inv_freq
    (pytorch_reference.py:5-6): compare a window at the end of a 65,600-row cache."""
    rng = np.random.default_rng(7)
    H, G, R = 32, 8, 32
    L = 65600
    a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16))
    b = torch.from_numpy((rng.standard_normal((H, R, 128)) / np.sqrt(R)).astype(np.float16))
    x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16))
    got = _abx()(a.cuda(), b.cuda(), x.cuda()).cpu()
    ref = oracle.abx_scores(a, b, x)
    exact = oracle.abx_scores_f64(a, b, x)
    scale = exact.abs().max().item()
    assert (got.double() - ref.double()).abs().max().item() / scale <= 1e-3
    e_mine = (got.double() - exact).abs().max().item() / scale
    e_ref = (ref.double() - exact).abs().max().item() / scale
    assert e_mine <= max(1.5 * e_ref, 2.0 ** -10), (e_mine, e_ref)


def _scores_f64_at(a, b, x, pos0):
    """fp64 evaluation of the abx math with the oracle's fp32-rounded angles fl32(l * inv_freq) at rows pos0 + l
    (pytorch_reference.py:5-6, abx_rope.py:152-171)"""
    H, R, D = b.shape
    G, L, _ = x.shape
    keys = torch.matmul(x.double()[:, None], b.double().reshape(G, H // G, R, D)).reshape(H, L, D)
    pos = torch.arange(pos0, pos0 + L, dtype=torch.int64).to(torch.float32)
    ang = torch.outer(pos, oracle.rope_inv_freq(D)).double()
    ang = torch.cat((ang, ang), dim=-1)
    keys = oracle.rope_rotate(keys, ang.cos(), ang.sin())
    return torch.matmul(a.double(), keys.transpose(-1, -2))


@pytest.mark.parametrize("R", [128, 64])
def test_abx_positions_beyond_2_18(R):
    """run_latency_attention.py builds its model with max_position_embeddings = 300000: past 2^18 positions the fast
    kernel switches to the second-order angle correction (ORDER2).  Window [290000, 300000) against the fp64 value of
    the oracle's formula; the first-order kernel on the same window (forced through a small pos0 + huge L is not
    possible, so the check is the error level itself) must stay at the fp16 output rounding."""
    rng = np.random.default_rng(R)
    H, G, L, pos0 = 32, 8, 10000, 290000
    a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16))
    b = torch.from_numpy((rng.standard_normal((H, R, 128)) / np.sqrt(R)).astype(np.float16))
    x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16))
    got = _abx()(a.cuda(), b.cuda(), x.cuda(), pos_offset=pos0).cpu()
    exact = _scores_f64_at(a, b, x, pos0)
    scale = exact.abs().max().item()
    err = (got.double() - exact).abs().max().item() / scale
    assert err <= 2.0 ** -10, err          # half an fp16 ulp of the largest score + the fp16 operand roundings
    # the same rows at small positions give the same error level (the angle correction is not the limiting term)
    got0 = _abx()(a.cuda(), b.cuda(), x.cuda(), pos_offset=0).cpu()
    err0 = (got0.double() - _scores_f64_at(a, b, x, 0)).abs().max().item() / scale
    assert err <= 1.5 * err0 + 1e-5, (err, err0)


@pytest.mark.parametrize("H,gs,R,L", [(32, 4, 128, 2049), (32, 4, 64, 700), (8, 2, 32, 333), (24, 3, 128, 1000), (32, 4, 128, 65537)])
def test_abx_shared_b_fast_path(H, gs, R, L):
    """N3: when the heads of a group share B (true-GQA checkpoints) abx reconstructs the keys once per group
    (palu_abx_rope_shared_f16, picked automatically by `abx`).  Against the oracle with the tied B, and against the
    per-head kernel fed the same tied B through the C ABI."""
    from palu_amd import _lib
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq, shared_b
    rng = np.random.default_rng(H + R + L)
    G = H // gs
    a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16))
    bg = torch.from_numpy((rng.standard_normal((G, 1, R, 128)) / np.sqrt(R)).astype(np.float16))
    b = bg.expand(G, gs, R, 128).reshape(H, R, 128).contiguous()
    x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16))
    ad, bd, xd = a.cuda(), b.cuda(), x.cuda()
    assert shared_b(bd, G) is not None
    got = _abx()(ad, bd, xd)
    # the per-head kernel on the same operands (q kept in fp32 like the shared kernel: fold off is the closest twin)
    frag = prepare_b(bd.clone(), G)
    inv = rope_inv_freq(xd.device)
    ref_k = torch.empty(H, 1, L, dtype=torch.float16, device="cuda")
    _lib.check(_lib.lib.palu_abx_rope_f16(ad.data_ptr(), ad.stride(0), ad.stride(2), frag.data_ptr(), xd.data_ptr(),
                                          xd.stride(0), xd.stride(1), ref_k.data_ptr(), ref_k.stride(0), H, G, L, R, 128,
                                          inv.data_ptr(), 0, torch.cuda.current_stream().cuda_stream), "abx")
    scale = ref_k.float().abs().max().item()
    assert (got.float() - ref_k.float()).abs().max().item() <= 1e-3 * scale
    if L <= 4096:
        o = oracle.abx_scores(a, b, x)
        exact = oracle.abx_scores_f64(a, b, x)
        sc = exact.abs().max().item()
        assert (got.cpu().double() - o.double()).abs().max().item() <= 1e-3 * sc
        e_mine = (got.cpu().double() - exact).abs().max().item() / sc
        e_ref = (o.double() - exact).abs().max().item() / sc
        assert e_mine <= max(1.5 * e_ref, 2.0 ** -10), (e_mine, e_ref)
    # a B that differs in one head is not "shared"
    b2 = bd.clone()
    b2[1, 0, 0] += 1
    assert shared_b(b2, G) is None


def test_abx_full_size_c2_properties():
    """BASELINE config 2 shape (H=32, R=128, L=65536): size-independent properties.
    (1) linearity in a; (2) tile independence: scores of rows [s, e) computed on the slice with
    pos_offset equal the full result; (3) oracle check on a strided sample of heads' groups."""
    torch.manual_seed(0)
    H, G, R, L = 32, 8, 128, 65536
    dev = "cuda"
    a1 = torch.randn(H, 1, 128, dtype=torch.float16, device=dev)
    a2 = torch.randn(H, 1, 128, dtype=torch.float16, device=dev)
    b = (torch.randn(H, R, 128, device=dev) / R ** 0.5).half()
    x = torch.randn(G, L, R, dtype=torch.float16, device=dev)
    abx = _abx()
    o1, o2, o12 = abx(a1, b, x), abx(a2, b, x), abx((a1.float() * 0.5 + a2.float() * 0.25).half(), b, x)
    lin = o1.float() * 0.5 + o2.float() * 0.25
    assert (lin - o12.float()).abs().max().item() <= 4e-3 * lin.abs().max().item()
    s, e = 40000, 41000
    sl = abx(a1, b, x[:, s:e].contiguous(), pos_offset=s)
    assert (sl.float() - o1[:, :, s:e].float()).abs().max().item() <= 2e-3 * o1.float().abs().max().item()
    # oracle on one group (4 heads), all 65536 positions
    g = 5
    ref = oracle.abx_scores(a1[4 * g:4 * g + 4].cpu(), b[4 * g:4 * g + 4].cpu(), x[g:g + 1].cpu())
    exact = oracle.abx_scores_f64(a1[4 * g:4 * g + 4].cpu(), b[4 * g:4 * g + 4].cpu(), x[g:g + 1].cpu())
    got = o1[4 * g:4 * g + 4].cpu()
    scale = exact.abs().max().item()
    assert (got.double() - ref.double()).abs().max().item() / scale <= 1e-3
    assert (got.double() - exact).abs().max().item() <= 1.5 * (ref.double() - exact).abs().max().item() + 1e-3 * scale / 4


def test_abx_errors():
    abx = _abx()
    a = torch.zeros(32, 1, 128, dtype=torch.float16, device="cuda")
    b = torch.zeros(32, 128, 128, dtype=torch.float16, device="cuda")
    x = torch.zeros(8, 64, 128, dtype=torch.float16, device="cuda")
    with pytest.raises(AssertionError):
        abx(a[:, 0], b, x)                                   # non-3-D (abx_rope.py:116-118)
    with pytest.raises(TypeError):
        abx(a.float(), b.float(), x.float())
    with pytest.raises(ValueError):
        abx(a, b, x[:, :, :64])
    with pytest.raises(RuntimeError):
        abx(a.cpu(), b.cpu(), x.cpu())
    out = abx(a, b, x[:, :0])
    assert out.shape == (32, 1, 0)


def test_abx_full_size_c5_tail_window():
    """BASELINE config 5 per-GPU slice (one group of 4 heads, R=128, L=262144): the last 8192 positions -- where the
    fp32 rounding of l*inv_freq is largest (2^-7 rad on the highest frequency) -- against the oracle's rounding
    points and against exact fp64 at the oracle's fp32 angles; plus tile independence at that offset."""
    torch.manual_seed(3)
    H, G, R, L = 4, 1, 128, 262144
    dev = "cuda"
    a = torch.randn(H, 1, 128, dtype=torch.float16, device=dev)
    b = (torch.randn(H, R, 128, device=dev) / R ** 0.5).half()
    x = torch.randn(G, L, R, dtype=torch.float16, device=dev)
    abx = _abx()
    full = abx(a, b, x)
    l0 = L - 8192
    win = abx(a, b, x[:, l0:].contiguous(), pos_offset=l0)
    assert (win.float() - full[:, :, l0:].float()).abs().max().item() <= 2e-3 * full.float().abs().max().item()
    ac, bc, xw = a.cpu(), b.cpu(), x[:, l0:].cpu()
    cos, sin = oracle.rope_cos_sin(L, 128, start=l0)
    keys = torch.matmul(xw[:, None], bc.reshape(G, H // G, R, 128)).reshape(H, L - l0, 128)
    ref = torch.matmul(ac, oracle.rope_rotate(keys, cos, sin).to(torch.float16).transpose(-1, -2))
    keys64 = torch.matmul(xw.double()[:, None], bc.double().reshape(G, H // G, R, 128)).reshape(H, L - l0, 128)
    ang = torch.outer(torch.arange(l0, L, dtype=torch.int64).to(torch.float32), oracle.rope_inv_freq(128)).double()
    ang = torch.cat((ang, ang), dim=-1)
    exact = torch.matmul(ac.double(), oracle.rope_rotate(keys64, ang.cos(), ang.sin()).transpose(-1, -2))
    got = full[:, :, l0:].cpu()
    scale = exact.abs().max().item()
    assert (got.double() - ref.double()).abs().max().item() / scale <= 1e-3
    assert (got.double() - exact).abs().max().item() <= 1.5 * (ref.double() - exact).abs().max().item() + 1e-3 * scale / 4


def test_abx_random_shapes_and_strides():
    """Seeded sweep over group sizes, ranks (fast and chunked paths), ragged lengths, position offsets and padded /
    interleaved latent layouts (row stride > R, groups interleaved in memory) against the oracle (P2)."""
    rng = np.random.default_rng(20260927)
    abx = _abx()
    for it in range(28):
        gs = int(rng.choice([1, 2, 3, 4, 8]))
        G = int(rng.choice([1, 2, 4]))
        H = G * gs
        R = int(rng.choice([32, 64, 128, 128, 96, 160, 256]))
        L = int(rng.integers(1, 2600))
        a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16))
        b = torch.from_numpy((rng.standard_normal((H, R, 128)) / np.sqrt(R)).astype(np.float16))
        x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16))
        layout = it % 3
        xd = x.cuda()
        if layout == 1:                                   # padded rows: stride(1) = R + 8
            buf = torch.zeros(G, L, R + 8, dtype=torch.float16, device="cuda")
            buf[:, :, :R] = xd
            xd = buf[:, :, :R]
        elif layout == 2:                                 # [L, G, R] storage viewed as [G, L, R]
            xd = xd.transpose(0, 1).contiguous().transpose(0, 1)
        off = int(rng.choice([0, 0, 1, 4097])) if R in (32, 64, 128) else 0
        got = abx(a.cuda(), b.cuda(), xd, pos_offset=off)
        if off == 0:
            _check(got, a, b, x)
        else:                                             # same rows placed at absolute positions off..off+L-1
            xz = torch.cat((torch.zeros(G, off, R, dtype=torch.float16), x), dim=1)
            ref = oracle.abx_scores(a, b, xz)[:, :, off:]
            exact = oracle.abx_scores_f64(a, b, xz)[:, :, off:]
            scale = exact.abs().max().item()
            assert (got.cpu().double() - ref.double()).abs().max().item() / scale <= 1e-3, (it, gs, G, R, L, off)


_COLD_START = r"""
import numpy as np, torch, sys
from palu_amd.kernel.abx_rope import abx
H, gs, R, L = 32, 4, 128, int(sys.argv[1])
G = H // gs
rng = np.random.default_rng(H + R + L)
a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16)).cuda()
bg = torch.from_numpy((rng.standard_normal((G, 1, R, 128)) / np.sqrt(R)).astype(np.float16))
b = bg.expand(G, gs, R, 128).reshape(H, R, 128).contiguous().cuda()
x = torch.from_numpy(rng.standard_normal((G, L, R)).astype(np.float16)).cuda()
first = abx(a, b, x); torch.cuda.synchronize()
second = abx(a, b, x); torch.cuda.synchronize()
sys.exit(0 if torch.equal(first, second) else 3)
"""


def test_abx_shared_b_cold_start():
    """The first launch of the shared-B kernel in a fresh process must equal every later one.  Regression test of
    profiles/r03_shared_b_cold_start.txt: with the dynamic wave priorities the kernel used to run with, 20-30 % of cold
    first launches had 16 wrong rows in one workgroup (98 of 98 fresh processes are right without them)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("PALU_ABX_PRIO_MODE", None)
    bad = []
    for batch in range(4):                 # 16 fresh processes, four at a time (4 draws miss a 25 % flake one time in three)
        procs = []
        for i in range(4):
            L = 65537 if i % 2 == 0 else 65536
            procs.append((L, subprocess.Popen([sys.executable, "-c", _COLD_START, str(L)], env=env, cwd=root)))
        for L, pr in procs:
            rc = pr.wait(timeout=600)
            assert rc in (0, 3), rc
            if rc == 3:
                bad.append((batch, L))
    assert not bad, bad
