"""Stress of the staging waits of the score-kernel family in the regime VERDICT r5 named (item 2b): caches of 8k .. 48k
positions in HOT loops -- most of X resident in L2 / MALL, part of it turned over between launches by copies of 64 .. 256 MB,
so that cache hits can overtake misses -- >= 2000 launches, every row checked.

Round 5's position-split kernel waited for its LDS-DMA requests with a count (`s_waitcnt vmcnt(N)`: requests retire in issue
order) and returned stale rows when that failed for clamped re-reads on cold launches.  Round 6 waits for every request in
every block (the count cost nothing: profiles/r06_abx_wait0.txt); this test is the guard the VERDICT asked for, for that
kernel (both fold forms) and for the fused attention core (csrc/decode_fused_kernel.h), whose steady state keeps a static
count over in-range requests to distinct addresses.  Reference: an fp64 evaluation of `torch_abx` (kernel/abx_rope.py:152-171)
/ of the decode branch's attention core (kernel/palu_attention.py:219-251) on the GPU; every launch is compared bit for bit
with a launch that was checked against it."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

D = 128


def _mods():
    from palu_amd import _lib, ops
    from palu_amd.kernel import abx_rope
    return _lib, abx_rope, ops


def _scores_f64(a, b, x, inv):
    H, R, _ = b.shape
    G, L, _ = x.shape
    out = torch.empty(H, L, dtype=torch.float64, device=x.device)
    bd, ad = b.double().reshape(G, H // G, R, D), a.double().reshape(G, H // G, D)
    for l0 in range(0, L, 8192):
        l1 = min(L, l0 + 8192)
        keys = torch.matmul(x[:, None, l0:l1].double(), bd)
        ang = torch.outer(torch.arange(l0, l1, device=x.device).float(), inv).double()
        c, s = ang.cos(), ang.sin()
        k1, k2 = keys[..., :64], keys[..., 64:]
        rot = torch.cat((k1 * c - k2 * s, k2 * c + k1 * s), -1)
        out[:, l0:l1] = torch.einsum("ghd,ghld->ghl", ad, rot).reshape(H, l1 - l0)
    return out


@pytest.mark.parametrize("fold", ["once_per_launch", "in_kernel"])
def test_position_split_kernel_in_hot_loops_with_partial_cache_turnover(fold):
    import contextlib
    _lib, ar, _ = _mods()
    dev = torch.device("cuda:0")
    inv = ar.rope_inv_freq(dev)
    junk = torch.empty(1 << 27, device=dev, dtype=torch.float16)          # 256 MB
    shapes = [(128, 8192), (128, 20000), (128, 49152), (64, 12345), (64, 32768), (64, 49152), (32, 8192), (32, 30001), (32, 49152)]
    launches = 0
    with contextlib.ExitStack() as st:
        st.enter_context(ar.position_split(0))
        if fold == "in_kernel":
            st.enter_context(ar.in_kernel_fold())
        for si, (R, L) in enumerate(shapes):
            g = torch.Generator().manual_seed(100 + si)
            sets = []
            for v in range(2):                                             # two input sets, alternating: stale rows of one show in the other
                a = torch.randn(32, 1, D, generator=g).half().to(dev)
                b = (torch.randn(32, R, D, generator=g) * R ** -0.5).half().to(dev)
                x = torch.randn(8, L, R, generator=g).half().to(dev)
                y = ar.abx(a, b, x).clone()
                ref = _scores_f64(a, b, x, inv)
                err = float((y.reshape(32, L).double() - ref).abs().max()) / float(ref.abs().max())
                assert err <= 2e-3, (R, L, v, err)
                sets.append((a, b, x, y))
            out = torch.empty_like(sets[0][3])
            bad = 0
            for it in range(120 if fold == "once_per_launch" else 60):
                a, b, x, y = sets[it & 1]
                if it % 3 == 0:                                            # 64 / 128 / 256 MB through the caches, every third launch
                    nb = (1 << 25) << (it // 3 % 3)
                    junk[:nb // 2].copy_(junk[nb // 2:nb])
                ar.abx(a, b, x, out=out)
                launches += 1
                bad += int(not torch.equal(out, y))
            assert bad == 0, (R, L, bad)
    assert launches >= 9 * 60


@pytest.mark.parametrize("L", [8192, 20001, 49152])
def test_fused_attention_core_in_hot_loops_with_partial_cache_turnover(L):
    _lib, ar, ops = _mods()
    lib = _lib.lib
    dev = torch.device("cuda:0")
    inv = ar.rope_inv_freq(dev)
    H, G, R, Rv = 4, 1, 128, 384                                          # one latent group per launch: the shape that selects the core
    assert lib.palu_decode_attn_preferred(H, G, L, R, Rv, D) == 1
    junk = torch.empty(1 << 27, device=dev, dtype=torch.float16)
    g = torch.Generator().manual_seed(L)
    ws = torch.empty(lib.palu_pv_workspace_bytes(H, G, L, Rv) + 4096, dtype=torch.uint8, device=dev)
    sets = []
    for v in range(2):
        q = torch.randn(H, D, generator=g).half().to(dev)
        b = (torch.randn(H, R, D, generator=g) * R ** -0.5).half().to(dev)
        k = torch.randn(G, L, R, generator=g).half().to(dev)
        vv = torch.randn(G, L, Rv, generator=g).half().to(dev)
        frag = ar.prepare_b(b, G)
        ctx = ops.decode_attn(q, frag, k, vv, inv, ws, H, L).clone()
        s = _scores_f64(q.view(H, 1, D), b, k, inv)
        s16 = (s.half() / math.sqrt(D)).float()                            # the reference's fp16 rounding points (:219, :238)
        p = torch.softmax(s16, dim=-1).half().double()
        ref = torch.matmul(p, vv[0].double())
        err = float((ctx.double() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 5e-3, (L, v, err)
        sets.append((q, b, frag, k, vv, ctx))
    bad = 0
    for it in range(240):
        q, b, frag, k, vv, ctx = sets[it & 1]
        if it % 3 == 0:
            nb = (1 << 25) << (it // 3 % 3)
            junk[:nb // 2].copy_(junk[nb // 2:nb])
        out = ops.decode_attn(q, frag, k, vv, inv, ws, H, L)
        bad += int(not torch.equal(out, ctx))
    assert bad == 0, (L, bad)
