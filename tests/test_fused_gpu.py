"""GPU parity of the single-kernel attention core of the decode step (palu_decode_attn_f16: abx scores -> /sqrt(D) ->
softmax -> latent P.V, csrc/decode_fused_kernel.h) against the CPU oracle (torch_abx + the decode branch of
kernel/palu_attention.py:219-251) and against the two-kernel HIP path, criterion P1: rtol = atol = 1e-3."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle

DEV = "cuda"
D = 128


def _lib():
    from palu_amd import _lib
    return _lib


def fused_attn(q, b, k, v, L, pos0=0):
    """q [H,D], b [H,Rk,D], k [G,cap,ldk] (first Rk columns), v [G,cap,ldv]: all cuda fp16 -> ctx [H,Rv], (max,sum) [H,2]"""
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq
    lib = _lib()
    H, Rk, _ = b.shape
    G = k.shape[0]
    Rv = v.shape[2]
    frag = prepare_b(b, G)
    inv = rope_inv_freq(torch.device(DEV))
    nbytes = lib.lib.palu_pv_workspace_bytes(H, G, k.shape[1], Rv)
    ws = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=DEV)
    ws[nbytes:] = 0x5A                                            # canary behind the workspace
    ctx = torch.full((H, Rv), float("nan"), dtype=torch.float16, device=DEV)
    lib.check(lib.lib.palu_decode_attn_f16(q.data_ptr(), q.stride(0), q.stride(1), frag.data_ptr(), k.data_ptr(),
                                           k.stride(0), k.stride(1), v.data_ptr(), v.stride(0), v.stride(1),
                                           ctx.data_ptr(), ws.data_ptr(), H, G, L, Rk, Rv, D, inv.data_ptr(), pos0,
                                           math.sqrt(D), torch.cuda.current_stream().cuda_stream), "decode_attn")
    torch.cuda.synchronize()
    assert bool((ws[nbytes:] == 0x5A).all()), "fused kernel wrote past its workspace"
    off = lib.lib.palu_decode_attn_stats_offset(H, G, L, Rv)
    stats = ws[off:off + H * 8].view(torch.float32).reshape(H, 2).clone()
    return ctx, stats


def oracle_attn(q, b, k, v):
    """CPU oracle: scores (torch_abx) / sqrt(D) -> softmax fp32 -> fp16 -> latent P.V (fp16 tensors on the CPU)"""
    H = q.shape[0]
    G, L, _ = k.shape
    scores = oracle.abx_scores(q.reshape(H, 1, D), b, k) / math.sqrt(D)
    probs = torch.softmax(scores, dim=-1, dtype=torch.float32).to(torch.float16)
    ctx = torch.matmul(probs.reshape(G, H // G, L), v)
    return ctx.reshape(H, -1), scores.reshape(H, L)


@pytest.mark.parametrize("H,G,Rk,Rv,L", [
    (32, 8, 128, 384, 1), (32, 8, 128, 384, 63), (32, 8, 128, 384, 64), (32, 8, 128, 384, 65), (32, 8, 128, 384, 129),
    (32, 8, 128, 384, 2049), (4, 1, 128, 384, 1500), (8, 2, 128, 384, 4097), (32, 8, 64, 192, 2500),
    (32, 8, 128, 256, 777), (32, 8, 64, 128, 1000), (24, 8, 128, 384, 900), (32, 8, 128, 192, 333),
])
def test_fused_attn_vs_oracle(H, G, Rk, Rv, L):
    rng = np.random.default_rng(H * 7 + L + Rv)
    q = torch.from_numpy(rng.standard_normal((H, D)).astype(np.float16))
    b = torch.from_numpy((rng.standard_normal((H, Rk, D)) * Rk ** -0.5).astype(np.float16))
    k = torch.from_numpy(rng.standard_normal((G, L, Rk)).astype(np.float16))
    v = torch.from_numpy(rng.standard_normal((G, L, Rv)).astype(np.float16))
    # device caches with capacity > L whose unused rows hold NaN: they must never reach the result
    cap = L + 70
    kd = torch.full((G, cap, Rk), float("nan"), dtype=torch.float16, device=DEV)
    vd = torch.full((G, cap, Rv), float("nan"), dtype=torch.float16, device=DEV)
    kd[:, :L] = k.to(DEV)
    vd[:, :L] = v.to(DEV)
    ctx, stats = fused_attn(q.to(DEV), b.to(DEV), kd, vd, L)
    rctx, rscores = oracle_attn(q, b, k, v)
    torch.testing.assert_close(ctx.cpu(), rctx, rtol=1e-3, atol=1e-3)
    # the softmax statistics the split-L callers merge with: max and sum of exp over the fp16 logits
    x = rscores.float()
    torch.testing.assert_close(stats[:, 0].cpu(), x.max(dim=-1).values, rtol=2e-3, atol=2e-3)
    ref_sum = torch.exp(x.double() - stats[:, 0].cpu().double()[:, None]).sum(dim=-1)
    torch.testing.assert_close(stats[:, 1].cpu().double(), ref_sum, rtol=5e-3, atol=1e-3)


def two_kernel_attn(q, b, k, v, L, pos0=0):
    """the two-kernel HIP path (abx -> softmax.PV) on the same device tensors -> (ctx [H,Rv], raw fp16 scores [H,L])"""
    from palu_amd.kernel.abx_rope import one_band, prepare_b, rope_inv_freq
    lib = _lib()
    H, Rk, _ = b.shape
    G, Rv = k.shape[0], v.shape[2]
    frag = prepare_b(b, G)
    inv = rope_inv_freq(torch.device(DEV))
    scores = torch.empty(H, L + 8, dtype=torch.float16, device=DEV)
    ws = torch.empty(lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx2 = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    # the fused core carries the ONE-band score pipeline: its scores are compared bit for bit with that kernel's
    with one_band():
        lib.check(lib.lib.palu_abx_rope_f16(q.data_ptr(), q.stride(0), 1, frag.data_ptr(), k.data_ptr(), k.stride(0),
                                            k.stride(1), scores.data_ptr(), scores.stride(0), H, G, L, Rk, D,
                                            inv.data_ptr(), pos0, s), "abx")
    lib.check(lib.lib.palu_softmax_pv_f16(scores.data_ptr(), scores.stride(0), 0, v.data_ptr(), v.stride(0),
                                          v.stride(1), ctx2.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv,
                                          math.sqrt(D), s), "pv")
    return ctx2, scores[:, :L]


def fp64_pv_of_scores(scores16, v, L):
    """softmax(fp16(fp16 scores / sqrt(D))) . V evaluated in fp64 (the reference's rounding points on the logits)"""
    H = scores16.shape[0]
    G, _, Rv = v.shape
    xs = (scores16.float() / math.sqrt(D)).half().double()
    return torch.einsum("ghl,glr->ghr", torch.softmax(xs, -1).view(G, H // G, L), v[:, :L].double()).reshape(H, Rv)


def test_fused_attn_matches_two_kernel_path_long():
    """C2-sized ranks at L = 20000, strided cache rows, key positions offset by pos0 (the split-L use)."""
    torch.manual_seed(3)
    H, G, Rk, Rv, L, pos0 = 32, 8, 128, 384, 20000, 4096
    q = torch.randn(H, D, device=DEV).half()
    b = (torch.randn(H, Rk, D, device=DEV) * Rk ** -0.5).half()
    kbuf = torch.randn(G, L + 64, Rk + 8, device=DEV).half()
    vbuf = torch.randn(G, L + 64, Rv + 8, device=DEV).half()
    k, v = kbuf[:, :, :Rk], vbuf[:, :, :Rv]
    ctx, _ = fused_attn(q, b, k, v, L, pos0=pos0)
    ctx2, scores = two_kernel_attn(q, b, k, v, L, pos0=pos0)
    torch.testing.assert_close(ctx, ctx2, rtol=1e-3, atol=2e-4)
    ref = fp64_pv_of_scores(scores, v, L)
    assert (ctx.double() - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


def test_fused_attn_config5_slice_full_size():
    """BASELINE config 5, what ONE of the 8 GPUs runs: G = 1 (4 heads), Rk = 128, Rv = 384, L = 262144 -- one latent group
    per launch is where the decode step SELECTS this kernel (palu_decode_attn_preferred: up to ~192k positions since round 5,
    beyond that the two kernels win again; the kernel stays checked at the config's full length; VERDICT r2 parity hole (i)).
      (a) full-size context against the two-kernel path (whose kernels are oracle-checked at this size by
          test_abx_full_size_c5_tail_window / test_softmax_pv_config5_slice) and against fp64 on its fp16 scores;
      (b) the LAST 8192 positions -- where the fp32 rounding of l * inv_freq is largest (2^-7 rad on the highest
          frequency) -- as a launch of their own at pos0 = L - 8192 against the CPU oracle's rounding points
          (oracle RoPE table at those positions, fp16 keys, softmax fp32 -> fp16, latent P.V);
      (c) split-merge: the LSE merge of [0, L - 8192) and the tail window (each its own launch, statistics from the
          workspace) reproduces the full launch -- the kernel's RoPE state / online softmax do not depend on where a
          range starts."""
    torch.manual_seed(55)
    H, G, Rk, Rv, L, W = 4, 1, 128, 384, 262144, 8192
    lib = _lib()
    assert lib.lib.palu_decode_attn_supported(H, G, Rk, Rv, D) == 1 and lib.lib.palu_decode_attn_preferred(H, G, 131072, Rk, Rv, D) == 1
    q = torch.randn(H, D, device=DEV).half()
    b = (torch.randn(H, Rk, D, device=DEV) * Rk ** -0.5).half()
    k = torch.randn(G, L + 64, Rk, device=DEV).half()
    v = torch.randn(G, L + 64, Rv, device=DEV).half()
    k[:, L:] = float("nan")
    v[:, L:] = float("nan")
    # (a)
    ctx, stats = fused_attn(q, b, k, v, L)
    ctx2, scores = two_kernel_attn(q, b, k, v, L)
    torch.testing.assert_close(ctx, ctx2, rtol=1e-3, atol=2e-4)
    ref64 = fp64_pv_of_scores(scores, v, L)
    assert (ctx.double() - ref64).abs().max().item() <= 1e-3 * max(1.0, ref64.abs().max().item())
    # (b) tail window vs the CPU oracle at the oracle's fp32 angles
    l0 = L - W
    kw, vw = k[:, l0:L].contiguous(), v[:, l0:L].contiguous()
    ctx_w, stats_w = fused_attn(q, b, kw, vw, W, pos0=l0)
    qc, bc, kc, vc = q.cpu(), b.cpu(), kw.cpu(), vw.cpu()
    cos, sin = oracle.rope_cos_sin(L, D, start=l0)
    keys = torch.matmul(kc[:, None], bc.reshape(G, H // G, Rk, D)).reshape(H, W, D)
    sc = torch.matmul(qc.reshape(H, 1, D), oracle.rope_rotate(keys, cos, sin).to(torch.float16).transpose(-1, -2))
    sc = sc / math.sqrt(D)
    pr = torch.softmax(sc, dim=-1, dtype=torch.float32).to(torch.float16)
    ref_w = torch.matmul(pr.reshape(G, H // G, W), vc).reshape(H, Rv)
    torch.testing.assert_close(ctx_w.cpu(), ref_w, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(stats_w[:, 0].cpu(), sc.reshape(H, W).float().max(dim=-1).values, rtol=2e-3, atol=2e-3)
    # (c) merge of head range + tail window == full launch
    ctx_h, stats_h = fused_attn(q, b, k, v, l0)
    m = torch.stack((stats_h[:, 0], stats_w[:, 0])).double()
    s_ = torch.stack((stats_h[:, 1], stats_w[:, 1])).double()
    wgt = s_ * torch.exp(m - m.max(dim=0, keepdim=True).values)
    merged = (wgt[0, :, None] * ctx_h.double() + wgt[1, :, None] * ctx_w.double()) / wgt.sum(0)[:, None]
    torch.testing.assert_close(ctx.double(), merged, rtol=2e-3, atol=3e-4)
    torch.testing.assert_close(stats[:, 0], torch.maximum(stats_h[:, 0], stats_w[:, 0]), rtol=0, atol=0)


def test_fused_attn_peaked_softmax_rescales():
    """Scores with a large spread whose maximum comes late in every range: the running-maximum rescale of the
    accumulators and of the partial sums is exercised in every workgroup.  With |score| ~ 100 the oracle's own fp16
    score error is visible in a peaked softmax (criterion P2 regime), so the reference here is the
    fp64 evaluation on the oracle-validated abx kernel's fp16 scores, which the fused kernel reproduces exactly."""
    rng = np.random.default_rng(11)
    H, G, Rk, Rv, L = 32, 8, 128, 384, 3000
    q = torch.from_numpy((rng.standard_normal((H, D)) * 4).astype(np.float16)).to(DEV)
    b = torch.from_numpy((rng.standard_normal((H, Rk, D)) * Rk ** -0.5).astype(np.float16)).to(DEV)
    k = rng.standard_normal((G, L, Rk)) * np.linspace(0.2, 2.0, L)[None, :, None]     # growing key norm
    k = torch.from_numpy(k.astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((G, L, Rv)).astype(np.float16)).to(DEV)
    ctx, stats = fused_attn(q, b, k, v, L)
    ctx2, scores = two_kernel_attn(q, b, k, v, L)
    ref = fp64_pv_of_scores(scores, v, L)
    assert (scores.float().abs().max() > 40).item()
    torch.testing.assert_close(ctx.double(), ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(ctx, ctx2, rtol=2e-3, atol=2e-3)
    xs = (scores.float() / math.sqrt(D)).half().float()
    torch.testing.assert_close(stats[:, 0], xs.max(dim=-1).values, rtol=0, atol=0)


@pytest.mark.parametrize("H,G,Rk,Rv,L", [(32, 8, 128, 384, 1500), (4, 1, 128, 384, 9001), (32, 8, 64, 192, 333)])
def test_fused_attn_with_additive_mask(H, G, Rk, Rv, L):
    """VERDICT r2 missing #5: the single-kernel core takes the additive attention mask of kernel/palu_attention.py:229-234
    (a left-padded prompt: the first columns at finfo.min, plus a few finite biases) -- against the CPU oracle with the
    same mask and against the two-kernel path's masked softmax."""
    import palu_amd.ops  # noqa: F401
    from palu_amd.kernel.abx_rope import prepare_b, rope_inv_freq
    rng = np.random.default_rng(L + Rv)
    q = torch.from_numpy(rng.standard_normal((H, D)).astype(np.float16))
    b = torch.from_numpy((rng.standard_normal((H, Rk, D)) * Rk ** -0.5).astype(np.float16))
    k = torch.from_numpy(rng.standard_normal((G, L, Rk)).astype(np.float16))
    v = torch.from_numpy(rng.standard_normal((G, L, Rv)).astype(np.float16))
    mask = torch.zeros(L, dtype=torch.float16)
    mask[:L // 5] = torch.finfo(torch.float16).min
    mask[L // 2:L // 2 + 7] = torch.tensor([-1.5, 0.25, -3.0, 2.0, -0.5, 1.0, -8.0], dtype=torch.float16)
    lib = _lib()
    ws = torch.empty(lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    kd, vd = k.to(DEV), v.to(DEV)
    ctx = torch.ops.palu.decode_attn(q.to(DEV), prepare_b(b.to(DEV), G), kd, vd, rope_inv_freq(torch.device(DEV)), ws, H, L, 0,
                                     mask.to(DEV))
    sc = oracle.abx_scores(q.reshape(H, 1, D), b, k) / math.sqrt(D)
    sc = sc + mask.reshape(1, 1, L)
    pr = torch.softmax(sc, dim=-1, dtype=torch.float32).to(torch.float16)
    ref = torch.matmul(pr.reshape(G, H // G, L), v).reshape(H, Rv)
    torch.testing.assert_close(ctx.cpu(), ref, rtol=1e-3, atol=1e-3)
    assert float(pr.reshape(H, L)[:, :L // 5].abs().max()) == 0.0            # the padded columns carry no weight
    # unmasked call on the same inputs differs (the mask did something) and still matches its own oracle
    ctx0 = torch.ops.palu.decode_attn(q.to(DEV), prepare_b(b.to(DEV), G), kd, vd, rope_inv_freq(torch.device(DEV)), ws, H, L)
    assert (ctx0.float() - ctx.float()).abs().max().item() > 1e-3


def test_fused_attn_rejects_uncovered_shapes():
    lib = _lib()
    q = torch.zeros(32, D, dtype=torch.float16, device=DEV)
    rc = lib.lib.palu_decode_attn_f16(q.data_ptr(), D, 1, q.data_ptr(), q.data_ptr(), 64, 32, q.data_ptr(), 64, 96,
                                      q.data_ptr(), q.data_ptr(), 32, 8, 2, 32, 96, D, q.data_ptr(), 0, 11.3, 0)
    assert rc == -2 and b"not covered" in lib.lib.palu_last_error()


@pytest.mark.parametrize("fused", [0, 1])
def test_decode_step_same_result_with_and_without_fused_core(fused, monkeypatch):
    """The whole step through palu_decode_step_f16 with the attention core forced to either implementation: both
    must agree with the oracle step (P1)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import math, sys, torch, numpy as np
sys.path.insert(0, %r)
import oracle
from palu_amd import _lib
from palu_amd.kernel import head_parallel as hp
torch.manual_seed(5)
H, G, D, HID, Rk, Rv, L = 32, 8, 128, 512, 128, 384, 1500
w = {"wq": (torch.randn(H * D, HID) / 16).half(), "vt_k": (torch.randn(G * Rk, HID) / 16).half(),
     "vt_v": (torch.randn(G * Rv, HID) / 16).half(), "b": (torch.randn(H, Rk, D) * Rk ** -0.5).half(),
     "wo": (torch.randn(HID, H * Rv) * 0.02).half()}
k = torch.randn(G, L, Rk).half(); v = torch.randn(G, L, Rv).half(); tok = torch.randn(HID).half()
ref, _, _, _ = oracle.decode_step(tok, L, w, k, v)
plan = hp.make_plan(1, 0, H, G, D, Rk, Rv)
wd = {n: t.cuda() for n, t in hp.shard_weights(plan, w).items()}
kc = torch.zeros(G, L + 64, Rk, dtype=torch.float16, device="cuda"); kc[:, :L] = k.cuda()
vc = torch.zeros(G, L + 64, Rv, dtype=torch.float16, device="cuda"); vc[:, :L] = v.cuda()
dec = hp.HeadParallelDecoder(plan, wd, kc, vc, HID)
out = dec.step(tok.cuda(), L, L)
torch.cuda.synchronize()
err = (out.cpu().float() - ref.float()).abs().max().item()
print("preferred", _lib.lib.palu_decode_attn_preferred(H, G, L + 1, Rk, Rv, D), "err", err)
torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-3)
''' % root
    env = dict(os.environ, PALU_FUSED_ATTN=str(fused))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"preferred {fused}" in r.stdout
