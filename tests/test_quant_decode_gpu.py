"""GPU: quantised-latent decode path (BASELINE configs 3/4 shapes at reduced L) against the oracle's
fake-quant semantics (project -> quantize_tensor -> attend: svd_linear.py:84-90,124-139), and the
offline Hadamard fusion (svd_linear.py:156-168)."""
import math

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

import oracle
from tests.golden import inputs as gi
from tests.test_decode_gpu import _module_from_palu_weights, softmax_pv

DEV = "cuda"


@pytest.mark.parametrize("bits,R,gs,H,L", [(4, 128, 4, 32, 1000), (3, 128, 4, 32, 777), (4, 64, 4, 32, 2049),
                                           (4, 32, 4, 32, 130), (3, 128, 2, 8, 300), (4, 64, 1, 4, 129),
                                           # the chunked quantised kernel: ranks the Fisher rank search emits
                                           # (palu/rank_search.py:11-17, multiples of 32 per group) and 3 bit at 32 / 64
                                           (3, 96, 4, 32, 1500), (4, 160, 4, 32, 700), (3, 224, 4, 16, 333), (4, 256, 4, 32, 515),
                                           (3, 64, 4, 32, 900), (3, 32, 2, 8, 260), (4, 96, 1, 4, 129), (3, 160, 8, 16, 410),
                                           (4, 40, 4, 8, 300)])
def test_abx_q_equals_fp16_kernel_on_dequantised_latents(bits, R, gs, H, L):
    """Same MFMA pipeline, same fp16 operand values -> bit-identical scores; plus the oracle bound."""
    from palu_amd import _lib
    from palu_amd.kernel import quant as q
    from palu_amd.kernel.abx_rope import abx, prepare_b, rope_inv_freq
    rng = np.random.default_rng(bits * 100 + R + L)
    G = H // gs
    a = torch.from_numpy(rng.standard_normal((H, 1, 128)).astype(np.float16)).to(DEV)
    b = torch.from_numpy((rng.standard_normal((H, R, 128)) / math.sqrt(R)).astype(np.float16)).to(DEV)
    x = torch.from_numpy((rng.standard_normal((G, L, R)) * rng.uniform(0.2, 3, (G, L, 1))).astype(np.float16)).to(DEV)
    codes, meta, deq = q.quantize_pack(x, bits, want_dequant=True)
    ref = abx(a, b, deq)
    out = torch.empty(H, 1, L, dtype=torch.float16, device=DEV)
    frag = prepare_b(b, G)
    inv = rope_inv_freq(x.device)
    # ranks above 128 run as passes of the 128-column kernel when an fp32 scratch is handed over (palu_abx_rope_qg)
    nscr = _lib.lib.palu_abx_scratch_bytes(H, G, L, R)
    scratch = torch.empty(max(nscr, 16), dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib.palu_abx_rope_qg(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                         codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                         out.data_ptr(), out.stride(0), H, G, L, R, 128, bits, 0, inv.data_ptr(), 0,
                                         scratch.data_ptr() if nscr else 0, _lib.current_stream()), "abx_qg")
    # The quantised kernel stages the SAME fp16 values the fp16 kernel reads; where both run the same pipeline -- the
    # 128-column kernel with q folded into B, one or several column windows -- the scores are bit-identical.  Only 3 bit
    # at R = 32 / 64 differs (no fast quantised kernel: quarter rows are not whole dwords; the codes take the chunked
    # kernel with q kept in fp32): equal to rounding, both within the oracle bound below.
    if not (bits == 3 and R in (32, 64)):
        assert torch.equal(out, ref)
    else:
        assert (out.float() - ref.float()).abs().max().item() <= 2e-3 * ref.float().abs().max().item()
    # without scratch a rank above 128 falls back to the chunked kernel: same scores to rounding
    if nscr:
        out2 = torch.empty_like(out)
        _lib.check(_lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                            codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                            out2.data_ptr(), out2.stride(0), H, G, L, R, 128, bits, inv.data_ptr(), 0,
                                            _lib.current_stream()), "abx_q")
        assert (out2.float() - ref.float()).abs().max().item() <= 2e-3 * ref.float().abs().max().item()
    o = oracle.abx_scores(a.cpu(), b.cpu(), oracle.quantize_rows(x.cpu().reshape(-1, R), bits)[0].reshape(G, L, R))
    scale = o.float().abs().max().item()
    assert (out.cpu().float() - o.float()).abs().max().item() <= 1e-3 * scale


@pytest.mark.parametrize("bits,Rv,gs,H,L", [(4, 192, 4, 32, 3001), (3, 384, 4, 32, 1500), (3, 96, 2, 8, 260),
                                            (4, 64, 1, 4, 129), (4, 384, 4, 32, 9000),
                                            # group sizes of GQA models (kv group size x n_rep): VALU kernel only
                                            (3, 160, 3, 12, 1100), (4, 224, 8, 32, 2500), (3, 96, 8, 16, 700)])
def test_softmax_pv_q(bits, Rv, gs, H, L):
    from palu_amd import _lib
    from palu_amd.kernel import quant as q
    rng = np.random.default_rng(bits + Rv + L)
    G = H // gs
    scores = torch.from_numpy((rng.standard_normal((H, L)) * 20).astype(np.float16)).to(DEV)
    v = torch.from_numpy((rng.standard_normal((G, L, Rv)) * rng.uniform(0.2, 3, (G, L, 1))).astype(np.float16)).to(DEV)
    codes, meta, deq = q.quantize_pack(v, bits, want_dequant=True)
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    probs = torch.empty(H, L, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, codes.data_ptr(), codes.stride(0),
                                          codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1), ctx.data_ptr(),
                                          probs.data_ptr(), probs.stride(0), ws.data_ptr(), H, G, L, Rv, bits,
                                          math.sqrt(128.0), _lib.current_stream()), "pv_q")
    ref_ctx, ref_p = softmax_pv(scores, deq, want_probs=True)                      # fp16 HIP kernel on dequantised V
    torch.testing.assert_close(ctx, ref_ctx, rtol=1e-3, atol=1e-3)
    # same logits, same global maximum; the normaliser is summed over a different split structure (the matrix-core
    # kernel uses larger ranges), so a weight may land on the other side of an fp16 rounding boundary: one ulp
    torch.testing.assert_close(probs, ref_p, rtol=1e-3, atol=1e-7)
    x = (scores.cpu() / math.sqrt(128.0))
    p64 = torch.softmax(x.double(), dim=-1)
    c64 = torch.matmul(p64.reshape(G, gs, L), deq.cpu().double()).reshape(H, Rv)
    assert (ctx.cpu().double() - c64).abs().max().item() <= 1e-3 * max(1.0, c64.abs().max().item())


@pytest.mark.parametrize("bits,rank_k,rank_v,L", [(4, 512, 1536, 1500), (3, 1024, 3072, 700), (4, 512, 1536, 131072),
                                                  (3, 768, 1280, 900), (4, 1280, 1792, 600), (3, 2048, 2560, 400)],
                         ids=["config4_short", "config3_short", "config4_full_size", "ranks_96_160_3bit",
                              "ranks_160_224_4bit", "ranks_256_320_3bit"])
def test_quantised_decode_step_vs_oracle(bits, rank_k, rank_v, L):
    """BASELINE config 3 (3-bit, 1024/3072) and 4 (4-bit, 512/1536) shapes: module forward on a
    QuantLatentCache == oracle.decode_step on fake-quantised caches with the new rows fake-quantised.  Reduced L, and
    config 4 at its FULL size (131072 cached positions, ~15 s of CPU for the oracle): VERDICT r2 parity hole (ii)."""
    from palu_amd.kernel.palu_attention import QuantLatentCache
    hidden, H, D, gs = 4096, 32, 128, 4
    G = H // gs
    w, k_lat, v_lat, tok, _ = gi.step_inputs(77 + bits, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = QuantLatentCache(bits)
    kd, vd = cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
    Rk, Rv = rank_k // G, rank_v // G
    kq = oracle.quantize_rows(k_lat.reshape(-1, Rk), bits)[0].reshape(G, L, Rk)
    vq = oracle.quantize_rows(v_lat.reshape(-1, Rv), bits)[0].reshape(G, L, Rv)
    assert torch.equal(kd[0].cpu(), kq) and torch.equal(vd[0].cpu(), vq)              # bit-exact fake-quant
    with torch.no_grad():
        out, probs, _ = m(tok.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L, L + 1),
                          past_key_value=cache, output_attentions=True)
    assert cache.get_seq_length(0) == L + 1
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    o2, p2, k2, v2 = oracle.decode_step(tok, L, wd, kq, vq, latent_bits=bits)
    torch.testing.assert_close(probs.cpu().reshape(H, L + 1), p2, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu().reshape(-1), o2, rtol=1e-3, atol=1e-3)
    kd2, vd2 = cache.dequantized(0)
    # the appended row: GEMV rounding may flip a code at a tie; values must agree to one quantisation step
    step_k = (k2[:, L].float().amax(-1) - k2[:, L].float().amin(-1)) / (2 ** bits - 1)
    assert ((kd2[0, :, L].cpu().float() - k2[:, L].float()).abs().amax(-1) <= step_k * 1.01 + 1e-3).all()


@pytest.mark.parametrize("L", [600, 65536], ids=["short", "config3_full_size"])
def test_config3_three_bit_with_hadamard_vs_oracle(L):
    """BASELINE config 3 as a whole: rank 1024/3072 (Rk = 128 = 2^7, Rv = 384 = 12 * 32 -> the had12 (x) H_32 branch),
    3-bit packed latents AND fuse_hadamard().  The module's rotated weights are handed to the oracle
    (oracle.decode_step(latent_bits=3) on the fake-quantised ROTATED latents): same probabilities and output (P1).
    L = 65536 is the configuration at FULL size (VERDICT r2 parity hole (ii); ~15 s of CPU for the oracle)."""
    from palu_amd.kernel.palu_attention import QuantLatentCache
    hidden, H, D, gs, rank_k, rank_v, bits = 4096, 32, 128, 4, 1024, 3072, 3
    G = H // gs
    Rk, Rv = rank_k // G, rank_v // G
    w, k_lat, v_lat, tok, _ = gi.step_inputs(31, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    m.fuse_hadamard()
    # the rotated latents of the prompt: rotate the un-rotated ones with the oracle's Hadamard (orthonormal, per group)
    k_rot = oracle.apply_hadamard(k_lat.float().reshape(-1, Rk)).reshape(G, L, Rk).half()
    v_rot = oracle.apply_hadamard(v_lat.float().reshape(-1, Rv)).reshape(G, L, Rv).half()
    cache = QuantLatentCache(bits)
    kd, vd = cache.update(k_rot.unsqueeze(0).to(DEV), v_rot.unsqueeze(0).to(DEV), 0)
    kq = oracle.quantize_rows(k_rot.reshape(-1, Rk), bits)[0].reshape(G, L, Rk)
    vq = oracle.quantize_rows(v_rot.reshape(-1, Rv), bits)[0].reshape(G, L, Rv)
    assert torch.equal(kd[0].cpu(), kq) and torch.equal(vd[0].cpu(), vq)
    with torch.no_grad():
        out, probs, _ = m(tok.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L, L + 1),
                          past_key_value=cache, output_attentions=True)
    wd = {"wq": m.q_proj.weight.detach().cpu().half(), "vt_k": m.k_proj.VT.weight.detach().cpu().half(),
          "vt_v": m.v_proj.VT.weight.detach().cpu().half(), "b": m.k_proj.B.detach().cpu().half(),
          "wo": m.o_proj.weight.detach().cpu().half()}
    o2, p2, _, _ = oracle.decode_step(tok, L, wd, kq, vq, latent_bits=bits)
    torch.testing.assert_close(probs.cpu().reshape(H, L + 1), p2, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu().reshape(-1), o2, rtol=1e-3, atol=1e-3)
    # and the rotation did what fused_hadamard_matrix promises: same step as the un-rotated module on fp16 latents
    m0 = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    b0 = oracle.build_b_from_u(w["u_k"], gs, D).half()
    b_rot = oracle.apply_hadamard(b0.float().transpose(1, 2).reshape(-1, Rk)).reshape(H, D, Rk).transpose(1, 2)
    torch.testing.assert_close(m.k_proj.B.detach().cpu().float(), b_rot, rtol=2e-3, atol=2e-3)
    assert m0.k_proj.B.shape == m.k_proj.B.shape


@pytest.mark.parametrize("bits,rank_k,rank_v,gsz,L", [(4, 1024, 3072, 32, 700), (3, 1024, 3072, 64, 500), (4, 512, 1536, 64, 1300),
                                                      (3, 768, 2304, 96, 400), (4, 1024, 3072, 128, 300)],
                         ids=["4bit_g32", "3bit_g64", "4bit_g64_c4ranks", "3bit_g96", "4bit_g128_k_whole_row"])
def test_quantised_decode_step_with_group_size_vs_oracle(bits, rank_k, rank_v, gsz, L):
    """VERDICT r2 missing #3: quantize_tensor's `group_size` (quant.py:11-13, --lt_group_size of utils.py:105) in the packed
    cache and the decode kernels: every `gsz` columns of a latent row carry their own (scale, zero).
    QuantLatentCache(bits, group_size=gsz) == oracle.quantize_rows(..., group_size=gsz) bit for bit, and the decode step
    (palu_decode_step_qg: grouped quantisation of the new rows, chunked score kernel with per-group metas, P.V per column
    group) == oracle.decode_step(latent_bits, latent_group_size) (P1)."""
    from palu_amd.kernel.palu_attention import QuantLatentCache
    hidden, H, D, gs = 4096, 32, 128, 4
    G = H // gs
    w, k_lat, v_lat, tok, _ = gi.step_inputs(91 + bits + gsz, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    cache = QuantLatentCache(bits, group_size=gsz)
    kd, vd = cache.update(k_lat.unsqueeze(0).to(DEV), v_lat.unsqueeze(0).to(DEV), 0)
    Rk, Rv = rank_k // G, rank_v // G
    kq = oracle.quantize_rows(k_lat.reshape(-1, Rk), bits, gsz)[0].reshape(G, L, Rk)
    vq = oracle.quantize_rows(v_lat.reshape(-1, Rv), bits, gsz)[0].reshape(G, L, Rv)
    assert torch.equal(kd[0].cpu(), kq) and torch.equal(vd[0].cpu(), vq)              # bit-exact grouped fake-quant
    st = cache.buffers(0)
    assert st["km"].shape[-1] == 2 * Rk // gsz and st["vm"].shape[-1] == 2 * Rv // gsz
    with torch.no_grad():
        out, probs, _ = m(tok.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L, L + 1),
                          past_key_value=cache, output_attentions=True)
    assert cache.get_seq_length(0) == L + 1
    wd = {"wq": w["wq"].half(), "vt_k": w["vt_k"].half(), "vt_v": w["vt_v"].half(),
          "b": oracle.build_b_from_u(w["u_k"], gs, D).half(), "wo": w["wo"].half()}
    o2, p2, k2, v2 = oracle.decode_step(tok, L, wd, kq, vq, latent_bits=bits, latent_group_size=gsz)
    torch.testing.assert_close(probs.cpu().reshape(H, L + 1), p2, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu().reshape(-1), o2, rtol=1e-3, atol=1e-3)
    # a second step through the dispatcher op (no attention weights): the appended rows are attended over as well
    tok2 = (tok.float() * 0.5 + 0.1).half()
    with torch.no_grad():
        out2, _, _ = m(tok2.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L + 1, L + 2), past_key_value=cache)
    kd2, vd2 = cache.dequantized(0)
    o3, _, _, _ = oracle.decode_step(tok2, L + 1, wd, kd2[0, :, :L + 1].cpu(), vd2[0, :, :L + 1].cpu(), latent_bits=bits,
                                     latent_group_size=gsz)
    torch.testing.assert_close(out2.cpu().reshape(-1), o3, rtol=1e-3, atol=1e-3)


def test_hadamard_fusion_is_output_invariant():
    """fuse_hadamard() rotates VT / U / B / W_o' offline (svd_linear.py:156-168): same decode output."""
    from palu_amd.kernel.palu_attention import LatentCache
    hidden, H, D, gs, rank_k, rank_v, L = 4096, 32, 128, 4, 1024, 3072, 300     # Rk=128 (2^7), Rv=384 (12*32)
    w, k_lat, v_lat, tok, _ = gi.step_inputs(5, hidden, H, D, gs, rank_k, rank_v, L, False)
    m = _module_from_palu_weights(hidden, H, D, gs, rank_k, rank_v, w)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((1, L, hidden)).astype(np.float16)).to(DEV)

    def run(mod):
        cache = LatentCache()
        with torch.no_grad():
            mod(x, past_key_value=cache, position_ids=torch.arange(L).unsqueeze(0))          # prefill fills the cache
            out, probs, _ = mod(tok.reshape(1, 1, hidden).to(DEV), position_ids=torch.arange(L, L + 1),
                                past_key_value=cache, output_attentions=True)
        return out, probs, cache
    o0, p0, c0 = run(m)
    m.fuse_hadamard()
    o1, p1, c1 = run(m)
    torch.testing.assert_close(p1, p0, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(o1, o0, rtol=2e-3, atol=2e-3)
    k0, k1 = c0.buffers(0)[0][0, :, :L].float(), c1.buffers(0)[0][0, :, :L].float()
    assert (k0 - k1).abs().max() > 0.05                                                       # latents really rotated
    torch.testing.assert_close(k0.norm(dim=-1), k1.norm(dim=-1), rtol=2e-2, atol=2e-2)        # by an orthogonal map


@pytest.mark.parametrize("bits,R,Rv,L", [(3, 128, 384, 65537), (4, 64, 192, 131073)], ids=["config3", "config4"])
def test_quantised_kernels_full_size(bits, R, Rv, L):
    """BASELINE configs 3 and 4 at full length: the quantised score kernel is BIT-IDENTICAL to the fp16 kernel on the
    de-quantised latents, the quantised P.V agrees with the fp16 P.V on the de-quantised values (size-independent
    properties; the oracle cannot hold these sizes in seconds)."""
    from palu_amd import _lib
    from palu_amd.kernel import quant as q
    from palu_amd.kernel.abx_rope import abx, pair_split, prepare_b, rope_inv_freq
    H, G = 32, 8
    g = torch.Generator(device=DEV).manual_seed(bits)
    a = torch.randn(H, 1, 128, dtype=torch.float16, device=DEV, generator=g)
    b = (torch.randn(H, R, 128, device=DEV, generator=g) / math.sqrt(R)).half()
    x = torch.randn(G, L, R, dtype=torch.float16, device=DEV, generator=g)
    codes, meta, deq = q.quantize_pack(x, bits, want_dequant=True)
    # (round 5: at this size fp16 latents take the position-split form of the two-band kernel, which sums a position's RoPE
    # pairs in another order; the packed kernels are the pair-split form: bit identity is with THAT form on the same rows,
    # and the default fp16 launch agrees with it to rounding)
    with pair_split():
        ref = abx(a, b, deq)
    dflt = abx(a, b, deq)
    assert (dflt.float() - ref.float()).abs().max().item() <= 1e-3 * ref.float().abs().max().item()
    out = torch.empty(H, 1, L, dtype=torch.float16, device=DEV)
    frag, inv = prepare_b(b, G), rope_inv_freq(x.device)
    _lib.check(_lib.lib.palu_abx_rope_q(a.data_ptr(), a.stride(0), a.stride(2), frag.data_ptr(), codes.data_ptr(),
                                        codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1),
                                        out.data_ptr(), out.stride(0), H, G, L, R, 128, bits, inv.data_ptr(), 0,
                                        _lib.current_stream()), "abx_q")
    assert torch.equal(out, ref)
    del x, deq, codes, meta
    v = torch.randn(G, L, Rv, dtype=torch.float16, device=DEV, generator=g)
    vc, vm, vdeq = q.quantize_pack(v, bits, want_dequant=True)
    scores = out[:, 0].contiguous()
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, vc.data_ptr(), vc.stride(0), vc.stride(1),
                                          vm.data_ptr(), vm.stride(0), vm.stride(1), ctx.data_ptr(), 0, 0, ws.data_ptr(),
                                          H, G, L, Rv, bits, math.sqrt(128.0), _lib.current_stream()), "pv_q")
    ref_ctx, _ = softmax_pv(scores, vdeq)
    torch.testing.assert_close(ctx, ref_ctx, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("bits,Rv,gs,H,L,masked", [
    (3, 64, 2, 4, 1, False), (4, 128, 4, 8, 33, False), (3, 512, 4, 4, 700, True), (4, 1024, 1, 2, 300, False),
    (3, 2048, 4, 4, 200, True), (4, 96, 2, 8, 5000, True), (3, 384, 4, 32, 4097, True), (4, 4096, 1, 1, 65, False)])
def test_softmax_pv_q_shapes_and_masks(bits, Rv, gs, H, L, masked):
    """The register-direct kernel's geometry cases: one / two row sets per unit, 1-8 column slices (R_v up to 4096), ranges
    shorter than a unit, an additive mask that blanks whole 64-row batches (a wave's running maximum stays -inf for a
    while) -- against the fp16 HIP kernel on the dequantised latents and the fp64 softmax."""
    from palu_amd import _lib
    from palu_amd.kernel import quant as q
    rng = np.random.default_rng(bits * 1000 + Rv + L)
    G = H // gs
    scores = torch.from_numpy((rng.standard_normal((H, L)) * 15).astype(np.float16)).to(DEV)
    v = torch.from_numpy((rng.standard_normal((G, L, Rv)) * rng.uniform(0.2, 3, (G, L, 1))).astype(np.float16)).to(DEV)
    codes, meta, deq = q.quantize_pack(v, bits, want_dequant=True)
    mask = None
    if masked:
        m = np.zeros(L, dtype=np.float16)
        m[: min(L // 2, 200)] = np.float16(-65504.0)                # whole leading batches masked out
        m[rng.random(L) < 0.2] = np.float16(-65504.0)
        m[L - 1] = 0
        mask = torch.from_numpy(m).to(DEV)
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0 if mask is None else mask.data_ptr(),
                                          codes.data_ptr(), codes.stride(0), codes.stride(1), meta.data_ptr(), meta.stride(0),
                                          meta.stride(1), ctx.data_ptr(), 0, 0, ws.data_ptr(), H, G, L, Rv, bits,
                                          math.sqrt(128.0), _lib.current_stream()), "pv_q")
    x = scores.cpu().float() / math.sqrt(128.0)
    x = x.half()
    if mask is not None:
        x = (x.float() + mask.cpu().float().reshape(1, L)).half()
    p64 = torch.softmax(x.double(), dim=-1)
    c64 = torch.matmul(p64.reshape(G, gs, L), deq.cpu().double()).reshape(H, Rv)
    assert torch.isfinite(ctx).all()
    assert (ctx.cpu().double() - c64).abs().max().item() <= 1.5e-3 * max(1.0, c64.abs().max().item())
    if gs in (1, 2, 4, 8) and Rv <= 2048:
        ref_ctx = softmax_pv(scores, deq, mask)[0]
        torch.testing.assert_close(ctx, ref_ctx, rtol=2e-3, atol=2e-3)


def test_softmax_pv_q_valu_fallback_kernel():
    """PALU_PVQ_DIRECT=0 selects the VALU kernel (pv_partial_q_kernel) that takes the shapes the register-direct kernel
    leaves (unaligned rows, > 2 GiB of codes per group, a divisor without an exact fast quotient).  The switch is read
    once per process: the quantised P.V cases are re-run in a child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PALU_PVQ_DIRECT="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_quant_decode_gpu.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_softmax_pv_q and not fallback and not shapes_and_masks"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


@pytest.mark.parametrize("bits,Rv,L,spikes", [(3, 384, 70000, (5, 40000, 69999)), (4, 192, 9000, (4500,)), (3, 128, 3000, (64, 2999))])
def test_softmax_pv_q_late_maximum_forces_rescale(bits, Rv, L, spikes):
    """The online statistics of the register-direct kernel rescale the accumulators when a later 64-row batch raises the
    running maximum -- a rare, data-dependent branch on ordinary inputs.  Here it is forced: flat logits with isolated
    spikes of increasing height deep inside wave ranges (different heads spike at different rows), checked against the
    fp64 softmax over the full tensor."""
    from palu_amd import _lib
    from palu_amd.kernel import quant as q
    rng = np.random.default_rng(bits + Rv + L)
    H, gs = 32, 4
    G = H // gs
    s = (rng.standard_normal((H, L)) * 0.5).astype(np.float32)
    for i, pos in enumerate(spikes):
        for h in range(H):
            p_h = min(L - 1, pos + 67 * h)                          # another batch / wave for every head
            s[h, p_h] += 40.0 * (i + 1) + h                         # each later spike beats the earlier ones
    scores = torch.from_numpy((s * math.sqrt(128.0)).astype(np.float16)).to(DEV)
    v = torch.from_numpy(rng.standard_normal((G, L, Rv)).astype(np.float16)).to(DEV)
    codes, meta, deq = q.quantize_pack(v, bits, want_dequant=True)
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, codes.data_ptr(), codes.stride(0),
                                          codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1), ctx.data_ptr(),
                                          0, 0, ws.data_ptr(), H, G, L, Rv, bits, math.sqrt(128.0), _lib.current_stream()),
               "pv_q")
    x = (scores.cpu().float() / math.sqrt(128.0)).half()
    p64 = torch.softmax(x.double(), dim=-1)
    c64 = torch.matmul(p64.reshape(G, gs, L), deq.cpu().double()).reshape(H, Rv)
    assert torch.isfinite(ctx).all()
    assert (ctx.cpu().double() - c64).abs().max().item() <= 1.5e-3 * max(1.0, c64.abs().max().item())


@pytest.mark.parametrize("Rv,gs,H,L", [(192, 4, 32, 131), (96, 2, 8, 4100), (288, 4, 8, 900), (96, 1, 4, 70), (576, 4, 8, 2500),
                                       (1536, 4, 4, 333)])
def test_softmax_pv_q4_in_24_code_chunks(Rv, gs, H, L):
    """4-bit rows whose 32-code chunks would leave MFMA lanes idle run in 12-byte
    chunks of 24 codes: against the fp64 softmax on the dequantised latents."""
    from palu_amd import _lib
    from palu_amd.kernel import quant as q
    rng = np.random.default_rng(Rv + L)
    G = H // gs
    assert _lib.lib.palu_pv_direct_nsplit(G, L, Rv, 4) > 0
    scores = torch.from_numpy((rng.standard_normal((H, L)) * 15).astype(np.float16)).to(DEV)
    v = torch.from_numpy((rng.standard_normal((G, L, Rv)) * rng.uniform(0.2, 3, (G, L, 1))).astype(np.float16)).to(DEV)
    codes, meta, deq = q.quantize_pack(v, 4, want_dequant=True)
    ws = torch.empty(_lib.lib.palu_pv_workspace_bytes(H, G, L, Rv), dtype=torch.uint8, device=DEV)
    ctx = torch.empty(H, Rv, dtype=torch.float16, device=DEV)
    _lib.check(_lib.lib.palu_softmax_pv_q(scores.data_ptr(), scores.stride(0), 0, codes.data_ptr(), codes.stride(0),
                                          codes.stride(1), meta.data_ptr(), meta.stride(0), meta.stride(1), ctx.data_ptr(),
                                          0, 0, ws.data_ptr(), H, G, L, Rv, 4, math.sqrt(128.0), _lib.current_stream()), "pv_q")
    x = (scores.cpu().float() / math.sqrt(128.0)).half()
    p64 = torch.softmax(x.double(), dim=-1)
    c64 = torch.matmul(p64.reshape(G, gs, L), deq.cpu().double()).reshape(H, Rv)
    assert torch.isfinite(ctx).all()
    assert (ctx.cpu().double() - c64).abs().max().item() <= 1.5e-3 * max(1.0, c64.abs().max().item())


@pytest.mark.parametrize("kind", ["pvq3", "pvq4"])
def test_quantised_pv_cold_start_is_deterministic(kind):
    """pv_partial_qr_kernel runs waves 4-7 at a raised (static) priority: its first launch in a FRESH process equals its second
    and third -- 8 processes per kind (the guard the score kernels have in test_two_band_gpu.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    outs = []
    for _ in range(2):
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "diag_cold_start.py"), kind], env=env, cwd=root,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(4)]
        outs += [p.communicate(timeout=600)[0] for p in procs]
    for o in outs:
        assert "first!=second: 0  second!=third: 0" in o, o
