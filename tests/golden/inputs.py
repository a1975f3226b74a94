"""Deterministic synthetic inputs shared by the golden-vector generator and the tests.

All randomness comes from ``numpy.random.default_rng(seed)`` (PCG64), never from torch's
global RNG, so the generator (build container, with /root/reference) and the tests (GPU box,
without it) construct byte-identical inputs; each fixture also stores a SHA-256 of the input
bytes so drift is detected instead of silently comparing different problems.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch

# (tag, seed, H, D, gs, R, L, regime)
ABX_CASES = [
    ("r32_l64_randn_s0", 0, 32, 128, 4, 32, 64, "randn"),
    ("r32_l2048_randn_s1", 1, 32, 128, 4, 32, 2048, "randn"),      # C1 shape
    ("r128_l64_randn_s2", 2, 32, 128, 4, 128, 64, "randn"),
    ("r128_l257_randn_s0", 0, 32, 128, 4, 128, 257, "randn"),      # ragged tail
    ("r512_l64_randn_s1", 1, 32, 128, 4, 512, 64, "randn"),        # reference-test rank
    ("r32_l2048_model_s0", 0, 32, 128, 4, 32, 2048, "model"),
    ("r128_l257_model_s1", 1, 32, 128, 4, 128, 257, "model"),
    ("r64_l130_gs2_randn_s3", 3, 8, 128, 2, 64, 130, "randn"),     # other group size
    ("r64_l96_gs1_model_s4", 4, 4, 128, 1, 64, 96, "model"),
]


def _f16(rng, shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float16))


def abx_inputs(seed, H, D, gs, R, L, regime):
    """(a [H,1,D], b [H,R,D], x [G,L,R]) fp16.  'randn' = abx_rope.py:200-202 statistics;
    'model' = unit-variance keys (b ~ N(0,1/R)) so scores/sqrt(D) are O(1) like a trained model."""
    rng = np.random.default_rng(1000 + seed)
    G = H // gs
    a = _f16(rng, (H, 1, D))
    b = _f16(rng, (H, R, D), 1.0 if regime == "randn" else 1.0 / np.sqrt(R))
    x = _f16(rng, (G, L, R))
    return a, b, x


def digest(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


# ---------------------------------------------------------------- decode-step cases
# (tag, seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask)
STEP_CASES = [
    ("small_gs2", 10, 512, 4, 128, 2, 64, 128, 96, False),
    ("small_gs2_mask", 11, 512, 4, 128, 2, 64, 128, 70, True),
    ("c1_h32", 12, 4096, 32, 128, 4, 256, 768, 2048, False),        # BASELINE configs[0]
]


def step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, L, with_mask):
    """Weights in Palu form + caches + token for one decode step (all fp16 but U in fp32).

    Linear weights are U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (nn.Linear default), the U
    reconstruction factors are N(0,1)/sqrt(R) so reconstructed keys are O(1); latent caches
    are randn like run_latency_attention.py:62-63; the token is randn (:70).
    """
    rng = np.random.default_rng(2000 + seed)
    G = H // gs
    Rk, Rv = rank_k // G, rank_v // G

    def uni(shape, fan_in):
        bound = 1.0 / np.sqrt(fan_in)
        return torch.from_numpy(rng.uniform(-bound, bound, shape).astype(np.float32))

    w = {
        "wq": uni((H * D, hidden), hidden),
        "vt_k": uni((rank_k, hidden), hidden),
        "vt_v": uni((rank_v, hidden), hidden),
        "u_k": [torch.from_numpy((rng.standard_normal((gs * D, Rk)) / np.sqrt(Rk)).astype(np.float32))
                for _ in range(G)],
        "wo": uni((hidden, H * Rv), H * Rv),
    }
    k_lat = _f16(rng, (G, L, Rk))
    v_lat = _f16(rng, (G, L, Rv))
    tok = _f16(rng, (hidden,))
    mask = None
    if with_mask:
        m = np.zeros((L + 1,), dtype=np.float16)
        m[rng.choice(L + 1, size=(L + 1) // 5, replace=False)] = np.float16(-65504.0)
        m[L] = 0
        mask = torch.from_numpy(m)
    return w, k_lat, v_lat, tok, mask


# ---------------------------------------------------------------- prefill cases
# (tag, seed, hidden, H, D, gs, rank_k, rank_v, T, causal_mask)
PREFILL_CASES = [
    ("pf_small_gs2_t48_causal", 20, 512, 4, 128, 2, 64, 128, 48, True),
    ("pf_small_gs2_t70_nomask", 21, 512, 4, 128, 2, 64, 128, 70, False),
    ("pf_c1_h32_t130_causal", 22, 4096, 32, 128, 4, 256, 768, 130, True),     # BASELINE configs[0] ranks, ragged T
]


def causal_mask(T, dtype=torch.float16):
    """The additive 4-D mask HF 4.37.2 hands to the attention module: finfo.min above the diagonal."""
    m = torch.zeros(T, T, dtype=dtype)
    m.masked_fill_(torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1), torch.finfo(dtype).min)
    return m


def prefill_inputs(seed, hidden, H, D, gs, rank_k, rank_v, T, causal):
    """Weights as in step_inputs (the caches/token drawn there are discarded) + a randn prompt [T, hidden]."""
    w, _, _, _, _ = step_inputs(seed, hidden, H, D, gs, rank_k, rank_v, 1, False)
    rng = np.random.default_rng(5000 + seed)
    prompt = _f16(rng, (T, hidden))
    return w, prompt, (causal_mask(T) if causal else None)


# ---------------------------------------------------------------- quantiser cases
QUANT_R = [32, 64, 128, 384]
QUANT_ROWS = 24


def quant_inputs(seed, R):
    """[QUANT_ROWS, R] fp16 rows incl. the edge rows the reference arithmetic is touchy about:
    constant row (max == min -> clamp(1e-5) subnormal path), all-zero row, tiny-range row,
    large-magnitude row, one-outlier row, exact .5 quotient rows (round-half-even)."""
    rng = np.random.default_rng(3000 + seed + R)
    x = rng.standard_normal((QUANT_ROWS, R)).astype(np.float32)
    x[0, :] = 0.75                        # constant
    x[1, :] = 0.0                         # zeros
    x[2, :] = 1.0 + 1e-3 * rng.standard_normal(R)   # tiny range on an offset
    x[3, :] *= 300.0                      # large
    x[4, :] *= 0.01
    x[4, 5] = 40.0                        # outlier
    x[5, :] = np.linspace(-1.0, 2.5, R)   # many exact ties on a 4-bit grid
    x[6, :] = np.linspace(0.0, 7.0, R)    # ties on a 3-bit grid
    x[7, :] = -np.abs(x[7, :])            # all negative
    x[8, :] = np.abs(x[8, :]) * 1e-4      # near the fp16 subnormal floor
    return torch.from_numpy(x.astype(np.float16))
