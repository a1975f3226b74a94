#!/usr/bin/env python3
"""Generate the non-power-of-two Hadamard factors hadK (K = 12 ... 244) as DATA by evaluating the reference.

Run only in the build container (needs /root/reference, read-only):

    python tests/golden/make_hadk.py

palu/model/modules/hadamard_utils.py:5-83 (`get_hadK`) selects, for a width n = K * 2^m, one of 18 literal K x K
+-1 matrices (:196-5046).  This script calls the reference's get_hadK for every K, checks H.H^T = K.I, and stores
  * palu_amd/kernel/hadk_tables.npz : the matrices, bit-packed (1 bit per entry, ~14 KB) -- numeric data the product's
    apply_hadamard loads (no reference source text travels);
  * tests/golden/g8_hadk_tables.npz : the same packed matrices for the oracle (test infrastructure);
  * tests/golden/g8_hadk.npz        : for every K two widths n = K * 2^m with seeded inputs and the reference's own
    matmul_hadU(x) (the in-tree pure-torch transform, hadamard_utils.py:92-113) as expected outputs.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import scipy.linalg
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

KS = [244, 180, 172, 156, 140, 108, 92, 84, 76, 68, 60, 52, 44, 36, 28, 40, 20, 12]   # get_hadK's test order


def main():
    stub = types.ModuleType("fast_hadamard_transform")
    stub.hadamard_transform = lambda x, scale=1.0: (x @ torch.from_numpy(
        scipy.linalg.hadamard(x.shape[-1]).astype(np.float32)).to(x.dtype)) * scale
    sys.modules["fast_hadamard_transform"] = stub
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_hadamard_utils", "/root/reference/palu/model/modules/hadamard_utils.py")
    rh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rh)

    tables, gold = {}, {}
    rng = np.random.default_rng(808)
    for K in KS:
        h, k = rh.get_hadK(K)              # n = K: K * 2^0
        assert k == K, (K, k)
        m = h.numpy().astype(np.int64)
        assert m.shape == (K, K) and set(np.unique(m)) <= {-1, 1}
        assert np.array_equal(m @ m.T, K * np.eye(K, dtype=np.int64)), K
        tables[f"had{K}"] = np.packbits((m > 0).astype(np.uint8).reshape(-1))
    # widths: every K itself, K * 8, and the multiples of 32 the Fisher rank search produces (rank_search.py:11-17);
    # which table a width selects is get_hadK's decision (e.g. 160 = 20 * 8 takes K = 40), stored beside the vectors
    widths = sorted(set(KS) | {K * 8 for K in KS} | {n for n in range(32, 801, 32) if n & (n - 1)})
    for n in widths:
        try:
            hk, kk = rh.get_hadK(n)
        except AssertionError:
            continue                           # not K * 2^m for any table: the reference refuses it too
        x = torch.from_numpy(rng.standard_normal((3, n)).astype(np.float32))
        gold[f"n{n}/K"] = np.array(kk)
        gold[f"n{n}/x"] = x.numpy()
        gold[f"n{n}/hadU"] = rh.matmul_hadU(x).numpy()
        gold[f"n{n}/hadUt"] = rh.matmul_hadU(x, transpose=True).numpy()
    np.savez_compressed(os.path.join(ROOT, "palu_amd", "kernel", "hadk_tables.npz"), ks=np.array(KS), **tables)
    np.savez_compressed(os.path.join(HERE, "g8_hadk_tables.npz"), ks=np.array(KS), **tables)   # the oracle's copy
    np.savez_compressed(os.path.join(HERE, "g8_hadk.npz"), ks=np.array(KS), **gold)
    print("wrote", sum(v.nbytes for v in tables.values()), "table bytes,", len(gold), "golden arrays")


if __name__ == "__main__":
    main()
