"""Shared helpers for the tests (seeded small datasets, index comparison)."""
import numpy as np

import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig


def random_dataset(seed, n_docs, dim, nnz_lo=4, nnz_hi=40, empty_every=0, value_scale=1.0):
    rng = np.random.default_rng(seed)
    vecs = []
    for d in range(n_docs):
        if empty_every and d % empty_every == empty_every - 1:
            vecs.append((np.zeros(0, np.uint32), np.zeros(0, np.float32)))
            continue
        n = int(rng.integers(nnz_lo, nnz_hi + 1))
        c = np.sort(rng.choice(dim, min(n, dim), replace=False)).astype(np.uint32)
        v = (rng.exponential(0.45, len(c)) * value_scale + 0.02).astype(np.float32)
        vecs.append((c, v))
    return orc.csr(vecs)


def random_queries(seed, nq, dim, nnz_lo=3, nnz_hi=30):
    rng = np.random.default_rng(seed)
    vecs = []
    for _ in range(nq):
        n = int(rng.integers(nnz_lo, nnz_hi + 1))
        c = np.sort(rng.choice(dim, min(n, dim), replace=False)).astype(np.uint32)
        v = (rng.exponential(0.45, len(c)) + 0.02).astype(np.float32)
        v = v + np.arange(len(c), dtype=np.float32) * 1e-4   # distinct weights
        vecs.append((c, v))
    return orc.csr(vecs)


def desc_equal(a, b):
    A, B = orc.desc_arrays(a), orc.desc_arrays(b)
    for f in ("comp_width", "n_docs", "dim", "nnz", "n_blocks", "n_postings", "n_rows", "n_entries"):
        assert getattr(a, f) == getattr(b, f), f
    for k in A:
        assert np.array_equal(A[k].view(np.uint8), B[k].view(np.uint8)), k
