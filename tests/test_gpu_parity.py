"""GPU parity: the HIP search path (through the C ABI) against the CPU oracle on
the same seeded inputs. Bit-exact: ids, scores (same canonical accumulation
order), summary dots. Run with `-m gpu` on an MI355X."""
import json
import os

import numpy as np
import pytest

import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import random_dataset, random_queries

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gpu_index(cw, dim, off, comps, vals, **cfg):
    ix = _native.NativeIndex.build(cw, dim, off, comps, vals, BuildConfig.defaults(**cfg))
    ix.upload(0)
    return ix


def _compare_batch(ix, q, k, query_cut, heap_factor, first_sorted):
    q_off, qc, qv = q
    gs, gi, gn = ix.batch_search(q_off, qc, qv, k, query_cut, heap_factor, first_sorted)
    os_, oi, on, _, _, _ = orc.batch_search(ix.desc, q_off, qc, qv, k, query_cut, heap_factor, first_sorted,
                                            num_threads=0)
    assert np.array_equal(gn, on)
    for i in range(len(gn)):
        n = int(gn[i])
        assert np.array_equal(gi[i, :n], oi[i, :n]), (i, gi[i, :n], oi[i, :n], gs[i, :n], os_[i, :n])
        assert np.array_equal(gs[i, :n].view(np.uint32), os_[i, :n].view(np.uint32)), i
    return gn


def test_kat_empty_vectors_gpu():
    g = json.load(open(os.path.join(GOLD, "kat_empty_vectors.json")))
    off, comps, vals = orc.csr([(d["components"], d["values"]) for d in g["docs"]])
    ix = _gpu_index(2, g["dim"], off, comps, vals)
    q = g["query"]
    sc, ids = ix.search(q["components"], q["values"], q["k"], q["query_cut"], q["heap_factor"], q["first_sorted"])
    assert ids.tolist() == g["expected_ids"] and sc.tolist() == g["expected_scores"]


def test_kat_rust_usage_gpu():
    g = json.load(open(os.path.join(GOLD, "kat_rust_usage.json")))
    off, comps, vals = orc.csr([(d["components"], d["values"]) for d in g["docs"]])
    ix = _gpu_index(2, g["dim"], off, comps, vals)
    q = g["query"]
    sc, ids = ix.search(q["components"], q["values"], q["k"], q["query_cut"], q["heap_factor"], False)
    assert ids.tolist() == g["expected_ids"] and sc.tolist() == g["expected_scores"]


@pytest.mark.parametrize("cw,dim", [(2, 300), (4, 70000)])
@pytest.mark.parametrize("first_sorted", [False, True])
@pytest.mark.parametrize("block", ["512", "1024"])
def test_search_matches_oracle_small(cw, dim, first_sorted, block, monkeypatch):
    monkeypatch.setenv("SGPU_BLOCK", block)
    off, comps, vals = random_dataset(11, 4000, dim, nnz_lo=8, nnz_hi=200, empty_every=97)
    ix = _gpu_index(cw, dim, off, comps, vals, n_postings=60 if dim == 300 else 1, centroid_fraction=0.2,
                    summary_energy=0.5, max_fraction=6.0)
    q = random_queries(12, 64, dim, 3, 60)
    for (k, qcut, hf) in [(10, 4, 1.0), (10, 10, 0.7), (1, 3, 0.9), (100, 8, 0.8), (10, 64, 0.0)]:
        _compare_batch(ix, q, k, qcut, hf, first_sorted)


def test_summary_distances_bit_exact():
    dim = 400
    off, comps, vals = random_dataset(21, 6000, dim, nnz_lo=10, nnz_hi=120)
    ix = _gpu_index(2, dim, off, comps, vals, n_postings=150, centroid_fraction=0.2, summary_energy=0.5,
                    max_fraction=6.0)
    q_off, qc, qv = random_queries(22, 16, dim, 5, 80)
    a = orc.desc_arrays(ix.desc)
    lists = [int(c) for c in np.argsort(np.diff(a["list_block_start"].astype(np.int64)))[-6:]]
    for qi in range(16):
        c = qc[q_off[qi]:q_off[qi + 1]]
        v = qv[q_off[qi]:q_off[qi + 1]]
        for l in lists:
            got = ix.summary_distances(l, c, v)
            exp = orc.summary_distances(ix.desc, l, c, v)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (qi, l)


def test_edge_cases():
    dim = 128
    off, comps, vals = random_dataset(31, 600, dim, nnz_lo=1, nnz_hi=30, empty_every=5)
    ix = _gpu_index(2, dim, off, comps, vals, n_postings=30)
    # empty query -> no results
    sc, ids = ix.search([], [], 10, 5, 0.7)
    assert len(ids) == 0
    # a query that matches fewer than k documents returns a short result
    q = (np.array([0, 3], np.uint64)[:0], None, None)
    s2, i2 = ix.search([7], [1.0], 1000, 5, 0.7)
    o2, oi2 = orc.search(ix.desc, [7], [1.0], 1000, 5, 0.7)
    assert np.array_equal(i2, oi2) and np.array_equal(s2, o2)
    # invalid arguments are rejected, never crash
    for bad in [dict(c=[5, 3], v=[1, 1]), dict(c=[3, 3], v=[1, 1]), dict(c=[dim], v=[1])]:
        with pytest.raises(_native.SeismicHipError):
            ix.search(bad["c"], bad["v"], 10, 5, 0.7)
    with pytest.raises(_native.SeismicHipError):
        ix.search([1], [1.0], 0, 5, 0.7)          # k == 0
    # n_knn without a graph is ignored, as in the reference (src/inverted_index.rs:215-216)
    a_, b_ = ix.search([7], [1.0], 10, 5, 0.7, n_knn=3), ix.search([7], [1.0], 10, 5, 0.7)
    assert np.array_equal(a_[1], b_[1]) and np.array_equal(a_[0], b_[0])
    # ragged batch incl. empty queries
    q_off = np.array([0, 0, 3, 3, 40], np.uint64)
    rng = np.random.default_rng(3)
    qc = np.concatenate([[1, 5, 9], np.sort(rng.choice(dim, 37, replace=False))]).astype(np.uint32)
    qv = (rng.random(40) + 0.1).astype(np.float32)
    _compare_batch(ix, (q_off, qc, qv), 10, 6, 0.8, True)
