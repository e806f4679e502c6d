"""kNN graph (reference Knn::new / Knn::refine, src/inverted_index.rs:430-594) on the GPU vs the oracle."""
import numpy as np
import pytest

import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import random_dataset, random_queries

pytestmark = pytest.mark.gpu


def _same(g, c):
    gs, gi, gn = g
    cs, ci, cn = c
    assert np.array_equal(gn, cn)
    for q in range(len(gn)):
        n = int(gn[q])
        assert np.array_equal(gi[q, :n], ci[q, :n]), (q, gi[q, :n], ci[q, :n])
        assert np.array_equal(gs[q, :n].view(np.uint32), cs[q, :n].view(np.uint32)), q


@pytest.mark.parametrize("cw,dim", [(2, 400), (4, 70000)])
def test_knn_build_and_refine_match_oracle(cw, dim):
    off, comps, vals = random_dataset(71, 3000, dim, nnz_lo=6, nnz_hi=150, empty_every=211)
    ix = _native.NativeIndex.build(cw, dim, off, comps, vals,
                                   BuildConfig.defaults(n_postings=60 if dim == 400 else 1, centroid_fraction=0.2,
                                                        summary_energy=0.5, max_fraction=4.0)).upload(0)
    nknn = 7
    ix.build_knn(nknn)                       # N_docs searches, batched through the GPU kernel
    nb_gpu, kdim = ix.get_knn()
    nb_cpu = orc.knn_build(ix.desc, nknn)    # Knn::new restated on the CPU
    assert kdim == nknn and np.array_equal(nb_gpu, nb_cpu)
    orc.knn_attach(nb_cpu, nknn)
    try:
        q = random_queries(72, 40, dim, 3, 50)
        for (k, qcut, hf, srt, n_knn) in [(10, 4, 1.0, False, 3), (10, 3, 0.9, True, 7), (5, 2, 1.0, False, 20),
                                         (100, 6, 0.8, False, 5)]:
            g = ix.batch_search(*q, k, qcut, hf, srt, n_knn=n_knn)
            c = orc.batch_search(ix.desc, *q, k, qcut, hf, srt, n_knn=n_knn)[:3]
            _same(g, c)
            # refinement changes results w.r.t. the plain search for at least one query here
        plain = ix.batch_search(*q, 10, 2, 1.0, False)
        refined = ix.batch_search(*q, 10, 2, 1.0, False, n_knn=7)
        assert (refined[0].sum(axis=1) >= plain[0].sum(axis=1) - 1e-3).all()   # refinement never hurts the top-k mass
        assert not np.array_equal(plain[1], refined[1])
        # the counted pass (visited bitmap) gives the same refined results
        b = _native.DeviceBatch(ix, *q, 10)
        b.run(10, 4, 1.0, False, n_knn=5)
        r1 = b.fetch(10)
        b.run_counted(10, 4, 1.0, False, n_knn=5)
        _same(r1, b.fetch(10))
    finally:
        orc.knn_attach(None, 0)


def test_knn_set_get_roundtrip_and_python_api(tmp_path):
    import seismic_amd
    dim = 300
    off, comps, vals = random_dataset(81, 800, dim, nnz_lo=5, nnz_hi=60)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, BuildConfig.defaults(n_postings=30)).upload(0)
    nb = orc.knn_build(ix.desc, 4)
    ix.set_knn(nb, 4)
    got, kd = ix.get_knn()
    assert kd == 4 and np.array_equal(got, nb)
    with pytest.raises(_native.SeismicHipError):
        ix.set_knn(np.array([10 ** 6], np.uint32), 1)     # neighbour id out of range


def test_knn_hand_computed_known_answer_on_gpu():
    """tests/golden/kat_knn_hand.json (derivation: tests/test_oracle_kat.py::test_knn_hand_computed):
    Knn::new through the GPU kernel and Knn::refine in the kernel give the hand-derived lists."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_knn_hand.json")))
    off, comps, vals = orc.csr([(d["components"], d["values"]) for d in g["docs"]])
    ix = _native.NativeIndex.build(2, g["dim"], off, comps, vals, BuildConfig.defaults(**g["build"])).upload(0)
    ix.build_knn(g["nknn"])
    nb, kd = ix.get_knn()
    assert kd == g["nknn"] and nb.tolist() == g["expected_neighbours"]
    q = g["refine_query"]
    s0, i0 = ix.search(q["components"], q["values"], q["k"], q["query_cut"], q["heap_factor"], False)
    assert i0.tolist() == g["expected_without_refine"]["ids"] and s0.tolist() == g["expected_without_refine"]["scores"]
    s1, i1 = ix.search(q["components"], q["values"], q["k"], q["query_cut"], q["heap_factor"], False, n_knn=q["n_knn"])
    assert i1.tolist() == g["expected_with_refine"]["ids"]
    assert s1.tolist() == [4.0, 1.0, float(np.float32(0.1) * np.float32(4.0))]
