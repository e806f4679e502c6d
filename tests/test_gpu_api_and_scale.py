"""GPU tests through the public API and at BASELINE sizes (run with -m gpu on an MI355X)."""
import json
import os

import numpy as np
import pytest

import orc
import seismic_amd
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from seismic_amd.index import _resolve
from util import random_dataset, random_queries

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _same(gpu, cpu):
    gs, gi, gn = gpu
    cs, ci, cn = cpu
    assert np.array_equal(gn, cn)
    for q in range(len(gn)):
        n = int(gn[q])
        assert np.array_equal(gi[q, :n], ci[q, :n]), q
        assert np.array_equal(gs[q, :n].view(np.uint32), cs[q, :n].view(np.uint32)), q


def test_toy_dataset_through_python_api():
    """BASELINE config 1 end to end on the GPU: SeismicIndex.build/search/batch_search with the
    reference's argument names and defaults, against the committed golden results."""
    ix = seismic_amd.SeismicIndex.build(os.path.join(GOLD, "toy", "documents.jsonl"))
    exp = json.load(open(os.path.join(GOLD, "toy", "expected.json")))
    qids, vecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "queries.jsonl"))
    comps = [np.array(list(v.keys()), dtype=seismic_amd.get_seismic_string()) for v in vecs]
    vals = [np.array(list(v.values()), dtype=np.float32) for v in vecs]
    for srt in (True, False):
        res = ix.batch_search(np.array(qids, dtype="U30"), comps, vals, k=10, query_cut=10, heap_factor=0.7,
                              sorted=srt)
        for row, r, qid in zip(exp["queries"], res, qids):
            assert [[d, s] for (_, s, d) in r] == row["seismic_sorted_%s" % srt]
            assert all(q == qid for (q, _, _) in r)
    one = ix.search(qids[0], comps[0], vals[0], k=10, query_cut=10, heap_factor=0.7)   # sorted=True default
    assert [[d, s] for (_, s, d) in one] == exp["queries"][0]["seismic_sorted_True"]
    # unknown tokens are dropped silently
    r2 = ix.search("q", np.array(["definitely-not-a-token"], dtype="U30"), np.array([1.0], np.float32), 10, 10, 0.7)
    assert r2 == []


def test_golden_synth_small_on_gpu():
    g = json.load(open(os.path.join(GOLD, "synth_small.json")))
    off, c, v = orc.csr([(d[0], d[1]) for d in g["docs"]])
    ix = _native.NativeIndex.build(2, g["dim"], off, c, v, BuildConfig.defaults(**g["build"])).upload(0)
    for r in g["results"]:
        for (qc, qv), e in zip(g["queries"], r["per_query"]):
            s, i = ix.search(qc, qv, r["k"], r["query_cut"], r["heap_factor"], r["first_sorted"])
            assert [int(x) for x in i] == e["ids"]
            assert [int(x) for x in s.view(np.uint32)] == e["score_bits"]


@pytest.mark.parametrize("env", [
    dict(SGPU_ITEMS_MAX="64", SGPU_ITEMS_INIT="16", SGPU_ITEMS_MIN="16", SGPU_RBLOCKS="1"),   # many rounds, oversize blocks
    dict(SGPU_NO_DENSE="1"),
    dict(SGPU_BLOCK="512"),
    dict(SGPU_BLOCK="512", SGPU_ITEMS_MAX="128", SGPU_ITEMS_INIT="32", SGPU_ITEMS_MIN="32"),
    dict(SGPU_FORCE_SPLIT="1"),
    dict(SGPU_BLOCK="1024", SGPU_STAGE_BYTES="8192"),                                        # many staging windows
    dict(SGPU_NO_LPT="1", SGPU_ITEMS_INIT="1024"),
    dict(SGPU_VISITED_BITMAP="1"),
    dict(SGPU_DOTS_CAP="1"),                                                                  # one list per group
    dict(SGPU_FWD_LAYOUT="doc"),                                                              # one record per document
    dict(SGPU_FWD_LAYOUT="doc", SGPU_REC_LINE="16"),
])
def test_kernel_paths_under_forced_small_buffers(env, monkeypatch):
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    dim = 300
    off, comps, vals = random_dataset(51, 6000, dim, nnz_lo=8, nnz_hi=300)
    # few huge clusters (centroid_fraction tiny) -> blocks larger than the item buffer
    ix = _native.NativeIndex.build(2, dim, off, comps, vals,
                                   BuildConfig.defaults(n_postings=400, centroid_fraction=0.01, summary_energy=0.5,
                                                        max_fraction=3.0)).upload(0)
    q = random_queries(52, 48, dim, 3, 70)
    for (k, qcut, hf, srt) in [(10, 4, 1.0, False), (10, 6, 0.8, True), (100, 5, 0.9, False), (200, 3, 0.7, True)]:
        g = ix.batch_search(*q, k, qcut, hf, srt)
        c = orc.batch_search(ix.desc, *q, k, qcut, hf, srt)[:3]
        _same(g, c)


@pytest.mark.parametrize("lookup_env", [dict(), dict(SGPU_FORCE_SPLIT="1"), dict(SGPU_NO_HASH="1"), dict(SGPU_FORCE_HASH="1", SGPU_BLOCK="1024")])
def test_large_vocabulary_u32_k100(lookup_env, monkeypatch):
    for k_, v_ in lookup_env.items():
        monkeypatch.setenv(k_, v_)
    """BASELINE config 5 shape at test size: u32 components, 200K vocabulary, k=100, heap_factor sweep."""
    dim = 200_000
    docs = _native.synth(60_000, dim, 42, 0)
    ix = _native.NativeIndex.build(4, dim, *docs, BuildConfig.defaults(n_postings=20, centroid_fraction=0.1,
                                                                        summary_energy=0.4, max_fraction=4.0,
                                                                        min_cluster_size=10)).upload(0)
    q = _native.synth(200, dim, 43, 1, docs)
    for hf in (0.7, 0.8, 0.9, 1.0):
        g = ix.batch_search(*q, 100, 10, hf, False)
        c = orc.batch_search(ix.desc, *q, 100, 10, hf, False)[:3]
        _same(g, c)
    # a query of more than 255 components cannot use the hashed byte table: the batch falls back
    rng = np.random.default_rng(9)
    big_c = np.sort(rng.choice(dim, 300, replace=False)).astype(np.uint32)
    big_v = (rng.random(300) + 0.05).astype(np.float32)
    q2 = (np.concatenate([q[0], [q[0][-1] + 300]]).astype(np.uint64), np.concatenate([q[1], big_c]), np.concatenate([q[2], big_v]))
    _same(ix.batch_search(*q2, 10, 6, 0.9, True), orc.batch_search(ix.desc, *q2, 10, 6, 0.9, True)[:3])


def test_full_size_config_properties():
    """BASELINE config 2 (1M docs x 30K vocab, 1K queries, k=10, best_configs parameters):
    identical to the oracle on every query, plus size-independent properties."""
    dim, n_docs, nq = 30_000, 1_000_000, 1000
    docs = _native.synth(n_docs, dim, 42, 0)
    ix = _native.NativeIndex.build(2, dim, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2,
                                                                        summary_energy=0.5, max_fraction=6.0, use_device=1))
    ix.upload(0)
    q = _native.synth(nq, dim, 43, 1, docs)
    b = _native.DeviceBatch(ix, *q, 10)
    b.run(10, 4, 1.0, False)                  # default path: heap-membership dedup, no visited bitmap
    g1 = b.fetch(10)
    b.run_counted(10, 4, 1.0, False)          # visited bitmap materialised: identical results, exact counters
    _same(g1, b.fetch(10))
    kernel_bytes, counters = b.algorithmic_bytes(10, 2)
    sc, ids, n, st, _, _ = orc.batch_search(ix.desc, *q, 10, 4, 1.0, False)
    _same(g1, (sc, ids, n))
    assert kernel_bytes == st["algo_bytes"]                         # the kernel's own work counters
    assert int(counters[:, 5].sum()) == st["docs_scored"]
    gs, gi, gn = g1
    assert (gn == 10).all()
    assert (np.diff(gs, axis=1) <= 0).all()                          # best first
    assert all(len(set(r.tolist())) == 10 for r in gi)              # no duplicate documents
    b.run(10, 4, 1.0, False)                                         # idempotence (scratch is reset)
    g2 = b.fetch(10)
    _same(g1, g2)
    # every returned score is the true inner product of that document (canonical order)
    rng = np.random.default_rng(0)
    for qi in rng.choice(nq, 20, replace=False):
        c = q[1][q[0][qi]:q[0][qi + 1]]
        v = q[2][q[0][qi]:q[0][qi + 1]]
        for j in (0, 9):
            assert np.float32(orc.score_doc(ix.desc, int(gi[qi, j]), c, v)) == gs[qi, j]
    # recall@10 vs exact is what the algorithm gives (identical to the oracle's by construction)
    es, ei, en = ix.exact_search(*q, 10)
    rec = np.mean([len(set(gi[i].tolist()) & set(ei[i].tolist())) / 10.0 for i in range(nq)])
    assert rec > 0.9
    # Python-default sorted=True on the same data
    b.run(10, 4, 1.0, True)
    _same(b.fetch(10), orc.batch_search(ix.desc, *q, 10, 4, 1.0, True)[:3])


def test_guidelines_operating_point_lists_of_thousands_of_blocks():
    """docs/Guidelines.md:44-70 (n_postings 3000, max_fraction 6, query_cut 10): posting lists capped
    at 18 000 postings, i.e. up to 3600 blocks at centroid_fraction 0.2, ten of them per query - 144 KB
    of block dots if they had to sit in LDS together. The kernel walks the lists in groups; the
    results are the oracle's. (Twelve "hot" components occur in a quarter of the documents each, so
    their lists hit the cap; min_cluster_size 0 keeps every cluster of these random documents, so a
    capped list has the full 0.2 x 18 000 = 3600 blocks.)"""
    rng = np.random.default_rng(5)
    n_docs, dim, nnz, hot = 100_000, 3000, 40, 12
    body = hot + np.sort(np.argsort(rng.random((n_docs, dim - hot)), axis=1)[:, :nnz], axis=1)
    hots = np.sort(np.argsort(rng.random((n_docs, hot)), axis=1)[:, :3], axis=1)
    comps = np.concatenate([hots, body], axis=1).astype(np.uint32)
    vals = (rng.exponential(0.45, comps.shape) + 0.02).astype(np.float32)
    off = (np.arange(n_docs + 1) * comps.shape[1]).astype(np.uint64)
    ix = _native.NativeIndex.build(2, dim, off, comps.ravel(), vals.ravel(),
                                   BuildConfig.defaults(n_postings=3000, centroid_fraction=0.2, summary_energy=0.4,
                                                        max_fraction=6.0, min_cluster_size=0))
    a = orc.desc_arrays(ix.desc)
    nb = np.diff(a["list_block_start"].astype(np.int64))
    assert np.sort(nb)[-10] >= 3000, np.sort(nb)[-12:]
    ix.upload(0)
    qs = []
    for _ in range(60):   # the ten heaviest components of every query are hot ones
        hc = np.sort(rng.choice(hot, 10, replace=False))
        oc = hot + np.sort(rng.choice(dim - hot, 20, replace=False))
        qs.append((np.concatenate([hc, oc]).astype(np.uint32),
                   np.concatenate([rng.random(10) + 2.0, rng.random(20) * 0.5 + 0.01]).astype(np.float32)))
    q = orc.csr(qs)
    for srt in (True, False):
        g = ix.batch_search(*q, 10, 10, 0.8, srt)
        c = orc.batch_search(ix.desc, *q, 10, 10, 0.8, srt)[:3]
        _same(g, c)
    g = ix.batch_search(*q, 100, 10, 0.7, True)
    _same(g, orc.batch_search(ix.desc, *q, 100, 10, 0.7, True)[:3])
