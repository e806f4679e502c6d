"""Fixed-u8 document values (the forward index of the reference's DotVByte / fixedu8 indexes) on the GPU:
bit-exact against the oracle's restatement of the same format (parity with vectorium itself is unpinned,
see include/seismic_hip.h). Run with `-m gpu`."""
import os

import numpy as np
import pytest

import orc
import seismic_amd
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import random_dataset, random_queries

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _same(gpu, cpu):
    gs, gi, gn = gpu
    cs, ci, cn = cpu
    assert np.array_equal(gn, cn)
    for q in range(len(gn)):
        n = int(gn[q])
        assert np.array_equal(gi[q, :n], ci[q, :n]), (q, gi[q, :n], ci[q, :n])
        assert np.array_equal(gs[q, :n].view(np.uint32), cs[q, :n].view(np.uint32)), q


@pytest.mark.parametrize("env", [dict(), dict(SGPU_NO_DENSE="1"), dict(SGPU_BLOCK="1024"), dict(SGPU_BLOCK="512"),
                                 dict(SGPU_FWD_LAYOUT="doc"),
                                 dict(SGPU_ITEMS_MAX="64", SGPU_ITEMS_INIT="16", SGPU_ITEMS_MIN="16", SGPU_RBLOCKS="1")])
def test_fixed_u8_search_matches_oracle(env, monkeypatch):
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    dim = 300
    off, comps, vals = random_dataset(101, 5000, dim, nnz_lo=8, nnz_hi=300, empty_every=89)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals,
                                   BuildConfig.defaults(n_postings=80, centroid_fraction=0.2, summary_energy=0.5,
                                                        max_fraction=6.0)).convert(1)
    assert ix.desc.value_type == 1
    ix.upload(0)
    q = random_queries(102, 64, dim, 3, 70)
    for (k, qcut, hf, srt) in [(10, 4, 1.0, False), (10, 10, 0.7, True), (1, 3, 0.9, False), (100, 8, 0.8, True),
                               (300, 5, 0.9, False)]:
        _same(ix.batch_search(*q, k, qcut, hf, srt), orc.batch_search(ix.desc, *q, k, qcut, hf, srt)[:3])
    b = _native.DeviceBatch(ix, *q, 10)
    b.run(10, 4, 1.0, False)
    g1 = b.fetch(10)
    b.run_counted(10, 4, 1.0, False)
    _same(g1, b.fetch(10))
    kb, counters = b.algorithmic_bytes(10, 2, 1)
    st = orc.batch_search(ix.desc, *q, 10, 4, 1.0, False)[3]
    assert kb == st["algo_bytes"]                      # 8 + nnz * 3 bytes per scored document


def test_fixed_u8_at_scale_and_knn():
    """Synthetic SPLADE shape, 300K docs: the converted index answers like the oracle's converted index;
    scores are within the quantisation step of the f16 index's; the kNN graph builds and refines on it."""
    dim, n_docs, nq = 30_000, 300_000, 500
    docs = _native.synth(n_docs, dim, 42, 0)
    f16 = _native.NativeIndex.build(2, dim, *docs, BuildConfig.defaults(n_postings=600, centroid_fraction=0.2,
                                                                         summary_energy=0.5, max_fraction=6.0))
    u8 = f16.convert(1).upload(0)
    q = _native.synth(nq, dim, 43, 1, docs)
    g = u8.batch_search(*q, 10, 4, 1.0, False)
    _same(g, orc.batch_search(u8.desc, *q, 10, 4, 1.0, False)[:3])
    _same(u8.batch_search(*q, 10, 4, 1.0, True), orc.batch_search(u8.desc, *q, 10, 4, 1.0, True)[:3])
    f16.upload(0)
    h = f16.batch_search(*q, 10, 4, 1.0, False)
    overlap = np.mean([len(set(g[1][i].tolist()) & set(h[1][i].tolist())) / 10.0 for i in range(nq)])
    assert overlap > 0.9, overlap                       # 8-bit values barely move the top-10
    qn = np.diff(q[0].astype(np.int64)).max()
    assert np.abs(g[0][:, 0] - h[0][:, 0]).max() <= qn * 3.5 * u8.desc.val_scale   # |ds| <= sum q * step / 2
    small = _native.NativeIndex.build(2, 400, *random_dataset(103, 3000, 400, nnz_lo=8, nnz_hi=100),
                                      BuildConfig.defaults(n_postings=60)).convert(1).upload(0)
    small.build_knn(4)
    nb = orc.knn_build(small.desc, 4)
    assert np.array_equal(small.get_knn()[0], nb)
    qs = random_queries(104, 30, 400, 3, 40)
    orc.knn_attach(nb, 4)
    try:
        exp = orc.batch_search(small.desc, *qs, 10, 3, 0.9, False, n_knn=3)[:3]
    finally:
        orc.knn_attach(None, 0)
    _same(small.batch_search(*qs, 10, 3, 0.9, False, n_knn=3), exp)


def test_dotvbyte_class_through_the_python_api():
    """SeismicIndexDotVByte (reference src/pylib/dotvbyte.rs): same build / search signatures as SeismicIndex."""
    path = os.path.join(GOLD, "toy", "documents.jsonl")
    ix = seismic_amd.SeismicIndexDotVByte.build(path)
    ref = seismic_amd.SeismicIndex.build(path)
    assert ix._ix.desc.value_type == 2 and ix.len == ref.len == 20 and ix.dim == ref.dim
    qids, vecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "queries.jsonl"))
    comps = [np.array(list(v.keys()), dtype=seismic_amd.get_seismic_string()) for v in vecs]
    vals = [np.array(list(v.values()), dtype=np.float32) for v in vecs]
    res = ix.batch_search(np.array(qids, dtype="U30"), comps, vals, k=10, query_cut=10, heap_factor=0.7)
    exp = ref.batch_search(np.array(qids, dtype="U30"), comps, vals, k=10, query_cut=10, heap_factor=0.7)
    for r, e in zip(res, exp):
        assert [d for _, _, d in r][:3] == [d for _, _, d in e][:3]                   # same leaders
        assert all(abs(a[1] - b[1]) < 0.2 for a, b in zip(r, e))                      # scores within the 8-bit step


def test_dotvbyte_graph_is_the_graph_of_the_index_as_built():
    """SeismicIndexDotVByte.build(nknn=...) builds the kNN graph on the u16/f16 index and converts the
    forward index afterwards (reference src/pylib/dotvbyte.rs:193-209: Index::from_file(..).knn(..), then
    convert_dataset_into): its graph is SeismicIndex's graph, not one computed on 8-bit values."""
    path = os.path.join(GOLD, "toy", "documents.jsonl")
    ix = seismic_amd.SeismicIndexDotVByte.build(path, nknn=3)
    ref = seismic_amd.SeismicIndex.build(path, nknn=3)
    a, da = ix._ix.get_knn()
    b, db = ref._ix.get_knn()
    assert ix._ix.desc.value_type == 2 and da == db == 3 and len(a) == len(b) > 0
    assert np.array_equal(a, b)
    qids, vecs, _ = seismic_amd.index.read_jsonl(os.path.join(GOLD, "toy", "queries.jsonl"))
    comps = [np.array(list(v.keys()), dtype=seismic_amd.get_seismic_string()) for v in vecs]
    vals = [np.array(list(v.values()), dtype=np.float32) for v in vecs]
    res = ix.batch_search(np.array(qids, dtype="U30"), comps, vals, k=5, query_cut=3, heap_factor=0.9, n_knn=2)
    assert len(res) == len(qids) and all(len(r) > 0 for r in res)


def test_fixed_u8_values_with_u32_components():
    """The reference's "fixedu8" value type also goes with u32 components (src/bin/perf_inverted_index.rs:125-126):
    large vocabulary, both lookup layouts the u32 kernels have for it, bit-identical to the oracle's restatement."""
    from util import random_dataset, random_queries
    dim = 90000
    off, comps, vals = random_dataset(31, 4000, dim, nnz_lo=1, nnz_hi=300, empty_every=53)
    cfg = BuildConfig.defaults(n_postings=300, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)
    ix = _native.NativeIndex.build(4, dim, off, comps, vals, cfg).convert(1)
    assert ix.desc.value_type == 1 and ix.desc.comp_width == 4
    ix.upload(0)
    o = orc.OracleIndex(4, dim, off, comps, vals, cfg).convert_fixedu8()
    from util import desc_equal
    desc_equal(ix.desc, o.desc)
    q = random_queries(9, 80, dim, 3, 60)
    for env in ({}, {"SGPU_FORCE_SPLIT": "1"}):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        try:
            for k, cut, hf, srt in ((10, 4, 1.0, False), (100, 10, 0.8, True)):
                _same(ix.batch_search(*q, k, cut, hf, srt), orc.batch_search(o.desc, *q, k, cut, hf, srt)[:3])
        finally:
            for k_ in env:
                os.environ.pop(k_, None)


def _gappy_dataset(seed, n_docs, dim):
    """Documents that exercise every record form of the DotVByte layout: lengths 0, 1, 7, 8, 9, 127 ... 300 (one, two
    and more passes of the 16-lane groups), vocabularies wide enough that some documents have a first component or a
    gap too wide for its field (raw fallback) and some do not."""
    rng = np.random.default_rng(seed)
    lens = [0, 1, 7, 8, 9, 16, 120, 127, 128, 129, 255, 256, 257, 300, 390]
    vecs = []
    for d in range(n_docs):
        n = lens[d % len(lens)] if d % 3 else int(rng.integers(1, 200))
        if d % 5 == 0:      # dense low ids: every gap small
            c = np.sort(rng.choice(min(dim, 3000), min(n, 3000), replace=False))
        else:               # anywhere in the vocabulary: wide gaps for short documents, small ones for long documents
            c = np.sort(rng.choice(dim, n, replace=False))
        vecs.append((c.astype(np.uint32), (rng.exponential(0.5, n) + 0.01).astype(np.float32)))
    return orc.csr(vecs)


@pytest.mark.parametrize("env", [dict(), dict(SGPU_NO_DENSE="1"), dict(SGPU_FORCE_HASH="1"), dict(SGPU_BLOCK="1024"),
                                 dict(SGPU_FWD_LAYOUT="doc"), dict(SGPU_COOP="force", SGPU_COOP_MIN_ITEMS="0"),
                                 dict(SGPU_ITEMS_MAX="64", SGPU_ITEMS_INIT="16", SGPU_ITEMS_MIN="16", SGPU_RBLOCKS="1")])
def test_dotvbyte_component_stream_is_lossless_on_the_gpu(env, monkeypatch):
    """SGPU_VAL_DOTVBYTE (fixed-u8 values + per slice a 16-bit first component, three 12-bit and four 11-bit gaps; raw fallback per document): the
    codec is lossless, so every search returns the fixed-u8 index's rows bit for bit - and the oracle's."""
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    dim = 20000
    off, comps, vals = _gappy_dataset(7, 6000, dim)
    f16 = _native.NativeIndex.build(2, dim, off, comps, vals,
                                    BuildConfig.defaults(n_postings=300, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0))
    u8 = f16.convert(1).upload(0)
    dvb = f16.convert(2)
    assert dvb.desc.value_type == 2 and dvb.convert(1).desc.value_type == 1
    dvb.upload(0)
    assert dvb.device_bytes() < u8.device_bytes()
    rng = np.random.default_rng(8)
    qs = []
    for i in range(64):
        n = int(rng.integers(3, 70))
        c = np.sort(rng.choice(dim if i % 2 else 3000, n, replace=False)).astype(np.uint32)
        qs.append((c, (rng.exponential(0.5, n) + 0.01).astype(np.float32)))
    q = orc.csr(qs)
    for (k, qcut, hf, srt) in [(10, 4, 1.0, False), (10, 10, 0.7, True), (1, 3, 0.9, False), (100, 8, 0.8, True), (300, 5, 0.0, False)]:
        got = dvb.batch_search(*q, k, qcut, hf, srt)
        _same(got, u8.batch_search(*q, k, qcut, hf, srt))
        _same(got, orc.batch_search(dvb.desc, *q, k, qcut, hf, srt)[:3])
    b = _native.DeviceBatch(dvb, *q, 10)
    b.run(10, 4, 1.0, False)
    g1 = b.fetch(10)
    b.run_counted(10, 4, 1.0, False)
    _same(g1, b.fetch(10))
    # kNN refinement reads the document-major records of the same layout
    dvb.build_knn(3)
    nb = orc.knn_build(dvb.desc, 3)
    assert np.array_equal(dvb.get_knn()[0], nb)
    orc.knn_attach(nb, 3)
    try:
        exp = orc.batch_search(dvb.desc, *q, 10, 3, 0.9, False, n_knn=2)[:3]
    finally:
        orc.knn_attach(None, 0)
    _same(dvb.batch_search(*q, 10, 3, 0.9, False, n_knn=2), exp)


def test_dotvbyte_at_scale_matches_fixed_u8():
    """Synthetic SPLADE shape, 1M docs (BASELINE configs[1] size), 1000 queries: rows identical to the fixed-u8 index
    in both traversal modes and through single-query (cooperative) launches; 2.5 instead of 3 bytes per element."""
    dim, n_docs, nq = 30_000, 1_000_000, 1000
    docs = _native.synth(n_docs, dim, 42, 0)
    f16 = _native.NativeIndex.build(2, dim, *docs, BuildConfig.defaults(n_postings=2000, centroid_fraction=0.2,
                                                                         summary_energy=0.5, max_fraction=6.0, use_device=1))
    u8 = f16.convert(1).upload(0)
    dvb = f16.convert(2).upload(0)
    q = _native.synth(nq, dim, 43, 1, docs)
    for srt in (False, True):
        want = u8.batch_search(*q, 10, 4, 1.0, srt)
        _same(dvb.batch_search(*q, 10, 4, 1.0, srt), want)
    _same(dvb.batch_search(*q, 10, 4, 1.0, False), orc.batch_search(dvb.desc, *q, 10, 4, 1.0, False, tuned=True)[:3])
    want = u8.batch_search(*q, 10, 4, 1.0, False)
    sc, ids, n, _, _ = dvb.search_sequential(q[0][:101], q[1], q[2], 10, 4, 1.0, False)
    _same((sc, ids, n), tuple(x[:100] for x in want))
    assert dvb.device_bytes() < 0.92 * u8.device_bytes()   # (the stream saves a sixth of the record bytes; postings, summaries and the row directory are the same)
