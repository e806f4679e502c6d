"""Product index builder (seismic_amd/csrc/builder.cpp, parallel) vs the oracle's
reference-following builder (oracle/seismic_oracle.cpp): byte-identical indexes."""
import numpy as np
import pytest

import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import desc_equal, random_dataset


@pytest.mark.parametrize("seed,n_docs,dim,cw,cfg", [
    (1, 300, 64, 2, dict()),
    (2, 2000, 500, 2, dict(n_postings=40, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)),
    (3, 1500, 70000, 4, dict(n_postings=1, centroid_fraction=0.1, summary_energy=0.4, max_fraction=4.0,
                              min_cluster_size=3)),
    (4, 800, 200, 2, dict(n_postings=10, centroid_fraction=0.5, min_cluster_size=1, doc_cut=3)),
    (5, 50, 40, 2, dict(n_postings=2, max_fraction=1.0)),
])
def test_builder_matches_oracle(seed, n_docs, dim, cw, cfg):
    off, comps, vals = random_dataset(seed, n_docs, dim, empty_every=17)
    c = BuildConfig.defaults(**cfg)
    o = orc.OracleIndex(cw, dim, off, comps, vals, c)
    p = _native.NativeIndex.build(cw, dim, off, comps, vals, BuildConfig.defaults(**cfg))
    desc_equal(o.desc, p.desc)


def test_builder_ties_and_duplicates():
    # many equal values: exercises every documented tie rule (value ties in pruning,
    # top-component selection, centroid argmax, summary ordering)
    rng = np.random.default_rng(9)
    vecs = []
    for _ in range(400):
        n = int(rng.integers(3, 12))
        c = np.sort(rng.choice(30, n, replace=False)).astype(np.uint32)
        v = rng.choice([0.5, 1.0, 1.5], n).astype(np.float32)
        vecs.append((c, v))
    off, comps, vals = orc.csr(vecs)
    cfg = dict(n_postings=20, centroid_fraction=0.3, summary_energy=0.6, max_fraction=2.0)
    o = orc.OracleIndex(2, 30, off, comps, vals, BuildConfig.defaults(**cfg))
    p = _native.NativeIndex.build(2, 30, off, comps, vals, BuildConfig.defaults(**cfg))
    desc_equal(o.desc, p.desc)


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_pruning_ties_across_thread_chunks(seed):
    """global_threshold_pruning keeps the FIRST entries (scan order) among those equal to the threshold
    (src/inverted_index.rs:354-389). The product builder selects chunk-parallel; with values drawn from
    five levels and few postings per list the threshold falls inside a large tie group that spans the
    chunks - the index must equal the oracle's sequential builder for every thread count."""
    rng = np.random.default_rng(seed)
    dim, n_docs = int(rng.integers(20, 200)), int(rng.integers(300, 3000))
    docs = []
    for _ in range(n_docs):
        n = int(rng.integers(0, min(dim, 30)))
        cc = np.sort(rng.choice(dim, n, replace=False))
        docs.append((cc.tolist(), rng.choice([0.25, 0.5, 0.75, 1.0, 1.5], n).tolist()))
    off, c, v = orc.csr(docs)
    npost = int(rng.integers(1, 40))
    assert int(off[-1]) > dim * npost   # the pruning is active
    for nt in (1, 3, 7, 0):
        cfg = BuildConfig.defaults(n_postings=npost, centroid_fraction=0.3, summary_energy=0.5, max_fraction=2.0,
                                   num_threads=nt)
        built, want = _native.NativeIndex.build(2, dim, off, c, v, cfg), orc.OracleIndex(2, dim, off, c, v, cfg)
        desc_equal(built.desc, want.desc)   # (the descriptors point into the two objects: both stay alive)


def test_builder_rejects_bad_input():
    off = np.array([0, 2], np.uint64)
    with pytest.raises(_native.SeismicHipError):   # unsorted components
        _native.NativeIndex.build(2, 10, off, np.array([3, 1], np.uint16), np.array([1, 1], np.float32))
    with pytest.raises(_native.SeismicHipError):   # component >= dim
        _native.NativeIndex.build(2, 10, off, np.array([3, 11], np.uint16), np.array([1, 1], np.float32))
    with pytest.raises(_native.SeismicHipError):   # NaN value
        _native.NativeIndex.build(2, 10, off, np.array([3, 5], np.uint16), np.array([1, np.nan], np.float32))


def test_save_load_roundtrip(tmp_path):
    off, comps, vals = random_dataset(7, 500, 100)
    p = _native.NativeIndex.build(2, 100, off, comps, vals, BuildConfig.defaults(n_postings=20))
    path = str(tmp_path / "ix.sgpu")
    p.save(path)
    q = _native.NativeIndex.load(path)
    desc_equal(p.desc, q.desc)
    r = _native.NativeIndex.from_desc(p.desc)
    desc_equal(p.desc, r.desc)
