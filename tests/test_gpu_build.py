"""Device-assisted index build (SURVEY f-1): the clustering step of the build on the GPU
(seismic_amd/csrc/build_assign.hip) gives the byte-identical index of the host builder, which is
byte-identical to the oracle's reference-following builder (tests/test_builder_parity.py).
Run with `-m gpu`."""
import time

import numpy as np
import pytest

import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from util import desc_equal, random_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cw,dim,cfg", [
    (2, 300, dict(n_postings=200, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)),
    (2, 300, dict(n_postings=200, centroid_fraction=0.3, summary_energy=0.4, max_fraction=2.0, min_cluster_size=0)),
    (2, 120, dict(n_postings=400, centroid_fraction=0.1, summary_energy=0.4, max_fraction=1.5, min_cluster_size=10)),
    (2, 64, dict(n_postings=20000, centroid_fraction=0.257, summary_energy=0.4, max_fraction=6.0, min_cluster_size=0)),  # ~4096 centroids per list: some lists stay on the host
    (4, 70_000, dict(n_postings=2, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)),
    (2, 200, dict(n_postings=100, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, doc_cut=3)),
])
def test_device_build_is_byte_identical(cw, dim, cfg):
    n_docs = 40_000 if dim == 64 else 6000
    off, comps, vals = random_dataset(111, n_docs, dim, nnz_lo=20 if dim == 64 else 8, nnz_hi=min(150, dim // 2), empty_every=53)
    host = _native.NativeIndex.build(cw, dim, off, comps, vals, BuildConfig.defaults(**cfg))
    dev = _native.NativeIndex.build(cw, dim, off, comps, vals, BuildConfig.defaults(use_device=1, **cfg))
    desc_equal(host.desc, dev.desc)
    if dim == 64:   # some lists have more centroids than a wavefront's LDS accumulators hold, some do not
        a = orc.desc_arrays(host.desc)
        lens = np.diff(a["block_post_start"][a["list_block_start"]].astype(np.int64))
        nc = np.floor(np.float32(0.257) * lens.astype(np.float32))
        assert nc.max() > 4096 and nc.min() <= 4096, (nc.min(), nc.max())


def test_device_build_at_scale_and_against_the_oracle_builder(capsys):
    dim, n_docs = 30_000, 500_000
    docs = _native.synth(n_docs, dim, 42, 0)
    cfg = dict(n_postings=1000, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0)
    t0 = time.time()
    host = _native.NativeIndex.build(2, dim, *docs, BuildConfig.defaults(**cfg))
    t1 = time.time()
    dev = _native.NativeIndex.build(2, dim, *docs, BuildConfig.defaults(use_device=1, **cfg))
    t2 = time.time()
    desc_equal(host.desc, dev.desc)
    with capsys.disabled():
        print("\n[build] 500K docs: host %.1f s, device-assisted %.1f s" % (t1 - t0, t2 - t1))
    small = (docs[0][:20001], docs[1][:int(docs[0][20000])], docs[2][:int(docs[0][20000])])
    o = orc.OracleIndex(2, dim, *small, BuildConfig.defaults(**cfg))
    d2 = _native.NativeIndex.build(2, dim, *small, BuildConfig.defaults(use_device=1, **cfg))
    desc_equal(o.desc, d2.desc)


def test_device_build_without_that_device_is_an_error():
    off, comps, vals = random_dataset(112, 200, 64)
    with pytest.raises(_native.SeismicHipError) as e:
        _native.NativeIndex.build(2, 64, off, comps, vals, BuildConfig.defaults(n_postings=20, use_device=64))
    assert e.value.status == 2   # SGPU_EDEVICE: never a silent host build
