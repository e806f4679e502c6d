"""Every query lookup layout returns the oracle's results: the hashed {component id, weight} entries (one LDS
read per document component; the u32 layout) asked for on the fuzz seeds - seeds with u32 components and f16 values run
them, the others (u16 components: the hashed families were dropped in r05; fixed-u8 over u32) their own layout -
512- and 1024-thread workgroups, cooperative and plain launches."""
import pytest

from test_gpu_fuzz import test_differential as _differential

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 4, 7, 9, 14, 15, 16, 18, 19, 20, 22, 24])
def test_hashed_entries_on_the_fuzz_seeds(seed, monkeypatch):
    monkeypatch.setenv("SGPU_FORCE_HASH", "1")
    if seed % 3 == 0:
        monkeypatch.setenv("SGPU_COOP", "force")
    _differential(seed, "default", monkeypatch)


@pytest.mark.parametrize("seed", [2, 9, 19])
def test_without_the_hashed_entries(seed, monkeypatch):
    monkeypatch.setenv("SGPU_NO_HASH", "1")
    _differential(seed, "default", monkeypatch)


@pytest.mark.parametrize("value_type", [0, 1, 2])
@pytest.mark.parametrize("coop", ["0", "force"])
def test_scaled_and_plain_dense_bytes_in_one_launch(value_type, coop, monkeypatch):
    """The dense lookup table holds SCALED bytes (4 * (1 + rank): the byte is the weight's LDS offset) for a query of
    at most 63 components and plain ones (1 + rank) above; the choice is made per query inside one launch, by owners
    and by cooperative helpers alike. Queries of 1, 62, 63, 64, 65, 200 and 255 components, interleaved, f16 /
    fixed-u8 / DotVByte documents, 512- and 1024-thread workgroups: bit-exact against the oracle."""
    import numpy as np
    import orc
    from seismic_amd import _native
    from seismic_amd._abi import BuildConfig
    monkeypatch.setenv("SGPU_COOP", coop)
    rng = np.random.default_rng(4242 + value_type)
    dim, n_docs = 3000, 4000
    vecs = []
    for _ in range(n_docs):
        n = int(rng.integers(4, 300))
        c = np.sort(rng.choice(dim, n, replace=False)).astype(np.uint32)
        vecs.append((c, (rng.exponential(0.5, n) + 0.01).astype(np.float32)))
    off, comps, vals = orc.csr(vecs)
    cfg = BuildConfig.defaults(n_postings=400, centroid_fraction=0.1, summary_energy=0.5, max_fraction=2.0,
                               min_cluster_size=2, doc_cut=10)
    ix = _native.NativeIndex.build(2, dim, off, comps, vals, cfg)
    if value_type:
        ix = ix.convert(value_type)
    ix.upload(0)
    qs = []
    for rep in range(3):
        for n in (1, 62, 63, 64, 65, 200, 255, 63, 64):
            c = np.sort(rng.choice(dim, n, replace=False)).astype(np.uint32)
            qs.append((c, (rng.exponential(0.5, n) + 0.01).astype(np.float32)))
    q = orc.csr(qs)
    for block in ("512", "1024"):
        monkeypatch.setenv("SGPU_BLOCK", block)
        for k, qcut, hf in ((10, 4, 1.0), (100, 8, 0.8)):
            gs, gi, gn = ix.batch_search(*q, k, qcut, hf, False)
            cs, ci, cn, _, _, _ = orc.batch_search(ix.desc, *q, k, qcut, hf, False)
            assert np.array_equal(gn, cn), (block, k)
            for i in range(len(gn)):
                n = int(gn[i])
                assert np.array_equal(gi[i, :n], ci[i, :n]), (block, k, i)
                assert np.array_equal(gs[i, :n].view(np.uint32), cs[i, :n].view(np.uint32)), (block, k, i)
