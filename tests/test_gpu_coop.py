"""The cooperative kernel variant (workgroups without a query help the owners of open wide rounds:
DESIGN.md "Cooperative mode") returns the reference's results bit for bit.

SGPU_COOP=force turns the variant on for every launch and lets an owner go wide without waiting for idle
workgroups, so the whole protocol (publish, claims, candidate lists, exact replay, the fallback to local
rounds when a round produces more candidates than the owner can sort) runs inside ordinary batches, where
owners and helpers of many queries interleave; auto mode is what small launches and single queries use."""
import numpy as np
import pytest

import orc
from seismic_amd import _native
from seismic_amd._abi import BuildConfig
from test_gpu_fuzz import test_differential as _differential

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(26))
def test_forced_cooperative_path_on_the_fuzz_seeds(seed, monkeypatch):
    monkeypatch.setenv("SGPU_COOP", "force")
    monkeypatch.setenv("SGPU_COOP_MIN_ITEMS", str([0, 1, 64, 300][seed % 4]))
    monkeypatch.setenv("SGPU_COOP_CHUNK", str([16, 64, 128, 1024][(seed // 2) % 4]))
    # positions per claim: the owner picks the round's figure between CHUNK_MIN and CHUNK from the idle count it
    # sees, so consecutive rounds of a slot differ in chunk size (the claim word counts chunks, r04: a claim that
    # lands on the next round's word is a valid claim of THAT round whatever the sizes)
    monkeypatch.setenv("SGPU_COOP_CHUNK_MIN", str([1, 4, 3, 7, 64][seed % 5]))
    if seed % 3 == 1:
        monkeypatch.setenv("SGPU_COOP_MAX_CAND", str([1, 4, 40][seed % 3]))   # rounds overflow: local rounds take over
    _differential(seed, "default", monkeypatch)


def _shape(n_docs, n_postings, nq, seed=43):
    docs = _native.synth(n_docs, 30000, 42, 0)
    ix = _native.NativeIndex.build(2, 30000, *docs, BuildConfig.defaults(
        n_postings=n_postings, centroid_fraction=0.2, summary_energy=0.5, max_fraction=6.0, use_device=1))
    ix.upload(0)
    return ix, _native.synth(nq, 30000, seed, 1, docs)


def _same(a, b):
    return (np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
            and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)))


def test_cooperative_modes_agree_with_the_oracle_at_config_2_size(monkeypatch):
    """BASELINE configs[1] (1M docs x 30K vocabulary, 1000 queries, k = 10): off / auto / forced
    cooperative launches of the whole batch, of 100-query launches and of single queries (one owner, every
    other workgroup helping) all return the oracle's results."""
    ix, q = _shape(1000000, 2000, 1000)
    q_off, qc, qv = q
    want = orc.batch_search(ix.desc, *q, 10, 4, 1.0, False, tuned=True)[:3]
    for srt in (False, True):
        ref = orc.batch_search(ix.desc, *q, 10, 4, 1.0, srt, tuned=True)[:3] if srt else want
        for mode in ("0", "1", "force"):
            monkeypatch.setenv("SGPU_COOP", mode)
            assert _same(ix.batch_search(*q, 10, 4, 1.0, srt), ref), (mode, srt)
            for lo in range(0, 1000, 100):   # 100-query launches: 1024-thread workgroups, 156 of 256 idle
                o0, o1 = int(q_off[lo]), int(q_off[lo + 100])
                got = ix.batch_search(q_off[lo:lo + 101] - q_off[lo], qc[o0:o1], qv[o0:o1], 10, 4, 1.0, srt)
                assert _same(got, tuple(x[lo:lo + 100] for x in ref)), (mode, srt, lo)
    monkeypatch.setenv("SGPU_COOP", "1")
    sc, ids, n, mean_us, _ = ix.search_sequential(q_off[:201], qc, qv, 10, 4, 1.0, False)
    assert _same((sc, ids, n), tuple(x[:200] for x in want))
    # heap_factor sweep and a large k through single-query launches
    for k, cut, hf in ((100, 10, 0.7), (10, 8, 0.9), (1, 4, 1.2), (10, 4, -1.0)):
        ref = orc.batch_search(ix.desc, q_off[:41], qc, qv, k, cut, hf, False, tuned=True)[:3]
        assert _same(ix.search_sequential(q_off[:41], qc, qv, k, cut, hf, False)[:3], ref), (k, cut, hf)


def test_cooperative_launches_keep_the_board_clean(monkeypatch):
    """The last workgroup of a cooperative launch zeroes the board for the next one: hundreds of launches of
    different sizes back to back on one lane, helped and unhelped, stay identical to a plain launch."""
    ix, q = _shape(200000, 800, 600, seed=47)
    q_off, qc, qv = q
    monkeypatch.setenv("SGPU_COOP", "0")
    want = ix.batch_search(*q, 10, 4, 1.0, False)
    monkeypatch.setenv("SGPU_COOP", "1")
    rng = np.random.default_rng(5)
    for it in range(150):
        nq = int(rng.choice([1, 2, 7, 31, 100, 300]))
        lo = int(rng.integers(0, 600 - nq + 1))
        o0, o1 = int(q_off[lo]), int(q_off[lo + nq])
        got = ix.batch_search(q_off[lo:lo + nq + 1] - q_off[lo], qc[o0:o1], qv[o0:o1], 10, 4, 1.0, False)
        assert _same(got, tuple(x[lo:lo + nq] for x in want)), (it, nq, lo)


@pytest.mark.parametrize("mode", ["auto", "force"])
def test_rows_are_complete_when_a_small_call_returns_early(mode, monkeypatch):
    """A cooperative call of at most 16 queries returns as soon as the launch's LAST query has stored the done word
    into the pinned host arena - the launch is still winding down then (r04). Every row of every query of the call
    must be there: the rows of a query and the done word can leave from different XCDs, so each workgroup releases its
    rows at system scope before its query counts as finished (a mere drain lost rows of earlier queries). Many calls of
    2 .. 16 queries back to back, each compared with the rows of one large launch; then the same with the early
    return switched off."""
    if mode == "force":
        monkeypatch.setenv("SGPU_COOP", "force")
    ix, q = _shape(300_000, 600, 1200)
    off, qc, qv = q
    want = ix.batch_search(off, qc, qv, 10, 4, 1.0, False)   # one plain launch of 1200 queries

    def call(a, b):
        o = (off[a:b + 1] - off[a]).astype(np.uint64)
        return ix.batch_search(o, qc[int(off[a]):int(off[b])], qv[int(off[a]):int(off[b])], 10, 4, 1.0, False)

    for early in ("1", "0"):
        monkeypatch.setenv("SGPU_EARLY_DONE", early)
        a = 0
        for it in range(150):
            n = (2, 5, 15, 16, 3, 1, 9)[it % 7]
            if a + n > 1200:
                a = 0
            gs, gi, gn = call(a, a + n)
            assert np.array_equal(gn, want[2][a:a + n]), (early, it, n, gn, want[2][a:a + n])
            assert np.array_equal(gi, want[1][a:a + n]), (early, it, n)
            assert np.array_equal(gs.view(np.uint32), want[0][a:a + n].view(np.uint32)), (early, it, n)
            a += n
