#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ (run in the build container,
where /root/reference exists; the GPU box only sees the committed outputs).

  toy/documents.jsonl, toy/queries.jsonl
      DATA of the reference's examples/toy_dataset (20 docs, 5 queries) with the free-text
      "content" field dropped (ids + sparse vectors only).
  toy/expected.json
      SELF-GENERATED, NOT REFERENCE-GENERATED (the reference cannot be built or imported here):
      exact top-10 and Seismic top-10 (Python-default build params, k=10, query_cut=10,
      heap_factor=0.7, sorted True/False) computed by the CPU oracle through the same
      token numbering (sorted tokens) the product uses.
  synth_small.json
      a small seeded synthetic case + the oracle's results (cross-box regression vector).
  toy_inner/documents.bin, toy_inner/queries.bin, toy_inner/token_to_id_mapping.json
      REFERENCE-GENERATED: the output of the reference's own scripts/convert_json_to_inner_format.py
      (run here, where /root/reference exists) on its toy dataset - the inner binary format that
      SeismicIndexRaw.build / batch_search read (src/pylib/mod.rs:987,1127). Pins read_inner_format,
      write_inner_format and the sorted token numbering against bytes the reference wrote.
  accuracy/case*_results.tsv, case*_groundtruth.tsv, expected.json
      REFERENCE-GENERATED expected values: seeded result / ground-truth TSV pairs in the layout of
      perf_inverted_index's dump (self-made inputs), and the accuracy the reference's own
      `compute_accuracy` (scripts/run_experiments.py:287-309) returns for each pair. That function is
      executed here straight from the reference's file (its module imports packages this image lacks,
      so only that one function is compiled, with pandas); nothing of its text is kept in this repository.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from seismic_amd._abi import BuildConfig  # noqa: E402
from seismic_amd.index import _resolve, _to_csr, _token_map, read_jsonl  # noqa: E402

REF = "/root/reference/examples/toy_dataset"


def strip(src, dst):
    with open(src) as f, open(dst, "w") as g:
        for line in f:
            row = json.loads(line)
            g.write(json.dumps({"id": row["id"], "vector": row["vector"]}) + "\n")


def reference_inner_format():
    import shutil
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([sys.executable, "/root/reference/scripts/convert_json_to_inner_format.py",
                               "--document-path-or-folder", os.path.join(REF, "documents.jsonl"),
                               "--query-path", os.path.join(REF, "queries.jsonl"), "--output-dir", tmp],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dst = os.path.join(HERE, "toy_inner")
        os.makedirs(dst, exist_ok=True)
        for f in ("documents.bin", "queries.bin", "token_to_id_mapping.json"):
            shutil.copy(os.path.join(tmp, "data", f), os.path.join(dst, f))


def reference_function(path, name, namespace):
    """The function `name` of a reference Python file, compiled from the file where it lies."""
    import ast
    tree = ast.parse(open(path).read(), path)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    exec(compile(ast.Module([node], []), path, "exec"), namespace)
    return namespace[name]


def reference_accuracy_cases():
    import contextlib
    import io
    import pandas as pd
    fn = reference_function("/root/reference/scripts/run_experiments.py", "compute_accuracy", {"pd": pd})
    dst = os.path.join(HERE, "accuracy")
    os.makedirs(dst, exist_ok=True)
    rng = np.random.default_rng(77)
    expected = {}
    for case in range(4):
        nq, k = [12, 30, 7, 1][case], [10, 10, 5, 3][case]
        gt, res = [], []
        for q in range(nq):
            truth = rng.choice(500, k, replace=False)
            for r, d in enumerate(truth):
                gt.append((q, int(d), r, float(100 - r)))
            if case == 2 and q == 3:          # a ground-truth row written twice counts twice in the denominator
                gt.append((q, int(truth[0]), 0, 100.0))
            if case == 1 and q % 7 == 0:      # queries the run did not answer
                continue
            n_ret = k if case != 2 else int(rng.integers(1, k + 1))   # fewer than k results
            keep = rng.random(k) < [1.0, 0.7, 0.5, 0.34][case]
            got = [int(d) if keep[i] else int(600 + rng.integers(0, 400)) for i, d in enumerate(truth)][:n_ret]
            for r, d in enumerate(got):
                res.append((q, d, r, float(50 - r)))
        if case == 1:                         # and queries the ground truth does not hold
            res += [(nq + 5, 1, 0, 1.0), (nq + 5, 2, 1, 0.5)]
        for tag, rows in (("results", res), ("groundtruth", gt)):
            with open(os.path.join(dst, "case%d_%s.tsv" % (case, tag)), "w") as f:
                for q, d, r, sc in rows:
                    f.write("%d\t%d\t%d\t%.6g\n" % (q, d, r, sc))
        with contextlib.redirect_stdout(io.StringIO()):
            expected["case%d" % case] = float(fn(os.path.join(dst, "case%d_results.tsv" % case),
                                                 os.path.join(dst, "case%d_groundtruth.tsv" % case)))
    json.dump({"note": "accuracy returned by the reference's compute_accuracy (scripts/run_experiments.py:287-309) "
                       "for each (results, groundtruth) pair of this directory", "accuracy": expected},
              open(os.path.join(dst, "expected.json"), "w"), indent=1)


def main():
    reference_accuracy_cases()
    reference_inner_format()
    os.makedirs(os.path.join(HERE, "toy"), exist_ok=True)
    strip(os.path.join(REF, "documents.jsonl"), os.path.join(HERE, "toy", "documents.jsonl"))
    strip(os.path.join(REF, "queries.jsonl"), os.path.join(HERE, "toy", "queries.jsonl"))
    ids, vecs, _ = read_jsonl(os.path.join(HERE, "toy", "documents.jsonl"))
    qids, qvecs, _ = read_jsonl(os.path.join(HERE, "toy", "queries.jsonl"))
    tm = _token_map(vecs)
    off, c, v = _to_csr(vecs, tm)
    ix = orc.OracleIndex(2, len(tm), off, c, v, BuildConfig.defaults())
    exp = {"note": "self-generated by the CPU oracle, not by the reference", "dim": len(tm), "n_docs": len(ids),
           "queries": []}
    for qid, qv in zip(qids, qvecs):
        qc, qw = _resolve(list(qv.keys()), list(qv.values()), tm)
        es, ei = orc.exact_search(ix.desc, qc, qw, 10, orc.ORDER_SEQ)
        row = {"query_id": qid, "exact": [[ids[int(i)], float(s)] for s, i in zip(es, ei) if s > 0]}
        for srt in (False, True):
            s, i = orc.search(ix.desc, qc, qw, 10, 10, 0.7, srt)
            row["seismic_sorted_%s" % srt] = [[ids[int(x)], float(y)] for y, x in zip(s, i)]
        exp["queries"].append(row)
    json.dump(exp, open(os.path.join(HERE, "toy", "expected.json"), "w"), indent=1)

    rng = np.random.default_rng(2024)
    dim, n_docs = 96, 400
    docs = []
    for _ in range(n_docs):
        n = int(rng.integers(2, 25))
        cc = np.sort(rng.choice(dim, n, replace=False))
        vv = np.round(rng.exponential(0.5, n) + 0.05, 3)
        docs.append((cc.tolist(), vv.tolist()))
    queries = []
    for _ in range(12):
        n = int(rng.integers(3, 20))
        cc = np.sort(rng.choice(dim, n, replace=False))
        vv = np.round(rng.exponential(0.5, n) + 0.05 + np.arange(n) * 1e-3, 4)
        queries.append((cc.tolist(), vv.tolist()))
    off, c, v = orc.csr(docs)
    cfg = dict(n_postings=30, centroid_fraction=0.2, summary_energy=0.5, max_fraction=2.0)
    ix = orc.OracleIndex(2, dim, off, c, v, BuildConfig.defaults(**cfg))
    out = {"dim": dim, "docs": docs, "queries": queries, "build": cfg, "results": []}
    for (k, qcut, hf, srt) in [(10, 4, 1.0, False), (10, 8, 0.8, True), (3, 20, 0.0, False)]:
        res = []
        for qc, qv in queries:
            s, i = orc.search(ix.desc, qc, qv, k, qcut, hf, srt)
            res.append({"ids": [int(x) for x in i], "score_bits": [int(np.float32(y).view(np.uint32)) for y in s]})
        out["results"].append({"k": k, "query_cut": qcut, "heap_factor": hf, "first_sorted": srt, "per_query": res})
    json.dump(out, open(os.path.join(HERE, "synth_small.json"), "w"))
    print("fixtures written")


if __name__ == "__main__":
    main()
