#!/usr/bin/env python3
"""Generates tests/golden/run_golden.json by RUNNING the reference's own helper functions of src/search.py.

`import search` fails in this container (faiss / pyserini / tevatron / nltk / peft are absent), but the four
helpers the path exposes — pickle_load, search_queries, get_run_dict, sparse_search (src/search.py:49-99; the names
score_statistic.py:44 imports) — are pure Python over numpy. This script parses the reference file with `ast`, compiles
exactly those four FunctionDef nodes (nothing else of the module runs: none of its imports, no stand-in for any missing
library) into a namespace that holds `np` and `pickle`, calls them on seeded inputs with small fake retrievers, and
records inputs and outputs. Data only; run in the build container (the reference never travels):
    python tests/golden/make_run_golden.py
"""
import ast
import json
import os
import pickle
import random
import tempfile
from types import SimpleNamespace

import numpy as np

REF = "/root/reference/src/search.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_golden.json")
WANTED = ("pickle_load", "search_queries", "get_run_dict", "sparse_search")


def load_reference_functions():
    tree = ast.parse(open(REF, encoding="utf-8").read(), filename=REF)
    ns = {"np": np, "pickle": pickle}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANTED:
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
    missing = [n for n in WANTED if n not in ns]
    assert not missing, missing
    return ns


class FakeSparse:
    """What LuceneImpactSearcher.batch_search hands back: {qid: [hit.score, hit.docid]} (src/search.py:86-98)."""

    def __init__(self, table):
        self.table = table

    def batch_search(self, topics, ids, depth, threads=1):
        return {qid: [SimpleNamespace(score=s, docid=d) for d, s in self.table[qid][:depth]] for qid in ids}


class FakeDense:
    """What FaissFlatSearcher returns: (scores [nq, depth], row indices [nq, depth]) (src/search.py:57-59)."""

    def __init__(self, scores, indices):
        self.scores, self.indices = scores, indices
        self.calls = []

    def batch_search(self, q_reps, depth, batch_size, quiet):
        self.calls.append(["batch_search", int(depth), int(batch_size), bool(quiet)])
        return self.scores[:, :depth], self.indices[:, :depth]

    def search(self, q_reps, depth):
        self.calls.append(["search", int(depth)])
        return self.scores[:, :depth], self.indices[:, :depth]


def main():
    ref = load_reference_functions()
    rng = random.Random(20250418)
    cases = []
    for ci in range(6):
        nq, depth = rng.randint(1, 6), rng.choice([1, 3, 10])
        pool = [str(x) for x in rng.sample(range(90, 160), 40)]
        qids = rng.sample(pool, nq) if ci % 2 else [str(500 + i) for i in range(nq)]   # odd cases: query ids ARE doc ids
        table = {}
        for qid in qids:
            n_hits = 0 if rng.random() < 0.2 else rng.randint(1, depth + 2)
            docs = rng.sample(pool, n_hits)
            if ci % 2 and n_hits and rng.random() < 0.7:
                docs[rng.randrange(n_hits)] = qid                                      # the query finds itself
                docs = list(dict.fromkeys(docs))
            scores = sorted((float(rng.randint(1, 40)) for _ in docs), reverse=True)    # repeated scores on purpose
            table[qid] = list(zip(docs, scores))
        order = qids[:]
        rng.shuffle(order)                                                              # batch order != dict order
        args = SimpleNamespace(depth=depth, threads=16)
        scores, rankings = ref["sparse_search"](FakeSparse(table), [f"topic {q}" for q in order], order, args)
        runs = {str(rq): ref["get_run_dict"](order, scores, rankings, rq) for rq in (False, True)}
        cases.append(dict(kind="sparse", table=table, batch_ids=order, depth=depth, scores=scores, rankings=rankings,
                          run_dict=runs))
    for ci, batch_size in enumerate((0, 2, 128)):
        nq, n, depth = 4, 30, 5
        nrng = np.random.default_rng(ci)
        s = np.sort(nrng.random((nq, 8)).astype(np.float32), axis=1)[:, ::-1].copy()
        idx = np.stack([nrng.permutation(n)[:8] for _ in range(nq)]).astype(np.int64)
        lookup = [str(1000 + 7 * i) for i in range(n)] if ci else list(range(2000, 2000 + n))   # ints become strings
        fake = FakeDense(s, idx)
        args = SimpleNamespace(depth=depth, batch_size=batch_size, quiet=True)
        out_scores, out_ids = ref["search_queries"](fake, nrng.random((nq, 16)).astype(np.float32), lookup, args)
        qids = [str(lookup[i]) for i in range(nq)]
        runs = {str(rq): ref["get_run_dict"](qids, out_scores, out_ids, rq) for rq in (False, True)}
        runs = {k: {q: {"docs": {d: float(v) for d, v in e["docs"].items()}, "min_score": float(e["min_score"]),
                        "max_score": float(e["max_score"])} for q, e in r.items()} for k, r in runs.items()}
        cases.append(dict(kind="dense", scores=s.tolist(), indices=idx.tolist(), lookup=lookup, depth=depth,
                          batch_size=batch_size, calls=fake.calls, out_scores=np.asarray(out_scores).tolist(),
                          out_ids=np.asarray(out_ids).tolist(), out_ids_dtype_kind=np.asarray(out_ids).dtype.kind,
                          batch_ids=qids, run_dict=runs))
    # pickle_load on a (reps, lookup) file as src/encode.py:405-410 writes it
    with tempfile.TemporaryDirectory() as d:
        reps = np.arange(12, dtype=np.float32).reshape(3, 4) / 7
        path = os.path.join(d, "corpus_0.pkl")
        with open(path, "wb") as f:
            pickle.dump((reps, ["5", "17", "9"]), f)
        got_reps, got_lookup = ref["pickle_load"](path)
        cases.append(dict(kind="pickle", reps=reps.tolist(), lookup=["5", "17", "9"], got_reps=np.asarray(got_reps).tolist(),
                          got_dtype=str(np.asarray(got_reps).dtype), got_lookup=list(got_lookup)))
    json.dump({"generator": "tests/golden/make_run_golden.py",
               "reference": "src/search.py:49-99 @ 2025-04-18 (four FunctionDef nodes compiled from the file's AST)",
               "cases": cases}, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
