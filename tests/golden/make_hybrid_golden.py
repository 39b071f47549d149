#!/usr/bin/env python3
"""Generates tests/golden/hybrid_golden.json by RUNNING the reference's own pure-Python fusion code.

Run in the build container only (it needs /root/reference, which never travels to the GPU box):
    python tests/golden/make_hybrid_golden.py
It imports /root/reference/src/hybrid.py (argparse + tqdm only) and records, for seeded random inputs:
  - fuse(runs, weights)                      src/hybrid.py:32-53
  - fuse_statistic(runs, weights)            src/hybrid.py:56-90  (as {qid: {doc: [score, type]}})
  - write_trec_run(run, file, name) text     src/hybrid.py:20-29
  - read_trec_run(file) of that text         src/hybrid.py:8-17
The JSON holds inputs and outputs only (data, no reference source).
"""
import importlib.util
import json
import os
import random
import sys
import tempfile

REF = "/root/reference/src/hybrid.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hybrid_golden.json")


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_hybrid", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_run(rng, qids, pool, depth, integer_scores, empty_prob):
    run = {}
    for qid in qids:
        if rng.random() < empty_prob:
            run[qid] = {"docs": {}, "min_score": 0, "max_score": 0}
            continue
        docs = rng.sample(pool, depth)
        if integer_scores:
            scores = sorted((float(rng.randint(1, 4000)) for _ in docs), reverse=True)
        else:
            scores = sorted((rng.uniform(-0.2, 1.0) for _ in docs), reverse=True)
        # get_run_dict semantics: min/max over ALL scores, 'docs' may miss the query's own id (remove_query)
        kept = {d: s for d, s in zip(docs, scores) if d != qid}
        run[qid] = {"docs": kept, "min_score": min(scores), "max_score": max(scores)}
    return run


def main():
    ref = load_ref()
    rng = random.Random(20250418)
    cases = []
    for ci, (nq, depth, alpha, empty_prob) in enumerate([(6, 5, 0.5, 0.0), (8, 12, 0.3, 0.2), (5, 1, 0.9, 0.0), (4, 7, 0.0, 0.0)]):
        qids = [str(100 + i) for i in range(nq)]
        pool = [str(i) for i in range(90, 140)]
        dense = make_run(rng, qids, pool, depth, False, empty_prob)
        sparse = make_run(rng, qids, pool, depth, True, empty_prob)
        if ci == 2:  # degenerate span: max == min -> the 1e-9 floor of src/hybrid.py:46
            for r in (dense, sparse):
                for q in r.values():
                    q["min_score"] = q["max_score"]
        weights = [alpha, 1 - alpha]
        fused = ref.fuse([dense, sparse], weights)
        stat = {q: {d: [r.score, r.type] for d, r in v.items()} for q, v in ref.fuse_statistic([dense, sparse], weights).items()}
        with tempfile.TemporaryDirectory() as d:
            f1 = os.path.join(d, "sparse.trec")
            ref.write_trec_run(sparse, f1, name="sparse")
            sparse_text = open(f1).read()
            sparse_back = ref.read_trec_run(f1)
            f2 = os.path.join(d, "fusion.trec")
            ref.write_trec_run(fused, f2)
            fused_text = open(f2).read()
        cases.append(dict(dense=dense, sparse=sparse, weights=weights, fused=fused, fused_statistic=stat, sparse_trec=sparse_text,
                          sparse_trec_read=sparse_back, fused_trec=fused_text))
    json.dump({"generator": "tests/golden/make_hybrid_golden.py", "reference": "src/hybrid.py @ 2025-04-18",
               "cases": cases}, open(OUT, "w"), indent=1)  # key order is data: it is the tie order of later stable sorts
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    sys.exit(main())
