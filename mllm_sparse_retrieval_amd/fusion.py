"""Hybrid fusion and TREC run files — the reference's src/hybrid.py:8-90 semantics.

fuse(): for every doc in the union of the runs' result lists, sum over runs of
    weight * (score - min_score) / max(max_score - min_score, 1e-9)   if the run retrieved the doc, else 0,
with min/max as recorded by get_run_dict (over the unfiltered list). Output {qid: {doc: fused score}} (no 'docs' level).
"""
from __future__ import annotations


def fuse(runs, weights):
    """src/hybrid.py:32-53."""
    qids = set()
    for run in runs:
        qids.update(run)
    fused = {}
    for qid in qids:
        acc = {}
        spans = [(r[qid]["min_score"], max(r[qid]["max_score"] - r[qid]["min_score"], 1e-9)) for r in runs]
        for run in runs:
            for doc in run[qid]["docs"]:
                if doc in acc:
                    continue
                total = 0
                for other, weight, (lo, den) in zip(runs, weights, spans):
                    s = other[qid]["docs"].get(doc)
                    if s is not None:
                        total += weight * ((s - lo) / den)
                acc[doc] = total
        fused[qid] = acc
    return fused


class ResultRecord:
    """src/hybrid.py:3-6: a fused score and where it came from ('dense', 'sparse' or 'fuse' = both lists held the doc)."""

    __slots__ = ("score", "type")

    def __init__(self, score, type):
        self.score = score
        self.type = type

    def __repr__(self):
        return f"ResultRecord(score={self.score}, type={self.type!r})"


def fuse_statistic(runs, weights):
    """src/hybrid.py:56-90 (used by the reference's score_statistic.py): fuse() that also says which run(s) a doc's
    score came from. A doc found in exactly one run is tagged by the POSITION of the run being walked when the doc is
    first met ('dense' while walking runs[0], 'sparse' afterwards), as the reference does; found in more: 'fuse'."""
    qids = set()
    for run in runs:
        qids.update(run)
    fused = {}
    for qid in qids:
        acc = {}
        spans = [(r[qid]["min_score"], max(r[qid]["max_score"] - r[qid]["min_score"], 1e-9)) for r in runs]
        for position, run in enumerate(runs, start=1):
            for doc in run[qid]["docs"]:
                if doc in acc:
                    continue
                total, found = 0, 0
                for other, weight, (lo, den) in zip(runs, weights, spans):
                    s = other[qid]["docs"].get(doc)
                    if s is not None:
                        total += weight * ((s - lo) / den)
                        found += 1
                kind = ("dense" if position == 1 else "sparse") if found == 1 else "fuse"
                acc[doc] = ResultRecord(total, kind)
        fused[qid] = acc
    return fused


def write_trec_run(run, file, name="fusion"):
    """src/hybrid.py:20-29: `qid Q0 doc rank score name`, docs by score descending (stable)."""
    with open(file, "w") as f:
        for qid, entry in run.items():
            docs = entry["docs"] if "docs" in entry else entry
            ranked = sorted(docs.items(), key=lambda kv: kv[1], reverse=True)
            for rank, (doc, score) in enumerate(ranked, start=1):
                f.write(f"{qid} Q0 {doc} {rank} {score} {name}\n")


def read_trec_run(file):
    """src/hybrid.py:8-17: max_score is the first line's score of a qid, min_score the last line's."""
    run = {}
    with open(file, "r") as f:
        for line in f:
            qid, _, docid, _rank, score, _ = line.strip().split()
            score = float(score)
            if qid not in run:
                run[qid] = {"docs": {}, "max_score": score, "min_score": score}
            run[qid]["docs"][docid] = score
            run[qid]["min_score"] = score
    return run
