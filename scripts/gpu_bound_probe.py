"""Where the production score_tiles time goes, measured on the production kernels by changing only the QUERIES
(no diagnostic build): the same batch with its dense-head terms removed, with only its dense-head terms, with no terms
at all (the fixed per-(tile, query) cost: accumulator init + selection), and at k = 1. Resident batches, kernel time
from the library's own events. usage: python scripts/gpu_bound_probe.py [flickr|c4|c3t2i|c3i2t] ..."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd import workloads  # noqa: E402


def filt(qp, qt, qw, keep):
    n = np.diff(qp)
    rows = np.repeat(np.arange(len(n)), n)
    sel = keep[qt]
    cnt = np.bincount(rows[sel], minlength=len(n))
    return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), qt[sel], qw[sel]


def rows_of(qp):
    return np.repeat(np.arange(len(qp) - 1), np.diff(qp))


def filt_entries(qp, qt, qw, sel):
    cnt = np.bincount(rows_of(qp)[sel], minlength=len(qp) - 1)
    return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), qt[sel], qw[sel]


def timed(ix, q, k, steps=5):
    b = ix.batch(*q, k)
    b.search(k)
    b.sync()
    b.timing_reset()
    for _ in range(steps):
        b.search(k)
    b.sync()
    calls, score_ms, merge_ms = b.timing_sum()
    return score_ms / calls, merge_ms / calls


def main():
    for name in (sys.argv[1:] or ["flickr", "c4"]):
        wl = {"flickr": workloads.flickr30k_t2i, "c4": workloads.c4_1m, "c3t2i": lambda **kw: workloads.coco5k("t2i", **kw),
              "c3i2t": lambda **kw: workloads.coco5k("i2t", **kw)}[name](threads=16)
        qp, qt, qw = (np.asarray(x) for x in wl.queries)
        qp, qt, qw = qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)
        tmp = tempfile.mkdtemp(prefix="msr_probe_")
        path = m.build_index_from_csr(os.path.join(tmp, "p.idx"), *wl.docs, wl.n_terms, threads=16)
        dp, dt, _ = wl.docs
        n_docs = len(dp) - 1
        df = np.bincount(dt, minlength=wl.n_terms)
        order = np.argsort(-df, kind="stable")[:16]
        dense = np.zeros(wl.n_terms, dtype=bool)
        dense[order[df[order] >= 0.4 * n_docs]] = True  # the build's default dense head: <= 16 terms of density >= 0.4
        with m.SparseIndex(path, device=0) as ix:
            if ix.n_dense != int(dense.sum()):  # (the builder's own rule decides; this guess only labels the variants)
                print(f"(note: the index holds {ix.n_dense} dense-head terms, this script's df >= 0.4 N guess {int(dense.sum())})")
            full = (qp, qt, qw)
            half = np.random.default_rng(0).random(len(qp) - 1) < 0.5
            variants = [("full query", full, 10), ("k = 1", full, 1),
                        ("without its dense-head terms", filt(qp, qt, qw, ~dense), 10),
                        # (a RANDOM half: regular patterns — every second query, every second block of 8 — line up with
                        # the round-robin placement of workgroups on XCDs / CUs, leave half of the CUs with only the
                        # expensive workgroups and show almost no effect: 0.99 x)
                        ("  ... for a random half of the queries", filt_entries(qp, qt, qw, ~dense[qt] | ~half[rows_of(qp)]), 10),
                        ("only its dense-head terms", filt(qp, qt, qw, dense), 10),
                        ("no terms at all", (np.zeros(len(qp), np.int64), qt[:0], qw[:0]), 10)]
            base = None
            print(f"== {wl.description}: {ix.n_tiles} tiles, {ix.n_dense} dense-head terms, "
                  f"{(dense[qt]).sum() / (len(qp) - 1):.1f} of {len(qt) / (len(qp) - 1):.1f} query terms in the dense head")
            for label, q, k in variants:
                s, mg = timed(ix, q, k)
                base = base or s
                print(f"{label:32s} score_tiles {s:8.3f} ms ({s / base:5.2f} x)   merge {mg:6.3f} ms")
        os.remove(path)


if __name__ == "__main__":
    main()
