"""How many candidates a TPUT-style threshold merge would have to exchange for term-range shards (SURVEY.md §8e option 1),
MEASURED on the config-4 corpus (CPU, numpy): for a sample of queries and G term ranges balanced by postings,
  phase 1: every shard reports its local top-k partial sums; tau = the k-th best of the sums known so far;
  phase 2: every shard must send each doc whose partial sum is >= tau / G.
Printed: docs sent in phase 2 per query (summed over the shards) and their bytes (8 B: ordinal + partial) next to the
4 B x N the reduce-scatter of accumulators moves per query. usage: python scripts/tput_candidates.py [n_docs] [queries]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd import workloads  # noqa: E402


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    k = 10
    wl = workloads.c4_1m(n_docs=n_docs, n_queries=max(n_sample, 64), threads=8)
    dp, dt, dw = wl.docs
    V = wl.n_terms
    doc_of = np.repeat(np.arange(n_docs, dtype=np.uint32), np.diff(dp).astype(np.int64))
    order = np.argsort(dt, kind="stable")
    t_sorted, d_sorted, w_sorted = dt[order], doc_of[order], dw[order].astype(np.int64)
    ptr = np.searchsorted(t_sorted, np.arange(V + 1))
    df = np.diff(ptr)
    qp, qt, qw = (np.asarray(x) for x in wl.queries)
    print(f"corpus: {n_docs} docs, {len(dt)} postings, V = {V}; sample of {n_sample} queries, k = {k}")
    for G in (2, 4, 8):
        # term ranges balanced by postings (the product's term_bounds rule: contiguous term ids)
        cum = np.concatenate([[0], np.cumsum(df)])
        bounds = [int(np.searchsorted(cum, cum[-1] * g / G)) for g in range(G)] + [V]
        sent, exact_rank_ok = [], 0
        for qi in range(n_sample):
            terms, weights = qt[qp[qi]:qp[qi + 1]], qw[qp[qi]:qp[qi + 1]].astype(np.int64)
            partial = np.zeros((G, n_docs), dtype=np.int64)
            for t, w in zip(terms, weights):
                if w <= 0 or df[t] == 0 or df[t] == n_docs:
                    continue
                g = int(np.searchsorted(bounds, t, side="right") - 1)
                sl = slice(ptr[t], ptr[t + 1])
                np.add.at(partial[g], d_sorted[sl], w * w_sorted[sl])
            total = partial.sum(axis=0)
            # phase 1: local top-k per shard, sums of what is known
            known = {}
            for g in range(G):
                top = np.argpartition(-partial[g], k)[:k]
                for d in top:
                    known[int(d)] = known.get(int(d), 0) + int(partial[g][d])
            tau = sorted(known.values(), reverse=True)[k - 1]
            # phase 2: every doc with a partial >= tau / G on some shard is sent by that shard
            thr = tau / G
            n_sent = int(sum(int((partial[g] >= thr).sum()) for g in range(G)))
            sent.append(n_sent)
            # (sanity: the true top-k is inside the phase-2 candidate set — TPUT's guarantee)
            cand = np.zeros(n_docs, dtype=bool)
            for g in range(G):
                cand |= partial[g] >= thr
            true_top = np.argpartition(-total, k)[:k]
            exact_rank_ok += bool(cand[true_top].all())
        sent = np.array(sent)
        rs_bytes = 4.0 * n_docs * (G - 1) / G
        print(f"G = {G}: phase-2 docs sent per query: median {int(np.median(sent))}, mean {sent.mean():.0f}, max {sent.max()} "
              f"= {100.0 * sent.mean() / (G * n_docs):.1f} % of all (shard, doc) pairs; {8.0 * sent.mean() / 1e6:.2f} MB per query "
              f"vs {rs_bytes / 1e6:.2f} MB for the reduce-scatter of accumulators; true top-{k} inside the candidates: "
              f"{exact_rank_ok}/{n_sample}")


if __name__ == "__main__":
    main()
