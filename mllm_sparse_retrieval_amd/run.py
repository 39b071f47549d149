"""The reference's own helper API around the searcher, with the same names and argument meaning
(score_statistic.py:44 imports exactly these from search.py): sparse_search, get_run_dict, search_queries, pickle_load.
"""
from __future__ import annotations

import pickle

import numpy as np


def sparse_search(sparse_retriever, batch_topics, batch_ids, search_args):
    """src/search.py:85-99: batch_search, then per query (in batch_ids order) the score list and the docid list."""
    by_qid = sparse_retriever.batch_search(batch_topics, batch_ids, search_args.depth, threads=search_args.threads)
    sparse_scores, sparse_rankings = [], []
    for qid in batch_ids:
        hits = by_qid[qid]
        sparse_scores.append([h.score for h in hits])
        sparse_rankings.append([h.docid for h in hits])
    return sparse_scores, sparse_rankings


def get_run_dict(batch_ids, batch_scores, batch_rankings, remove_query):
    """src/search.py:66-82. 'docs' skips the query's own id when remove_query; min/max cover ALL returned scores."""
    run = {}
    for qid, scores, docs in zip(batch_ids, batch_scores, batch_rankings):
        kept = {}
        for s, d in zip(scores, docs):
            if not (remove_query and d == qid):
                kept[d] = s
        lo, hi = (min(scores), max(scores)) if len(scores) else (0, 0)
        run[qid] = {"docs": kept, "min_score": lo, "max_score": hi}
    return run


def search_queries(retriever, q_reps, p_lookup, args):
    """src/search.py:55-63 (dense side of the hybrid path): row indices -> external ids as a str array."""
    if args.batch_size > 0:
        all_scores, all_indices = retriever.batch_search(q_reps, args.depth, args.batch_size, args.quiet)
    else:
        all_scores, all_indices = retriever.search(q_reps, args.depth)
    psg_indices = np.array([[str(p_lookup[x]) for x in row] for row in all_indices])
    return all_scores, psg_indices


def pickle_load(path):
    """src/search.py:49-52: (reps, lookup) written by the encode step. Only for files this package wrote itself."""
    with open(path, "rb") as f:
        reps, lookup = pickle.load(f)
    return np.array(reps), lookup
