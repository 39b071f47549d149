"""The reference's own helper API around the searcher, with the same names and argument meaning
(score_statistic.py:44 imports exactly these from search.py): sparse_search, get_run_dict, search_queries, pickle_load.
"""
from __future__ import annotations

import io
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


# What a (reps, lookup) file of the encode step needs to come back to life (src/encode.py:405-410 pickles
# `(np.ndarray, list[str])`): the numpy array reconstructors and nothing else. Containers, str, int, float, bytes are
# pickle opcodes, not globals. Any other global — i.e. anything that could run code — is refused.
_PICKLE_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("_codecs", "encode"),  # protocol <= 2 carries an array's bytes as a latin-1 string
}


class _RepsUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _PICKLE_ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refused global '{module}.{name}': a reps file may only hold numpy arrays, "
                                     f"lists, tuples, str, int and float")

    def persistent_load(self, pid):
        raise pickle.UnpicklingError("refused persistent id in a reps file")


def pickle_load(path):
    """src/search.py:49-52: (reps, lookup) written by the encode step (src/encode.py:405-410) — read with an unpickler
    that executes nothing from the file: only numpy array reconstruction is allowed, so a reference-written
    corpus*.pkl / query.pkl loads and anything else raises pickle.UnpicklingError."""
    with open(path, "rb") as f:
        obj = _RepsUnpickler(io.BytesIO(f.read())).load()
    if not (isinstance(obj, tuple) and len(obj) == 2):
        raise pickle.UnpicklingError("a reps file holds a (reps, lookup) pair")
    reps, lookup = obj
    reps = np.array(reps)
    if reps.dtype == object:
        raise pickle.UnpicklingError("reps must be a numeric array")
    return reps, lookup
