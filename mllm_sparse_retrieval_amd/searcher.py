"""Drop-in for the three pyserini calls the reference makes on its sparse path:

    r = LuceneImpactSearcher(os.path.join(idx, 'index'), None)     src/search.py:273
    r.set_analyzer(JWhiteSpaceAnalyzer())                          src/search.py:274-275
    res = r.batch_search(queries, qids, depth, threads=threads)    src/search.py:86-87
    for hit in res[qid]: hit.score, hit.docid                      src/search.py:96-98

Query handling follows pyserini's behaviour with query_encoder=None as recorded in SURVEY.md §8a A3 ([3P-UNVERIFIED]
there, declared contract here): the query string is split on whitespace and counted (so a token repeated v times has
weight v, src/search.py:419-422), tokens outside the index vocabulary are ignored, tokens present in every doc are
dropped while `min_idf` is 0, and the rest is scored on the GPU by libmsr.so. `threads` is accepted and ignored: the
GPU scores the whole batch at once.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

from .index import SparseIndex


class JWhiteSpaceAnalyzer:
    """Placeholder for pyserini's analyzer object (src/search.py:274): queries are always whitespace-split here."""


class Hit:
    """One search hit: `docid` is the external id string, `score` a Python float holding the f32 score."""

    __slots__ = ("docid", "score")

    def __init__(self, docid, score):
        self.docid = docid
        self.score = score

    def __repr__(self):
        return f"Hit(docid={self.docid!r}, score={self.score})"


def tokenize_queries(queries):
    """list[str] -> (q_ptr int64, tokens list[str], weights int32): token-frequency encoding, first-seen order."""
    ptr = [0]
    toks, ws = [], []
    for q in queries:
        cnt = Counter(q.split())
        toks.extend(cnt.keys())
        ws.extend(cnt.values())
        ptr.append(len(toks))
    return np.asarray(ptr, dtype=np.int64), toks, np.asarray(ws, dtype=np.int32)


class LuceneImpactSearcher:
    """PARITY UNPINNED for the scorer: T1-T5 below restate pyserini / Lucene impact search as recalled (SURVEY.md §8a
    A3), the reference pins none of it and Lucene cannot run here. The two points most likely to differ from a real
    pyserini run are explicit switches, so an integrator can flip them when checking:
      drop_df_eq_n  None: follow min_idf (terms present in EVERY doc are dropped while min_idf >= 0, contract T3);
                    True / False: force the df == N filter on / off.
      tie rule      a build-time property of the index: `set_build_option("tie_order", 1)` before building numbers the
                    docs in input order (a score tie goes to the doc indexed first, like Lucene's internal doc ids
                    under one indexing thread) instead of doc-id string order (contract T1, the default)."""

    def __init__(self, index_dir, query_encoder=None, min_idf=0, device=0, drop_df_eq_n=None):
        if query_encoder is not None:
            raise NotImplementedError("only query_encoder=None (token-frequency queries) is supported, as in the "
                                      "reference (src/search.py:273)")
        self.index = SparseIndex(index_dir, device=device)
        self.min_idf = min_idf
        self.drop_df_eq_n = drop_df_eq_n
        self.num_docs = self.index.n_docs

    def set_analyzer(self, analyzer):
        return None

    def close(self):
        self.index.close()

    def _encode(self, queries):
        q_ptr, toks, ws = tokenize_queries(queries)
        return q_ptr, self.index.lookup(toks), ws

    def search(self, q, k=10):
        return self.batch_search([q], ["_q"], k)["_q"]

    def batch_search(self, queries, qids, k=10, threads=1, fields=None):
        if len(queries) != len(qids):
            raise ValueError("queries and qids differ in length")
        # idf = log(N/df) > min_idf: with the default min_idf = 0 exactly the df == N terms go (contract T3);
        # a positive min_idf is applied here on the host (Python tokenisation), everything else tokenises in C.
        if self.min_idf > 0:
            q_ptr, term_ids, ws = self._encode(queries)
            df = self.index.df(term_ids).astype(np.float64)
            with np.errstate(divide="ignore"):
                idf = np.log(self.index.n_docs / np.maximum(df, 1e-300))
            ws = np.where((df > 0) & (idf > self.min_idf), ws, 0).astype(np.int32)
            drop = True if self.drop_df_eq_n is None else bool(self.drop_df_eq_n)
            ords, scores, _, n = self.index.search_csr(q_ptr, term_ids, ws, k, drop_df_eq_n=drop)
        else:
            drop = self.min_idf >= 0 if self.drop_df_eq_n is None else bool(self.drop_df_eq_n)
            ords, scores, _, n = self.index.search_text(queries, k, drop_df_eq_n=drop)
        out = {}
        table = self.index.docid_table() if len(qids) * k > 64 else None
        score_rows = scores.tolist()
        for i, qid in enumerate(qids):  # duplicate qids collapse, like the qid-keyed map pyserini returns
            cnt = int(n[i])
            ids = table[ords[i, :cnt]].tolist() if table is not None else self.index.docids(ords[i, :cnt])
            row = score_rows[i]
            out[qid] = [Hit(ids[j], row[j]) for j in range(cnt)]
        return out
