"""ctypes front of oracle_taat.c (the C restatement / CPU baseline).  TEST INFRASTRUCTURE ONLY — see oracle.py."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_taat.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.otaat_build.restype = C.c_void_p
        L.otaat_build.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.otaat_free.argtypes = [C.c_void_p]
        L.otaat_search.restype = C.c_int
        L.otaat_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.otaat_search_mode.restype = C.c_int
        L.otaat_search_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.otaat_n_postings.restype = C.c_uint64
        L.otaat_n_postings.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


class TaatIndex:
    """Plain CSR inverted index + exhaustive term-at-a-time scorer. Rows must be given in doc-ORDINAL order."""

    def __init__(self, doc_ptr, term, weight, n_terms):
        self._doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.uint64)
        self._term = np.ascontiguousarray(term, dtype=np.uint32)
        self._weight = np.ascontiguousarray(weight, dtype=np.uint32)
        self.n_docs = len(self._doc_ptr) - 1
        self.n_terms = int(n_terms)
        self._h = lib().otaat_build(self.n_docs, self.n_terms, self._doc_ptr.ctypes.data, self._term.ctypes.data,
                                    self._weight.ctypes.data)
        if not self._h:
            raise MemoryError("otaat_build failed")

    @classmethod
    def from_rows_by_docid(cls, doc_ptr, term, weight, n_terms, doc_ids=None):
        """Permute doc-major rows into ordinal order (doc-id string ascending, T1); returns (index, row_of_ordinal)."""
        doc_ptr = np.asarray(doc_ptr, dtype=np.int64)
        n = len(doc_ptr) - 1
        ids = [str(i) for i in range(n)] if doc_ids is None else list(doc_ids)
        order = np.asarray(sorted(range(n), key=lambda i: ids[i].encode("utf-8")), dtype=np.int64)
        lens = np.diff(doc_ptr)[order]
        new_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        # entry j of the permuted CSR comes from doc_ptr[row] + (j - new_ptr[ordinal])
        starts = np.repeat(doc_ptr[:-1][order], lens)
        within = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(new_ptr[:-1].astype(np.int64), lens)
        gather = starts + within
        return cls(new_ptr, np.asarray(term)[gather], np.asarray(weight)[gather], n_terms), order

    MODES = {"exhaustive": 0, "touched": 1, "maxscore": 2}

    def search(self, q_ptr, q_term, q_w, k, drop_df_eq_n=True, threads=1, mode="exhaustive"):
        """mode: 'exhaustive' (TAAT, full accumulator scan: the checker), 'touched' (TAAT, only touched docs scanned and
        cleared) or 'maxscore' (document-at-a-time MaxScore pruning). All three return the same hits."""
        q_ptr = np.ascontiguousarray(q_ptr, dtype=np.int64)
        q_term = np.ascontiguousarray(q_term, dtype=np.int32)
        q_w = np.ascontiguousarray(q_w, dtype=np.int32)
        nq = len(q_ptr) - 1
        ords = np.empty((nq, k), dtype=np.int64)
        scores = np.empty((nq, k), dtype=np.int64)
        n = np.zeros(nq, dtype=np.int32)
        rc = lib().otaat_search_mode(self._h, q_ptr.ctypes.data, q_term.ctypes.data, q_w.ctypes.data, nq, k,
                                     1 if drop_df_eq_n else 0, int(threads), self.MODES[mode], ords.ctypes.data,
                                     scores.ctypes.data, n.ctypes.data)
        if rc != 0:
            raise OverflowError("otaat_search: score bound exceeds u32 or allocation failed")
        return ords, scores, n

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().otaat_free(self._h)
                self._h = None
        except Exception:
            pass
