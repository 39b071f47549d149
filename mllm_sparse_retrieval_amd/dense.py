"""Dense half of the reference's hybrid search: flat inner product over passage vectors, top-`depth`.

Drop-in for tevatron's FaissFlatSearcher as src/search.py uses it:
    dense_retriever = FaissFlatSearcher(p_reps_0); dense_retriever.add(p_reps_0)        src/search.py:232-237
    all_scores, all_indices = retriever.batch_search(q_reps, depth, batch_size, quiet)  src/search.py:57
    (or retriever.search(q_reps, depth), src/search.py:59)
The reference keeps the vectors in fp16 on the GPU (co.useFloat16, src/search.py:257,267); so does this class, with
f32 accumulation on MFMA (msr_dense_search). Ties in score are broken by the lower row index.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._cabi import check, lib, ptr


def keys_to_f32(keys):
    """Inverse of the order-preserving f32 -> u32 map of the kernels (include/msr.h); key 0 (padding) -> -inf."""
    keys = np.asarray(keys, dtype=np.uint32)
    bits = np.where(keys & np.uint32(0x80000000), keys & np.uint32(0x7FFFFFFF), ~keys)
    out = bits.astype(np.uint32).view(np.float32).copy()
    out[keys == 0] = -np.inf
    return out


def _as_fp16_rows(x, dim=None, holder=None):
    """[rows, dim] matrix -> C-contiguous fp16 (round to nearest even), columns padded to a multiple of 32.
    `holder`: an object whose `_q16` attribute keeps the output buffer between calls (a fresh 200 MB array costs more in
    page faults than the conversion itself) — which makes the holder single-threaded, see DenseIndex.
    msr_f32_to_f16 rounds like numpy's astype(float16) for every non-NaN value (NaN payloads differ: F16C sets the
    quiet bit, numpy keeps the payload)."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("expected a [rows, dim] matrix")
    h = x.shape[1]
    pad = (-h) % 32  # the GEMM steps K by 32; zero columns do not change inner products
    if dim is not None and h != dim:
        raise ValueError(f"dimension mismatch: {h} vs {dim}")
    if x.dtype == np.float32 and x.flags.c_contiguous and x.size >= (1 << 16):
        # (numpy converts ~0.2 G values/s on one core: for a big query matrix that is longer than the GPU search)
        x16 = getattr(holder, "_q16", None) if holder is not None and not pad else None
        if x16 is None or x16.shape != x.shape:
            x16 = np.empty(x.shape, dtype=np.float16)
            if holder is not None and not pad:
                holder._q16 = x16
        check(lib().msr_f32_to_f16(ptr(x), ptr(x16.view(np.uint16)), x.size, 0))
    else:
        x16 = x.astype(np.float16)
    if pad:
        x16 = np.concatenate([x16, np.zeros((x16.shape[0], pad), np.float16)], axis=1)
    return np.ascontiguousarray(x16)


class DenseIndex:
    """fp16 passage matrix resident in HBM.

    NOT thread-safe per handle (like every libmsr handle: one in-flight call per handle): the fp16 copy of the query
    matrix, the device scratch and the HIP stream all live on the handle and are reused by the next call. Threads that
    search concurrently each open their own DenseIndex."""

    def __init__(self, p_reps, device=0):
        p16 = _as_fp16_rows(p_reps)
        self.dim = np.asarray(p_reps).shape[1]
        self.n = p16.shape[0]
        self._h = C.c_void_p()
        check(lib().msr_dense_open(ptr(p16.view(np.uint16)), self.n, p16.shape[1], int(device), C.byref(self._h)))
        self.last_ms = (0.0, 0.0)

    def search(self, q_reps, k):
        """-> (scores float32 [nq,k], indices int64 [nq,k]); rows past the hit count hold (-inf, -1) like faiss."""
        q16 = _as_fp16_rows(q_reps, self.dim, self)   # (consumed by the synchronous call below)
        nq = q16.shape[0]
        idx = np.empty((nq, k), dtype=np.uint32)
        key = np.empty((nq, k), dtype=np.uint32)
        n = np.zeros(nq, dtype=np.int32)
        a, b = C.c_float(), C.c_float()
        check(lib().msr_dense_search(self._h, ptr(q16.view(np.uint16)), nq, int(k), ptr(idx), ptr(key), ptr(n),
                                     C.byref(a), C.byref(b)))
        self.last_ms = (a.value, b.value)
        valid = np.arange(k)[None, :] < n[:, None]
        scores = np.where(valid, keys_to_f32(key), -np.inf).astype(np.float32)
        indices = np.where(valid, idx.astype(np.int64), -1)
        return scores, indices

    def stats(self):
        """{'device_allocs': hipMalloc calls made for this handle's scratch, 'scratch_bytes': what it holds}."""
        out = np.zeros(4, dtype=np.uint64)
        check(lib().msr_dense_stats(self._h, ptr(out)))
        return dict(device_allocs=int(out[0]), scratch_bytes=int(out[1]))

    def close(self):
        if self._h:
            lib().msr_dense_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FaissFlatSearcher:
    """Same call shape as tevatron.retriever.searcher.FaissFlatSearcher (src/search.py:10,232-237,57-59)."""

    def __init__(self, init_reps, device=0):
        self._dim = np.asarray(init_reps).shape[1]
        self._parts = []
        self._index = None
        self._device = device

    def add(self, p_reps):
        p = np.asarray(p_reps, dtype=np.float32)
        if p.shape[1] != self._dim:
            raise ValueError("dimension mismatch")
        self._parts.append(p)
        if self._index is not None:
            self._index.close()
            self._index = None

    def _ready(self):
        if self._index is None:
            if not self._parts:
                raise RuntimeError("FaissFlatSearcher: no vectors were added")
            self._index = DenseIndex(np.concatenate(self._parts, axis=0), device=self._device)
        return self._index

    def search(self, q_reps, k):
        return self._ready().search(q_reps, k)

    def batch_search(self, q_reps, k, batch_size, quiet=False):
        q = np.asarray(q_reps)
        ix = self._ready()
        scores, indices = [], []
        for i in range(0, q.shape[0], max(int(batch_size), 1)):
            s, j = ix.search(q[i:i + batch_size], k)
            scores.append(s)
            indices.append(j)
        return np.concatenate(scores, axis=0), np.concatenate(indices, axis=0)


def row_to_ordinal(sparse_index, lookup):
    """dense row r carries external id lookup[r] (the pkl's lookup list, src/search.py:61) -> sparse doc ordinal."""
    ord_of = {sparse_index.docid(o): o for o in range(sparse_index.n_docs)}
    return np.asarray([ord_of[str(x)] for x in lookup], dtype=np.uint32)


def hybrid_search(sparse_index, dense_index, q_ptr, q_term, q_w, q_reps, depth, k, alpha, row2ord, self_ord=None,
                  drop_df_eq_n=True):
    """Sparse top-`depth` + dense top-`depth` + min-max fusion (src/hybrid.py:32-53, weights [alpha, 1-alpha]) + top-k,
    all on the GPU. -> (doc ordinals uint32 [nq,k], fused scores float32 [nq,k], n [nq], kernel ms per stage)."""
    from . import _cabi
    from ._cabi import MSR_F_DROP_DF_EQ_N

    q_ptr, q_term, q_w = _cabi.as_csr(q_ptr, q_term, q_w)
    q16 = _as_fp16_rows(q_reps, dense_index.dim, dense_index)   # (consumed by the synchronous call below)
    nq = len(q_ptr) - 1
    if q16.shape[0] != nq:
        raise ValueError("sparse and dense query counts differ")
    row2ord = np.ascontiguousarray(row2ord, dtype=np.uint32)
    so = None if self_ord is None else np.ascontiguousarray(self_ord, dtype=np.int32)
    ords = np.empty((nq, k), dtype=np.uint32)
    sc = np.empty((nq, k), dtype=np.float32)
    n = np.zeros(nq, dtype=np.int32)
    ms = (C.c_float * 4)()
    check(lib().msr_hybrid_search(sparse_index._h, dense_index._h, ptr(q_ptr), ptr(q_term), ptr(q_w),
                                  ptr(q16.view(np.uint16)), nq, int(depth), int(k), float(alpha),
                                  MSR_F_DROP_DF_EQ_N if drop_df_eq_n else 0, ptr(row2ord), ptr(so), ptr(ords), ptr(sc),
                                  ptr(n), ms))
    return ords, sc, n, dict(sparse=ms[0], dense_gemm=ms[1], dense_select=ms[2], fusion=ms[3])
