"""Host-side handle classes over the C-ABI: index build / open, CSR search, device-resident query batches.

Index build replaces scripts/sparse_index.sh:12-18 (pyserini.index.lucene --impact --pretokenized); opening
replaces LuceneImpactSearcher(index_dir, None) (src/search.py:273); searching replaces batch_search
(src/search.py:86-87). The directory layout of the reference is kept: the index of `<dir>` lives at `<dir>/index`
(src/search.py:273 joins 'index'), here as the single file `<dir>/index/msr.idx`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _cabi
from ._cabi import MSR_F_DROP_DF_EQ_N, MsrInfo, check, lib, ptr

INDEX_FILE = "msr.idx"


def index_file_of(path):
    """Accept the index file itself, the `index/` directory, or the encode directory that holds `index/`."""
    if os.path.isfile(path):
        return path
    for cand in (os.path.join(path, INDEX_FILE), os.path.join(path, "index", INDEX_FILE)):
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError(f"no {INDEX_FILE} under '{path}' (build it with the `index` command first)")


def auto_tile_docs(n_docs):
    """4096-doc tiles for tiny corpora, else 8192: the tile whose 32 KiB of accumulators lets four workgroups
    (32 waves) share a CU, which measured fastest on MI355X (profiles/)."""
    return 4096 if n_docs <= 4096 else 8192


def set_build_option(key, value):
    """'dense_min_density' (default 0.4) / 'dense_max_terms' (default 16, 0 = no dense head) / 'tie_order' (0: ties go
    to the lower doc-id string, contract T1; 1: to the doc that came first in the input) for later builds."""
    check(lib().msr_set_build_option(key.encode(), float(value)))


def build_index_from_jsonl(jsonl_dir, out_file=None, threads=16, tile_docs=0):
    """corpus_*.jsonl under `jsonl_dir` -> `<jsonl_dir>/index/msr.idx` (scripts/sparse_index.sh:14-15 layout)."""
    if out_file is None:
        os.makedirs(os.path.join(jsonl_dir, "index"), exist_ok=True)
        out_file = os.path.join(jsonl_dir, "index", INDEX_FILE)
    if tile_docs == 0:
        # count lines cheaply to pick the tile; the builder re-validates everything
        n = 0
        for fn in sorted(os.listdir(jsonl_dir)):
            if fn.endswith((".jsonl", ".json")) and os.path.isfile(os.path.join(jsonl_dir, fn)):
                with open(os.path.join(jsonl_dir, fn), "rb") as f:
                    n += sum(1 for line in f if line.strip())
        tile_docs = auto_tile_docs(n)
    check(lib().msr_index_build(os.fsencode(jsonl_dir), os.fsencode(out_file), int(threads), int(tile_docs)))
    return out_file


def build_index_from_csr(out_file, doc_ptr, term, weight, n_terms, doc_ids=None, term_strs=None, threads=16,
                         tile_docs=0):
    """Doc-major CSR (numeric term ids) -> index file. doc_ids / term_strs default to decimal numbers."""
    doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.uint64)
    term = np.ascontiguousarray(term, dtype=np.uint32)
    weight = np.ascontiguousarray(weight, dtype=np.uint32)
    n_docs = len(doc_ptr) - 1
    if tile_docs == 0:
        tile_docs = auto_tile_docs(n_docs)
    ids_arr = keep1 = terms_arr = keep2 = None
    if doc_ids is not None:
        if len(doc_ids) != n_docs:
            raise ValueError("doc_ids length mismatch")
        ids_arr, keep1 = _cabi.c_str_array(doc_ids)
    if term_strs is not None:
        if len(term_strs) != n_terms:
            raise ValueError("term_strs length mismatch")
        terms_arr, keep2 = _cabi.c_str_array(term_strs)
    os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)
    check(lib().msr_index_build_csr(
        os.fsencode(out_file), n_docs, int(n_terms), ptr(doc_ptr), ptr(term), ptr(weight),
        C.cast(ids_arr, C.c_void_p) if ids_arr is not None else None,
        C.cast(terms_arr, C.c_void_p) if terms_arr is not None else None, int(threads), int(tile_docs)))
    del keep1, keep2
    return out_file


class SparseIndex:
    """An opened index. device >= 0: postings resident in that GPU's HBM; device = -1: metadata only (cannot search)."""

    def __init__(self, path, device=0, shard=0, n_shards=1, term_shard=None):
        """shard / n_shards: doc-range shard (contiguous tiles). term_shard=(g, G): term-range shard g of G — every
        doc tile, but only the postings of the owned term range resident (serves the term-range protocol only)."""
        self._h = C.c_void_p()
        self.path = index_file_of(path)
        self.term_shard = None if term_shard is None else (int(term_shard[0]), int(term_shard[1]))
        if term_shard is not None:
            check(lib().msr_index_open_termshard(os.fsencode(self.path), int(device), self.term_shard[0],
                                                 self.term_shard[1], C.byref(self._h)))
        elif n_shards == 1:
            check(lib().msr_index_open(os.fsencode(self.path), int(device), C.byref(self._h)))
        else:
            check(lib().msr_index_open_shard(os.fsencode(self.path), int(device), int(shard), int(n_shards),
                                             C.byref(self._h)))
        info = MsrInfo()
        check(lib().msr_index_info(self._h, C.byref(info)))
        self.info = info
        self._small = {}                       # (nq, k) -> result buffers of small search_csr calls
        self._search_csr = lib().msr_search_csr
        self.n_docs = int(info.n_docs)
        self.n_terms = int(info.n_terms)
        self.n_postings = int(info.n_postings)
        self.tile_docs = int(info.tile_docs)
        self.n_tiles = int(info.n_tiles)
        self.device = int(info.device)
        self.shard_tile0 = int(info.shard_tile0)
        self.shard_ntiles = int(info.shard_ntiles)
        self.n_dense = int(info.n_dense)
        self.term_lo, self.term_hi = int(info.term_lo), int(info.term_hi)
        self.resident_bytes = int(info.resident_bytes)

    # ---- metadata
    def close(self):
        if self._h:
            lib().msr_index_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def lookup(self, tokens):
        """tokens -> int32 term ids, -1 for tokens outside the index vocabulary (contract T2)."""
        n = len(tokens)
        out = np.empty(n, dtype=np.int32)
        if n:
            arr, keep = _cabi.c_str_array(tokens)
            check(lib().msr_term_lookup(self._h, C.cast(arr, C.c_void_p), n, ptr(out)))
            del keep
        return out

    def df(self, term_ids):
        term_ids = np.ascontiguousarray(term_ids, dtype=np.int32)
        out = np.empty(len(term_ids), dtype=np.uint32)
        if len(term_ids):
            check(lib().msr_term_df(self._h, ptr(term_ids), len(term_ids), ptr(out)))
        return out

    def term_str(self, term_id):
        s = C.c_char_p()
        check(lib().msr_term_str(self._h, int(term_id), C.byref(s)))
        return s.value.decode("utf-8", "surrogateescape")

    def docid(self, ordinal):
        s = C.c_char_p()
        check(lib().msr_docid_str(self._h, int(ordinal), C.byref(s)))
        return s.value.decode("utf-8", "surrogateescape")

    def docid_table(self):
        """All external doc ids by ordinal as a numpy object array (built once; one C call per doc)."""
        if getattr(self, "_docid_table", None) is None:
            self._docid_table = np.array([self.docid(o) for o in range(self.n_docs)], dtype=object)
        return self._docid_table

    def docids(self, ordinals):
        ordinals = np.asarray(ordinals, dtype=np.int64)
        if self.n_docs <= 4_000_000 and (ordinals.size > 64 or getattr(self, "_docid_table", None) is not None):
            return self.docid_table()[ordinals].tolist()
        return [self.docid(o) for o in ordinals]

    # ---- search
    def search_csr(self, q_ptr, q_term, q_w, k, drop_df_eq_n=True):
        """CSR queries -> (ordinals [nq,k] uint32, scores [nq,k] float32, exact scores [nq,k] uint32, n [nq])."""
        q_ptr, q_term, q_w = _cabi.as_csr(q_ptr, q_term, q_w)
        nq = len(q_ptr) - 1
        flags = MSR_F_DROP_DF_EQ_N if drop_df_eq_n else 0
        if nq * k <= 1024:
            # the reference's call shape (4 queries per batch_search, scripts/search_sparse.sh:16): result buffers and
            # their addresses are kept per (nq, k) — allocating four arrays and taking seven addresses costs such a call
            # a fifth of its time — and the caller gets copies
            bufs = self._small.get((nq, k))
            if bufs is None:
                arrs = (np.empty((nq, k), np.uint32), np.empty((nq, k), np.float32), np.empty((nq, k), np.uint32),
                        np.zeros(nq, np.int32))
                bufs = self._small[(nq, k)] = (arrs, tuple(a.ctypes.data for a in arrs))
            arrs, addr = bufs
            check(self._search_csr(self._h, q_ptr.ctypes.data, q_term.ctypes.data, q_w.ctypes.data, nq, int(k), flags, *addr))
            return tuple(a.copy() for a in arrs)
        ords = np.empty((nq, k), dtype=np.uint32)
        sc = np.empty((nq, k), dtype=np.float32)
        su = np.empty((nq, k), dtype=np.uint32)
        n = np.zeros(nq, dtype=np.int32)
        check(lib().msr_search_csr(self._h, ptr(q_ptr), ptr(q_term), ptr(q_w), nq, int(k), flags, ptr(ords), ptr(sc),
                                   ptr(su), ptr(n)))
        return ords, sc, su, n

    def search_text(self, queries, k, drop_df_eq_n=True):
        """Query strings (tokens repeated `weight` times, src/search.py:419-422) -> the same outputs as search_csr;
        tokenisation, counting and dictionary lookup run in C inside the call."""
        nq = len(queries)
        arr, keep = _cabi.c_str_array(queries)
        ords = np.empty((nq, k), dtype=np.uint32)
        sc = np.empty((nq, k), dtype=np.float32)
        su = np.empty((nq, k), dtype=np.uint32)
        n = np.zeros(nq, dtype=np.int32)
        flags = MSR_F_DROP_DF_EQ_N if drop_df_eq_n else 0
        check(lib().msr_search_text(self._h, C.cast(arr, C.c_void_p), nq, int(k), flags, ptr(ords), ptr(sc), ptr(su), ptr(n)))
        del keep
        return ords, sc, su, n

    def encode_queries(self, queries):
        """Query strings (src/search.py:419-422) -> CSR (q_ptr int64, term ids int32, counts int32), tokenised and looked
        up in C on several threads; out-of-vocabulary tokens are dropped (contract T2)."""
        nq = len(queries)
        arr, keep = _cabi.c_str_array(queries)
        q_ptr = np.zeros(nq + 1, dtype=np.int64)
        need = C.c_int64()
        check(lib().msr_encode_queries(self._h, C.cast(arr, C.c_void_p), nq, ptr(q_ptr), None, None, 0, C.byref(need)))
        q_term = np.empty(need.value, dtype=np.int32)
        q_w = np.empty(need.value, dtype=np.int32)
        if need.value:
            check(lib().msr_encode_queries(self._h, C.cast(arr, C.c_void_p), nq, ptr(q_ptr), ptr(q_term), ptr(q_w),
                                           need.value, C.byref(need)))
        del keep
        return q_ptr, q_term, q_w

    def batch(self, q_ptr, q_term, q_w, kmax, drop_df_eq_n=True, term_shard=None):
        if term_shard is None:
            term_shard = self.term_shard  # a term-shard handle only takes batches of its own term range
        return QueryBatch(self, q_ptr, q_term, q_w, kmax, drop_df_eq_n, term_shard)

    def search_termshard_emulated(self, q_ptr, q_term, q_w, k, n_shards, drop_df_eq_n=True):
        """Term-range sharded search protocol for `n_shards` logical shards on this one GPU."""
        q_ptr, q_term, q_w = _cabi.as_csr(q_ptr, q_term, q_w)
        nq = len(q_ptr) - 1
        ords = np.empty((nq, k), dtype=np.uint32)
        sc = np.empty((nq, k), dtype=np.float32)
        su = np.empty((nq, k), dtype=np.uint32)
        n = np.zeros(nq, dtype=np.int32)
        flags = MSR_F_DROP_DF_EQ_N if drop_df_eq_n else 0
        check(lib().msr_search_termshard_emulated(self._h, ptr(q_ptr), ptr(q_term), ptr(q_w), nq, int(k), flags,
                                                  int(n_shards), ptr(ords), ptr(sc), ptr(su), ptr(n)))
        return ords, sc, su, n

    def comm_info(self):
        """(ranks, rank, device) as the RCCL communicator of this handle reports them."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().msr_comm_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def merge_lists(self, ords, scores_u32, n, k):
        """Exact top-k merge (same tie rule) of `L` per-shard lists: ords/scores [L,nq,k], n [L,nq]."""
        ords = np.ascontiguousarray(ords, dtype=np.uint32)
        scores_u32 = np.ascontiguousarray(scores_u32, dtype=np.uint32)
        n = np.ascontiguousarray(n, dtype=np.int32)
        L, nq = n.shape
        o = np.empty((nq, k), dtype=np.uint32)
        sf = np.empty((nq, k), dtype=np.float32)
        su = np.empty((nq, k), dtype=np.uint32)
        on = np.zeros(nq, dtype=np.int32)
        check(lib().msr_merge_lists(self._h, L, nq, int(k), ptr(ords), ptr(scores_u32), ptr(n), ptr(o), ptr(sf), ptr(su),
                                    ptr(on)))
        return o, sf, su, on

    # ---- exchange (doc-range shards)
    def comm_init(self, n_ranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), _cabi.MSR_COMM_ID_BYTES)
        check(lib().msr_comm_init(self._h, int(n_ranks), int(rank), buf))

    def comm_destroy(self):
        check(lib().msr_comm_destroy(self._h))


def search_termshard_emulated_handles(shards, q_ptr, q_term, q_w, k, drop_df_eq_n=True):
    """The term-range protocol over `shards` = [SparseIndex(path, term_shard=(g, G)) for g in range(G)] on one GPU."""
    q_ptr, q_term, q_w = _cabi.as_csr(q_ptr, q_term, q_w)
    nq = len(q_ptr) - 1
    ords = np.empty((nq, k), dtype=np.uint32)
    sc = np.empty((nq, k), dtype=np.float32)
    su = np.empty((nq, k), dtype=np.uint32)
    n = np.zeros(nq, dtype=np.int32)
    hs = (C.c_void_p * len(shards))(*[s._h for s in shards])
    check(lib().msr_search_termshard_emulated_handles(hs, len(shards), ptr(q_ptr), ptr(q_term), ptr(q_w), nq, int(k),
                                                      MSR_F_DROP_DF_EQ_N if drop_df_eq_n else 0, ptr(ords), ptr(sc),
                                                      ptr(su), ptr(n)))
    return ords, sc, su, n


def runtime_info():
    """{'hip_runtime', 'hip_lib', 'rccl', 'rccl_header', 'rccl_lib'} of the libraries libmsr.so resolved in this process."""
    buf = C.create_string_buffer(2048)
    check(lib().msr_runtime_info(buf, 2048))
    return dict(kv.split("=", 1) for kv in buf.value.decode().split())


def search_laps():
    """Host-side laps (microseconds) of this thread's last msr_search_csr call: see include/msr.h."""
    out = np.zeros(8, dtype=np.float64)
    check(lib().msr_search_laps(ptr(out)))
    names = ("prepare_upload", "enqueue_kernels", "wait_stream", "download", "release", "call_total", "score_kernel",
             "merge_kernel")
    return dict(zip(names, (round(float(v), 2) for v in out)))


def device_sync(device):
    check(lib().msr_device_sync(int(device)))


def device_copy_gbs(device, nbytes=1 << 30, reps=8):
    """Measured bandwidth of a device-to-device copy (read + write bytes / time, GB/s)."""
    g = C.c_double()
    check(lib().msr_device_copy_gbs(int(device), int(nbytes), int(reps), C.byref(g)))
    return g.value


def device_peak_rates(device):
    """Measured pipe peaks at the scoring kernel's launch shape: {'ds_add_per_s', 'dot2_per_s', 'lds_bytes_per_s', 'cus'}."""
    out = np.zeros(4, dtype=np.float64)
    check(lib().msr_device_peak_rates(int(device), ptr(out)))
    return dict(ds_add_per_s=float(out[0]), dot2_per_s=float(out[1]), lds_bytes_per_s=float(out[2]), cus=int(out[3]))


def comm_unique_id():
    buf = C.create_string_buffer(_cabi.MSR_COMM_ID_BYTES)
    check(lib().msr_comm_unique_id(buf))
    return bytes(buf.raw)


class QueryBatch:
    """Queries uploaded once; searches run on the index's HIP stream with inputs and outputs resident in HBM."""

    def __init__(self, index, q_ptr, q_term, q_w, kmax, drop_df_eq_n=True, term_shard=None):
        q_ptr, q_term, q_w = _cabi.as_csr(q_ptr, q_term, q_w)
        self.index = index
        self.nq = len(q_ptr) - 1
        self.kmax = int(kmax)
        self._h = C.c_void_p()
        flags = MSR_F_DROP_DF_EQ_N if drop_df_eq_n else 0
        if term_shard is None:
            check(lib().msr_batch_create(index._h, ptr(q_ptr), ptr(q_term), ptr(q_w), self.nq, self.kmax, flags,
                                         C.byref(self._h)))
        else:  # (shard, n_shards): keep only the query terms of that term range
            check(lib().msr_batch_create_termshard(index._h, ptr(q_ptr), ptr(q_term), ptr(q_w), self.nq, self.kmax,
                                                   flags, int(term_shard[0]), int(term_shard[1]), C.byref(self._h)))
        self._k = 0

    def search(self, k, sharded=False):
        """Enqueue one search (asynchronous). sharded=True adds the RCCL all-gather + global merge (doc-range shards);
        sharded="terms" runs the term-range protocol (reduce-scatter of accumulators, then the same tail)."""
        if sharded == "terms":
            check(lib().msr_batch_search_termshard(self._h, int(k)))
        elif sharded:
            check(lib().msr_batch_search_sharded(self._h, int(k)))
        else:
            check(lib().msr_batch_search(self._h, int(k)))
        self._k = int(k)

    def sync(self):
        check(lib().msr_batch_sync(self._h))

    def fetch(self):
        k = self._k
        ords = np.empty((self.nq, k), dtype=np.uint32)
        sc = np.empty((self.nq, k), dtype=np.float32)
        su = np.empty((self.nq, k), dtype=np.uint32)
        n = np.zeros(self.nq, dtype=np.int32)
        check(lib().msr_batch_fetch(self._h, ptr(ords), ptr(sc), ptr(su), ptr(n)))
        return ords, sc, su, n

    def kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        check(lib().msr_batch_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing_reset(self):
        check(lib().msr_batch_timing_reset(self._h))

    def timing_sum(self):
        """(calls, scoring-kernel ms, merge-kernel ms) summed over every search since timing_reset()."""
        n, a, b = C.c_int(), C.c_float(), C.c_float()
        check(lib().msr_batch_timing_sum(self._h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value

    def debug_stamps(self):
        out = np.zeros(8, dtype=np.uint64)
        check(lib().msr_batch_debug_stamps(self._h, ptr(out)))
        return out

    def work(self):
        """The work of one search of this batch (msr_batch_work): postings by pipe, workgroups, accumulator LDS bytes."""
        out = np.zeros(6, dtype=np.uint64)
        check(lib().msr_batch_work(self._h, ptr(out)))
        return dict(sparse_postings=int(out[0]), dense_head_postings=int(out[1]), workgroups=int(out[2]),
                    acc_init_bytes=int(out[3]), acc_select_bytes=int(out[4]), query_entries=int(out[5]))

    def algo_bytes(self, k):
        by, po = C.c_uint64(), C.c_uint64()
        check(lib().msr_batch_algo_bytes(self._h, int(k), C.byref(by), C.byref(po)))
        return by.value, po.value

    def close(self):
        if self._h:
            lib().msr_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_vectors(n, nnz, n_terms, zipf_s=0.8, seed=1, threads=16):
    """Synthetic encode step (SURVEY.md §8d generator) -> doc-major CSR (ptr uint64, term uint32, weight uint32)."""
    p = np.empty(n + 1, dtype=np.uint64)
    t = np.empty(n * nnz, dtype=np.uint32)
    w = np.empty(n * nnz, dtype=np.uint32)
    check(lib().msr_synth_vectors(int(n), int(nnz), int(n_terms), float(zipf_s), int(seed), int(threads), ptr(p), ptr(t),
                                  ptr(w)))
    return p, t, w
