"""Shared test helpers: seeded workloads, an independent reader of the index file, oracle adapters."""
from __future__ import annotations

import struct

import numpy as np

SECTIONS = ["term_off", "term_str", "term_sorted", "df", "maxw", "doc_off", "doc_str", "seg_ptr", "postings",
            "dense_terms", "dense"]


def read_index_file(path):
    """Decode an MSRIDX01 file independently of libmsr.so (layout: csrc/msr_internal.h)."""
    raw = np.fromfile(path, dtype=np.uint8)
    b = raw.tobytes()
    magic = b[:8]
    assert magic == b"MSRIDX01", magic
    version, tile_docs = struct.unpack_from("<II", b, 8)
    n_docs, n_postings, n_vecs = struct.unpack_from("<QQQ", b, 16)
    n_terms, n_tiles, max_weight, n_dense = struct.unpack_from("<IIII", b, 40)
    ns = len(SECTIONS)
    off = struct.unpack_from(f"<{ns}Q", b, 56)
    size = struct.unpack_from(f"<{ns}Q", b, 56 + 8 * ns)
    (file_size,) = struct.unpack_from("<Q", b, 56 + 16 * ns)
    assert file_size == len(b)
    sec = {name: b[off[i] : off[i] + size[i]] for i, name in enumerate(SECTIONS)}
    term_off = np.frombuffer(sec["term_off"], dtype=np.uint64)
    doc_off = np.frombuffer(sec["doc_off"], dtype=np.uint64)
    terms = [sec["term_str"][int(term_off[i]) : int(term_off[i + 1]) - 1].decode("utf-8", "surrogateescape")
             for i in range(n_terms)]
    docs = [sec["doc_str"][int(doc_off[i]) : int(doc_off[i + 1]) - 1].decode("utf-8", "surrogateescape")
            for i in range(n_docs)]
    seg_ptr = np.frombuffer(sec["seg_ptr"], dtype=np.uint32).reshape(n_tiles, n_terms + 1)
    postings = np.frombuffer(sec["postings"], dtype=np.uint32)
    return dict(version=version, tile_docs=tile_docs, n_docs=n_docs, n_postings=n_postings, n_vecs=n_vecs,
                n_terms=n_terms, n_tiles=n_tiles, max_weight=max_weight, terms=terms, docs=docs, n_dense=n_dense,
                dense_terms=np.frombuffer(sec["dense_terms"], dtype=np.uint32),
                dense=np.frombuffer(sec["dense"], dtype=np.uint32).reshape(n_tiles, n_dense // 2, tile_docs),
                term_sorted=np.frombuffer(sec["term_sorted"], dtype=np.uint32),
                df=np.frombuffer(sec["df"], dtype=np.uint32), maxw=np.frombuffer(sec["maxw"], dtype=np.uint32),
                seg_ptr=seg_ptr, postings=postings)


def index_file_to_dense(ix):
    """Tile-major postings -> dense [n_docs, n_terms] int64 matrix (small indexes only)."""
    D = np.zeros((ix["n_docs"], ix["n_terms"]), dtype=np.int64)
    for tile in range(ix["n_tiles"]):
        for t in range(ix["n_terms"]):
            a, b = int(ix["seg_ptr"][tile, t]) * 4, int(ix["seg_ptr"][tile, t + 1]) * 4
            seg = ix["postings"][a:b]
            # a segment is a run of chunks (256 postings = 64 vecs); the order of postings INSIDE a chunk is free (the
            # builder arranges them by LDS bank, csrc/msr_index.cpp pass C), chunks follow each other in ordinal order
            real = []
            prev_max = -1
            for c0 in range(0, len(seg), 256):
                flat = seg[c0 : c0 + 256]
                nz = np.sort(flat[(flat >> 16) != 0] & 0xFFFF)  # weight 0 = padding
                assert len(flat) - len(nz) <= 3 or c0 + 256 >= len(seg), "only the last chunk is partial"
                assert len(nz) and nz[0] > prev_max, "chunks partition the segment in ordinal order"
                prev_max = int(nz[-1])
                real.append(flat[(flat >> 16) != 0])
            real = np.concatenate(real) if real else np.zeros(0, np.uint32)
            loc = (real & 0xFFFF).astype(np.int64)
            assert len(np.unique(loc)) == len(loc), "one posting per doc"
            D[tile * ix["tile_docs"] + loc, t] += (real >> 16).astype(np.int64)
    # dense head: slot s of pair s // 2 holds the weights of term dense_terms[s]; those terms have no segments
    for s, t in enumerate(ix["dense_terms"]):
        if t == 0xFFFFFFFF:
            assert not ((ix["dense"][:, s // 2, :] >> (16 * (s & 1))) & 0xFFFF).any()
            continue
        assert not D[:, t].any(), "a dense-head term must not have inverted lists"
        w = ((ix["dense"][:, s // 2, :] >> (16 * (s & 1))) & 0xFFFF).reshape(-1)[: ix["n_docs"]]
        D[:, t] = w.astype(np.int64)
    return D


def synth(n_docs, doc_nnz, n_queries, q_nnz, n_terms, seed):
    import mllm_sparse_retrieval_amd as m

    dp, dt, dw = m.synth_vectors(n_docs, doc_nnz, n_terms, seed=seed, threads=8)
    qp, qt, qw = m.synth_vectors(n_queries, q_nnz, n_terms, seed=seed + 1000003, threads=8)
    return (dp, dt, dw), (qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32))


def taat_oracle(docs, n_terms, doc_ids=None):
    from oracle import taat

    dp, dt, dw = docs
    ix, order = taat.TaatIndex.from_rows_by_docid(dp, dt, dw, n_terms, doc_ids)
    return ix, order


def assert_same_results(got, want, k):
    """got = (ords u32, f32 scores, u32 scores, n) from libmsr; want = (ords i64 / -1 pad, scores i64, n)."""
    g_ord, g_f32, g_u32, g_n = got
    w_ord, w_sc, w_n = want
    assert (g_n == w_n).all(), np.flatnonzero(g_n != w_n)[:10]
    mask = np.arange(k)[None, :] < w_n[:, None]
    assert (g_ord.astype(np.int64)[mask] == w_ord[mask]).all(), "doc ordinals differ"
    assert (g_u32.astype(np.int64)[mask] == w_sc[mask]).all(), "exact scores differ"
    # f32 scores: |delta| <= 1e-5 (north_star); exact below 2^24
    assert np.abs(g_f32.astype(np.float64)[mask] - w_sc.astype(np.float64)[mask]
                  ).max(initial=0) <= np.where(w_sc.max(initial=0) < 2**24, 1e-5, np.inf)
    assert (g_u32[~mask] == 0).all()


def dense_oracle(q, p, k, tie_key=None):
    """numpy f32 inner products of the fp16-ROUNDED inputs (the storage precision of the reference's GPU faiss,
    src/search.py:257), ranked by (-score, row) — or by (-score, tie_key[row]) when a tie key is given (the fused
    hybrid kernel breaks exact dense ties by doc ordinal: DESIGN.md §5). Tolerance on scores: 1e-5 (f32 accumulation order)."""
    s = q.astype(np.float16).astype(np.float32) @ p.astype(np.float16).astype(np.float32).T
    tk = np.arange(s.shape[1]) if tie_key is None else np.asarray(tie_key, dtype=np.int64)
    order = np.lexsort((np.broadcast_to(tk, s.shape), -s), axis=1)[:, :k]
    return np.take_along_axis(s, order, axis=1), order


def oracle_hybrid(docs, n_terms, ids, qp, qt, qw, q, p, depth, alpha, sample, remove_query=False, qids=None,
                  dense_tie_key=None):
    """The reference's hybrid pipeline driven by the oracles for the queries in `sample`: C oracle sparse top-depth +
    numpy dense top-depth -> oracle.get_run_dict -> oracle.fuse (pinned to src/hybrid.py:32-53).
    -> ({qid: {doc: fused}}, qids of the sample)"""
    from oracle import oracle

    oix, order = taat_oracle(docs, n_terms, ids)
    sorted_ids = [ids[r] for r in order]
    sample = np.asarray(sample)
    sel = np.concatenate([np.arange(qp[i], qp[i + 1]) for i in sample]) if len(sample) else np.zeros(0, np.int64)
    sp = np.concatenate([[0], np.cumsum(qp[sample + 1] - qp[sample])]).astype(np.int64)
    wo, wsc, wn = oix.search(sp, qt[sel], qw[sel], depth, threads=16)
    sq = [str(int(i)) for i in sample] if qids is None else [qids[int(i)] for i in sample]
    o_sparse = oracle.get_run_dict(sq, [[float(np.float32(x)) for x in wsc[j, :wn[j]]] for j in range(len(sample))],
                                   [[sorted_ids[int(d)] for d in wo[j, :wn[j]]] for j in range(len(sample))], remove_query)
    dsc, didx = dense_oracle(q[sample], p, min(depth, p.shape[0]), dense_tie_key)
    o_dense = oracle.get_run_dict(sq, dsc, np.array([[ids[j] for j in row] for row in didx]), remove_query)
    return oracle.fuse([o_dense, o_sparse], [alpha, 1 - alpha]), sq


def assert_hybrid_matches(want, sq, sample, ords, fs, cnt, docid_of, k, tol=1e-5, tie=2e-6):
    worst = 0.0
    for j, i in enumerate(sample):
        ranked = sorted(want[sq[j]].items(), key=lambda kv: (-float(kv[1]), kv[0].encode()))[:k]
        assert cnt[i] == len(ranked), (sq[j], cnt[i], len(ranked))
        for r, (doc, score) in enumerate(ranked):
            worst = max(worst, abs(float(fs[i, r]) - float(score)))
            g = docid_of(int(ords[i, r]))
            if g != doc:  # only a near-tie in the fused score may swap neighbours
                assert g in want[sq[j]] and abs(float(want[sq[j]][g]) - float(score)) <= tie, (sq[j], r, g, doc)
    assert worst <= tol, worst
