"""Synthetic workloads of BASELINE.md / SURVEY.md §8d, produced with the synthetic encode step (msr_synth_vectors).

No dataset or checkpoint is reachable here, so the corpus and query vectors have the SHAPE of the reference's encoder
output (src/encode.py:69-75: top-128 vocabulary entries, weights rint(100*log(1+relu(logit)))) with random content.
Retrieval quality is made non-trivial by planting signal: every caption shares a few terms with its image.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .index import synth_vectors
from .qrels import CrossModalQrels


@dataclass
class Workload:
    name: str
    n_terms: int
    docs: tuple        # (ptr uint64, term uint32, weight uint32), doc-major CSR; row i has doc id str(i)
    queries: tuple     # (ptr int64, term int32, weight int32)
    k: int
    qrels: CrossModalQrels | None = None
    query_type: str = "text"
    description: str = ""


def planted_captions(doc_csr, n_terms, captions_per_image, nnz_lo, nnz_hi, n_shared, seed, threads=16,
                     first_image=0, n_images=None):
    """Caption queries for images [first_image, first_image + n_images): `n_shared` terms copied from the image
    (with fresh weights) plus Zipf-drawn filler up to a length uniform in [nnz_lo, nnz_hi]."""
    dp, dt, _ = doc_csr
    n_docs = len(dp) - 1
    if n_images is None:
        n_images = n_docs - first_image
    nq = n_images * captions_per_image
    rng = np.random.default_rng(seed)
    img = first_image + np.arange(nq) // captions_per_image
    doc_nnz = int(dp[1] - dp[0])
    assert (np.diff(dp.astype(np.int64)) == doc_nnz).all(), "planted_captions expects fixed-length doc rows"
    # shared terms: n_shared distinct positions of the image's row
    pos = np.argsort(rng.random((nq, doc_nnz)), axis=1)[:, :n_shared]
    shared = dt.reshape(n_docs, doc_nnz)[img[:, None], pos].astype(np.int32)
    lens = rng.integers(nnz_lo, nnz_hi + 1, size=nq)
    fill_w = nnz_hi - n_shared
    _, ft, fw = synth_vectors(nq, max(fill_w, 1), n_terms, seed=seed * 7919 + 13, threads=threads)
    ft = ft.reshape(nq, -1).astype(np.int32)
    fw = fw.reshape(nq, -1).astype(np.int32)
    sw = np.clip(np.rint(100.0 * np.log1p(np.exp(0.5 + 0.6 * rng.standard_normal((nq, n_shared))))), 1, 400).astype(np.int32)
    terms = np.concatenate([shared, ft], axis=1)
    weights = np.concatenate([sw, fw], axis=1)
    keep = np.arange(terms.shape[1])[None, :] < lens[:, None]
    q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return q_ptr, terms[keep].astype(np.int32), weights[keep].astype(np.int32)


def flickr30k_t2i(n_images=31014, captions_per_image=5, n_terms=32064, seed=1, threads=16, query_images=None,
                  first_query_image=0):
    """BASELINE config 1/2: Flickr30K text->image. docs = images (128 nnz), queries = captions (8-15 nnz,
    text w/o --sparse_manual, src/encode.py:128-129), V = 32 064 (vicuna vocabulary), top-10."""
    docs = synth_vectors(n_images, 128, n_terms, seed=seed, threads=threads)
    if query_images is None:
        query_images = n_images
    q = planted_captions(docs, n_terms, captions_per_image, 8, 15, 6, seed + 1, threads, first_query_image, query_images)
    qrels = CrossModalQrels()
    for j in range(query_images * captions_per_image):
        qrels.add(first_query_image + j // captions_per_image, first_query_image * captions_per_image + j)
    return Workload("flickr30k_t2i", n_terms, docs, q, 10, qrels, "text",
                    f"Flickr30K-shape text->image: {n_images} docs x128 nnz, {query_images * captions_per_image} "
                    f"queries x8-15 nnz, V={n_terms}, top-10, synthetic Zipf(0.8) vectors with planted signal")


def coco5k(direction="i2t", n_terms=30000, seed=2, threads=16):
    """BASELINE config 3: COCO-5K, ~120-nnz queries, V = 30 000."""
    if direction == "i2t":  # queries = 5 000 images, docs = 25 010 captions
        docs = synth_vectors(25010, 128, n_terms, seed=seed, threads=threads)
        qp, qt, qw = synth_vectors(5000, 120, n_terms, seed=seed + 1, threads=threads)
    else:
        docs = synth_vectors(5000, 128, n_terms, seed=seed, threads=threads)
        qp, qt, qw = synth_vectors(25010, 120, n_terms, seed=seed + 1, threads=threads)
    return Workload(f"coco5k_{direction}", n_terms, docs, (qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)),
                    10, None, "image" if direction == "i2t" else "text",
                    f"COCO-5K-shape {direction}: {len(docs[0]) - 1} docs x128 nnz, {len(qp) - 1} queries x120 nnz, V={n_terms}")


def c4_1m(n_docs=1_000_000, n_queries=10_000, n_terms=30000, seed=3, threads=16):
    """BASELINE config 4: 1 M docs x 128 nnz (128 M postings), 10 000 queries x 120 nnz, V = 30 000, top-10."""
    docs = synth_vectors(n_docs, 128, n_terms, seed=seed, threads=threads)
    qp, qt, qw = synth_vectors(n_queries, 120, n_terms, seed=seed + 1, threads=threads)
    return Workload("c4_1m", n_terms, docs, (qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)), 10, None,
                    "text", f"synthetic {n_docs} docs x128 nnz, {n_queries} queries x120 nnz, V={n_terms}, top-10")


def hybrid_vectors(n_docs, n_queries, h=4096, n_terms=30000, seed=4, threads=16):
    """BASELINE config 5 and the reference's own hybrid run (scripts/search.sh:16-33): docs and queries with a sparse
    (128 / 120 nnz) and a dense (h-d, unit norm: src/encode.py:301, src/search.py:342) vector each.
    -> (doc CSR, query CSR, passage matrix f32 [n_docs, h], query matrix f32 [n_queries, h])"""
    docs = synth_vectors(n_docs, 128, n_terms, seed=seed, threads=threads)
    qp, qt, qw = synth_vectors(n_queries, 120, n_terms, seed=seed + 1, threads=threads)
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((n_docs, h), dtype=np.float32)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    q = rng.standard_normal((n_queries, h), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return docs, (qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)), p, q


def recall_at(ords, n, docids_of_ord, qrels, query_ids, query_type, ks=(1, 5, 10)):
    """Recall@k over result arrays (any-target-in-top-k, src/metrices.py:76-84) without building run dicts."""
    hits = {k: 0 for k in ks}
    for i, qid in enumerate(query_ids):
        t = qrels.get_target(qid, query_type)
        tset = set(t) if isinstance(t, list) else {t}
        ranked = [docids_of_ord[int(o)] for o in ords[i, : int(n[i])]]
        for k in ks:
            if tset.intersection(ranked[:k]):
                hits[k] += 1
    return {k: hits[k] / max(len(query_ids), 1) for k in ks}
