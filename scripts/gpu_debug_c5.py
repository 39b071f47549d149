"""debug: near-tie swaps of the fused hybrid path vs the oracle pipeline (prints the first few)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m
from mllm_sparse_retrieval_amd.dense import DenseIndex, hybrid_search, row_to_ordinal
from tests import helpers

n, nq, h, depth, k, alpha, n_terms = 5000, 2000, 4096, 1000, 10, 0.5, 30000
docs = m.synth_vectors(n, 128, n_terms, seed=4, threads=16)
qp, qt, qw = m.synth_vectors(nq, 120, n_terms, seed=5, threads=16)
qp, qt, qw = qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)
rng = np.random.default_rng(4)
p = rng.standard_normal((n, h), dtype=np.float32); p /= np.linalg.norm(p, axis=1, keepdims=True)
q = rng.standard_normal((nq, h), dtype=np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
ids = [str(i) for i in range(n)]
tmp = tempfile.mkdtemp()
path = m.build_index_from_csr(os.path.join(tmp, "c5.idx"), *docs, n_terms, doc_ids=ids)
ix = m.SparseIndex(path, device=0); dix = DenseIndex(p)
r2o = row_to_ordinal(ix, ids)
ords, fs, cnt, ms = hybrid_search(ix, dix, qp, qt, qw, q, depth, k, alpha, r2o)
sample = np.arange(0, nq, 10)
want, sq = helpers.oracle_hybrid(docs, n_terms, ids, qp, qt, qw, q, p, depth, alpha, sample)
shown = 0
for j, i in enumerate(sample):
    ranked = sorted(want[sq[j]].items(), key=lambda kv: (-float(kv[1]), kv[0].encode()))[:k]
    got = [ix.docid(int(o)) for o in ords[i, :cnt[i]]]
    if got != [d for d, _ in ranked] and shown < 5:
        shown += 1
        print("query", i)
        for r in range(k):
            print("  ", r, "gpu", got[r], f"{fs[i, r]:.9f}", "| oracle", ranked[r][0], f"{ranked[r][1]:.9f}", "| oracle score of gpu doc", f"{want[sq[j]].get(got[r], float('nan')):.9f}")
print("done; swapped queries shown:", shown)
