import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, tempfile, os
import mllm_sparse_retrieval_amd as m
from mllm_sparse_retrieval_amd import workloads
for direction in ("i2t", "t2i"):
    wl = workloads.coco5k(direction)
    d = tempfile.mkdtemp()
    path = m.build_index_from_csr(os.path.join(d, "c.idx"), *wl.docs, wl.n_terms)
    with m.SparseIndex(path, device=0) as ix:
        qp, qt, qw = wl.queries
        b = ix.batch(qp, qt, qw, 10)
        for _ in range(3): b.search(10)
        b.sync(); b.timing_reset()
        t0 = time.perf_counter()
        for _ in range(10): b.search(10)
        b.sync(); dt = (time.perf_counter() - t0) / 10
        calls, sms, mms = b.timing_sum()
        nq = len(qp) - 1
        print(direction, "docs", ix.n_docs, "queries", nq, "tiles", ix.n_tiles, "ms/step %.3f" % (dt * 1e3), "q/s %.0f" % (nq / dt), "kernel ms %.3f merge %.3f" % (sms / calls, mms / calls))
        b.close()
