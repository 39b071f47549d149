"""Per-rank compute of the doc-range sharded config-4 search, measured on ONE GPU: for G in 1, 2, 4, 8 every shard s of G
is opened by itself (the tiles a rank of an 8-GPU run would hold) and the resident 10 000-query batch is timed on it.
The slowest shard bounds the sharded step from below (the exchange — one all-gather of G x 10 000 x 10 keys and the
merge — comes on top; it cannot be run here). usage: python scripts/gpu_c4_shard_probe.py"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd import workloads  # noqa: E402


def timed(ix, q, k, steps=4):
    b = ix.batch(*q, k)
    b.search(k)
    b.sync()
    b.timing_reset()
    for _ in range(steps):
        b.search(k)
    b.sync()
    calls, score_ms, merge_ms = b.timing_sum()
    b.close()
    return score_ms / calls, merge_ms / calls


def main():
    wl = workloads.c4_1m(threads=16)
    qp, qt, qw = (np.asarray(x) for x in wl.queries)
    q = (qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32))
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = m.build_index_from_csr(os.path.join(shm, f"msr_probe_{os.getpid()}.idx"), *wl.docs, wl.n_terms, threads=16)
    try:
        base = None
        for G in (1, 2, 4, 8):
            rows = []
            for s in range(G):
                with m.SparseIndex(path, device=0, shard=s, n_shards=G) as ix:
                    sc, mg = timed(ix, q, 10)
                    rows.append((ix.shard_ntiles, sc, mg))
            worst = max(r[1] + r[2] for r in rows)
            base = base or worst
            print(f"G = {G}: tiles per shard {[r[0] for r in rows]}, score_tiles + merge per shard (ms) "
                  f"{[round(r[1] + r[2], 3) for r in rows]}; slowest {worst:.3f} ms = {base / worst:.2f} x the unsharded step")
    finally:
        os.remove(path)


if __name__ == "__main__":
    main()
