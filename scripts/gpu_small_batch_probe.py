"""The reference's call shape (src/search.py:278,423-425: batch_search with per_device_batch_size = 4 queries,
scripts/search_sparse.sh:16; 2 in scripts/search.sh:16) on the headline index: where do the microseconds of one call go?
usage: python scripts/gpu_small_batch_probe.py [batch sizes ...]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd import workloads  # noqa: E402
from mllm_sparse_retrieval_amd.compat import LuceneImpactSearcher  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 16, 64]
    wl = workloads.flickr30k_t2i(n_images=31014, seed=1, threads=16)
    tmp = tempfile.mkdtemp(prefix="msr_small_")
    path = m.build_index_from_csr(os.path.join(tmp, "f.idx"), *wl.docs, wl.n_terms, threads=16)
    qp, qt, qw = wl.queries
    with m.SparseIndex(path, device=0) as ix:
        for bs in sizes:
            q = (qp[: bs + 1] - qp[0]), qt[: qp[bs]], qw[: qp[bs]]
            for _ in range(20):
                ix.search_csr(*q, 10)
            reps = 200
            acc = None
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.search_csr(*q, 10)
                laps = m.search_laps()
                acc = laps if acc is None else {k: acc[k] + v for k, v in laps.items()}
            wall = (time.perf_counter() - t0) / reps * 1e6
            avg = {k: round(v / reps, 1) for k, v in acc.items()}
            print(f"CSR, {bs} queries per call: python wall {wall:.1f} us (incl. the laps query); C call {avg['call_total']} us = "
                  f"prepare+upload {avg['prepare_upload']} + enqueue {avg['enqueue_kernels']} + wait {avg['wait_stream']} + "
                  f"download {avg['download']} + release {avg['release']}; kernel spans: score {avg['score_kernel']} us, "
                  f"merge {avg['merge_kernel']} us", flush=True)
    s = LuceneImpactSearcher(path, None, device=0)
    for bs in sizes:
        strings = [" ".join(" ".join([str(int(t))] * int(w)) for t, w in zip(qt[qp[i]:qp[i + 1]], qw[qp[i]:qp[i + 1]]))
                   for i in range(bs)]
        ids = [str(i) for i in range(bs)]
        for _ in range(20):
            s.batch_search(strings, ids, 10, threads=16)
        t0 = time.perf_counter()
        for _ in range(200):
            s.batch_search(strings, ids, 10, threads=16)
        wall = (time.perf_counter() - t0) / 200 * 1e6
        laps = m.search_laps()
        print(f"drop-in class, {bs} query strings ({sum(len(x.split()) for x in strings) // bs} tokens each): {wall:.1f} us per "
              f"batch_search; last C search call {laps['call_total']} us")
    s.close()
    os.remove(path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
