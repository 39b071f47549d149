"""Handle / memory leak check: open, search, hybrid-search and close many times; the VRAM in use (rocm-smi) and the
process's resident host memory after the loop must be back at their values before it. usage: python scripts/gpu_leak_check.py [iterations]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd.dense import DenseIndex, hybrid_search, row_to_ordinal  # noqa: E402


def vram_used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    mo = re.search(r"VRAM Total Used Memory \(B\):\s*(\d+)", out)
    return int(mo.group(1)) if mo else -1


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n, n_terms, h = 5000, 3000, 256
    docs = m.synth_vectors(n, 64, n_terms, seed=1, threads=8)
    qp, qt, qw = m.synth_vectors(300, 40, n_terms, seed=2, threads=8)
    qp, qt, qw = qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)
    path = m.build_index_from_csr(os.path.join(tempfile.mkdtemp(), "l.idx"), *docs, n_terms, threads=8)
    rng = np.random.default_rng(0)
    p = rng.standard_normal((n, h)).astype(np.float32)
    q = rng.standard_normal((300, h)).astype(np.float32)
    ids = [str(i) for i in range(n)]
    # one warm round first: the runtime's own pools and code objects are resident from here on
    def round_():
        with m.SparseIndex(path, device=0) as ix:
            b = ix.batch(qp, qt, qw, 10)
            b.search(10)
            b.fetch()
            ix.search_csr(qp, qt, qw, 100)
            dix = DenseIndex(p)
            dix.search(q, 10)
            r2o = row_to_ordinal(ix, ids)
            hybrid_search(ix, dix, qp, qt, qw, q, 100, 10, 0.5, r2o)     # one tile, k <= 64: the fused kernel
            hybrid_search(ix, dix, qp, qt, qw, q, 100, 100, 0.5, r2o)    # k > 64: candidate kernels + per-query fusion
            ix.search_csr(qp[:5], qt[: qp[4]], qw[: qp[4]], 10)          # a small call: mapped host blocks
            dix.close()
            # (b is left to the index: closing the index detaches live batches)
    for _ in range(20):  # warm rounds: the runtime's own pools and code objects reach their steady size
        round_()
    import psutil

    proc = psutil.Process()
    base = vram_used()
    rss0 = proc.memory_info().rss
    for i in range(iters):
        round_()
        if (i + 1) % 50 == 0:
            print(f"after {i + 1} rounds: VRAM used {vram_used() - base:+d} B vs after the warm rounds", flush=True)
    end = vram_used()
    rss1 = proc.memory_info().rss
    print(f"VRAM used: {base} B after 20 warm rounds, {end} B after {iters} more rounds ({end - base:+d} B); "
          f"host RSS {rss0 >> 20} -> {rss1 >> 20} MiB")
    sys.exit(0 if (base < 0 or end - base < (16 << 20)) and rss1 - rss0 < (64 << 20) else 1)


if __name__ == "__main__":
    main()
