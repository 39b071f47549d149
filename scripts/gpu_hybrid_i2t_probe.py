"""The reference's OWN hybrid run shape on one GPU (scripts/search.sh:5,16-33: TARGET_TYPE=text, --query_type image,
--depth 1000, --remove_query, --alpha 0.5): 5 000 image queries over 25 010 caption docs, H = 4096. Prints the kernel
laps of msr_hybrid_search for the fused multi-tile path and for the list-based path (MSR_NO_FUSED_HYBRID=1 in a child).
usage: python scripts/gpu_hybrid_i2t_probe.py [k]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd import workloads  # noqa: E402
from mllm_sparse_retrieval_amd.dense import DenseIndex, hybrid_search, row_to_ordinal  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n, nq, h, depth, alpha = 25010, 5000, 4096, 1000, 0.5
    docs, (qp, qt, qw), p, q = workloads.hybrid_vectors(n, nq, h)
    tmp = tempfile.mkdtemp(prefix="msr_i2t_")
    path = m.build_index_from_csr(os.path.join(tmp, "i2t.idx"), *docs, 30000)
    with m.SparseIndex(path, device=0) as ix:
        dix = DenseIndex(p)
        r2o = row_to_ordinal(ix, [str(i) for i in range(n)])
        self_ord = r2o[:nq].astype(np.int32)  # query j is "the same id" as doc j (remove_query)
        for rep in range(3):
            t0 = time.perf_counter()
            ords, fs, cnt, ms = hybrid_search(ix, dix, qp, qt, qw, q, depth, k, alpha, r2o, self_ord)
            wall = time.perf_counter() - t0
            print(f"[{os.environ.get('MSR_NO_FUSED_HYBRID', 'fused')}] rep {rep}: k={k} kernel ms {dict((a, round(b, 3)) for a, b in ms.items())} "
                  f"sum {sum(ms.values()):.3f} ms -> {nq / sum(ms.values()) * 1e3:.0f} q/s; host wall {wall * 1e3:.1f} ms", flush=True)
        print("checksum", int(ords.astype(np.int64).sum()), float(fs.sum()), int(cnt.sum()))
        dix.close()
    os.remove(path)
    os.rmdir(tmp)
    if "MSR_NO_FUSED_HYBRID" not in os.environ and "--no-child" not in sys.argv:
        env = dict(os.environ, MSR_NO_FUSED_HYBRID="1")
        subprocess.run([sys.executable, os.path.abspath(__file__), str(k), "--no-child"], env=env, check=False)


if __name__ == "__main__":
    main()
