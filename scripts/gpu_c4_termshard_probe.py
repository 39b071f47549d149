"""Per-rank compute of the TERM-range sharded config-4 search (BASELINE.json configs[3]), measured on ONE GPU: for G in
2, 4, 8 the G term-shard handles (msr_index_open_termshard: the postings a rank of a G-GPU run would hold) play the exact
protocol in turn — every shard dumps its partial accumulator tiles (score_tiles<MODE 1>), the sums of every doc range
are selected (select_tiles, 16 tiles per range at G = 8), the range lists are merged — and libmsr prints the HIP-event
time of every shard's dump and every range's selection (MSR_DEBUG_TERMSHARD). What cannot run here is the exchange
itself: ncclReduceScatter of nq x N x 4 B = 40 GB per step; its volume is printed next to the compute so that DESIGN.md §6
can price the 8-GPU step as compute + bytes / xGMI rate. usage: python scripts/gpu_c4_termshard_probe.py [n_queries]"""
import os
import sys
import tempfile
import time

os.environ["MSR_DEBUG_TERMSHARD"] = "1"
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mllm_sparse_retrieval_amd as m  # noqa: E402
from mllm_sparse_retrieval_amd import workloads  # noqa: E402
from mllm_sparse_retrieval_amd.index import search_termshard_emulated_handles  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    wl = workloads.c4_1m(n_queries=nq, threads=16)
    q = tuple(np.asarray(x) for x in wl.queries)
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = m.build_index_from_csr(os.path.join(shm, f"msr_tprobe_{os.getpid()}.idx"), *wl.docs, wl.n_terms, threads=16)
    try:
        with m.SparseIndex(path, device=0) as ix:
            b = ix.batch(*q, 10)
            b.search(10)
            b.sync()
            b.timing_reset()
            for _ in range(3):
                b.search(10)
            calls, sc, mg = b.timing_sum()
            want = b.fetch()
            b.close()
            n_docs, n_tiles, tile = ix.n_docs, ix.n_tiles, ix.tile_docs
            print(f"unsharded step ({nq} queries, {n_tiles} tiles, doc-range path): {(sc + mg) / calls:.3f} ms; index resident "
                  f"{ix.resident_bytes / 1e6:.0f} MB", flush=True)
        for G in (2, 4, 8):
            shards = [m.SparseIndex(path, device=0, term_shard=(g, G)) for g in range(G)]
            print(f"G = {G}: resident MB per shard {[round(sh.resident_bytes / 1e6) for sh in shards]}, term ranges "
                  f"{[(sh.term_lo, sh.term_hi) for sh in shards]}", flush=True)
            sys.stdout.flush()
            t0 = time.perf_counter()
            got = search_termshard_emulated_handles(shards, *q, 10)
            wall = (time.perf_counter() - t0) * 1e3
            same = all((a == b2).all() for a, b2 in zip(got, want))
            for sh in shards:
                sh.close()
            xchg = nq * (n_tiles * tile) * 4 * (G - 1) / G   # bytes a rank sends (and receives) in a ring reduce-scatter
            print(f"       host wall of the whole emulation {wall:.0f} ms; identical to the unsharded result: {same}; a rank's "
                  f"reduce-scatter traffic per {nq}-query step: {xchg / 1e9:.1f} GB = {xchg / 150e9 * 1e3:.0f} ms at ~150 GB/s per "
                  f"xGMI link direction (ring: per-link bound)", flush=True)
    finally:
        os.remove(path)


if __name__ == "__main__":
    main()
