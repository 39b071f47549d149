#!/usr/bin/env python3
"""Benchmark of the sparse-search hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic queries: the scoring kernel + top-k merge for every
query of the batch, with the index and the CSR queries already resident in HBM.

Headline workload (BASELINE.json metric, configs[1]): Flickr30K-shape text->image sparse search, top-10.
  N > 1: the reference's own data parallelism (src/search.py:180-182): every rank holds the (16 MB) index and scores
  its own queries; no data-path collective; "scaling": "weak".
Extra object "c4_1m" (BASELINE.json configs[3], the north-star target): 1 M docs / 10 000 queries; at N > 1 the index
  is doc-range sharded over the ranks and the per-shard top-k lists are merged after ONE RCCL all-gather (exact);
  "scaling": "strong".

rank 0 prints ONE JSON line. torch is used only as launcher plumbing (gloo barrier / max-reduce); the search path is
ctypes -> libmsr.so -> HIP.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Ranks:
    """Launcher plumbing: rank info, barrier and max-over-ranks (gloo; the data path never touches torch)."""

    def __init__(self, n_gpus):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        try:  # rehearsals with more ranks than GPUs (1-GPU box) wrap around; a real node has one GPU per rank
            import torch

            n_dev = torch.cuda.device_count()
            if n_dev > 0:
                self.local_rank %= n_dev
        except Exception:
            pass
        self.dist = None
        if self.world > 1:
            import datetime

            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=20))
            self.dist = dist
        if n_gpus != self.world:
            log(f"[bench] note: --gpus {n_gpus} but WORLD_SIZE={self.world}; using WORLD_SIZE")

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, x):
        if not self.dist:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_bytes(self, b, n):
        if not self.dist:
            return b
        obj = [b if self.rank == 0 else None]
        self.dist.broadcast_object_list(obj, src=0)
        return obj[0]

    def any_failed(self, failed):
        """True on every rank if any rank reports a failure (keeps the ranks in step instead of deadlocking)."""
        return self.max(1.0 if failed else 0.0) > 0

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


_SYNC_DEVICE = 0


def device_sync():
    """Contract: torch.cuda.synchronize() on both sides of the timed region (device-wide, covers libmsr's stream).
    Always on THIS rank's GPU: a bare synchronize() would create a context on GPU 0 from every rank."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize(_SYNC_DEVICE)
    except Exception:
        pass


_HBM_COPY_GBS = None


def hbm_copy_gbs():
    """Measured HBM bandwidth of a device-to-device copy on this rank's GPU (read + write bytes / time), quoted beside
    the vendor peak as SURVEY.md §8d asks. torch is only the allocator and the timer here."""
    global _HBM_COPY_GBS
    if _HBM_COPY_GBS is None:
        try:
            import torch

            dev = torch.device("cuda", _SYNC_DEVICE if isinstance(_SYNC_DEVICE, int) else 0)
            n = 1 << 30
            a = torch.empty(n, dtype=torch.uint8, device=dev)
            b = torch.empty(n, dtype=torch.uint8, device=dev)
            a.zero_()
            for _ in range(2):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 8
            for _ in range(reps):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize(dev)
            _HBM_COPY_GBS = round(2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del a, b
        except Exception:
            _HBM_COPY_GBS = 0.0
    return _HBM_COPY_GBS or None


def timed_steps(batch, k, steps, warmup, ranks, sharded=False, step=None):
    """`step` (optional) replaces batch.search for exchanges that run on the host (gloo fallback)."""
    run = step if step is not None else (lambda: batch.search(k, sharded=sharded))
    for _ in range(warmup):
        run()
    batch.sync()
    batch.timing_reset()
    ranks.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    batch.sync()
    device_sync()
    ranks.barrier()
    dt = time.perf_counter() - t0
    calls, score_ms, merge_ms = batch.timing_sum()
    if os.environ.get("MSR_DEBUG_FLAGS"):
        st = batch.debug_stamps().astype(np.float64)
        if st.sum() > 0:
            names = ["init", "stage", "stream", "wait", "maxima", "cand", "rank", "resolve"]
            log("[bench] wave-0 phase shares: " + ", ".join(f"{n}={v / st.sum():.3f}" for n, v in zip(names, st))
                + f"; cycles/WG-launch total={st.sum() / max(calls + warmup, 1):.3e}")
    return ranks.max(dt), score_ms / max(calls, 1), merge_ms / max(calls, 1)


def roofline(batch, k, score_ms_avg, workload_name):
    by, postings = batch.algo_bytes(k)
    achieved = by / (score_ms_avg * 1e-3) / 1e9 if score_ms_avg > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from rocprofv3 --pmc runs
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(workload_name)
        except Exception:
            traffic = None
    # One step = the staged search's two score_tiles launches (first 1/16 of the tiles, then the rest with the
    # thresholds those gave): bytes, traffic and kernel_ms are all per step, i.e. summed over both.
    return {"bound": "hbm", "kernel": "score_tiles", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": by, "postings_per_launch": postings,
            "hbm_copy_measured": hbm_copy_gbs(), "kernel_ms": round(score_ms_avg, 4), "launches_per_step": 2 if batch.index.n_tiles >= 2 else 1,
            "note": "achieved = SURVEY §8d algorithmic bytes / kernel time; it can exceed the HBM peak because a tile's "
                    "postings are re-read by every query from the XCDs' L2, not from HBM (traffic = measured HBM-side "
                    "bytes per step); the kernel's own limiters are VALU issue (64-74 % busy) and L2->CU bandwidth "
                    "(DESIGN.md §7)"}


def cpu_baseline(wl, got, target_seconds, threads):
    """The oracle's C restatement (multithreaded term-at-a-time) timed on this host on a bounded query sample, and
    used as the checker of the GPU results on that sample. Checker only: nothing here feeds the GPU path."""
    from oracle import taat

    dp, dt, dw = wl.docs
    qp, qt, qw = wl.queries
    nq = len(qp) - 1
    t0 = time.perf_counter()
    oix, _ = taat.TaatIndex.from_rows_by_docid(dp, dt, dw, wl.n_terms)
    build_s = time.perf_counter() - t0

    def sub(a, b):
        return (qp[a : b + 1] - qp[a]), qt[qp[a] : qp[b]], qw[qp[a] : qp[b]]

    probe = min(nq, max(4 * threads, 64))
    t0 = time.perf_counter()
    oix.search(*sub(0, probe), wl.k, threads=threads)
    per_q = (time.perf_counter() - t0) / probe
    n = int(min(nq, max(probe, target_seconds / max(per_q, 1e-9))))
    t0 = time.perf_counter()
    w_ord, w_sc, w_n = oix.search(*sub(0, n), wl.k, threads=threads)
    dt_s = time.perf_counter() - t0
    g_ord, _, g_u32, g_n = got
    mask = np.arange(wl.k)[None, :] < w_n[:, None]
    mism = int((g_n[:n] != w_n).sum() + (g_ord[:n].astype(np.int64)[mask] != w_ord[mask]).sum()
               + (g_u32[:n].astype(np.int64)[mask] != w_sc[mask]).sum())
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # (one thread count only: the 1-GPU box grants this job 16 CPUs, which is also the reference's setting,
    # scripts/search_sparse.sh:17 — 128 threads on the same share measured SLOWER, 202 K vs 231 K q/s)
    return {"value": round(n / dt_s, 1), "unit": "queries/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
            "sample": f"first {n} of {nq} queries of the same workload, {threads} threads, exhaustive term-at-a-time "
                      f"C restatement (oracle/oracle_taat.c); index build {build_s:.1f}s not timed",
            "seconds": round(dt_s, 2)}, {"checked_queries": n, "mismatches": mism}


def run_headline(args, ranks, m, wlmod):
    seed = 1
    t0 = time.perf_counter()
    wl = wlmod.flickr30k_t2i(n_images=args.flickr_images, seed=seed, threads=args.host_threads)
    if ranks.world > 1:  # every rank scores its own captions (DP over queries, src/search.py:180-182)
        wl.queries = wlmod.planted_captions(wl.docs, wl.n_terms, 5, 8, 15, 6, seed + 1 + 100 * ranks.rank,
                                            args.host_threads)
    log(f"[bench r{ranks.rank}] workload {wl.name} generated in {time.perf_counter() - t0:.1f}s")
    tmp = tempfile.mkdtemp(prefix="msr_bench_")
    path = m.build_index_from_csr(os.path.join(tmp, f"flickr_{ranks.rank}.idx"), *wl.docs, wl.n_terms,
                                  threads=args.host_threads, tile_docs=args.tile_docs)
    ix = m.SparseIndex(path, device=ranks.local_rank)
    qp, qt, qw = wl.queries
    nq = len(qp) - 1
    batch = ix.batch(qp, qt, qw, wl.k)
    wall, score_ms, merge_ms = timed_steps(batch, wl.k, args.steps, args.warmup, ranks)
    out = {
        "metric": "queries/sec, Flickr30K-shape text->image sparse search (top-10), MI355X",
        "value": round(ranks.world * nq * args.steps / wall, 1),
        "unit": "queries/s",
        "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": wl.description, "queries_per_step_per_gpu": nq, "k": wl.k,
                   "tile_docs": ix.tile_docs, "n_tiles": ix.n_tiles,
                   "parallelism": "1 GPU" if ranks.world == 1 else f"dp{ranks.world} over queries, index replicated"},
    }
    if ranks.rank == 0:
        out["roofline"] = roofline(batch, wl.k, score_ms, wl.name)
        out["roofline"]["merge_kernel_ms"] = round(merge_ms, 4)
        got = batch.fetch()
        # Recall@1/5/10 of the GPU results (qrels: caption j <-> image j // 5)
        doc_int = np.array([int(ix.docid(o)) for o in range(ix.n_docs)], dtype=np.int64)
        ranked = np.where(np.arange(wl.k)[None, :] < got[3][:, None], doc_int[np.minimum(got[0], ix.n_docs - 1)], -1)
        target = (np.arange(nq) // 5)[:, None]
        out["recall"] = {f"R@{k}": round(float((ranked[:, :k] == target).any(axis=1).mean()), 5) for k in (1, 5, 10)}
        if ranks.world == 1 and not args.no_cpu:
            cb, par = cpu_baseline(wl, got, args.cpu_seconds, args.cpu_threads)
            out["cpu_baseline"] = cb
            out["parity"] = par
            out["speedup_vs_cpu"] = round(out["value"] / cb["value"], 1)
        if ranks.world == 1 and not args.no_cpu:  # (--no-cpu = profiling runs: only the timed launches)
            # PCIe-inclusive figures (never `value`): host CSR in -> host results out through msr_search_csr, and the
            # reference's own call shape of 4 queries per batch_search call (scripts/search_sparse.sh:16)
            ix.search_csr(qp, qt, qw, wl.k)
            t0 = time.perf_counter()
            ix.search_csr(qp, qt, qw, wl.k)
            e2e = time.perf_counter() - t0
            q4 = (qp[:5] - qp[0]), qt[: qp[4]], qw[: qp[4]]
            ix.search_csr(*q4, wl.k)
            t0 = time.perf_counter()
            for _ in range(50):
                ix.search_csr(*q4, wl.k)
            small = (time.perf_counter() - t0) / 50
            # ... and the drop-in class itself on query STRINGS (tokens repeated weight times, src/search.py:419-422)
            from mllm_sparse_retrieval_amd.compat import LuceneImpactSearcher

            strings = [" ".join(" ".join([str(int(t))] * int(w)) for t, w in zip(qt[qp[i]:qp[i + 1]], qw[qp[i]:qp[i + 1]]))
                       for i in range(4)]
            searcher = LuceneImpactSearcher(path, None, device=ranks.local_rank)
            searcher.batch_search(strings, ["0", "1", "2", "3"], wl.k, threads=16)
            t0 = time.perf_counter()
            for _ in range(50):
                searcher.batch_search(strings, ["0", "1", "2", "3"], wl.k, threads=16)
            dropin = (time.perf_counter() - t0) / 50
            searcher.close()
            out["host_inclusive"] = {"queries_per_s_one_call": round(nq / e2e, 1),
                                     "ms_per_call_4_queries": round(small * 1e3, 4),
                                     "queries_per_s_4_per_call": round(4 / small, 1),
                                     "ms_per_batch_search_4_query_strings": round(dropin * 1e3, 4),
                                     "tokens_per_query_string": round(sum(len(x.split()) for x in strings) / 4, 1)}
    batch.close()
    ix.close()
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    return out


def run_c4(args, ranks, m, wlmod):
    """configs[3]: 1 M docs; doc-range shards + one RCCL all-gather of the per-shard top-k (exact)."""
    t0 = time.perf_counter()
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(shm, f"msr_c4_{os.environ.get('MASTER_PORT', 'p')}_{os.getuid()}.idx")
    wl = None
    err = None
    if ranks.rank == 0:
        try:
            wl = wlmod.c4_1m(n_docs=args.c4_docs, n_queries=args.c4_queries, threads=args.host_threads)
            m.build_index_from_csr(path, *wl.docs, wl.n_terms, threads=args.host_threads, tile_docs=args.c4_tile_docs)
            log(f"[bench] c4 corpus + index in {time.perf_counter() - t0:.1f}s -> {path}")
        except Exception as e:
            err = e
    if ranks.any_failed(err is not None):
        raise RuntimeError(f"c4 corpus/index build failed on rank 0: {err}")
    try:
        ix = batch = None
        sharded = ranks.world > 1
        try:
            qp, qt, qw = m.synth_vectors(args.c4_queries, 120, 30000, seed=4, threads=args.host_threads)
            qp, qt, qw = qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)
            ix = m.SparseIndex(path, device=ranks.local_rank, shard=ranks.rank, n_shards=ranks.world)
        except Exception as e:
            err = e
        if ranks.any_failed(err is not None):
            raise RuntimeError(f"opening the index shard failed on some rank (this rank: {err})")
        exchange = "one RCCL all-gather of per-shard top-k + exact merge"
        step = None
        fallback_result = {}
        if sharded:
            uid = ranks.bcast_bytes(m.comm_unique_id() if ranks.rank == 0 else None, 128)
            try:
                ix.comm_init(ranks.world, ranks.rank, uid)
            except Exception as e:
                err = e
            if ranks.any_failed(err is not None):
                # RCCL could not be brought up (this cannot be rehearsed on a 1-GPU box): keep the doc-range shards and
                # move the per-shard lists through torch.distributed (gloo) instead, merged by msr_merge_lists
                log(f"[bench r{ranks.rank}] RCCL init failed ({err}); falling back to a gloo all-gather of the lists")
                from mllm_sparse_retrieval_amd import dist as mdist

                exchange = f"FALLBACK: gloo all-gather of host lists + msr_merge_lists (RCCL init failed: {err})"
                sharded = False  # batch.search without the in-library exchange

                def step():
                    batch.search(10)
                    o, _, su, cnt = batch.fetch()
                    g = mdist.all_gather_lists(ranks.dist, o, su, cnt)
                    fallback_result["r"] = ix.merge_lists(g[0], g[1], g[2], 10)
        batch = ix.batch(qp, qt, qw, 10)
        wall, score_ms, merge_ms = timed_steps(batch, 10, args.steps, args.warmup, ranks, sharded=sharded, step=step)
        nq = len(qp) - 1
        out = {"workload": f"synthetic {args.c4_docs} docs x128 nnz ({ix.n_postings} postings), {nq} queries x120 nnz, "
                           f"V=30000, top-10",
               "value": round(nq * args.steps / wall, 1), "unit": "queries/s", "n_gpus": ranks.world,
               "ms_per_step": round(wall / args.steps * 1e3, 3), "scaling": "strong",
               "parallelism": "1 GPU" if ranks.world == 1 else
               f"index doc-range sharded over {ranks.world} GPUs ({ix.shard_ntiles} of {ix.n_tiles} tiles on rank 0), "
               + exchange}
        if ranks.rank == 0:
            out["roofline"] = roofline(batch, 10, score_ms, "c4_1m" if ranks.world == 1 else f"c4_1m_shard{ranks.world}")
            out["roofline"]["merge_and_exchange_ms"] = round(merge_ms, 4)
            if ranks.world == 1 and not args.no_cpu:
                cb, par = cpu_baseline(wl, batch.fetch(), args.cpu_seconds, args.cpu_threads)
                out["cpu_baseline"] = cb
                out["parity"] = par
                out["speedup_vs_cpu"] = round(out["value"] / cb["value"], 1)
        doc_sharded_result = batch.fetch() if sharded else fallback_result.get("r")
        batch.close()
        if sharded:
            ix.comm_destroy()
        ix.close()
        if sharded and not args.no_term_shards:
            # the north star's literal partition: term-range shards. Exact, but the exchange is a reduce-scatter of
            # u32 accumulators (nq x N x 4 B), so it is exchange-bound by construction.
            terr = None
            try:
                ixf = m.SparseIndex(path, device=ranks.local_rank)              # every doc tile, own term range
                uid = ranks.bcast_bytes(m.comm_unique_id() if ranks.rank == 0 else None, 128)
                ixf.comm_init(ranks.world, ranks.rank, uid)
                tb = ixf.batch(qp, qt, qw, 10, term_shard=(ranks.rank, ranks.world))
            except Exception as e:
                terr = e
            if ranks.any_failed(terr is not None):
                out["term_range_shards"] = {"error": f"setup failed on some rank (this rank: {terr})"}
            else:
                twall, _, _ = timed_steps(tb, 10, max(1, min(args.steps, 3)), 1, ranks, sharded="terms")
                tsteps = max(1, min(args.steps, 3))
                tres = tb.fetch()
                same = all((x == y).all() for x, y in zip(tres, doc_sharded_result))
                out["term_range_shards"] = {
                    "value": round(nq * tsteps / twall, 1), "unit": "queries/s", "ms_per_step": round(twall / tsteps * 1e3, 2),
                    "steps": tsteps, "identical_to_doc_range_result": bool(same),
                    "exchange": "ncclReduceScatter(sum) of u32 accumulator tiles, then ncclAllGather of per-range top-k"}
                tb.close()
                ixf.comm_destroy()
                ixf.close()
    finally:
        ranks.barrier()
        if ranks.rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass
    return out


def run_c5(args, ranks, m):
    """configs[4]: hybrid dense + sparse, COCO-5K t->i shape: N = 5 000 docs (128 nnz + 4096-d fp16 unit vectors),
    25 010 queries (120 nnz + 4096-d), depth 1000 -> fused top-10, alpha 0.5 (scripts/search.sh:25,32). One GPU."""
    from mllm_sparse_retrieval_amd.dense import DenseIndex, hybrid_search, row_to_ordinal

    n, nq, h, depth, k, alpha, n_terms = args.c5_docs, args.c5_queries, 4096, 1000, 10, 0.5, 30000
    docs = m.synth_vectors(n, 128, n_terms, seed=4, threads=args.host_threads)
    qp, qt, qw = m.synth_vectors(nq, 120, n_terms, seed=5, threads=args.host_threads)
    rng = np.random.default_rng(4)
    p = rng.standard_normal((n, h), dtype=np.float32)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    q = rng.standard_normal((nq, h), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    tmp = tempfile.mkdtemp(prefix="msr_c5_")
    path = m.build_index_from_csr(os.path.join(tmp, "c5.idx"), *docs, n_terms, threads=args.host_threads)
    ix = m.SparseIndex(path, device=ranks.local_rank)
    dix = DenseIndex(p, device=ranks.local_rank)
    r2o = row_to_ordinal(ix, [str(i) for i in range(n)])
    hybrid_search(ix, dix, qp, qt, qw, q, depth, k, alpha, r2o)                      # warm-up
    t0 = time.perf_counter()
    ords, fs, cnt, ms = hybrid_search(ix, dix, qp, qt, qw, q, depth, k, alpha, r2o)
    wall = time.perf_counter() - t0
    kern = sum(ms.values())
    out = {"workload": f"hybrid: {n} docs x (128 nnz + {h}-d fp16), {nq} queries x (120 nnz + {h}-d), depth {depth} -> "
                       f"fused top-{k}, alpha {alpha}",
           "value": round(nq / (kern * 1e-3), 1), "unit": "queries/s (kernel time, inputs resident)",
           "host_inclusive_queries_per_s": round(nq / wall, 1),
           "kernel_ms": {k2: round(v, 3) for k2, v in ms.items()},
           "dense_tflops": round(2.0 * nq * n * h / (ms["dense_gemm"] * 1e-3) / 1e12, 1) if ms["dense_gemm"] > 0 else None,
           "dtype": "f16 in / f32 accumulate (dense), u32 (sparse), f32 (fusion)"}
    dix.close()
    ix.close()
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    return out


def main():
    # stdout carries exactly ONE JSON line: keep a private handle to it and point fd 1 at stderr, so that banners
    # printed by native libraries (RCCL prints its version block at communicator init) cannot pollute it
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--flickr-images", type=int, default=31014)
    ap.add_argument("--tile-docs", type=int, default=0)
    ap.add_argument("--c4-docs", type=int, default=1_000_000)
    ap.add_argument("--c4-queries", type=int, default=10_000)
    ap.add_argument("--c4-tile-docs", type=int, default=0)
    ap.add_argument("--dense-max", type=int, default=-1, help="index build option dense_max_terms (-1: library default)")
    ap.add_argument("--dense-density", type=float, default=-1.0, help="index build option dense_min_density")
    ap.add_argument("--c4-timeout", type=float, default=900.0, help="watchdog for the 1 M-doc object at N > 1 (s)")
    ap.add_argument("--no-c4", action="store_true", help="skip the 1 M-doc extra object")
    ap.add_argument("--no-term-shards", action="store_true", help="skip the term-range sharded variant at N > 1")
    ap.add_argument("--only-c4", action="store_true", help="(profiling) run only the 1 M-doc workload")
    ap.add_argument("--c5-docs", type=int, default=5000)
    ap.add_argument("--c5-queries", type=int, default=25010)
    ap.add_argument("--no-c5", action="store_true", help="skip the hybrid (config 5) extra object")
    ap.add_argument("--only-c5", action="store_true", help="(profiling) run only the hybrid workload")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity sample")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--host-threads", type=int, default=min(16, os.cpu_count() or 1))
    args = ap.parse_args()

    ranks = Ranks(args.gpus)
    global _SYNC_DEVICE
    _SYNC_DEVICE = ranks.local_rank
    import mllm_sparse_retrieval_amd as m  # raises if libmsr.so is missing: there is no fallback scorer
    from mllm_sparse_retrieval_amd import workloads as wlmod

    if args.dense_max >= 0:
        m.set_build_option("dense_max_terms", args.dense_max)
    if args.dense_density >= 0:
        m.set_build_option("dense_min_density", args.dense_density)

    out = {}
    if not args.only_c4 and not args.only_c5:
        out = run_headline(args, ranks, m, wlmod)
    if ranks.world == 1 and not args.no_c5 and not args.only_c4:
        try:
            out["c5_hybrid"] = run_c5(args, ranks, m)
        except Exception as e:
            out["c5_hybrid"] = {"error": f"{type(e).__name__}: {e}"}
            log(f"[bench] c5_hybrid failed: {out['c5_hybrid']['error']}")
    hung = False
    if not args.no_c4 and not args.only_c5:
        # The extra object must never take the headline line down: exceptions are caught, and at N > 1 (RCCL paths that
        # cannot be rehearsed on a 1-GPU box) a watchdog bounds the wait so that the JSON line is printed regardless.
        box = {}

        def work():
            try:
                box["c4"] = run_c4(args, ranks, m, wlmod)
            except Exception as e:
                box["c4"] = {"error": f"{type(e).__name__}: {e}"}
                log(f"[bench r{ranks.rank}] c4_1m failed: {box['c4']['error']}")

        if ranks.world > 1:
            import threading

            t = threading.Thread(target=work, daemon=True)
            t.start()
            t.join(args.c4_timeout)
            if t.is_alive():
                hung = True
                box["c4"] = {"error": f"no completion within {args.c4_timeout:.0f}s (multi-rank exchange hung?)"}
        else:
            work()
        out["c4_1m"] = box["c4"]
    if ranks.rank == 0:
        print(json.dumps(out), file=real_stdout, flush=True)
    if hung:
        real_stdout.flush()
        os._exit(0)  # a native call is stuck: leave without joining it
    ranks.close()


if __name__ == "__main__":
    main()
