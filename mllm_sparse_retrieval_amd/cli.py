"""encode -> index -> search, with the reference's flag names (src/arguments.py:57-68, scripts/*.sh).

    python -m mllm_sparse_retrieval_amd encode --synthetic flickr --sparse_output_dir out/        (no MLLM here)
    python -m mllm_sparse_retrieval_amd index  --input out/ [--index out/index] --threads 16       (sparse_index.sh)
    python -m mllm_sparse_retrieval_amd search --sparse_index out/ --depth 10 --query_type text \\
           --dataset_name flickr --queries out/query.tsv --qrels out/qrels.csv --save_dir runs/   (search_sparse.sh)

    python -m mllm_sparse_retrieval_amd eval   --runs_dir runs/ --qrels out/qrels.csv --dataset_name flickr \
           --query_type text [--compat-denominator --world_size 4]                                (recall of TREC runs)

`encode` writes exactly the files src/encode.py:412-426 writes (corpus_{shard}.jsonl, query.tsv); `index` accepts and
ignores pyserini's --collection/--generator/--impact/--pretokenized; `search` prints recall in the reference's format
(src/metrices.py:103-137) and writes TREC runs (src/hybrid.py:20-29) under --save_dir. Under a launcher
(`python -m torch.distributed.run --nproc-per-node N -m mllm_sparse_retrieval_amd search ...`, the counterpart of
`deepspeed --num_gpus=4 src/search.py`, scripts/search_sparse.sh:14) `search` is the reference's DP over queries
(src/search.py:180-182): every rank opens the index on its own GPU, searches its DistributedSampler shard of the
queries, and only recall fractions are gathered (src/metrices.py:86-100). `eval` is the recall reporter on its own.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace


def _cmd_encode(a):
    import numpy as np

    from . import workloads

    if a.synthetic is None:
        sys.exit("encode: only --synthetic {flickr,coco} is available (the MLLM encoder is out of scope here)")
    if a.synthetic == "flickr":
        wl = workloads.flickr30k_t2i(n_images=a.n_images or 1000, seed=a.seed, threads=a.threads)
        name = "flickr"
    else:
        wl = workloads.flickr30k_t2i(n_images=a.n_images or 5000, n_terms=30000, seed=a.seed, threads=a.threads)
        name = "coco"
    out = a.sparse_output_dir
    os.makedirs(out, exist_ok=True)
    dp, dt, dw = wl.docs
    with open(os.path.join(out, f"corpus_{a.dataset_shard_index}.jsonl"), "w") as f:
        for i in range(len(dp) - 1):
            vec = {f"t{int(t)}": int(w) for t, w in zip(dt[dp[i]:dp[i + 1]], dw[dp[i]:dp[i + 1]])}
            f.write(json.dumps(dict(id=str(i), content="", vector=vec)) + "\n")
    qp, qt, qw = wl.queries
    with open(os.path.join(out, "query.tsv"), "w") as f:
        for i in range(len(qp) - 1):
            toks = " ".join(" ".join([f"t{int(t)}"] * int(w)) for t, w in zip(qt[qp[i]:qp[i + 1]], qw[qp[i]:qp[i + 1]]))
            if toks.strip():
                f.write(f"{i}\t{toks}\n")
    if a.dense_output_dir:  # dense reps in the reference's pickle layout (reps, lookup) (src/encode.py:405-410)
        import pickle

        os.makedirs(a.dense_output_dir, exist_ok=True)
        rng = np.random.default_rng(a.seed)
        n_docs, n_q = len(dp) - 1, len(qp) - 1
        p = rng.standard_normal((n_docs, a.dense_dim)).astype(np.float32)
        # planted signal: a caption's vector is its image's vector plus noise
        qv = p[np.arange(n_q) // 5] + 1.5 * rng.standard_normal((n_q, a.dense_dim)).astype(np.float32)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        with open(os.path.join(a.dense_output_dir, f"corpus_{a.dataset_shard_index}.pkl"), "wb") as f:
            pickle.dump((p, [str(i) for i in range(n_docs)]), f)
        with open(os.path.join(a.dense_output_dir, "query.pkl"), "wb") as f:
            pickle.dump((qv, [str(i) for i in range(n_q)]), f)
    with open(os.path.join(out, "qrels.csv"), "w") as f:  # flickr csv schema (src/dataset.py:86-102)
        f.write("imgid,filename,caption,sentid\n")
        for i in range(len(qp) - 1):
            f.write(f"{i // 5},{i // 5}.jpg,,{i}\n")
    print(f"encode: {len(dp) - 1} docs, {len(qp) - 1} queries ({name}-shape, synthetic) -> {out}")


def _cmd_index(a):
    from .index import build_index_from_jsonl

    out = os.path.join(a.index, "msr.idx") if a.index else None
    if a.index:
        os.makedirs(a.index, exist_ok=True)
    t0 = time.time()
    path = build_index_from_jsonl(a.input, out, threads=a.threads, tile_docs=a.tile_docs)
    print(f"index: {path} ({os.path.getsize(path) / 1e6:.1f} MB) in {time.time() - t0:.2f}s")


def read_queries(path):
    """query.tsv (`id\\ttok tok tok`, src/encode.py:418-424) or query jsonl ({"id","vector"}) -> (ids, strings)."""
    ids, texts = [], []
    with open(path, encoding="utf-8") as f:
        if path.endswith((".jsonl", ".json")):
            for line in f:
                if line.strip():
                    o = json.loads(line)
                    ids.append(str(o["id"]))
                    texts.append(" ".join(" ".join([t] * int(v)) for t, v in o["vector"].items() if int(v) > 0))
        else:
            for line in f:
                line = line.rstrip("\n")
                if line:
                    qid, _, text = line.partition("\t")
                    ids.append(qid)
                    texts.append(text)
    return ids, texts


def _cmd_search(a):
    from .fusion import write_trec_run
    from .qrels import CrossModalQrels
    from .recall import RecallMetrics
    from .run import get_run_dict, sparse_search
    from .searcher import JWhiteSpaceAnalyzer, LuceneImpactSearcher

    dense_retriever = p_reps = p_lookup = q_reps_by_id = None
    if a.passage_reps is not None:  # dense side of the hybrid search (src/search.py:227-237)
        import glob

        import numpy as np

        from .run import pickle_load, search_queries

        files = sorted(glob.glob(os.path.join(a.passage_reps, "corpus*.pkl")))
        if not files:
            sys.exit(f"search: no corpus*.pkl under {a.passage_reps}")
        p_reps, p_lookup = pickle_load(files[0])     # files written by this package's own encode step
        q_reps, q_lookup = pickle_load(os.path.join(a.passage_reps, "query.pkl"))
        q_reps_by_id = {str(i): r for i, r in zip(q_lookup, q_reps)}
    # one process per GPU under a launcher (src/search.py:115-129); gloo carries the recall fractions only
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("gloo")
        a.device = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("MSR_SHARE_GPU"):  # rehearsal on a box with fewer GPUs than ranks: ranks share the devices
            import torch

            a.device %= max(torch.cuda.device_count(), 1)
    searcher = LuceneImpactSearcher(os.path.join(a.sparse_index, "index") if os.path.isdir(
        os.path.join(a.sparse_index, "index")) else a.sparse_index, None, device=a.device)
    searcher.set_analyzer(JWhiteSpaceAnalyzer())
    qids, texts = read_queries(a.queries or os.path.join(a.sparse_index, "query.tsv"))
    n_queries = len(qids)
    if world > 1:  # the reference's DistributedSampler split, padding included (src/search.py:180-182)
        from .sampler import distributed_sampler_indices

        pos = distributed_sampler_indices(n_queries, world, rank)
        if not a.compat_denominator:  # every query once: drop the padded repeats
            pos = [p for j, p in enumerate(pos) if rank + j * world < n_queries]
        qids, texts = [qids[p] for p in pos], [texts[p] for p in pos]
    bs = a.batch_size if a.batch_size > 0 else len(qids)
    args = SimpleNamespace(depth=a.depth, threads=a.threads, query_type=a.query_type)
    sparse_run, dense_run, fusion_run = {}, {}, {}
    args.batch_size, args.quiet = a.batch_size, True
    t0 = time.time()
    if p_reps is not None and not a.host_fusion:
        # Hybrid on the GPU (the default): this rank's WHOLE query shard in one msr_hybrid_search call — sparse scores,
        # both depth lists, the reference's fusion (src/hybrid.py:32-53) and the top --fusion_k never leave HBM.
        sparse_run, dense_run, fusion_run = _hybrid_on_gpu(a, searcher, p_reps, p_lookup, q_reps_by_id, qids, texts)
    else:
        if p_reps is not None:  # --host_fusion: the reference's own structure, batch by batch (src/search.py:455-461)
            from .dense import FaissFlatSearcher

            dense_retriever = FaissFlatSearcher(p_reps, device=a.device)
            dense_retriever.add(p_reps)
        for i in range(0, len(qids), bs):
            scores, rankings = sparse_search(searcher, texts[i:i + bs], qids[i:i + bs], args)
            sparse_run.update(get_run_dict(qids[i:i + bs], scores, rankings, a.remove_query))
            if dense_retriever is not None:
                q = np.stack([q_reps_by_id[x] for x in qids[i:i + bs]])
                q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)    # F.normalize, src/search.py:342
                d_scores, d_ids = search_queries(dense_retriever, q, p_lookup, args)
                dense_run.update(get_run_dict(qids[i:i + bs], d_scores, d_ids, a.remove_query))
        if dense_retriever is not None:
            from .fusion import fuse

            fusion_run = fuse([dense_run, sparse_run], [a.alpha, 1 - a.alpha])         # src/search.py:455-461
    dt = time.time() - t0
    if not a.quiet:
        print(f"search: {len(qids)} queries in {dt:.3f}s ({len(qids) / max(dt, 1e-9):.0f} q/s end-to-end incl. host)")
    if a.save_dir:
        os.makedirs(a.save_dir, exist_ok=True)
        sfx = f".rank{rank}" if world > 1 else ""
        write_trec_run(sparse_run, os.path.join(a.save_dir, "sparse.trec" + sfx), name="sparse")
        if dense_run:
            write_trec_run(dense_run, os.path.join(a.save_dir, "dense.trec" + sfx), name="dense")
            write_trec_run(fusion_run, os.path.join(a.save_dir, "fusion.trec" + sfx))
        if world > 1:  # rank 0 joins the per-rank files (one node, shared directory)
            dist.barrier()
            if rank == 0:
                for name in ("sparse.trec", "dense.trec", "fusion.trec"):
                    parts = [os.path.join(a.save_dir, f"{name}.rank{r}") for r in range(world)]
                    missing = [p for p in parts if not os.path.exists(p)]
                    if missing and len(missing) < len(parts):  # (all missing: that run was not produced at all)
                        print(f"search: WARNING: {name} not joined, rank files missing (one node and a shared --save_dir "
                              f"are assumed): {missing}", file=sys.stderr)
                    if not missing:
                        done = set()  # queries an earlier rank's file already holds (the sampler's padded repeats)
                        with open(os.path.join(a.save_dir, name), "w") as out:
                            for p in parts:
                                mine = set()
                                for line in open(p):
                                    qid = line.split(" ", 1)[0]
                                    if qid in done:
                                        continue
                                    mine.add(qid)
                                    out.write(line)
                                done |= mine
                        for p in parts:
                            os.remove(p)
    if a.qrels:
        ds = CrossModalQrels(a.qrels, a.dataset_name)
        m = RecallMetrics(ds, dense_run, sparse_run, fusion_run, p_lookup or [], qids, args,
                          denominator=None if (a.compat_denominator or world == 1) else n_queries)
        m.sort_and_count()
        m.all_gather_object()
        m.print_recall()
    searcher.close()
    if world > 1:
        dist.barrier()


def _hybrid_on_gpu(a, searcher, p_reps, p_lookup, q_reps_by_id, qids, texts):
    """search --passage_reps on the GPU: ONE msr_hybrid_search call for the fused run; the sparse and the dense run the
    reference also reports (src/metrices.py:105-126) come from one whole-shard call each. Returns the three run dicts in
    the reference's layouts (src/search.py:66-82; the fused run without the 'docs' level, src/hybrid.py:32-53).
    Difference to --host_fusion: the fused run holds the best --fusion_k docs of the union (the recall reporter reads
    200, src/metrices.py:9), and an exact tie of two DENSE scores at the depth boundary goes to the lower doc ordinal
    instead of the lower row (include/msr.h)."""
    import numpy as np

    from .dense import DenseIndex, hybrid_search, row_to_ordinal
    from .run import get_run_dict

    ix = searcher.index
    depth = min(a.depth, 1024)
    k_f = max(1, min(a.fusion_k, 1024))
    q_ptr, q_term, q_w = ix.encode_queries(texts)
    q = np.stack([q_reps_by_id[x] for x in qids]).astype(np.float32) if qids else np.zeros((0, p_reps.shape[1]), np.float32)
    q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)                # F.normalize, src/search.py:342
    dix = DenseIndex(p_reps, device=ix.device)
    try:
        table = ix.docid_table()
        ord_of = {d: o for o, d in enumerate(table.tolist())}
        r2o = np.asarray([ord_of[str(x)] for x in p_lookup], dtype=np.uint32)
        self_ord = np.asarray([ord_of.get(x, -1) for x in qids], dtype=np.int32) if a.remove_query else None
        drop = searcher.min_idf >= 0 if searcher.drop_df_eq_n is None else bool(searcher.drop_df_eq_n)
        ords, fs, cnt, _ = hybrid_search(ix, dix, q_ptr, q_term, q_w, q, depth, k_f, a.alpha, r2o, self_ord, drop_df_eq_n=drop)
        fs_rows = fs.tolist()
        fusion_run = {qid: dict(zip(table[ords[i, :cnt[i]]].tolist(), fs_rows[i][:cnt[i]])) for i, qid in enumerate(qids)}
        sparse_run, dense_run = {}, {}
        if not a.fusion_only:
            o, f, _, c = ix.search_csr(q_ptr, q_term, q_w, depth, drop_df_eq_n=drop)
            f_rows = f.tolist()
            sparse_run = get_run_dict(qids, [f_rows[i][:c[i]] for i in range(len(qids))],
                                      [table[o[i, :c[i]]].tolist() for i in range(len(qids))], a.remove_query)
            d_scores, d_idx = dix.search(q, min(depth, dix.n))
            lookup = np.asarray([str(x) for x in p_lookup], dtype=object)
            dense_run = get_run_dict(qids, d_scores, lookup[d_idx], a.remove_query)
    finally:
        dix.close()
    return sparse_run, dense_run, fusion_run


def _cmd_eval(a):
    """Standalone recall reporter over TREC runs: sparse/dense/fusion.trec (src/hybrid.py:8-29
    format) + the dataset csv -> the reference's print-out (src/metrices.py:103-137). Default denominator: the true
    number of queries; --compat-denominator replays --world_size ranks of the reference (DistributedSampler shards,
    padded repeats searched and counted twice, len(lookup_indices) * world, src/metrices.py:92)."""
    from .fusion import read_trec_run
    from .qrels import CrossModalQrels
    from .recall import replay_ranks

    def load(explicit, name):
        path = explicit or (os.path.join(a.runs_dir, name) if a.runs_dir else None)
        return read_trec_run(path) if path and os.path.exists(path) else {}

    sparse_run, dense_run = load(a.sparse_run, "sparse.trec"), load(a.dense_run, "dense.trec")
    fusion_run = {q: v["docs"] for q, v in load(a.fusion_run, "fusion.trec").items()}  # no 'docs' level (src/metrices.py:70-73)
    if not (sparse_run or dense_run or fusion_run):
        sys.exit("eval: no run file found (--runs_dir with sparse.trec / dense.trec / fusion.trec, or --sparse_run ...)")
    ds = CrossModalQrels(a.qrels, a.dataset_name)
    # the query universe in dataset order: captions ('full' mode) or images ('single'), src/dataset.py:104-110
    query_ids = ds.text_id_list if a.query_type == "text" else ds.img_id_list
    if a.queries:
        query_ids = read_queries(a.queries)[0]
    # the reference prints len(look_up) = the number of corpus passages (src/metrices.py:106): take it from
    # --n_passages, else from the dataset's passage universe, else from the docs the dense run happens to hold
    if a.n_passages:
        look_up = list(range(a.n_passages))
    else:
        universe = ds.img_id_list if a.query_type == "text" else ds.text_id_list
        look_up = list(universe) or sorted({d for v in dense_run.values() for d in v["docs"]})
    args = SimpleNamespace(query_type=a.query_type)
    world = a.world_size if a.compat_denominator else 1
    m = replay_ranks(ds, dense_run, sparse_run, fusion_run, look_up, query_ids, args, world, compat=a.compat_denominator)
    m.print_recall()
    return m


def main(argv=None):
    ap = argparse.ArgumentParser(prog="mllm_sparse_retrieval_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)

    e = sub.add_parser("encode")
    e.add_argument("--synthetic", choices=["flickr", "coco"])
    e.add_argument("--sparse_output_dir", default="./sparse_output/")
    e.add_argument("--dense_output_dir", default=None)
    e.add_argument("--dense_dim", type=int, default=256)
    e.add_argument("--dataset_shard_index", type=int, default=0)
    e.add_argument("--n_images", type=int, default=0)
    e.add_argument("--seed", type=int, default=1)
    e.add_argument("--threads", type=int, default=16)
    e.set_defaults(fn=_cmd_encode)

    i = sub.add_parser("index")
    i.add_argument("--input", required=True)
    i.add_argument("--index", default=None)
    i.add_argument("--threads", type=int, default=16)
    i.add_argument("--tile_docs", type=int, default=0)
    for flag in ("--collection", "--generator"):  # pyserini flags of scripts/sparse_index.sh, accepted and ignored
        i.add_argument(flag, default=None)
    i.add_argument("--impact", action="store_true")
    i.add_argument("--pretokenized", action="store_true")
    i.set_defaults(fn=_cmd_index)

    s = sub.add_parser("search")
    s.add_argument("--sparse_index", required=True)
    s.add_argument("--passage_reps", default=None)
    s.add_argument("--depth", type=int, default=1000)
    s.add_argument("--threads", type=int, default=1)
    s.add_argument("--batch_size", type=int, default=128)
    s.add_argument("--alpha", type=float, default=0.5)
    s.add_argument("--remove_query", action="store_true")
    s.add_argument("--query_type", default="text")
    s.add_argument("--dataset_name", default="flickr")
    s.add_argument("--save_dir", default=None)
    s.add_argument("--quiet", action="store_true")
    s.add_argument("--use_gpu", action="store_true")
    s.add_argument("--queries", default=None)
    s.add_argument("--qrels", default=None)
    s.add_argument("--device", type=int, default=0)
    s.add_argument("--host_fusion", action="store_true",
                   help="hybrid: the reference's own structure (dense + sparse lists per batch, fuse() on the host) "
                        "instead of one msr_hybrid_search call over the whole query file")
    s.add_argument("--fusion_k", type=int, default=1000, help="hybrid on the GPU: docs kept of each query's fused union (<= 1024)")
    s.add_argument("--fusion_only", action="store_true", help="hybrid on the GPU: skip the separate sparse and dense runs")
    s.add_argument("--compat-denominator", "--compat_denominator", dest="compat_denominator", action="store_true",
                   help="multi-rank: keep the sampler's padded repeats and divide by len(shard) * world like the reference")
    s.set_defaults(fn=_cmd_search)

    v = sub.add_parser("eval")
    v.add_argument("--runs_dir", default=None)
    v.add_argument("--sparse_run", default=None)
    v.add_argument("--dense_run", default=None)
    v.add_argument("--fusion_run", default=None)
    v.add_argument("--qrels", required=True, help="dataset csv (data/flickr/flickr_test.csv schema, src/dataset.py:86-102)")
    v.add_argument("--dataset_name", default="flickr")
    v.add_argument("--query_type", default="text")
    v.add_argument("--n_passages", type=int, default=0, help="corpus size for the dense report's header line "
                                                            "(default: the dataset's passage universe)")
    v.add_argument("--queries", default=None, help="query.tsv giving the query universe / order (default: the csv's)")
    v.add_argument("--compat-denominator", "--compat_denominator", dest="compat_denominator", action="store_true")
    v.add_argument("--world_size", type=int, default=4, help="ranks to replay with --compat-denominator "
                                                             "(scripts/search_sparse.sh:14 uses 4)")
    v.set_defaults(fn=_cmd_eval)

    a = ap.parse_args(argv)
    a.fn(a)


if __name__ == "__main__":
    main()
