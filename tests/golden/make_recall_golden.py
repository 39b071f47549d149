#!/usr/bin/env python3
"""Generates tests/golden/recall_golden.json by RUNNING the reference's own RecallMetrics (src/metrices.py:6-137).

Run in the build container only (it needs /root/reference, which never travels to the GPU box):
    python tests/golden/make_recall_golden.py
It imports /root/reference/src/metrices.py (torch, torch.distributed and tqdm only) and records, for seeded inputs and
for world sizes 1 and 2 (gloo, one process per rank):
  - the hit counts after sort_and_count()                          src/metrices.py:37-84
  - every rank's fractions after all_gather_object()               src/metrices.py:86-100
  - the text print_recall() writes on rank 0                       src/metrices.py:102-137
The one thing changed for the run: the build container has no GPU, and the class moves its id tensors there
(`torch.tensor(...).cuda()`, :33,43) before `torch.isin`. This script makes Tensor.cuda() return the tensor itself for
its own process — device placement only; sorted(), int() and torch.isin give the same values on any device. The
dataset object is a stub with the one method the class calls, get_target(qid, query_type) (src/dataset.py:164-168:
an int-able id for text queries, a list of five for image queries). The JSON holds inputs and outputs only.
"""
import contextlib
import importlib.util
import io
import json
import os
import random
import socket
import sys
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REF = "/root/reference/src/metrices.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "recall_golden.json")


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_metrices", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class StubDataset:
    def __init__(self, targets):
        self.targets = targets

    def get_target(self, qid, query_type):
        return self.targets[qid]


def make_case(rng, nq, query_type, depth, with_dense, with_fusion, empty_prob, tie_prob):
    qids = [str(1000 + i) for i in range(nq)]
    pool = [str(i) for i in range(1, 400)]
    targets = {}
    for q in qids:
        targets[q] = rng.choice(pool) if query_type == "text" else rng.sample(pool, 5)

    def docs_for(q, integer):
        if rng.random() < empty_prob:
            return {}
        ids = rng.sample(pool, depth)
        t = targets[q]
        if rng.random() < 0.7:  # plant a target somewhere in the list (else it may still be there by chance)
            ids[rng.randrange(len(ids))] = t if isinstance(t, str) else rng.choice(t)
            ids = list(dict.fromkeys(ids))
        scores = [float(rng.randint(1, 30)) if integer else round(rng.uniform(0, 1), 3) for _ in ids]
        if rng.random() < tie_prob:  # many equal scores: the stable sort keeps hit order
            scores = [scores[0]] * len(scores)
        return dict(zip(ids, scores))

    sparse = {q: {"docs": docs_for(q, True), "min_score": 0, "max_score": 0} for q in qids}
    dense = {q: {"docs": docs_for(q, False), "min_score": 0, "max_score": 0} for q in qids} if with_dense else {}
    fusion = {q: docs_for(q, False) for q in qids} if with_fusion else {}
    return {"query_type": query_type, "qids": qids, "targets": targets, "dense_run": dense, "sparse_run": sparse,
            "fusion_run": fusion, "look_up": pool}


def shard(qids, world, rank):
    """The reference's own split: torch's DistributedSampler(num_replicas, shuffle=True, rank), no set_epoch
    (src/search.py:180)."""
    from torch.utils.data import DistributedSampler

    s = DistributedSampler(list(range(len(qids))), num_replicas=world, rank=rank, shuffle=True)
    return [qids[i] for i in s]


def run_rank(rank, world, port, case, out_path):
    torch.Tensor.cuda = lambda self, *a, **k: self  # (see the module docstring)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = load_ref()
    mine = shard(case["qids"], world, rank)
    pick = lambda run: {q: run[q] for q in dict.fromkeys(mine) if q in run}  # noqa: E731 (a padded repeat is one dict key)
    m = ref.RecallMetrics(StubDataset(case["targets"]), pick(case["dense_run"]), pick(case["sparse_run"]),
                          pick(case["fusion_run"]), case["look_up"], mine, types.SimpleNamespace(query_type=case["query_type"]))
    m.sort_and_count()
    counts = {"dense": dict(m.dense_counts), "sparse": dict(m.sparse_counts), "fusion": dict(m.fusion_counts)}
    m.all_gather_object()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        m.print_recall()
    res = {"rank": rank, "shard": mine, "counts": counts,
           "lists": {"dense": m.dense_recall_lists, "sparse": m.sparse_recall_lists, "fusion": m.fusion_recall_lists},
           "printed": buf.getvalue()}
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    rng = random.Random(20251005)
    specs = [(11, "text", 12, False, False, 0.0, 0.0), (9, "image", 30, True, True, 0.2, 0.3), (7, "text", 3, True, False, 0.3, 0.5),
             (10, "image", 250, False, True, 0.1, 0.0)]
    out = []
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"recall_golden_{os.getpid()}")
    for spec in specs:
        case = make_case(rng, *spec)
        case["by_world"] = {}
        for world in (1, 2):
            mp.spawn(run_rank, args=(world, free_port(), case, tmp), nprocs=world, join=True)
            ranks = []
            for r in range(world):
                ranks.append(json.load(open(f"{tmp}.{r}")))
                os.remove(f"{tmp}.{r}")
            case["by_world"][str(world)] = ranks
        out.append(case)
    json.dump({"reference": "src/metrices.py RecallMetrics", "cases": out}, open(OUT, "w"), indent=1)
    print(f"wrote {OUT}: {len(out)} cases x world sizes 1, 2")


if __name__ == "__main__":
    sys.exit(main())
