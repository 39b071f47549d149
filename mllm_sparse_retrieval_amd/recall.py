"""Recall@{1,5,10,100,200} with the reference's semantics (src/metrices.py:6-137), without its GPU tensors.

Kept exactly: k list (:9); per query the {doc: score} dict is sorted by score descending with Python's stable sort, so
equal scores keep hit order (:30); doc ids and targets are int()ed (:33,41-44); a query with no docs is skipped but
still counts in the denominator (:45-46); a hit is "any target id in the top-k" (:76-84); every rank divides its count
by len(lookup_indices) * world_size (:87,92,97) and rank 0 sums the per-rank fractions (:106,118,128); print format
(:103-137). Changed: counting is plain Python sets instead of torch.isin on .cuda() tensors, and a missing process
group means world_size 1 instead of an exception (defect 1 of SURVEY.md §3.1).
"""
from __future__ import annotations

RECALL_KS = [1, 5, 10, 100, 200]


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


class RecallMetrics:
    def __init__(self, dataset, dense_run, sparse_run, fusion_run, look_up, lookup_indices, search_args,
                 denominator=None, world_size=None, rank=None):
        """denominator: None = the reference's len(lookup_indices) * world_size (src/metrices.py:92, which counts the
        sampler's padding twice); an int = that many queries (the true query count of a sharded run).
        world_size / rank: override the process group (the `eval` step replays W ranks in one process)."""
        self.recall_k_setting_list = list(RECALL_KS)
        d = _dist() if world_size is None else None
        self.world_size = world_size if world_size is not None else (d.get_world_size() if d else 1)
        self.rank = rank if rank is not None else (d.get_rank() if d else 0)
        self._dist = d
        self.denominator = denominator
        ks = self.recall_k_setting_list
        self.dense_counts = {k: 0 for k in ks}
        self.sparse_counts = {k: 0 for k in ks}
        self.fusion_counts = {k: 0 for k in ks}
        self.dense_recall_lists = {k: [None] * self.world_size for k in ks}
        self.sparse_recall_lists = {k: [None] * self.world_size for k in ks}
        self.fusion_recall_lists = {k: [None] * self.world_size for k in ks}
        self.dataset = dataset
        self.dense_run = dense_run
        self.sparse_run = sparse_run
        self.fusion_run = fusion_run
        self.look_up = look_up
        self.lookup_indices = lookup_indices
        self.search_args = search_args

    def _sort(self, dictionary):
        ranked = [int(doc) for doc, _ in sorted(dictionary.items(), key=lambda kv: kv[1], reverse=True)]
        return {k: ranked[:k] for k in self.recall_k_setting_list}

    def _count(self, counts, search_results, target):
        for k, top in search_results.items():
            if target.intersection(top):
                counts[k] += 1

    def _target(self, qid):
        t = self.dataset.get_target(qid, self.search_args.query_type)
        return {int(i) for i in t} if isinstance(t, list) else {int(t)}

    def sort_and_count(self):
        for run, counts, has_docs_level in ((self.dense_run, self.dense_counts, True),
                                            (self.sparse_run, self.sparse_counts, True),
                                            (self.fusion_run, self.fusion_counts, False)):
            for qid, v in run.items():
                target = self._target(qid)
                docs = v["docs"] if has_docs_level else v
                if len(docs) == 0:
                    continue
                self._count(counts, self._sort(docs), target)

    def all_gather_object(self):
        denom = self.denominator if self.denominator is not None else len(self.lookup_indices) * self.world_size
        denom = max(denom, 1)
        d = self._dist
        for counts_name, lists in (("dense_counts", self.dense_recall_lists), ("sparse_counts", self.sparse_recall_lists),
                                   ("fusion_counts", self.fusion_recall_lists)):
            counts = {k: getattr(self, counts_name)[k] / denom for k in self.recall_k_setting_list}
            setattr(self, counts_name, counts)
            for k in self.recall_k_setting_list:
                if d:
                    d.all_gather_object(object_list=lists[k], obj=counts[k])
                else:
                    lists[k][self.rank] = counts[k]  # (a replayed rank fills its own slot)

    def recalls(self):
        """{'dense'|'sparse'|'fusion': {k: recall summed over ranks}} for the runs that are present."""
        out = {}
        for name, run, lists in (("dense", self.dense_run, self.dense_recall_lists),
                                 ("sparse", self.sparse_run, self.sparse_recall_lists),
                                 ("fusion", self.fusion_run, self.fusion_recall_lists)):
            if len(run) > 0:
                out[name] = {k: sum(lists[k]) for k in self.recall_k_setting_list}
        return out

    def print_recall(self):
        if self.rank != 0:
            return
        print(self.denominator if self.denominator is not None else len(self.lookup_indices) * self.world_size)
        labels = (("dense", "Dense recall @ {}: {}", "Dense reps recall", self.dense_run, self.dense_recall_lists),
                  ("sparse", "Sparse recall @ {}: {}", "Sparse reps recall", self.sparse_run, self.sparse_recall_lists),
                  ("fusion", "Fusion/Hybrid recall @ {}: {}", "Fusion/Hybrid reps recall", self.fusion_run,
                   self.fusion_recall_lists))
        for name, per_k, summary, run, lists in labels:
            if len(run) == 0:
                continue
            if name == "dense":
                print(len(self.look_up))
            total = {k: sum(lists[k]) for k in self.recall_k_setting_list}
            for k in self.recall_k_setting_list:
                print(per_k.format(k, lists[k]))
            print("{}: r@1 {}, r@5 {}, r@10 {}, r@100 {}, r@200 {}".format(summary, total[1], total[5], total[10],
                                                                           total[100], total[200]))


def replay_ranks(dataset, dense_run, sparse_run, fusion_run, look_up, query_ids, search_args, world_size, compat=True):
    """Recall of complete runs as `world_size` ranks of the reference would have reported it (src/search.py:180-182 +
    src/metrices.py:86-100), replayed in one process: rank r owns the DistributedSampler shard of `query_ids`
    (padding included), counts its hits and divides by len(shard) * world_size (compat) or by the true query count.
    Returns the rank-0 RecallMetrics object with every rank's fractions in its lists (print_recall() works on it)."""
    from .sampler import shard_query_ids

    n = len(query_ids)
    first = None
    for r in range(world_size):
        shard = shard_query_ids(query_ids, world_size, r) if world_size > 1 else list(query_ids)
        if not compat:  # the true denominator: every query once — drop the sampler's padding
            total = -(-n // world_size) * world_size
            shard = [q for j, q in enumerate(shard) if r + j * world_size < n] if total != n else shard
        mine = set(shard)
        pick = lambda run: {q: v for q, v in run.items() if q in mine}  # noqa: E731
        m = RecallMetrics(dataset, pick(dense_run), pick(sparse_run), pick(fusion_run), look_up, shard, search_args,
                          denominator=None if compat else n, world_size=world_size, rank=r)
        m.sort_and_count()
        m.all_gather_object()
        if first is None:
            first = m
        else:
            for lists_name in ("dense_recall_lists", "sparse_recall_lists", "fusion_recall_lists"):
                for k in m.recall_k_setting_list:
                    getattr(first, lists_name)[k][r] = getattr(m, lists_name)[k][r]
    # runs that are empty on rank 0 but not overall still print
    first.dense_run, first.sparse_run, first.fusion_run = dense_run, sparse_run, fusion_run
    return first
