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
    def __init__(self, dataset, dense_run, sparse_run, fusion_run, look_up, lookup_indices, search_args):
        self.recall_k_setting_list = list(RECALL_KS)
        d = _dist()
        self.world_size = d.get_world_size() if d else 1
        self.rank = d.get_rank() if d else 0
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
        denom = len(self.lookup_indices) * self.world_size
        d = _dist()
        for counts_name, lists in (("dense_counts", self.dense_recall_lists), ("sparse_counts", self.sparse_recall_lists),
                                   ("fusion_counts", self.fusion_recall_lists)):
            counts = {k: getattr(self, counts_name)[k] / denom for k in self.recall_k_setting_list}
            setattr(self, counts_name, counts)
            for k in self.recall_k_setting_list:
                if d:
                    d.all_gather_object(object_list=lists[k], obj=counts[k])
                else:
                    lists[k][0] = counts[k]

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
        print(len(self.lookup_indices) * self.world_size)
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
