"""Multi-GPU search over doc-range shards of one index: one process per GPU, one exchange per query batch.

The reference never shards its sparse index (one Lucene index per rank, src/search.py:216,273); its only parallelism
on this path is DP over queries (src/search.py:180-182). Sharding is the north-star extension:
rank r holds tiles [T*r/R, T*(r+1)/R) of the tile-major index, scores every query against its docs, and the per-shard
exact top-k lists are merged after ONE all-gather. Doc-range shards make that merge exact by construction: the global
top-k is a subset of the union of the per-shard top-k lists.

Two exchanges:
  "rccl"  : libmsr.so calls ncclAllGather on its own stream over xGMI (msr_batch_search_sharded); the host only
            ships the 128-byte communicator id.
  "torch" : per-shard lists are gathered with torch.distributed (gloo on CPU, nccl = RCCL on GPU) and merged by
            msr_merge_lists on the GPU. Used by the CPU (gloo) tests of the plumbing and as a diagnostic cross-check.
"""
from __future__ import annotations

import numpy as np

from .index import SparseIndex, comm_unique_id


def shard_tile_range(n_tiles, shard, n_shards):
    """Tiles [t0, t1) of shard `shard` — mirrors open_common() in csrc/msr_index.cpp."""
    return (n_tiles * shard) // n_shards, (n_tiles * (shard + 1)) // n_shards


def all_gather_lists(dist, ords, scores_u32, n, device=None):
    """Gather every rank's [nq,k] result arrays -> ([R,nq,k], [R,nq,k], [R,nq]) on every rank."""
    import torch

    world = dist.get_world_size()

    def gather(a, dtype):
        t = torch.from_numpy(np.ascontiguousarray(a).astype(dtype))
        if device is not None:
            t = t.to(device)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.stack([o.cpu().numpy() for o in outs])

    # int64 carriers: gloo/nccl have no uint32
    return (gather(ords, np.int64).astype(np.uint32), gather(scores_u32, np.int64).astype(np.uint32),
            gather(n, np.int64).astype(np.int32))


def merge_lists_host(ords, scores_u32, n, k):
    """Reference merge on the host (numpy) with the same tie rule — checker for tests, not used by the product."""
    R, nq, _ = ords.shape
    o = np.full((nq, k), 0xFFFFFFFF, dtype=np.uint32)
    s = np.zeros((nq, k), dtype=np.uint32)
    cnt = np.zeros(nq, dtype=np.int32)
    for q in range(nq):
        items = [(-int(scores_u32[r, q, j]), int(ords[r, q, j])) for r in range(R) for j in range(int(n[r, q]))]
        items.sort()
        items = items[:k]
        cnt[q] = len(items)
        for j, (ns, d) in enumerate(items):
            o[q, j] = d
            s[q, j] = -ns
    return o, s, cnt


class ShardedSearcher:
    """Rank-local view of a doc-range sharded index; search_csr returns the GLOBAL top-k on every rank."""

    def __init__(self, index_path, rank, world_size, device, exchange="rccl", dist=None):
        self.rank, self.world = rank, world_size
        self.exchange = exchange
        self.dist = dist
        self.index = SparseIndex(index_path, device=device, shard=rank, n_shards=world_size)
        if exchange == "rccl" and world_size >= 1:
            uid = comm_unique_id() if rank == 0 else None
            if world_size > 1:
                if dist is None:
                    raise ValueError("a torch.distributed module is needed to ship the communicator id")
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                uid = box[0]
            self.index.comm_init(world_size, rank, uid)

    def search_csr(self, q_ptr, q_term, q_w, k, drop_df_eq_n=True):
        batch = self.index.batch(q_ptr, q_term, q_w, k, drop_df_eq_n)
        try:
            if self.exchange == "rccl":
                batch.search(k, sharded=True)
                return batch.fetch()
            batch.search(k)
            ords, _, su, n = batch.fetch()
        finally:
            batch.close()
        if self.world == 1:
            g = (ords[None], su[None], n[None])
        else:
            g = all_gather_lists(self.dist, ords, su, n)
        return self.index.merge_lists(g[0], g[1], g[2], k)

    def close(self):
        if self.exchange == "rccl":
            self.index.comm_destroy()
        self.index.close()
