"""The query split of the reference's search loop, restated without a DataLoader:

    sampler = DistributedSampler(dataset, num_replicas=world_size, shuffle=True, rank=rank)      src/search.py:180
    (no set_epoch -> seed 0, epoch 0)

Rank r searches dataset positions perm[r::world] of a torch.randperm(n, seed 0) permutation PADDED with its own head
up to ceil(n / world) * world entries — so with n % world != 0 a few queries are searched (and counted) twice, and
the reference's recall denominator is len(lookup_indices) * world = the padded total (src/metrices.py:92; defect 6 of
SURVEY.md §3.1). `eval --compat-denominator` and the multi-rank `search` command reproduce exactly this split.
"""
from __future__ import annotations

import math


def distributed_sampler_indices(n, world, rank, shuffle=True, seed=0, epoch=0):
    """Dataset positions rank `rank` of `world` iterates, in order (torch.utils.data.DistributedSampler, drop_last=False)."""
    if shuffle:
        import torch  # the permutation must be torch's own generator stream to match the reference's ranks

        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = math.ceil(n / world) * world if n else 0
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * math.ceil(pad / max(len(idx), 1)))[:pad]
    return idx[rank:total:world]


def shard_query_ids(query_ids, world, rank, shuffle=True, seed=0):
    return [query_ids[i] for i in distributed_sampler_indices(len(query_ids), world, rank, shuffle, seed)]
