"""world_size-2 gloo tests (CPU) of the multi-rank plumbing: shard ranges, the list all-gather, the merge rule,
and the recall gather. The shard-local scoring is played by the oracle restricted to the shard's docs; on the GPU
box the same plumbing is driven by libmsr.so (tests/test_gpu_parity.py::test_resident_batch_and_shards)."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, idx_path, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mllm_sparse_retrieval_amd as m
        from mllm_sparse_retrieval_amd import dist as mdist
        from mllm_sparse_retrieval_amd.qrels import CrossModalQrels
        from mllm_sparse_retrieval_amd.recall import RecallMetrics
        from oracle import oracle
        from tests import helpers

        docs, (qp, qt, qw) = helpers.synth(9000, 16, 40, 8, 300, seed=3)
        ix = m.SparseIndex(idx_path, device=-1, shard=rank, n_shards=world)        # host-only shard handle
        t0, t1 = mdist.shard_tile_range(ix.n_tiles, rank, world)
        assert (ix.shard_tile0, ix.shard_ntiles) == (t0, t1 - t0)
        # shard-local exact top-k from the oracle: keep only docs whose ordinal falls into this shard
        oi = oracle.OracleIndex.from_csr(*docs, 300)
        lo, hi = t0 * ix.tile_docs, min(t1 * ix.tile_docs, ix.n_docs)
        D = oi.D.tolil()
        mask = np.ones(oi.n_docs, bool)
        mask[lo:hi] = False
        D[np.flatnonzero(mask)] = 0
        oi_shard = oracle.OracleIndex.__new__(oracle.OracleIndex)
        oi_shard.__dict__.update(oi.__dict__)
        oi_shard.D = D.tocsr()
        queries = [{int(t): int(w) for t, w in zip(qt[qp[i]:qp[i + 1]], qw[qp[i]:qp[i + 1]])} for i in range(40)]
        oi_shard.df = oi.df  # the df == N filter uses GLOBAL df on every shard
        ords, scores, n = oracle.search(oi_shard, queries, 10)
        g = mdist.all_gather_lists(dist, np.where(ords < 0, 0, ords).astype(np.uint32), scores.astype(np.uint32),
                                   n.astype(np.int32))
        assert g[0].shape == (world, 40, 10) and (g[2][rank] == n).all()
        mo, ms, mn = mdist.merge_lists_host(*g, 10)
        wo, ws, wn = oracle.search(oi, queries, 10)
        assert (mn == wn).all()
        valid = np.arange(10)[None, :] < wn[:, None]
        assert (mo.astype(np.int64)[valid] == wo[valid]).all() and (ms.astype(np.int64)[valid] == ws[valid]).all()

        # recall gather: each rank scores half of the queries (DP over queries, src/search.py:180-182)
        q = CrossModalQrels.synthetic(4, 2)
        mine = [str(i) for i in range(8) if i % world == rank]
        run = {qid: {"docs": {str(int(qid) // 2): 3.0} if int(qid) < 6 else {"0": 1.0}} for qid in mine}
        rm = RecallMetrics(q, {}, run, {}, [], mine, SimpleNamespace(query_type="text"))
        rm.sort_and_count()
        rm.all_gather_object()
        r = rm.recalls()["sparse"]
        assert abs(r[1] - 6 / 8) < 1e-12 and len(rm.sparse_recall_lists[1]) == world
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_and_recall(built, tmp_path):
    import mllm_sparse_retrieval_amd as m
    from tests import helpers

    docs, _ = helpers.synth(9000, 16, 40, 8, 300, seed=3)
    idx = m.build_index_from_csr(str(tmp_path / "d.idx"), *docs, 300, tile_docs=4096)
    mp.spawn(_worker, args=(2, _free_port(), idx, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_merge_rule_ties():
    from mllm_sparse_retrieval_amd import dist as mdist

    ords = np.array([[[5, 9, 0]], [[2, 7, 0]]], dtype=np.uint32)
    sc = np.array([[[8, 3, 0]], [[8, 3, 0]]], dtype=np.uint32)
    n = np.array([[2], [2]], dtype=np.int32)
    o, s, c = mdist.merge_lists_host(ords, sc, n, 3)
    assert o[0].tolist() == [2, 5, 7] and s[0].tolist() == [8, 8, 3] and c[0] == 3
