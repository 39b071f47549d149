"""GPU parity tests proper: the HIP path, called through the C-ABI, against the oracle on the same seeded inputs.

Bar: bit-exact doc ordinals and exact integer scores; f32 scores within 1e-5 (north_star) — in fact equal.
"""
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m(built):
    import mllm_sparse_retrieval_amd as m

    return m


def _case(m, tmp_path, n_docs, doc_nnz, nq, q_nnz, n_terms, seed, tile_docs, ks, doc_ids=None):
    docs, (qp, qt, qw) = helpers.synth(n_docs, doc_nnz, nq, q_nnz, n_terms, seed)
    path = m.build_index_from_csr(str(tmp_path / "c.idx"), *docs, n_terms, doc_ids=doc_ids, tile_docs=tile_docs)
    oix, _ = helpers.taat_oracle(docs, n_terms, doc_ids)
    with m.SparseIndex(path, device=0) as ix:
        for k in ks:
            got = ix.search_csr(qp, qt, qw, k)
            want = oix.search(qp, qt, qw, k, threads=8)
            helpers.assert_same_results(got, want, k)


@pytest.mark.parametrize("tile_docs", [4096, 8192, 16384, 32768])
def test_single_and_multi_tile(m, tmp_path, tile_docs):
    # 20k docs: 5 / 3 / 2 / 1 tiles, last tile ragged
    _case(m, tmp_path, 20000, 64, 200, 40, 5000, seed=21, tile_docs=tile_docs, ks=[1, 10, 100])


def test_flickr_shape_c2(m, tmp_path):
    # BASELINE config 2: 1 000 docs x 128 nnz, 5 000 queries, V = 32 064, top-10
    _case(m, tmp_path, 1000, 128, 5000, 12, 32064, seed=1, tile_docs=0, ks=[10])


def test_coco_shape_c3_depth1000(m, tmp_path):
    # COCO-5K t->i shape with the hybrid script's depth 1000 (scripts/search.sh:25): large-k path
    _case(m, tmp_path, 5000, 128, 300, 120, 30000, seed=2, tile_docs=0, ks=[10, 1000])


def test_many_query_terms(m, tmp_path):
    # > 256 terms per query exercises the staged term rounds
    _case(m, tmp_path, 9000, 32, 50, 700, 4000, seed=5, tile_docs=4096, ks=[10, 37])


def test_heavy_ties_and_string_order(m, tmp_path):
    # every doc holds the same single term with the same weight -> all scores tie; order must be doc-id STRING order
    n, V = 10000, 8
    dp = np.arange(n + 1, dtype=np.uint64)
    dt = np.full(n, 3, dtype=np.uint32)
    dw = np.full(n, 7, dtype=np.uint32)
    dt[::2] = 4  # half the docs hold term 4 instead, so neither term has df == N
    ids = [str(i) for i in range(n)]
    path = m.build_index_from_csr(str(tmp_path / "t.idx"), dp, dt, dw, V, doc_ids=ids, tile_docs=4096)
    qp = np.array([0, 1, 3], dtype=np.int64)
    qt = np.array([3, 3, 4], dtype=np.int32)
    qw = np.array([2, 1, 1], dtype=np.int32)
    oix, _ = helpers.taat_oracle((dp, dt, dw), V, ids)
    with m.SparseIndex(path, device=0) as ix:
        for k in (1, 10, 300, 1000):
            got = ix.search_csr(qp, qt, qw, k)
            helpers.assert_same_results(got, oix.search(qp, qt, qw, k), k)
        ords, _, su, cnt = ix.search_csr(qp, qt, qw, 5)
        got_ids = ix.docids(ords[0, : cnt[0]])
    odd = sorted((str(i) for i in range(1, n, 2)), key=lambda s: s.encode())
    assert got_ids == odd[:5] and (su[0] == 14).all()


def test_edge_queries(m, tmp_path):
    docs, _ = helpers.synth(3000, 16, 1, 4, 500, seed=9)
    path = m.build_index_from_csr(str(tmp_path / "e.idx"), *docs, 500, tile_docs=4096)
    oix, _ = helpers.taat_oracle(docs, 500)
    # empty query, all-OOV query, zero/negative weights, duplicate terms (add), one ordinary query
    qp = np.array([0, 0, 2, 5, 8, 10], dtype=np.int64)
    qt = np.array([-1, -1, 7, 7, 9, 11, 11, 11, 3, 4], dtype=np.int32)
    qw = np.array([5, 9, 0, -3, 4, 2, 3, 1, 6, 2], dtype=np.int32)
    with m.SparseIndex(path, device=0) as ix:
        got = ix.search_csr(qp, qt, qw, 10)
        helpers.assert_same_results(got, oix.search(qp, qt, qw, 10), 10)
        assert got[3][0] == 0 and got[3][1] == 0
        # out-of-range term id and k are refused, not computed
        with pytest.raises(Exception):
            ix.search_csr(np.array([0, 1]), np.array([500]), np.array([1]), 10)
        with pytest.raises(Exception):
            ix.search_csr(qp, qt, qw, 2000)
        # nq = 0
        e = ix.search_csr(np.array([0]), np.array([], dtype=np.int32), np.array([], dtype=np.int32), 10)
        assert e[0].shape == (0, 10)


def test_df_eq_n_switch(m, tmp_path):
    # term 0 is in every doc: dropped by default (contract T3), kept when the switch is off
    n, V = 500, 20
    rng = np.random.default_rng(5)
    rows = [[(0, int(rng.integers(1, 50)))] + [(int(t), int(rng.integers(1, 50))) for t in rng.choice(np.arange(1, V), 3, replace=False)]
            for _ in range(n)]
    dp = np.arange(0, 4 * n + 1, 4, dtype=np.uint64)
    dt = np.array([t for r in rows for t, _ in r], dtype=np.uint32)
    dw = np.array([w for r in rows for _, w in r], dtype=np.uint32)
    path = m.build_index_from_csr(str(tmp_path / "d.idx"), dp, dt, dw, V, tile_docs=4096)
    oix, _ = helpers.taat_oracle((dp, dt, dw), V)
    qp = np.array([0, 2, 3], dtype=np.int64)
    qt = np.array([0, 5, 0], dtype=np.int32)
    qw = np.array([3, 2, 1], dtype=np.int32)
    with m.SparseIndex(path, device=0) as ix:
        for drop in (True, False):
            got = ix.search_csr(qp, qt, qw, 10, drop_df_eq_n=drop)
            helpers.assert_same_results(got, oix.search(qp, qt, qw, 10, drop_df_eq_n=drop), 10)
        assert ix.search_csr(qp, qt, qw, 10, drop_df_eq_n=True)[3][1] == 0


def test_overflow_is_refused(m, tmp_path):
    dp = np.array([0, 1, 2], dtype=np.uint64)
    dt = np.array([0, 1], dtype=np.uint32)
    dw = np.array([65535, 65535], dtype=np.uint32)
    path = m.build_index_from_csr(str(tmp_path / "o.idx"), dp, dt, dw, 2, tile_docs=4096)
    with m.SparseIndex(path, device=0) as ix:
        with pytest.raises(Exception, match="OVERFLOW"):
            ix.search_csr(np.array([0, 1]), np.array([0]), np.array([70000]), 1)
        ords, f32, u32, n = ix.search_csr(np.array([0, 1]), np.array([0]), np.array([65535]), 1)
        assert n[0] == 1 and u32[0, 0] == 65535 * 65535 and f32[0, 0] == np.float32(65535 * 65535)


def test_resident_batch_and_shards(m, tmp_path):
    # doc-range shards searched one by one on the same GPU, merged by msr_merge_lists == unsharded result
    docs, (qp, qt, qw) = helpers.synth(30000, 32, 300, 30, 3000, seed=31)
    path = m.build_index_from_csr(str(tmp_path / "s.idx"), *docs, 3000, tile_docs=4096)
    oix, _ = helpers.taat_oracle(docs, 3000)
    want = oix.search(qp, qt, qw, 10, threads=8)
    with m.SparseIndex(path, device=0) as ix:
        b = ix.batch(qp, qt, qw, 16)
        b.search(10)
        helpers.assert_same_results(b.fetch(), want, 10)
        ms = b.kernel_ms()
        by, po = b.algo_bytes(10)
        assert ms[0] > 0 and by > po * 6
        b.close()
        lists = []
        for s in range(3):
            with m.SparseIndex(path, device=0, shard=s, n_shards=3) as sh:
                assert sh.shard_ntiles in (2, 3)
                lists.append(sh.search_csr(qp, qt, qw, 10))
        o, sf, su, n = ix.merge_lists(np.stack([l[0] for l in lists]), np.stack([l[2] for l in lists]),
                                      np.stack([l[3] for l in lists]), 10)
        helpers.assert_same_results((o, sf, su, n), want, 10)


def test_rccl_single_rank(m, tmp_path):
    # the RCCL exchange path with a 1-rank communicator (all a 1-GPU box can run)
    docs, (qp, qt, qw) = helpers.synth(12000, 32, 100, 30, 3000, seed=41)
    path = m.build_index_from_csr(str(tmp_path / "r.idx"), *docs, 3000, tile_docs=4096)
    oix, _ = helpers.taat_oracle(docs, 3000)
    with m.SparseIndex(path, device=0) as ix:
        ix.comm_init(1, 0, m.comm_unique_id())
        b = ix.batch(qp, qt, qw, 10)
        b.search(10, sharded=True)
        helpers.assert_same_results(b.fetch(), oix.search(qp, qt, qw, 10), 10)
        b.close()
        ix.comm_destroy()
