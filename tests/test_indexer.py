"""The product's host-side indexer (jsonl reader + tile-major builder, no GPU) against the oracle's reading of the
same files, decoded from the index file by an independent reader (tests/helpers.py)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sparse_small")


@pytest.fixture(scope="module")
def m(built):
    import mllm_sparse_retrieval_amd as m

    return m


def test_jsonl_index_equals_oracle(m, tmp_path):
    out = m.build_index_from_jsonl(GOLD, str(tmp_path / "g.idx"), threads=3)
    ixf = helpers.read_index_file(out)
    oi = oracle.OracleIndex(oracle.read_corpus_dir(GOLD))
    assert ixf["docs"] == oi.doc_ids
    assert ixf["terms"] == oi.vocab
    assert (ixf["df"] == oi.df).all()
    assert (helpers.index_file_to_dense(ixf) == oi.D.toarray()).all()
    assert ixf["n_postings"] == oi.D.nnz and ixf["tile_docs"] == 4096
    assert (ixf["maxw"] == np.asarray(oi.D.max(axis=0).todense()).ravel()).all()


@pytest.mark.parametrize("threads", [1, 2, 7])
def test_thread_count_does_not_change_the_file(m, tmp_path, threads):
    a = m.build_index_from_jsonl(GOLD, str(tmp_path / "a.idx"), threads=1)
    b = m.build_index_from_jsonl(GOLD, str(tmp_path / f"b{threads}.idx"), threads=threads)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_multi_tile_csr_index(m, tmp_path):
    docs, _ = helpers.synth(9000, 16, 1, 4, 300, seed=3)
    ids = [str(i * 7) for i in range(9000)]
    out = m.build_index_from_csr(str(tmp_path / "c.idx"), *docs, 300, doc_ids=ids, tile_docs=4096)
    ixf = helpers.read_index_file(out)
    oi = oracle.OracleIndex.from_csr(*docs, 300, ids)
    assert ixf["n_tiles"] == 3 and ixf["docs"] == oi.doc_ids
    assert (helpers.index_file_to_dense(ixf) == oi.D.toarray()).all()
    # every segment starts on a 16-byte vec and tiles are contiguous
    assert (np.diff(ixf["seg_ptr"].reshape(-1).astype(np.int64)) >= 0).all()
    with m.SparseIndex(out, device=-1) as ix:
        assert ix.n_docs == 9000 and ix.n_tiles == 3
        assert ix.docid(0) == oi.doc_ids[0] and ix.docid(8999) == oi.doc_ids[-1]
        assert ix.lookup(["17", "299", "300", "nope"]).tolist() == [17, 299, -1, -1]
        assert (ix.df(np.arange(300)) == oi.df).all()
    for s in range(3):
        with m.SparseIndex(out, device=-1, shard=s, n_shards=3) as sh:
            assert (sh.shard_tile0, sh.shard_ntiles) == (s, 1)


def test_repeated_term_in_csr_row_adds(m, tmp_path):
    dp = np.array([0, 3, 4], dtype=np.uint64)
    dt = np.array([2, 2, 1, 2], dtype=np.uint32)
    dw = np.array([5, 6, 1, 0], dtype=np.uint32)
    out = m.build_index_from_csr(str(tmp_path / "r.idx"), dp, dt, dw, 3, tile_docs=4096)
    D = helpers.index_file_to_dense(helpers.read_index_file(out))
    assert D.tolist() == [[0, 1, 11], [0, 0, 0]]


def _write(tmp_path, lines):
    d = tmp_path / "corpus"
    d.mkdir()
    (d / "corpus_0.jsonl").write_text("\n".join(lines) + "\n", encoding="utf-8")
    return str(d)


@pytest.mark.parametrize("bad,msg", [
    ('{"id": "1", "vector": {"a": 1}', "corpus_0.jsonl:2"),
    ('{"id": "1"}', "missing \"vector\""),
    ('{"vector": {"a": 1}}', "missing \"id\""),
    ('{"id": "1", "vector": {"a": "x"}}', "expected number"),
    ('{"id": "1", "vector": {"a": 70000}}', "65535"),
    ('[1, 2]', "not a JSON object"),
    ('{"id": "1", "vector": {"a": 1}} trailing', "trailing"),
])
def test_malformed_input_is_refused(m, tmp_path, bad, msg):
    d = _write(tmp_path, ['{"id": "0", "content": "", "vector": {"ok": 1}}', bad])
    with pytest.raises(Exception) as e:
        m.build_index_from_jsonl(d, str(tmp_path / "x.idx"), threads=2)
    assert msg in str(e.value)


def test_unicode_and_escapes(m, tmp_path):
    lines = [json.dumps({"id": "a", "content": "", "vector": {"ġdog": 3, "▁chat": 2, "emoji\U0001F600": 1}}),
             '{"id": "b", "content": "", "vector": {"tab\\there": 4, "q\\"uote": 5, "sl\\/ash": 6}}']
    d = _write(tmp_path, lines)
    out = m.build_index_from_jsonl(d, str(tmp_path / "u.idx"), threads=1)
    ixf = helpers.read_index_file(out)
    oi = oracle.OracleIndex(oracle.read_corpus_dir(d))
    assert ixf["terms"] == oi.vocab and set(ixf["terms"]) >= {"ġdog", "▁chat", "emoji\U0001F600", "tab", "here",
                                                              'q"uote', "sl/ash"}
    assert (helpers.index_file_to_dense(ixf) == oi.D.toarray()).all()


def test_empty_vectors_and_no_files(m, tmp_path):
    d = _write(tmp_path, ['{"id": "0", "content": "", "vector": {}}', '{"id": "1", "content": "", "vector": {"a": 2}}'])
    ixf = helpers.read_index_file(m.build_index_from_jsonl(d, str(tmp_path / "e.idx")))
    assert ixf["n_docs"] == 2 and ixf["n_postings"] == 1
    empty = tmp_path / "none"
    empty.mkdir()
    with pytest.raises(Exception, match="no \\*.jsonl"):
        m.build_index_from_jsonl(str(empty), str(tmp_path / "n.idx"))
    with pytest.raises(Exception):
        m.SparseIndex(str(tmp_path / "missing.idx"), device=-1)
    (tmp_path / "junk.idx").write_bytes(b"not an index" * 50)
    with pytest.raises(Exception, match="MSRIDX01"):
        m.SparseIndex(str(tmp_path / "junk.idx"), device=-1)


def test_synth_is_deterministic_and_shaped(m):
    a = m.synth_vectors(500, 32, 1000, seed=5, threads=1)
    b = m.synth_vectors(500, 32, 1000, seed=5, threads=7)
    assert all((x == y).all() for x, y in zip(a, b))
    p, t, w = a
    assert (np.diff(p) == 32).all() and w.min() >= 1 and w.max() <= 400
    rows = t.reshape(500, 32)
    assert all(len(set(r)) == 32 for r in rows) and (np.diff(rows, axis=1) > 0).all()
    assert (t == 0).mean() > (t == 500).mean()  # Zipf: low ranks dominate


def test_tie_order_switch_numbers_docs_in_input_order(tmp_path):
    """Build option tie_order = 1: doc ordinals follow the INPUT order (a score tie goes to the doc indexed first)
    instead of the doc-id string order of contract T1 — the switch an integrator flips when checking against pyserini."""
    import mllm_sparse_retrieval_amd as m

    ids = ["10", "9", "100", "2"]
    dp = np.arange(5, dtype=np.uint64)
    dt = np.zeros(4, dtype=np.uint32)
    dw = np.full(4, 3, dtype=np.uint32)
    a = m.build_index_from_csr(str(tmp_path / "a.idx"), dp, dt, dw, 2, doc_ids=ids, tile_docs=4096)
    m.set_build_option("tie_order", 1)
    try:
        b = m.build_index_from_csr(str(tmp_path / "b.idx"), dp, dt, dw, 2, doc_ids=ids, tile_docs=4096)
    finally:
        m.set_build_option("tie_order", 0)
    assert helpers.read_index_file(a)["docs"] == sorted(ids, key=lambda s: s.encode())   # "10" < "100" < "2" < "9"
    assert helpers.read_index_file(b)["docs"] == ids


def test_encode_queries_in_c_equals_the_python_tokeniser(m, tmp_path):
    """msr_encode_queries (query strings -> CSR, tokenised and looked up in C, on several threads from 64 queries on)
    against the Python restatement of pyserini's token-frequency encoding (searcher.tokenize_queries + lookup): tokens
    repeated `weight` times in a row (src/search.py:419-422), repeats that are NOT adjacent, out-of-vocabulary tokens,
    every ASCII whitespace, empty strings. No GPU needed: the handle is opened with device = -1."""
    from mllm_sparse_retrieval_amd.searcher import tokenize_queries

    rng = np.random.default_rng(3)
    vocab = [f"t{i}" for i in range(200)] + ["ġdog", "▁cat", "a-b", "7"]
    n_terms = len(vocab)
    dp = np.arange(0, 2 * 300 + 1, 2, dtype=np.uint64)
    dt = rng.integers(0, n_terms, 600).astype(np.uint32)
    dt[:n_terms] = np.arange(n_terms)                      # every term occurs (else it is not in the dictionary)
    dw = np.ones(600, dtype=np.uint32)
    path = m.build_index_from_csr(str(tmp_path / "q.idx"), dp, dt, dw, n_terms, term_strs=vocab, tile_docs=4096)
    queries = []
    for i in range(150):
        toks = []
        for _ in range(int(rng.integers(0, 12))):
            tok = vocab[int(rng.integers(0, n_terms))] if rng.random() < 0.8 else f"oov{int(rng.integers(0, 5))}"
            toks += [tok] * int(rng.integers(1, 6))         # the reference writes a token `weight` times in a row
        if rng.random() < 0.3:
            rng.shuffle(toks)                               # ... and a query whose repeats are scattered
        sep = [" ", "  ", "\t", "\n ", " \r"][i % 5]
        queries.append(sep.join(toks) + ("  " if i % 7 == 0 else ""))
    queries[10] = ""
    queries[11] = "   \t "
    with m.SparseIndex(path, device=-1) as ix:
        for qs in (queries, queries[:5], []):               # threaded, serial, empty
            q_ptr, q_term, q_w = ix.encode_queries(qs)
            w_ptr, w_toks, w_w = tokenize_queries(qs)
            w_term = ix.lookup(w_toks)
            assert len(q_ptr) == len(qs) + 1 and q_ptr[0] == 0
            for i in range(len(qs)):
                got = dict(zip(q_term[q_ptr[i]:q_ptr[i + 1]].tolist(), q_w[q_ptr[i]:q_ptr[i + 1]].tolist()))
                want = {}
                for t, w in zip(w_term[w_ptr[i]:w_ptr[i + 1]].tolist(), w_w[w_ptr[i]:w_ptr[i + 1]].tolist()):
                    if t >= 0:                              # the C encoder drops out-of-vocabulary tokens (contract T2)
                        want[t] = want.get(t, 0) + w
                assert got == want, (i, qs[i])
                assert len(got) == q_ptr[i + 1] - q_ptr[i]   # one entry per distinct term
