"""Host logic that mirrors the reference's own Python around the searcher: run dicts, fusion, TREC IO, recall.

fuse / write_trec_run / read_trec_run are pinned against outputs of the REFERENCE's src/hybrid.py
(tests/golden/hybrid_golden.json, produced by tests/golden/make_hybrid_golden.py in the build container)."""
import json
import os
from types import SimpleNamespace

import pytest

from mllm_sparse_retrieval_amd import fusion, run
from mllm_sparse_retrieval_amd.qrels import CrossModalQrels
from mllm_sparse_retrieval_amd.recall import RecallMetrics
from mllm_sparse_retrieval_amd.searcher import Hit, tokenize_queries
from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hybrid_cases():
    return json.load(open(os.path.join(GOLD, "hybrid_golden.json")))["cases"]


def test_fuse_matches_reference_bit_for_bit(hybrid_cases):
    for c in hybrid_cases:
        got = fusion.fuse([c["dense"], c["sparse"]], c["weights"])
        assert got == c["fused"]                       # float equality: same operations in the same order
        assert oracle.fuse([c["dense"], c["sparse"]], c["weights"]) == c["fused"]
        for qid in got:                                # union order: dense docs first, then sparse-only docs
            assert list(got[qid]) == list(c["fused"][qid])


def test_trec_io_matches_reference(hybrid_cases, tmp_path):
    for i, c in enumerate(hybrid_cases):
        f = tmp_path / f"s{i}.trec"
        fusion.write_trec_run(c["sparse"], str(f), name="sparse")
        assert f.read_text() == c["sparse_trec"]
        assert fusion.read_trec_run(str(f)) == c["sparse_trec_read"]
        g = tmp_path / f"f{i}.trec"
        fusion.write_trec_run(c["fused"], str(g))
        assert g.read_text() == c["fused_trec"]


class _FakeSearcher:
    def __init__(self, table):
        self.table = table
        self.calls = []

    def batch_search(self, topics, ids, depth, threads=1):
        self.calls.append((list(topics), list(ids), depth, threads))
        return {qid: [Hit(d, s) for d, s in self.table[qid][:depth]] for qid in ids}


def test_sparse_search_and_run_dict():
    table = {"7": [("3", 9.0), ("7", 8.0), ("5", 8.0)], "8": [], "9": [("1", 2.0)]}
    s = _FakeSearcher(table)
    args = SimpleNamespace(depth=2, threads=16)
    scores, rankings = run.sparse_search(s, ["a", "b", "c"], ["9", "7", "8"], args)
    assert s.calls == [(["a", "b", "c"], ["9", "7", "8"], 2, 16)]
    assert rankings == [["1"], ["3", "7"], []] and scores == [[2.0], [9.0, 8.0], []]
    rd = run.get_run_dict(["9", "7", "8"], scores, rankings, remove_query=True)
    assert rd["7"] == {"docs": {"3": 9.0}, "min_score": 8.0, "max_score": 9.0}   # own id dropped, min over ALL scores
    assert rd["8"] == {"docs": {}, "min_score": 0, "max_score": 0}
    assert run.get_run_dict(["7"], [[9.0, 8.0]], [["3", "7"]], False)["7"]["docs"] == {"3": 9.0, "7": 8.0}
    assert rd == oracle.get_run_dict(["9", "7", "8"], scores, rankings, True)


def test_query_tokenization_counts_repeats():
    q_ptr, toks, ws = tokenize_queries(["dog dog  cat", "", "a\tb a\n"])
    assert q_ptr.tolist() == [0, 2, 2, 4]
    assert toks == ["dog", "cat", "a", "b"] and ws.tolist() == [2, 1, 2, 1]
    assert oracle.encode_query(oracle.query_string(["x", "y", "z"], [2, 0, 1])) == {"x": 2, "z": 1}


def _metrics(runs, qrels, lookup, qtype="text"):
    m = RecallMetrics(qrels, runs.get("dense", {}), runs.get("sparse", {}), runs.get("fusion", {}), [0] * 3, lookup,
                      SimpleNamespace(query_type=qtype))
    m.sort_and_count()
    m.all_gather_object()
    return m


def test_recall_hand_computed(capsys):
    q = CrossModalQrels.synthetic(3, captions_per_image=2)          # captions 0,1 -> img 0; 2,3 -> img 1; 4,5 -> img 2
    sparse = {
        "0": {"docs": {"0": 5.0, "1": 4.0}},                         # hit@1
        "1": {"docs": {"2": 5.0, "1": 5.0, "0": 5.0}},               # all tie: stable sort keeps hit order -> 0 is 3rd
        "2": {"docs": {}},                                            # skipped, still in the denominator
        "3": {"docs": {"2": 9.0, "0": 1.0}},                         # miss (target img 1)
    }
    m = _metrics({"sparse": sparse}, q, ["0", "1", "2", "3"])
    r = m.recalls()["sparse"]
    assert r[1] == 1 / 4 and r[5] == 2 / 4 and r[10] == 2 / 4 and r[200] == 2 / 4
    assert oracle.recall_fractions(oracle.recall_counts(sparse, lambda qid: q.get_target(qid, "text")), 4) == r
    m.print_recall()
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "4" and out[1] == "Sparse recall @ 1: [0.25]"
    assert out[-1] == "Sparse reps recall: r@1 0.25, r@5 0.5, r@10 0.5, r@100 0.5, r@200 0.5"
    # image -> text: any of the image's captions counts; fusion runs have no 'docs' level
    fus = {"1": {"9": 0.9, "3": 0.8}, "2": {"0": 1.0}}
    m2 = _metrics({"fusion": fus}, q, ["1", "2"], qtype="image")
    assert m2.recalls()["fusion"][1] == 0.0 and m2.recalls()["fusion"][5] == 0.5


def test_qrels_csv_schemas(tmp_path):
    f = tmp_path / "flickr.csv"
    f.write_text("imgid,filename,caption,sentid\n25,a.jpg,\"x, y\",125\n25,a.jpg,z,126\n30,b.jpg,w,150\n")
    q = CrossModalQrels(str(f), "flickr")
    assert q.get_target("125", "text") == "25" and q.get_target("25", "image") == ["125", "126"]
    assert q.img_id_list == ["25", "30"] and q.text_id_list == ["125", "126", "150"]
    c = tmp_path / "coco.csv"
    c.write_text("imgid,filepath,filename,caption,sentid\n7,val,f.jpg,cap,770\n")
    assert CrossModalQrels(str(c), "coco").get_target("770", "text") == "7"


def test_cli_query_readers(tmp_path):
    from mllm_sparse_retrieval_amd.cli import read_queries

    t = tmp_path / "query.tsv"
    t.write_text("5\tdog dog cat\n6\tfish\n")
    assert read_queries(str(t)) == (["5", "6"], ["dog dog cat", "fish"])
    j = tmp_path / "q.jsonl"
    j.write_text('{"id": 5, "vector": {"dog": 2, "cat": 1, "zero": 0}}\n')
    assert read_queries(str(j)) == (["5"], ["dog dog cat "]) or read_queries(str(j))[1][0].split() == ["dog", "dog", "cat"]
