"""Host logic that mirrors the reference's own Python around the searcher: run dicts, fusion, TREC IO, recall.

fuse / write_trec_run / read_trec_run are pinned against outputs of the REFERENCE's src/hybrid.py
(tests/golden/hybrid_golden.json, produced by tests/golden/make_hybrid_golden.py in the build container)."""
import json
import os
from types import SimpleNamespace

import numpy as np
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


def test_fuse_statistic_matches_reference(hybrid_cases):
    """src/hybrid.py:56-90 (the variant score_statistic.py uses): the same fused scores as fuse(), each tagged 'dense' /
    'sparse' / 'fuse'; recorded by running the reference's own function (tests/golden/make_hybrid_golden.py)."""
    for c in hybrid_cases:
        got = fusion.fuse_statistic([c["dense"], c["sparse"]], c["weights"])
        assert {q: {d: [r.score, r.type] for d, r in v.items()} for q, v in got.items()} == c["fused_statistic"]
        assert {q: {d: r.score for d, r in v.items()} for q, v in got.items()} == c["fused"]
        for qid in got:
            assert list(got[qid]) == list(c["fused_statistic"][qid])      # union order: dense docs first
            for doc, rec in got[qid].items():
                in_d, in_s = doc in c["dense"][qid]["docs"], doc in c["sparse"][qid]["docs"]
                assert rec.type == ("fuse" if in_d and in_s else "dense" if in_d else "sparse")


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


def test_run_helpers_against_the_reference_functions(tmp_path):
    """sparse_search / get_run_dict / search_queries / pickle_load against the outputs of the REFERENCE's own functions
    (src/search.py:49-99), recorded by tests/golden/make_run_golden.py, which compiles exactly those four function bodies
    out of the reference file's AST (the module itself cannot be imported here: faiss, pyserini, … are absent) and runs
    them on seeded inputs: batch order != dict order, queries that find themselves (remove_query), empty hit lists,
    repeated scores, integer lookup ids that become strings, batch_size 0 -> .search(), > 0 -> .batch_search()."""
    import pickle

    cases = json.load(open(os.path.join(GOLD, "run_golden.json")))["cases"]
    assert [c["kind"] for c in cases].count("sparse") == 6
    for c in cases:
        if c["kind"] == "sparse":
            table = {q: [(d, s) for d, s in hits] for q, hits in c["table"].items()}
            args = SimpleNamespace(depth=c["depth"], threads=16)
            for mod in (run, oracle):
                topics = [f"topic {q}" for q in c["batch_ids"]]
                scores, rankings = (run.sparse_search(_FakeSearcher(table), topics, c["batch_ids"], args) if mod is run else
                                    oracle.sparse_search(_FakeSearcher(table), topics, c["batch_ids"], c["depth"], 16))
                assert scores == c["scores"] and rankings == c["rankings"]
                for rq in (False, True):
                    got = mod.get_run_dict(c["batch_ids"], scores, rankings, rq)
                    assert got == c["run_dict"][str(rq)]
                    assert [list(got[q]["docs"]) for q in got] == [list(v["docs"]) for v in c["run_dict"][str(rq)].values()]
        elif c["kind"] == "dense":
            calls = []

            class Fake:
                def batch_search(self, q_reps, depth, batch_size, quiet):
                    calls.append(["batch_search", int(depth), int(batch_size), bool(quiet)])
                    return np.asarray(c["scores"], np.float32)[:, :depth], np.asarray(c["indices"])[:, :depth]

                def search(self, q_reps, depth):
                    calls.append(["search", int(depth)])
                    return np.asarray(c["scores"], np.float32)[:, :depth], np.asarray(c["indices"])[:, :depth]

            args = SimpleNamespace(depth=c["depth"], batch_size=c["batch_size"], quiet=True)
            out_scores, out_ids = run.search_queries(Fake(), np.zeros((4, 16), np.float32), c["lookup"], args)
            assert calls == c["calls"]
            assert np.asarray(out_scores).tolist() == c["out_scores"]
            assert out_ids.tolist() == c["out_ids"] and out_ids.dtype.kind == c["out_ids_dtype_kind"]
            for rq in (False, True):
                got = run.get_run_dict(c["batch_ids"], out_scores, out_ids, rq)
                got = {q: {"docs": {d: float(v) for d, v in e["docs"].items()}, "min_score": float(e["min_score"]),
                           "max_score": float(e["max_score"])} for q, e in got.items()}
                assert got == c["run_dict"][str(rq)]
        else:
            path = tmp_path / "corpus_0.pkl"
            with open(path, "wb") as f:
                pickle.dump((np.asarray(c["reps"], np.float32), c["lookup"]), f)
            reps, lookup = run.pickle_load(str(path))
            assert reps.tolist() == c["got_reps"] and str(reps.dtype) == c["got_dtype"] and list(lookup) == c["got_lookup"]


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


# ------------------------------------------------------------------------------------------------ reps files
def test_pickle_load_reads_reps_files_without_executing_anything(tmp_path):
    """pickle_load (src/search.py:49-52) on files like src/encode.py:405-410 writes — (float32 [N,H], list of ids) —
    in every protocol; a pickle that names any other global (here: os.system) is refused, not run."""
    import pickle

    from mllm_sparse_retrieval_amd.run import pickle_load

    reps = np.arange(12, dtype=np.float32).reshape(3, 4)
    for proto in (2, 3, 4, 5):
        f = tmp_path / f"corpus_{proto}.pkl"
        with open(f, "wb") as fh:
            pickle.dump((reps, ["10", "11", "12"]), fh, protocol=proto)
        got, lookup = pickle_load(str(f))
        assert got.dtype == np.float32 and (got == reps).all() and lookup == ["10", "11", "12"]
    with open(tmp_path / "ints.pkl", "wb") as fh:  # ids as ints / numpy scalars, reps as a list of rows
        pickle.dump(([[1.0, 2.0], [3.0, 4.0]], [np.int64(7), 8]), fh)
    got, lookup = pickle_load(str(tmp_path / "ints.pkl"))
    assert got.shape == (2, 2) and [int(x) for x in lookup] == [7, 8]

    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, (f"touch {marker}",))

    with open(tmp_path / "evil.pkl", "wb") as fh:
        pickle.dump((Evil(), []), fh)
    with pytest.raises(pickle.UnpicklingError, match="refused global"):
        pickle_load(str(tmp_path / "evil.pkl"))
    assert not marker.exists()
    with open(tmp_path / "obj.pkl", "wb") as fh:
        pickle.dump((np.array([{"a": 1}], dtype=object), []), fh)
    with pytest.raises(pickle.UnpicklingError):
        pickle_load(str(tmp_path / "obj.pkl"))
    with open(tmp_path / "shape.pkl", "wb") as fh:
        pickle.dump([1, 2, 3], fh)
    with pytest.raises(pickle.UnpicklingError, match="pair"):
        pickle_load(str(tmp_path / "shape.pkl"))


# ------------------------------------------------------------------------------------------------ sampler / eval
@pytest.mark.parametrize("n,world", [(35, 4), (36, 4), (5, 4), (25010, 4), (1, 2), (7, 8), (10, 1)])
def test_sampler_restatement_equals_torch_distributed_sampler(n, world):
    from torch.utils.data import DistributedSampler

    from mllm_sparse_retrieval_amd.sampler import distributed_sampler_indices

    for rank in range(world):
        want = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True))  # seed 0, epoch 0
        assert distributed_sampler_indices(n, world, rank) == want
        want = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False))
        assert distributed_sampler_indices(n, world, rank, shuffle=False) == want


def test_eval_subcommand_true_and_compat_denominator(tmp_path, capsys):
    """`eval` over TREC runs + the dataset csv (fixture: the first 35 rows of data/flickr/flickr_test.csv = 7 images x 5
    captions): true-nq denominator by default; --compat-denominator replays the reference's 4 ranks — 35 captions are
    padded to 36, the repeated caption is searched and counted on two ranks, every rank divides by 9 * 4
    (src/metrices.py:92, src/search.py:180-182)."""
    from torch.utils.data import DistributedSampler

    from mllm_sparse_retrieval_amd import cli
    from mllm_sparse_retrieval_amd.fusion import write_trec_run
    from mllm_sparse_retrieval_amd.qrels import CrossModalQrels

    csv_path = os.path.join(os.path.dirname(__file__), "golden", "flickr_test_head.csv")
    ds = CrossModalQrels(csv_path, "flickr")
    caps = ds.text_id_list
    assert len(caps) == 35 and len(ds.img_id_list) == 7
    imgs = ds.img_id_list
    # a sparse run: caption j ranks its own image at position (j % 4) + 1 among other images; every 7th caption misses
    run = {}
    for j, c in enumerate(caps):
        own = ds.get_target(c, "text")
        others = [i for i in imgs if i != own]
        ranked = others[: j % 4] + ([own] if j % 7 else []) + others[j % 4:]
        run[c] = {"docs": {d: 100.0 - r for r, d in enumerate(ranked[:6])}}
    os.makedirs(tmp_path / "runs")
    write_trec_run(run, str(tmp_path / "runs" / "sparse.trec"), name="sparse")

    def hit_at(c, k):
        ranked = sorted(run[c]["docs"].items(), key=lambda kv: kv[1], reverse=True)[:k]
        return ds.get_target(c, "text") in [d for d, _ in ranked]

    cli.main(["eval", "--runs_dir", str(tmp_path / "runs"), "--qrels", csv_path, "--dataset_name", "flickr",
              "--query_type", "text"])
    out = capsys.readouterr().out.splitlines()
    r1, r5 = sum(hit_at(c, 1) for c in caps) / 35, sum(hit_at(c, 5) for c in caps) / 35
    assert out[0] == "35" and out[1] == f"Sparse recall @ 1: [{r1}]"
    assert out[-1].startswith(f"Sparse reps recall: r@1 {r1}, r@5 {r5}, ")

    cli.main(["eval", "--runs_dir", str(tmp_path / "runs"), "--qrels", csv_path, "--dataset_name", "flickr",
              "--query_type", "text", "--compat-denominator", "--world_size", "4"])
    out = capsys.readouterr().out.splitlines()
    per_rank = []
    for rank in range(4):
        shard = [caps[i] for i in DistributedSampler(range(35), num_replicas=4, rank=rank, shuffle=True)]
        assert len(shard) == 9
        per_rank.append(sum(hit_at(c, 1) for c in set(shard)) / (9 * 4))
    assert out[0] == "36" and out[1] == f"Sparse recall @ 1: {per_rank}"
    assert out[-1].startswith(f"Sparse reps recall: r@1 {sum(per_rank)}, ")
    # the padded repeat is counted twice: compat != true whenever the repeated caption is a hit
    rep = [caps[i] for i in DistributedSampler(range(35), num_replicas=4, rank=3, shuffle=True)][-1]
    assert (sum(per_rank) * 36 - r1 * 35) == pytest.approx(1.0 if hit_at(rep, 1) else 0.0)


def test_f32_to_f16_matches_numpy_bit_for_bit(monkeypatch):
    """msr_f32_to_f16 (the host conversion in front of msr_dense_search / msr_hybrid_search): round to nearest even like
    numpy's astype(float16) — normal values, ties, the subnormal range, overflow to inf, infinities; NaNs stay NaNs.
    Both code paths: F16C and the portable bit arithmetic."""
    from mllm_sparse_retrieval_amd._cabi import check, lib, ptr

    rng = np.random.default_rng(3)
    special = np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 7e4, -7e4, np.inf, -np.inf, np.nan, 6.1035e-5, 6.0e-5,
                        5.96e-8, 2.98e-8, 2.9802322e-8, 2.9802326e-8, 1e-10, 0.1, 1.0009765, 1.00048828125, 1.0014648],
                       dtype=np.float32)
    anybits = rng.integers(0, 2 ** 32, 400_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    # every fp16 value, the midpoints between neighbours (ties) and their f32 neighbours
    halves = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float32)
    mids = (halves[:-1].astype(np.float64) + halves[1:].astype(np.float64)) / 2
    ties = np.concatenate([mids.astype(np.float32), np.nextafter(mids.astype(np.float32), np.float32(np.inf)),
                           np.nextafter(mids.astype(np.float32), np.float32(-np.inf))])
    x = np.concatenate([special, anybits, halves, ties, -ties, rng.standard_normal(200_000).astype(np.float32)])
    with np.errstate(all="ignore"):
        want = x.astype(np.float16)
    for no_f16c in (False, True):
        if no_f16c:
            monkeypatch.setenv("MSR_NO_F16C", "1")
        for threads in (1, 0):
            got = np.empty(x.shape, dtype=np.float16)
            check(lib().msr_f32_to_f16(ptr(x), ptr(got.view(np.uint16)), x.size, threads))
            nan = np.isnan(want)
            assert (np.isnan(got) == nan).all()
            assert (got.view(np.uint16)[~nan] == want.view(np.uint16)[~nan]).all(), (no_f16c, threads)
    # the wrapper takes the C path for big f32 matrices and numpy otherwise: same bits
    from mllm_sparse_retrieval_amd.dense import _as_fp16_rows

    big = rng.standard_normal((300, 256)).astype(np.float32)
    assert (_as_fp16_rows(big).view(np.uint16) == big.astype(np.float16).view(np.uint16)).all()
    odd = rng.standard_normal((300, 250)).astype(np.float32)          # padded to 256 columns
    out = _as_fp16_rows(odd)
    assert out.shape == (300, 256) and (out[:, 250:] == 0).all()
    assert (out[:, :250].view(np.uint16) == odd.astype(np.float16).view(np.uint16)).all()


def test_recall_metrics_against_the_reference_class():
    """recall.RecallMetrics / replay_ranks against outputs of the reference's own RecallMetrics (src/metrices.py),
    recorded by tests/golden/make_recall_golden.py for world sizes 1 and 2 (gloo): hit counts, every rank's fractions
    with the reference's padded denominator, the sampler's split, and the text rank 0 prints — bit for bit."""
    import contextlib
    import io
    from types import SimpleNamespace

    from mllm_sparse_retrieval_amd.recall import RecallMetrics, replay_ranks
    from mllm_sparse_retrieval_amd.sampler import shard_query_ids

    class Stub:
        def __init__(self, targets):
            self.targets = targets

        def get_target(self, qid, query_type):
            return self.targets[qid]

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "recall_golden.json")))
    assert len(gold["cases"]) >= 4
    for case in gold["cases"]:
        args = SimpleNamespace(query_type=case["query_type"])
        ds = Stub(case["targets"])
        # ---- one rank: the class itself, stage by stage
        ref = case["by_world"]["1"][0]
        m = RecallMetrics(ds, case["dense_run"], case["sparse_run"], case["fusion_run"], case["look_up"], case["qids"], args)
        m.sort_and_count()
        for name, counts in (("dense", m.dense_counts), ("sparse", m.sparse_counts), ("fusion", m.fusion_counts)):
            assert {str(k): v for k, v in counts.items()} == ref["counts"][name], name
        m.all_gather_object()
        for name, lists in (("dense", m.dense_recall_lists), ("sparse", m.sparse_recall_lists), ("fusion", m.fusion_recall_lists)):
            assert {str(k): v for k, v in lists.items()} == ref["lists"][name], name
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m.print_recall()
        assert buf.getvalue() == ref["printed"]
        # ---- two ranks: the sampler's split (with its padded repeat) and the reference's denominator, replayed
        two = case["by_world"]["2"]
        for r in (0, 1):
            assert shard_query_ids(case["qids"], 2, r) == two[r]["shard"]
        m2 = replay_ranks(ds, case["dense_run"], case["sparse_run"], case["fusion_run"], case["look_up"], case["qids"], args,
                          world_size=2, compat=True)
        for name, lists in (("dense", m2.dense_recall_lists), ("sparse", m2.sparse_recall_lists), ("fusion", m2.fusion_recall_lists)):
            assert {str(k): v for k, v in lists.items()} == two[0]["lists"][name], name
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            m2.print_recall()
        assert buf.getvalue() == two[0]["printed"]
