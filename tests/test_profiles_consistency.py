"""The committed measurement artefacts agree with each other: the bench line under profiles/ follows the contract, every
roofline fraction is a fraction (<= 1) and can be recomputed — the useful-work fraction from the work counts and measured
peaks the line itself carries, the utilisation figures from profiles/r03_counters.json with bench.py's own formulas
(profiles/README.md) at the kernel time the line reports. And a counters file that was collected on other kernels is
flagged (`counters_stale`) instead of being replayed. No GPU, no oracle."""
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r03_bench_default.json")


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (top level of bench.py only defines things; main() runs under __main__)
    return mod


@pytest.fixture(scope="module")
def line():
    return json.loads(open(LINE).read().strip().splitlines()[-1])


def test_contract_keys(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert {"affinity_cpus", "rows"} <= set(cb)
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "lds_atomic", "valu_dot2", "lds_bw") and isinstance(rf["unit"], str) and rf["unit"]
    assert {"achieved", "peak", "frac", "traffic", "useful", "utilisation", "counters_stale"} <= set(rf)
    # whole-job throughput = queries of all steps / timed region
    nq = line["config"]["queries_per_step_per_gpu"]
    assert line["value"] == pytest.approx(nq / (line["ms_per_step"] * 1e-3), rel=1e-3)


def _sparse_rooflines(line):
    yield "flickr30k_t2i", line["roofline"]
    for d in ("i2t", "t2i"):
        yield f"coco5k_{d}", line["c3_coco5k"][d]["roofline"]
    yield "c4_1m", line["c4_1m"]["roofline"]


def test_the_committed_line_was_not_made_from_stale_counters(line):
    for name, rf in _sparse_rooflines(line):
        assert rf["counters_stale"] is False and rf["frac"] is not None, name
    assert line["c5_hybrid"]["counters_stale"] is False and line["c5_hybrid_i2t"]["counters_stale"] is False


def test_useful_fraction_recomputes_from_the_line(line):
    """frac = max over pipes of (work / measured peak) / kernel time, every term in the line."""
    for name, rf in _sparse_rooflines(line):
        u, w, pk = rf["useful"], rf["useful"]["work_per_step"], rf["useful"]["peaks_measured"]
        t_ms = rf["kernel_ms"]
        mins = {"lds_atomic": w["sparse_postings"] / pk["ds_add_u32_lane_ops_per_s"] * 1e3,
                "valu_dot2": w["dense_head_postings"] / 2 / pk["v_dot2_u32_u16_lane_ops_per_s"] * 1e3,
                "lds_bw": (w["acc_init_bytes"] + w["acc_select_bytes"]) / pk["lds_bytes_per_s_1w2r"] * 1e3,
                "hbm": rf["traffic"] / 8e12 * 1e3}
        for pipe, v in mins.items():
            assert u["min_ms_per_pipe"][pipe] == pytest.approx(v, rel=2e-3, abs=2e-4), (name, pipe)
        bound = max(mins, key=mins.get)
        assert rf["bound"] == bound, name
        assert rf["frac"] == pytest.approx(mins[bound] / t_ms, rel=3e-3, abs=2e-4), name
        assert 0.0 < rf["frac"] <= 1.0, (name, rf["frac"])
        assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=5e-3), name
        # the work counts are the batch's: postings split into inverted-list and dense-head postings
        assert w["sparse_postings"] + w["dense_head_postings"] == rf["algorithmic"]["postings_per_step"], name


def test_every_utilisation_figure_is_a_fraction_and_recomputes(bench, line):
    if bench.counters_state()["stale"]:
        pytest.skip("profiles/r03_counters.json is stale for this tree: " + bench.counters_state().get("reason", ""))
    cases = [(w, "score_tiles", rf["utilisation"], rf["kernel_ms"]) for w, rf in _sparse_rooflines(line)]
    for wl in ("c5_hybrid", "c5_hybrid_i2t"):
        for stage, rf in line[wl]["roofline"].items():
            kre = {"dense_gemm": "dense_scores", "hybrid_tiles": "hybrid_tiles", "hybrid_tiles_mode1": "hybrid_tiles",
                   "hybrid_fuse_query": "hybrid_fuse_query"}[stage]
            cases.append((wl, kre, rf["utilisation"], rf["kernel_ms"]))
    for workload, kernel_re, ut, kernel_ms in cases:
        c = bench.counters(workload, kernel_re)
        assert c is not None, (workload, kernel_re)
        again = bench.binding_fractions(c, kernel_ms)
        for key in ("hbm_frac", "l2_frac", "valu_busy", "lds_issue_busy"):
            if ut.get(key) is not None:
                assert 0.0 <= ut[key] <= 1.0, (workload, kernel_re, key, ut[key])
                # (the line rounds kernel_ms to 4 digits: the recomputation sees that rounding, nothing else)
                assert again[key] == pytest.approx(ut[key], rel=2e-3, abs=2e-4), (workload, kernel_re, key)


def test_gemm_fraction_is_flops_over_time_over_peak(line):
    for wl in ("c5_hybrid", "c5_hybrid_i2t"):
        c5 = line[wl]
        g = c5["roofline"]["dense_gemm"]
        assert g["bound"] == "mfma" and g["peak"] == 2500.0
        assert g["frac"] == pytest.approx(g["achieved"] / g["peak"], abs=1e-4)
        # 2 * queries * docs * hidden / kernel time; the workload string names the three sizes
        m = re.search(r"hybrid: (\d+) docs x \(128 nnz \+ (\d+)-d fp16\), (\d+) queries", c5["workload"])
        n, h, nq = (int(x) for x in m.groups())
        assert g["achieved"] == pytest.approx(2.0 * n * h * nq / (g["kernel_ms"] * 1e-3) / 1e12, rel=2e-3)
        assert c5["parity"]["id_mismatches"] == 0 and c5["parity"]["max_abs_score_diff"] <= 1e-5


class _FakeBatch:
    """Duck type of QueryBatch for sparse_roofline (no GPU): a step of 1 000 workgroups."""

    class index:
        n_tiles = 4

    def algo_bytes(self, k):
        return 6_000_000, 1_000_000

    def work(self):
        return dict(sparse_postings=400_000, dense_head_postings=600_000, workgroups=1000, acc_init_bytes=32_768_000,
                    acc_select_bytes=65_536_000, query_entries=11_000)


def test_stale_counters_are_flagged_not_replayed(bench, tmp_path, monkeypatch):
    """Change a kernel, skip re-profiling: the counter-backed figures must disappear and the line must say why."""
    from mllm_sparse_retrieval_amd import _buildinfo

    monkeypatch.setattr(bench, "pipe_peaks", lambda: dict(ds_add_per_s=1e13, dot2_per_s=3e13, lds_bytes_per_s=5e13, cus=256))
    monkeypatch.setattr(bench, "hbm_copy_gbs", lambda: 5000.0)
    fake = {"_stamp": _buildinfo.stamp(),
            "wl": {"steps": 1, "kernels": {"msr::score_tiles": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 10.0, "TCC_HIT_sum": 1e6,
                                                                "TCC_MISS_sum": 1e4}}}}
    path = tmp_path / "counters.json"
    monkeypatch.setattr(bench, "COUNTERS_FILE", str(path))

    def roofline():
        bench._COUNTERS = bench._COUNTERS_STATE = None
        return bench.sparse_roofline(_FakeBatch(), 10, 1.0, "wl")

    path.write_text(json.dumps(fake))
    fresh = roofline()
    assert fresh["counters_stale"] is False and fresh["traffic"] == int((2 * 1000.0 + 10.0) * 1024)
    assert fresh["frac"] is not None and fresh["utilisation"]["hbm_frac"] is not None
    assert fresh["bound"] == "lds_bw" and fresh["frac"] == pytest.approx((32_768_000 + 65_536_000) / 5e13 * 1e3 / 1.0, abs=1e-4)
    # the same file after an edit of a kernel source
    monkeypatch.setattr(_buildinfo, "kernel_source_sha256", lambda *a: "0" * 64)
    stale = roofline()
    assert stale["counters_stale"] is True and "kernel sources changed" in stale["counters"]["reason"]
    assert stale["frac"] is None and stale["traffic"] is None and stale["utilisation"]["hbm_frac"] is None
    # ... and a file without a stamp (round 2's format)
    monkeypatch.undo()
    monkeypatch.setattr(bench, "pipe_peaks", lambda: dict(ds_add_per_s=1e13, dot2_per_s=3e13, lds_bytes_per_s=5e13, cus=256))
    monkeypatch.setattr(bench, "hbm_copy_gbs", lambda: 5000.0)
    monkeypatch.setattr(bench, "COUNTERS_FILE", str(path))
    del fake["_stamp"]
    path.write_text(json.dumps(fake))
    unstamped = roofline()
    assert unstamped["counters_stale"] is True and unstamped["frac"] is None
    bench._COUNTERS = bench._COUNTERS_STATE = None


def test_buildinfo_hashes_follow_the_sources(tmp_path):
    from mllm_sparse_retrieval_amd import _buildinfo

    d = tmp_path / "csrc"
    d.mkdir()
    (d / "a.hip").write_text("kernel 1")
    (d / "notes.txt").write_text("ignored")
    h1 = _buildinfo.kernel_source_sha256(str(d))
    (d / "notes.txt").write_text("still ignored")
    assert _buildinfo.kernel_source_sha256(str(d)) == h1
    (d / "a.hip").write_text("kernel 2")
    assert _buildinfo.kernel_source_sha256(str(d)) != h1
    assert _buildinfo.code_object_sha256(str(d / "missing.so")) is None
    co = _buildinfo.code_object_sha256()
    assert co is None or re.fullmatch(r"[0-9a-f]{64}", co)
