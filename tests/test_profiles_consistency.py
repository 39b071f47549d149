"""The committed measurement artefacts agree with each other: the bench line under profiles/ follows the contract, every
roofline fraction is a fraction (<= 1), and the counter-backed fractions can be recomputed from profiles/r02_counters.json
with bench.py's own formulas (profiles/README.md) at the kernel time the line reports. No GPU, no oracle."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r02_bench_default.json")


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
    assert rf["bound"] in ("hbm", "l2", "valu", "mfma") and isinstance(rf["unit"], str) and rf["unit"]
    # whole-job throughput = queries of all steps / timed region
    nq = line["config"]["queries_per_step_per_gpu"]
    assert line["value"] == pytest.approx(nq / (line["ms_per_step"] * 1e-3), rel=1e-3)


def _rooflines(line):
    yield "headline", line["roofline"]
    for d in ("i2t", "t2i"):
        yield f"c3 {d}", line["c3_coco5k"][d]["roofline"]
    yield "c4", line["c4_1m"]["roofline"]
    for k, v in line["c5_hybrid"]["roofline"].items():
        yield f"c5 {k}", v


def test_every_fraction_is_a_fraction(line):
    for name, rf in _rooflines(line):
        for key in ("frac", "hbm_frac", "l2_frac", "valu_busy", "lds_issue_busy"):
            if rf.get(key) is not None:
                assert 0.0 <= rf[key] <= 1.0, (name, key, rf[key])
        # `frac` is the largest of the ceilings the object lists
        ceil = [rf[k] for k in ("hbm_frac", "l2_frac", "valu_busy") if rf.get(k) is not None]
        if rf.get("bound") != "mfma" and ceil:
            assert rf["frac"] == pytest.approx(max(ceil), abs=1e-4), name


def test_fractions_recompute_from_committed_counters(bench, line):
    cases = [("flickr30k_t2i", "score_tiles", line["roofline"]),
             ("c4_1m", "score_tiles", line["c4_1m"]["roofline"]),
             ("coco5k_i2t", "score_tiles", line["c3_coco5k"]["i2t"]["roofline"]),
             ("coco5k_t2i", "score_tiles", line["c3_coco5k"]["t2i"]["roofline"]),
             ("c5_hybrid", "dense_scores", line["c5_hybrid"]["roofline"]["dense_gemm"]),
             ("c5_hybrid", "hybrid_tiles", line["c5_hybrid"]["roofline"]["hybrid_tiles"])]
    for workload, kernel_re, rf in cases:
        c = bench.counters(workload, kernel_re)
        assert c is not None, workload
        again = bench.binding_fractions(c, rf["kernel_ms"])
        for key in ("hbm_frac", "l2_frac", "valu_busy", "traffic"):
            if key in again and rf.get(key) is not None:
                # (the line rounds kernel_ms to 4 digits: the recomputation sees that rounding, nothing else)
                assert again[key] == pytest.approx(rf[key], rel=2e-3, abs=2e-4), (workload, kernel_re, key)


def test_gemm_fraction_is_flops_over_time_over_peak(line):
    c5 = line["c5_hybrid"]
    g = c5["roofline"]["dense_gemm"]
    assert g["bound"] == "mfma" and g["peak"] == 2500.0
    assert g["frac"] == pytest.approx(g["achieved"] / g["peak"], abs=1e-4)
    # 2 * queries * docs * hidden / kernel time; the workload string names the three sizes
    import re

    m = re.search(r"hybrid: (\d+) docs x \(128 nnz \+ (\d+)-d fp16\), (\d+) queries", c5["workload"])
    n, h, nq = (int(x) for x in m.groups())
    assert g["achieved"] == pytest.approx(2.0 * n * h * nq / (g["kernel_ms"] * 1e-3) / 1e12, rel=2e-3)
