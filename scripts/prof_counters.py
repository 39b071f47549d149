#!/usr/bin/env python3
"""rocprofv3 csv output dirs of ONE workload -> its entry in a counters JSON file (read by bench.py's roofline).

usage: prof_counters.py <prefix> <workload> <steps> <out.json>
    <prefix>   gpurun_out/prof_<tag>   (reads <prefix>_stats, _fetch, _l2, _sq, _sq2 as written by scripts/gpu_prof.sh)
    <steps>    bench steps the profiled command ran (timed + warm-up): counter sums / steps = per-step figures
Per kernel (name cut at the template arguments): dispatches, summed duration (kernel trace of the stats pass) and the
SUM of every counter over the dispatches. One pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass,
MI355X_MICROARCH.md 'rocprofv3 PMC slots'); units stay rocprofv3's (FETCH_SIZE / WRITE_SIZE in KiB)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def newest(pattern):  # gpurun_out/ keeps the csv files of earlier calls: take the most recent one per pass
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.split("(")[0]
    m = re.match(r"_ZN3msr(\d+)", name)  # unmangled-less names of extern kernels
    if m:
        n = int(m.group(1))
        name = "msr::" + name[len(m.group(0)):len(m.group(0)) + n]
    return name


prefix, workload, steps, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
kern = collections.defaultdict(lambda: collections.defaultdict(float))
for f in newest(prefix + "_stats/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        kern[k]["dispatches"] += 1
        kern[k]["duration_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for suffix in ("fetch", "l2", "sq", "sq2"):
    for f in newest(f"{prefix}_{suffix}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            kern[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    # kernel time under THIS pass's counter set (some sets slow a kernel down: cycle counters of a pass must be read
    # against that pass's own duration, e.g. clock = GRBM_GUI_ACTIVE / 8 / duration_ns_sq2)
    for f in newest(f"{prefix}_{suffix}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            kern[short(r["Kernel_Name"])]["duration_ns_" + suffix] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
try:
    doc = json.load(open(out_path))
except Exception:
    doc = {"_comment": "per-workload SUMS over all dispatches of a profiled bench command (rocprofv3 --pmc, one pass per "
                       "counter group); divide by `steps` for per-step figures. Written by scripts/prof_counters.py."}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mllm_sparse_retrieval_amd import _buildinfo  # noqa: E402

# which binary these counters describe: bench.py withholds every counter-backed figure (`counters_stale`) when the
# kernels of the tree it runs in hash differently
stamp = _buildinfo.stamp()
if doc.get("_stamp") and doc["_stamp"].get("kernel_source_sha256") != stamp["kernel_source_sha256"]:
    sys.exit(f"{out_path} was started on other kernel sources: delete it and profile every workload again")
doc["_stamp"] = stamp
doc[workload] = {"steps": steps, "kernels": {k: dict(v) for k, v in sorted(kern.items()) if k.startswith("msr::")}}
json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
print(f"{workload}: {len(doc[workload]['kernels'])} kernels -> {out_path}")
