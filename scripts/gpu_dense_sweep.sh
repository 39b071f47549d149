# usage: bash scripts/gpu_dense_sweep.sh  -> C4 (1M docs) time per dense-head setting
cd $GRAFT_REPO_ROOT
for cfg in "16 0.4" "16 0.5" "16 0.6" "16 0.75" "8 0.4" "8 0.6" "32 0.3" "32 0.4" "4 0.5" "0 0.4"; do
  set -- $cfg
  python bench.py --no-cpu --only-c4 --no-term-shards --dense-max $1 --dense-density $2 2> gpurun_out/ds.err > gpurun_out/ds.json
  python - "$1" "$2" <<'PY'
import sys, json
d = json.loads(open("gpurun_out/ds.json").read())
c = d.get("c4_1m", d)
print("dense_max", sys.argv[1], "density", sys.argv[2], "c4_ms", c.get("ms_per_step"), "qps", c.get("value"), "n_dense", c.get("n_dense"), "roof", (c.get("roofline") or {}).get("achieved"))
PY
done
