# usage: bash scripts/gpu_sweep_env.sh VAR "v1 v2 ..." [bench args]  -> one line per value: Flickr ms, q/s, C4 ms
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu --no-c5 --no-term-shards "$@" 2> gpurun_out/sweep.err > gpurun_out/sweep_$v.json
  python - "$VAR" "$v" gpurun_out/sweep_$v.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[3]).read())
print(sys.argv[1], sys.argv[2], "flickr_ms", d["ms_per_step"], "qps", round(d["value"]), "c4_ms", d.get("c4_1m", {}).get("ms_per_step"), "c4_qps", d.get("c4_1m", {}).get("value"))
PY
done
