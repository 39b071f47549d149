cd $GRAFT_REPO_ROOT
for t in 4096 8192 12288 16384 32768; do
  python bench.py --no-cpu --no-c4 --no-c5 --tile-docs $t 2>/dev/null > gpurun_out/ts_$t.json
  python - $t <<'PY'
import sys, json
d = json.loads(open(f"gpurun_out/ts_{sys.argv[1]}.json").read())
print("tile", sys.argv[1], "flickr_ms", d["ms_per_step"], "qps", round(d["value"]), "kernel_ms", d["roofline"]["kernel_ms"])
PY
done
