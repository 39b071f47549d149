# usage: bash scripts/gpu_dense_sweep_small.sh  -> dense-head build options on the 5 000 / 25 010-doc workloads (C3 both ways, C5)
cd $GRAFT_REPO_ROOT
for cfg in "16 0.4" "32 0.4" "32 0.2" "32 0.1" "64 0.2" "64 0.1" "64 0.05"; do
  set -- $cfg
  python bench.py --no-cpu --only-c3 --dense-max $1 --dense-density $2 2> gpurun_out/ds.err > gpurun_out/ds3.json
  python bench.py --no-cpu --only-c5 --dense-max $1 --dense-density $2 2> gpurun_out/ds.err > gpurun_out/ds5.json
  python - "$1" "$2" <<'PY'
import sys, json
d3 = json.loads(open("gpurun_out/ds3.json").read().strip().splitlines()[-1])["c3_coco5k"]
d5 = json.loads(open("gpurun_out/ds5.json").read().strip().splitlines()[-1])["c5_hybrid"]
print("dense_max", sys.argv[1], "density", sys.argv[2], "| c3 i2t ms", d3["i2t"].get("ms_per_step"), "t2i ms", d3["t2i"].get("ms_per_step"),
      "| c5 kernels", d5.get("kernel_ms"), "parity mismatches", (d5.get("parity") or {}).get("id_mismatches"))
PY
done
