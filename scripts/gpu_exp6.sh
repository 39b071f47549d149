cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for t in 32768 16384 8192 4096; do
  echo "== tile $t"; $B --c4-tile-docs $t 2>/dev/null | python3 -c "$P"
done
echo "== headline"; python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"])'
for t in 16384 8192; do
echo "== headline tile $t"; python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 --tile-docs $t 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"])'
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_x_fetch -- $B --c4-tile-docs 16384 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_x_l2 -- $B --c4-tile-docs 16384 > /dev/null 2>&1
python3 scripts/summarize_prof.py gpurun_out/prof_x | grep score_tiles
