cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for t in 16384 8192; do
  echo "== stamps tile $t"; MSR_DEBUG_FLAGS=8 $B --c4-tile-docs $t 2>gpurun_out/e.err | python3 -c "$P"; grep "phase shares" gpurun_out/e.err
done
echo "== headline stamps tile 8192"; MSR_DEBUG_FLAGS=8 python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 --tile-docs 8192 2>gpurun_out/e.err | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"])'; grep "phase shares" gpurun_out/e.err
