cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1 --c4-tile-docs 16384"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for f in 276 260 68; do
  echo "== dbg $f tile 16384"; MSR_DEBUG_FLAGS=$f $B 2>/dev/null | python3 -c "$P"
done
