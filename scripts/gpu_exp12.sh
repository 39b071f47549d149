cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for v in 0 1 2 3 4; do
  echo "== variant $v"; MSR_VARIANT=$v $B 2>/dev/null | python3 -c "$P"
  MSR_VARIANT=$v python3 bench.py --no-c4 --no-c5 --no-cpu --steps 10 --warmup 2 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"])'
done
