cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for t in 32768 16384 12288 8192 4096; do
  echo "== tile $t"; $B --c4-tile-docs $t 2>/dev/null | python3 -c "$P"
done
for t in 32768 16384 8192; do
echo "== headline tile $t"; python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 --tile-docs $t 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"])'
done
