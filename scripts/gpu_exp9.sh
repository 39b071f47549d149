cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1 --c4-tile-docs 16384"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for cfg in "0 0.4" "8 0.5" "16 0.4" "24 0.3" "32 0.2" "32 0.1"; do
  set -- $cfg
  echo "== dense max $1 density $2"; $B --dense-max $1 --dense-density $2 2>/dev/null | python3 -c "$P"
done
for cfg in "0 0.4" "16 0.4" "32 0.2"; do
  set -- $cfg
  echo "== headline dense max $1 density $2";  python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 --dense-max $1 --dense-density $2 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"])'
done
