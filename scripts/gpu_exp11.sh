cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P='import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["merge_kernel_ms"])'
for t in 4096 8192 16384; do for dm in 8 16; do
echo "== headline tile $t dense $dm"; python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 --tile-docs $t --dense-max $dm 2>/dev/null | python3 -c "$P"
done; done
