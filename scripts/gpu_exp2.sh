set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for t in 32768 16384 8192 4096; do
  echo "== tile $t"; $B --c4-tile-docs $t 2>/dev/null | python3 -c "$P"
done
for f in 1 2 3 4 7; do
  echo "== dbg $f tile 32768"; MSR_DEBUG_FLAGS=$f $B 2>/dev/null | python3 -c "$P"
done
for f in 1 3 4 7; do
  echo "== dbg $f tile 16384"; MSR_DEBUG_FLAGS=$f $B --c4-tile-docs 16384 2>/dev/null | python3 -c "$P"
done
echo "== headline"; python3 bench.py --no-c4 --no-cpu --steps 10 --warmup 2 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(d["value"], d["roofline"]["kernel_ms"], d["recall"])'
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/prof2_c4_sq -- $B > /dev/null 2>&1
