set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1"
for t in 32768 16384 8192 4096; do
  echo "== tile $t"; $B --c4-tile-docs $t 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin)['c4_1m']; print(d['value'], d['roofline']['kernel_ms'])"
done
for f in 1 2 3 4 5 7; do
  echo "== dbg $f tile 32768"; MSR_DEBUG_FLAGS=$f $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin)['c4_1m']; print(d['value'], d['roofline']['kernel_ms'])"
done
for f in 1 2 4; do
  echo "== dbg $f tile 8192"; MSR_DEBUG_FLAGS=$f $B --c4-tile-docs 8192 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin)['c4_1m']; print(d['value'], d['roofline']['kernel_ms'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4_stats -- $B > /dev/null 2>gpurun_out/prof_c4_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_c4_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_c4_l2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_c4_sq -- $B > /dev/null 2>&1
find gpurun_out -name "*.csv" | head -30
