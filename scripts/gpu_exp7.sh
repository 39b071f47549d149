cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --only-c4 --no-cpu --steps 5 --warmup 1 --c4-tile-docs 16384"
P='import json,sys; d=json.load(sys.stdin)["c4_1m"]; print(d["value"], d["roofline"]["kernel_ms"])'
for f in 32 1 2 3 4 16 17 19 23; do
  echo "== dbg $f tile 16384"; MSR_DEBUG_FLAGS=$f $B 2>/dev/null | python3 -c "$P"
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_y_sq -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_y_sq2 -- $B > /dev/null 2>&1
python3 scripts/summarize_prof.py gpurun_out/prof_y | grep score_tiles
