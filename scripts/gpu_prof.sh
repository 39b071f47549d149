# usage: bash scripts/gpu_prof.sh <tag> <workload-key> [bench args...]
#   -> gpurun_out/prof_<tag>_{stats,fetch,l2,sq,sq2}, gpurun_out/prof_<tag>.txt (summary), gpurun_out/r03_counters.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; WL=$2; shift; shift
STEPS=5; WARM=1
B="python3 bench.py --no-cpu --steps $STEPS --warmup $WARM $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- $B > gpurun_out/prof_${TAG}.json 2>gpurun_out/prof_${TAG}.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_${TAG}_l2 -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_${TAG}_sq -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${TAG}_sq2 -- $B > /dev/null 2>&1
# hybrid_search runs twice per bench call (warm-up + timed) whatever --steps says
case "$WL" in c5_hybrid*) N=2;; *) N=$((STEPS+WARM));; esac
python3 scripts/summarize_prof.py gpurun_out/prof_${TAG} > gpurun_out/prof_${TAG}.txt
python3 scripts/prof_counters.py gpurun_out/prof_${TAG} $WL $N gpurun_out/r03_counters.json
