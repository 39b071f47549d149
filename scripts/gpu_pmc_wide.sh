# usage: bash scripts/gpu_pmc_wide.sh <tag> [bench args] -> gpurun_out/pmcw_<tag>.txt : wide counter sweep of score_tiles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
i=0
for set in \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" \
 "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_WAVES" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcw_${TAG}_$i -- python3 bench.py --no-cpu --steps 2 --warmup 1 "$@" > /dev/null 2> gpurun_out/pmcw_${TAG}_$i.err
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
out = open(f'gpurun_out/pmcw_{tag}.txt', 'w')
for i in range(1, 6):
    fs = glob.glob(f'gpurun_out/pmcw_{tag}_{i}/*/*counter_collection.csv')
    if not fs:
        print(i, 'no data', file=out); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        kn = r['Kernel_Name'][:36]
        agg[kn][r['Counter_Name']] += float(r['Counter_Value']); disp[kn].add(r['Dispatch_Id'])
    for kn in agg:
        if 'score_tiles' not in kn: continue
        n = len(disp[kn])
        for c, v in sorted(agg[kn].items()):
            print(f'{kn} launches={n} {c} {v / n:.4g}', file=out)
print(open(f'gpurun_out/pmcw_{tag}.txt').read())
PY
