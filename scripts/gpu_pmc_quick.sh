# usage: bash scripts/gpu_pmc_quick.sh <tag> [bench args] -> executed-instruction counters per kernel (one PMC pass) + kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
B="python3 bench.py --no-cpu --steps 3 --warmup 1 $@"
rm -rf gpurun_out/pq_${TAG}_a gpurun_out/pq_${TAG}_s
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pq_${TAG}_s -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pq_${TAG}_a -- $B > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/pq_${TAG}_a/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-48:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k, c in agg.items():
    w = max(c['SQ_WAVES'], 1)
    print(f"{k:50s} dispatches {n[k]:3d} waves/disp {w/max(n[k],1):9.0f}  per wave: VALU {c['SQ_INSTS_VALU']/w:7.0f} SALU {c['SQ_INSTS_SALU']/w:7.0f} LDS {c['SQ_INSTS_LDS']/w:6.0f} VMEM_RD {c['SQ_INSTS_VMEM_RD']/w:6.0f} VMEM_WR {c['SQ_INSTS_VMEM_WR']/w:6.0f}  valu_busy {4*c['SQ_ACTIVE_INST_VALU']/max(c['GRBM_GUI_ACTIVE']/8*1024,1):.3f}")
s = sorted(glob.glob('gpurun_out/pq_${TAG}_s/*/*kernel_stats.csv'))[-1]
print(open(s).read())
PY
