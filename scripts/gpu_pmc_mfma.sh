# usage: bash scripts/gpu_pmc_mfma.sh <tag> [bench args] -> MFMA busy, clock, wait breakdown per kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
B="python3 bench.py --no-cpu --steps 3 --warmup 1 $@"
rm -rf gpurun_out/pm_${TAG}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pm_${TAG} -- $B > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/pm_${TAG}/*/*counter_collection.csv'))[-1]
kt = sorted(glob.glob('gpurun_out/pm_${TAG}/*/*kernel_trace.csv'))[-1]
dur = collections.defaultdict(float)
for r in csv.DictReader(open(kt)):
    dur[r['Kernel_Name'].split('(')[0][-48:]] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][-48:]][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in agg.items():
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    print(f"{k:50s} ns {dur[k]:.0f} clock GHz {cyc/max(dur[k],1):.2f} mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/max(cyc*1024,1):.3f} wave_cycles/simd-cycle {4*c['SQ_WAVE_CYCLES']/max(cyc*1024,1):.2f} wait_any {c['SQ_WAIT_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} wait_inst {c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} active {c['SQ_ACTIVE_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} mops {c['SQ_INSTS_VALU_MFMA_MOPS_F16']:.3g}")
PY
