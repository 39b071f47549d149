# usage: bash scripts/gpu_util.sh [bench args] -> busy fractions of the score_tiles launches (VALU / scalar / LDS / waves)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/util_a gpurun_out/util_b
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/util_a -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > /dev/null 2> gpurun_out/util_a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/util_b -- python3 bench.py --no-cpu --steps 3 --warmup 1 "$@" > /dev/null 2> gpurun_out/util_b.err
python3 - <<'PY'
import csv, glob, collections
def load(d):
    f = sorted(glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv'))[-1]
    kt = sorted(glob.glob(f'gpurun_out/{d}/*/*kernel_trace.csv'))[-1]
    dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt)) if 'score_tiles' in r['Kernel_Name']}
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'score_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
    return agg, sum(dur.values())
a, ta = load('util_a'); b, tb = load('util_b')
cyc_a = ta * 2.4  # ns -> cycles at 2.4 GHz (kernel time under the profiler)
simd_quads = cyc_a / 4 * 1024; cu_cycles = cyc_a * 256
print(f"kernel time under profiler: {ta/1e6:.2f} ms (all score_tiles launches)")
print(f"VALU busy {a['SQ_ACTIVE_INST_VALU']/simd_quads:.2%}  scalar {a['SQ_INSTS_SALU']/cu_cycles:.2%}  LDS {a['SQ_LDS_IDX_ACTIVE']/cu_cycles:.2%}  waves/SIMD {a['SQ_WAVE_CYCLES']/simd_quads:.2f}")
print({k: f"{v:.4g}" for k, v in sorted(a.items())})
print({k: f"{v:.4g}" for k, v in sorted(b.items())})
PY
