# usage: bash scripts/gpu_instr_phases.sh [bench args]  -> gpurun_out/phase_<flags>/  (instruction counts per ablation)
# MSR_DEBUG_FLAGS: 128 launch only, 64 stop after staging, 256 stop after first chunk resolution, 4 no select,
# 16 no dense head, 32 = plain diagnostic instance (all phases)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for F in 128 64 256 4 20 32; do
  MSR_DEBUG_FLAGS=$F rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/phase_$F -- python3 bench.py --no-cpu --steps 2 --warmup 1 "$@" > /dev/null 2> gpurun_out/phase_$F.err
done
python3 - <<'PY'
import csv, glob, collections
for F in (128, 64, 256, 4, 20, 32):
    fs = glob.glob(f'gpurun_out/phase_{F}/*/*counter_collection.csv')
    if not fs: print(F, 'no data'); continue
    agg = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(fs[0])):
        if 'score_tiles' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n = max(1, len(disp))
    print(F, {k: round(v / n / 1e6, 1) for k, v in sorted(agg.items())})
PY
