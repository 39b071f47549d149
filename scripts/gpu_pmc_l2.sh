# usage: bash scripts/gpu_pmc_l2.sh <tag> [bench args] -> L2 hit/miss + fetch bytes per kernel (two PMC passes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
B="python3 bench.py --no-cpu --steps 3 --warmup 1 $@"
rm -rf gpurun_out/pl_${TAG}_a gpurun_out/pl_${TAG}_b
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE --output-format csv -d gpurun_out/pl_${TAG}_a -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pl_${TAG}_b -- $B > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ('a','b'):
    f = sorted(glob.glob('gpurun_out/pl_${TAG}_'+d+'/*/*counter_collection.csv'))[-1]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-48:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('TCC_HIT_sum',): n[k] += 1
for k, c in agg.items():
    print(f"{k:50s} disp {n[k]:3d} L2 hit {c['TCC_HIT_sum']/max(c['TCC_HIT_sum']+c['TCC_MISS_sum'],1):.3f} hitGB/disp {c['TCC_HIT_sum']*128/1e9/max(n[k],1):.3f} fetchGB/disp {2*c['FETCH_SIZE']*1024/1e9/max(n[k],1):.3f} writeGB/disp {c['WRITE_SIZE']*1024/1e9/max(n[k],1):.3f}")
PY
