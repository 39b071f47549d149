# usage (on the GPU box): bash scripts/gpu_prof_all.sh   -> rocprofv3 stats + counter passes for every profiled workload, the
# measured pipe peaks and launch latencies, then the default bench line — ONE session, one binary (stamped into the counters)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
R=r03
rm -f $O/${R}_counters.json $O/${R}_prof.log
bash scripts/gpu_prof.sh ${R}_flickr flickr30k_t2i --no-c3 --no-c4 --no-c5 && echo "prof flickr ok" | tee -a $O/${R}_prof.log
bash scripts/gpu_prof.sh ${R}_c3_i2t coco5k_i2t --only-c3 --c3-dir i2t && echo "prof c3 i2t ok" | tee -a $O/${R}_prof.log
bash scripts/gpu_prof.sh ${R}_c3_t2i coco5k_t2i --only-c3 --c3-dir t2i && echo "prof c3 t2i ok" | tee -a $O/${R}_prof.log
bash scripts/gpu_prof.sh ${R}_c5 c5_hybrid --only-c5 --c5-shape t2i && echo "prof c5 ok" | tee -a $O/${R}_prof.log
bash scripts/gpu_prof.sh ${R}_c5_i2t c5_hybrid_i2t --only-c5 --c5-shape i2t && echo "prof c5 i2t ok" | tee -a $O/${R}_prof.log
bash scripts/gpu_prof.sh ${R}_c4 c4_1m --only-c4 && echo "prof c4 ok" | tee -a $O/${R}_prof.log
python3 -c "
import mllm_sparse_retrieval_amd as m
for rep in range(3):
    p = m.device_peak_rates(0)
    print(f'run {rep}: ds_add_u32 {p[\"ds_add_per_s\"] / 1e12:.3f} T lane-ops/s, v_dot2_u32_u16 {p[\"dot2_per_s\"] / 1e12:.3f} T lane-ops/s, '
          f'LDS (1 write : 2 reads, b128) {p[\"lds_bytes_per_s\"] / 1e12:.2f} TB/s, {p[\"cus\"]} CUs')
" > $O/${R}_peak_rates.txt 2>&1
[ -x build/latency_lab ] && build/latency_lab > $O/${R}_latency_lab.txt 2>&1
mkdir -p profiles && cp $O/${R}_counters.json profiles/${R}_counters.json
python3 bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err; echo "bench rc=$?" | tee -a $O/${R}_prof.log
