# usage (on the GPU box): bash scripts/gpu_prof_all.sh   -> counter passes for every profiled workload + the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/r02_counters.json $O/r02_prof.log
bash scripts/gpu_prof.sh r02_flickr flickr30k_t2i --no-c3 --no-c4 --no-c5 && echo "prof flickr ok" | tee -a $O/r02_prof.log
bash scripts/gpu_prof.sh r02_c3_i2t coco5k_i2t --only-c3 --c3-dir i2t && echo "prof c3 i2t ok" | tee -a $O/r02_prof.log
bash scripts/gpu_prof.sh r02_c3_t2i coco5k_t2i --only-c3 --c3-dir t2i && echo "prof c3 t2i ok" | tee -a $O/r02_prof.log
bash scripts/gpu_prof.sh r02_c5 c5_hybrid --only-c5 && echo "prof c5 ok" | tee -a $O/r02_prof.log
bash scripts/gpu_prof.sh r02_c4 c4_1m --only-c4 && echo "prof c4 ok" | tee -a $O/r02_prof.log
mkdir -p profiles && cp $O/r02_counters.json profiles/r02_counters.json
python3 bench.py > $O/r02_bench.json 2> $O/r02_bench.err; echo "bench rc=$?" | tee -a $O/r02_prof.log
