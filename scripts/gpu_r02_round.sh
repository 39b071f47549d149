# one GPU-box session: default bench line, launcher checks, counter passes for every profiled workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
python3 bench.py > $O/r02_bench.json 2> $O/r02_bench.err; echo "bench rc=$?" | tee -a $O/r02_round.log
python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/r02_l2.json 2> $O/r02_l2.err; echo "gpus2 (must be 2) rc=$?" | tee -a $O/r02_round.log
timeout -k 10 300 python3 bench.py --gpus 2 --allow-oversubscribe --steps 2 --warmup 1 --no-c3 --no-c5 --c4-docs 100000 --c4-queries 500 --c4-timeout 120 > $O/r02_l2o.json 2> $O/r02_l2o.err; echo "gpus2 oversubscribed (RCCL on one GPU: expect 4) rc=$?" | tee -a $O/r02_round.log
rm -f $O/r02_counters.json
bash scripts/gpu_prof.sh r02_flickr flickr30k_t2i --no-c3 --no-c4 --no-c5 && echo "prof flickr ok" | tee -a $O/r02_round.log
bash scripts/gpu_prof.sh r02_c3_i2t coco5k_i2t --only-c3 --c3-dir i2t && echo "prof c3 i2t ok" | tee -a $O/r02_round.log
bash scripts/gpu_prof.sh r02_c3_t2i coco5k_t2i --only-c3 --c3-dir t2i && echo "prof c3 t2i ok" | tee -a $O/r02_round.log
bash scripts/gpu_prof.sh r02_c5 c5_hybrid --only-c5 && echo "prof c5 ok" | tee -a $O/r02_round.log
bash scripts/gpu_prof.sh r02_c4 c4_1m --only-c4 && echo "prof c4 ok" | tee -a $O/r02_round.log
