# usage (on the GPU box): [SEEDS="1 2 3"] [CASES=60] bash scripts/gpu_fuzz.sh  -> gpurun_out/fuzz.log
# Longer sessions of the three randomised GPU tests with other seeds than the suite's (one pytest process per seed).
cd $GRAFT_REPO_ROOT
: > gpurun_out/fuzz.log
for s in ${SEEDS:-101 102 103}; do
  echo "== seed $s" >> gpurun_out/fuzz.log
  MSR_FUZZ_SEED=$s MSR_FUZZ_CASES=${CASES:-60} timeout -k 10 ${PER_SEED_TIMEOUT:-500} python -m pytest tests/test_gpu_parity.py -m gpu -x -q \
    -k "test_randomised_configurations or test_fused_hybrid_randomised or test_multitile_hybrid_randomised" >> gpurun_out/fuzz.log 2>&1 || { echo "FAILED seed $s rc=$?" >> gpurun_out/fuzz.log; break; }
done
grep -n "passed\|failed\|FAILED\|seed\|Error" gpurun_out/fuzz.log | tail -30
