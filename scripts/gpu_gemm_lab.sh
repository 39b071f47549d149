# usage (on the GPU box, via gpurun): [RUNS="4:0 5:0 4:1"] bash scripts/gpu_gemm_lab.sh  -> gpurun_out/gemm_lab.txt
# scripts/_lab/gemm_lab is built in the container: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/gemm_lab.hip -o scripts/_lab/gemm_lab
cd $GRAFT_REPO_ROOT
: > gpurun_out/gemm_lab.txt
for r in ${RUNS:-4:0}; do
  nb=${r%%:*}; lab=${r##*:}
  timeout -k 10 120 scripts/_lab/gemm_lab 25010 5000 4096 20 $nb $lab >> gpurun_out/gemm_lab.txt 2>&1 || echo "rc=$? (nbuf=$nb lab=$lab)" >> gpurun_out/gemm_lab.txt
done
cat gpurun_out/gemm_lab.txt
