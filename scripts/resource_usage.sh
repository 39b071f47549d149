# usage: bash scripts/resource_usage.sh [file.hip] [name filter] -> one line per kernel: VGPRs, SGPRs, scratch, occupancy, LDS
cd "$(dirname "$0")/../mllm_sparse_retrieval_amd/csrc"
F=${1:-msr_device.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c $F -o /dev/null 2>&1 \
 | grep remark | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n, v; n=$3; v=""} /VGPRs:|TotalSGPRs|ScratchSize|Occupancy|LDS Size|VGPRs Spill/{v=v" | "$0} END{print n, v}' \
 | while read -r line; do name=$(echo "$line" | cut -d' ' -f1 | c++filt | cut -c1-70); echo "$name $(echo "$line" | cut -d' ' -f2-)"; done | grep -E "${2:-.}"
