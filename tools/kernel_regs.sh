#!/bin/bash
# Register / scratch / LDS use of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each:
#   tools/kernel_regs.sh llm-guided-multimodal-mil_amd/csrc/gated_pool_bf16.hip [name filter]
f=$1; pat=${2:-.}
cd "$(dirname "$f")" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$(basename "$f")" -o /dev/null 2>&1 \
 | grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' \
 | awk '/^Function Name:/{if(n)print n, v, a, sc, sp, oc, l; n=$3; next} /^VGPRs:/{v="vgpr="$2} /^AGPRs:/{a="agpr="$2} /^ScratchSize/{sc="scratch="$4} /^VGPRs Spill:/{sp="spill="$3} /^Occupancy/{oc="occ="$3} /^LDS Size/{l="lds="$4} END{print n, v, a, sc, sp, oc, l}' \
 | while read n rest; do echo "$(echo $n | c++filt | cut -c1-60) $rest"; done | grep -E "$pat"
