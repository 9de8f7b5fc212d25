#!/bin/bash
# Build experimental variants of the kernel library into tools/variants/ (scratch; not product).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared $v \
     llm-guided-multimodal-mil_amd/csrc/gated_pool.hip llm-guided-multimodal-mil_amd/csrc/gated_pool_bf16.hip llm-guided-multimodal-mil_amd/csrc/head_loss.hip llm-guided-multimodal-mil_amd/csrc/linear.hip llm-guided-multimodal-mil_amd/csrc/linear_x.hip llm-guided-multimodal-mil_amd/csrc/small_linear.hip llm-guided-multimodal-mil_amd/csrc/mid_linear.hip llm-guided-multimodal-mil_amd/csrc/attention.hip llm-guided-multimodal-mil_amd/csrc/absorbed_attn.hip \
     -o tools/variants/lib$name.so &
done
wait
ls tools/variants
