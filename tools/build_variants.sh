#!/bin/bash
# Build experimental variants of the kernel library into tools/variants/ (scratch; not product).
# usage: tools/build_variants.sh "-DFLAG_A" "-DFLAG_B -DFLAG_C" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
CS=llm-guided-multimodal-mil_amd/csrc
SRCS=$(sed -n 's/^SRCS = //p' $CS/Makefile)
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  ( cd $CS && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -shared $v $SRCS -o ../../tools/variants/lib$name.so ) &
done
wait
ls tools/variants
