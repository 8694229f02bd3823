#!/bin/sh
# developer helper: an extra build of the product library with -D flags, for A/B runs on the GPU box
# (OSOT_MI355X_LIB=tools/bin/NAME.so python tools/bench_configs.py).  usage: tools/build_variant.sh NAME [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -pragma-unroll-threshold=1000000 \
  -Iopensot_amd/csrc -Iinclude "$@" opensot_amd/csrc/osot_mi355x.hip -o tools/bin/$name.so -lrccl
