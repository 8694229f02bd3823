#!/bin/sh
# developer helper: per-kernel register / scratch / spill figures of the gfx950 build (compiler remarks)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-value \
  -Iinclude -Iopensot_amd/csrc --cuda-device-only -c opensot_amd/csrc/osot_mi355x.hip -o /tmp/osot_dev.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name:| VGPRs:|AGPRs:|ScratchSize|SGPRs Spill|VGPRs Spill|error" \
  | sed 's/.*remark: *//; s/ \[-Rpass.*//; s/Function Name: /\n/' | tr '\n' '\t' | sed 's/\t_ZN/\n_ZN/g'; echo
