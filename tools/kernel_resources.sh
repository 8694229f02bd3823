#!/bin/sh
# developer helper: per-kernel register / scratch / spill figures of the gfx950 build (compiler remarks)
# usage: tools/kernel_resources.sh [extra hipcc flags, e.g. -DOSOT_X_NO_FUSED_NS]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-value \
  -Iinclude -Iopensot_amd/csrc --cuda-device-only -c opensot_amd/csrc/osot_mi355x.hip -o /tmp/osot_dev.o "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name:| VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill" \
  | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
  | awk '/Function Name:/ { if (name != "") print name " | " line; name = $3; line = ""; next } { gsub(/^ +/, ""); line = line (line == "" ? "" : " | ") $0 } END { print name " | " line }' \
  | while IFS= read -r l; do n=$(echo "$l" | cut -d' ' -f1 | c++filt | sed 's/(.*//'); echo "$n |$(echo "$l" | cut -d'|' -f2-)"; done
