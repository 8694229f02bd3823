#!/bin/sh
# developer helper, runs ON the GPU box: the sweeps of tools/stress_round.sh with FURTHER seeds (round 5: parity 5..12, qp 2..4, frontends 2..3 + the wide mode 5..7,
# closed loop 43 / 44 in both modes and both eps) -> gpurun_out/stress_extra_<tag>/
TAG=${1:-r05}
OUT=$PWD/gpurun_out/stress_extra_$TAG
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
for s in 5 6 7 8 9 10 11 12; do python tests/stress_parity.py $s 800 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_parity.txt
for s in 2 3 4; do python tests/stress_qp.py $s 120 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_qp.txt
{ for s in 2 3; do python tests/stress_frontends.py $s 160 2>&1 | grep -v amdgpu.ids; done; for s in 5 6 7; do python tests/stress_frontends.py $s 120 wide 2>&1 | grep -v amdgpu.ids; done; } > $OUT/stress_frontends.txt
for s in 43 44 45 46; do for e in 1e6 200; do for m in tasks ttc; do python tests/stress_closed_loop.py $s 1024 300 $e $m; done; done; done 2>&1 | grep -v amdgpu.ids | grep -v "^BUG" > $OUT/stress_closed_loop.txt
tail -n 2 $OUT/*.txt
# round 6: the 40-lane null-space paths with further seeds, the explicit-QP surface beyond 64 variables
for s in 7 8 9 10; do python tests/stress_parity.py $s 400 coman40 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_parity_coman40.txt
for s in 7 8 9; do python tests/stress_qp.py $s 200 wide 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_qp_wide.txt
tail -n 2 $OUT/stress_parity_coman40.txt $OUT/stress_qp_wide.txt
