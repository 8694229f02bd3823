#!/bin/sh
# developer helper, runs ON the GPU box (through gpurun): every randomised sweep of tests/stress_*.py against the witnesses
# (qpOASES reference build + the restatements), one output file per sweep under gpurun_out/stress_<tag>/ ; the files are
# copied into profiles/<tag>_stress_*.txt by hand.  The seeds are the ones every round since round 3 has used.
TAG=${1:-r05}
OUT=$PWD/gpurun_out/stress_$TAG
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
for s in 1 2 3 4; do python tests/stress_parity.py $s 800 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_parity.txt
python tests/stress_qp.py 1 60 2>&1 | grep -v amdgpu.ids > $OUT/stress_qp.txt
python tests/stress_frontends.py 1 160 2>&1 | grep -v amdgpu.ids > $OUT/stress_frontends.txt
for s in 11 12 13; do python tests/stress_hotstart.py $s 90 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_hotstart.txt
{ python tests/stress_closed_loop.py 41 1024 300 1e6 tasks; python tests/stress_closed_loop.py 42 1024 300 200 tasks;
  python tests/stress_closed_loop.py 41 1024 300 1e6 ttc; python tests/stress_closed_loop.py 42 1024 300 200 ttc; } 2>&1 | grep -v amdgpu.ids > $OUT/stress_closed_loop.txt
tail -n 3 $OUT/*.txt
# round 6: the 40-lane layout's null-space / low-rank paths (33 .. 38 variables, many equality rows)
for s in 5 6; do python tests/stress_parity.py $s 400 coman40 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_parity_coman40.txt
tail -n 3 $OUT/stress_parity_coman40.txt
# round 6: the explicit-QP surface beyond 64 variables (osot_qp_big.h)
for s in 3 4 5; do python tests/stress_qp.py $s 200 wide 2>&1 | grep -v amdgpu.ids; done > $OUT/stress_qp_wide.txt
tail -n 3 $OUT/stress_qp_wide.txt
# round 6: the reference's own robot in closed loop (the 40-lane layout's null-space paths under real drift); the second block is the
# out-of-reach regime whose census DESIGN.md section 5 discusses
for st in S1 S2 S3 S4; do for sd in 1 2; do python tests/stress_closed_loop_coman.py $sd 1024 200 $st 0.3 2>&1 | grep -v amdgpu.ids | tail -4; done; done > $OUT/stress_closed_loop_coman35.txt
for st in S1 S2 S3 S4; do for sd in 3 4; do python tests/stress_closed_loop_coman.py $sd 4096 400 $st 0.5 2>&1 | grep -v amdgpu.ids | tail -6; done; done > $OUT/stress_closed_loop_coman35_unreachable_goals.txt
tail -n 3 $OUT/stress_closed_loop_coman35.txt
