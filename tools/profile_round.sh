#!/bin/sh
# developer helper, runs ON the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench.py run (all
# configurations of the bench line: the headline lanes, C2 / C4 / C5, nHQP, eHQP, ADMM, kinematics), then separate --pmc
# passes (one counter group each, as MI355X_MICROARCH.md prescribes; --kernel-trace only beside them) over the same command.
# Output under gpurun_out/prof_<tag>/ ; the summaries are copied into profiles/ by hand.
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
T0=$(date +%s); python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt; echo "default bench.py run: $(( $(date +%s) - T0 )) s" > $OUT/bench_wall_seconds.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2>> $OUT/bench_stderr.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $OUT/bench_stderr.txt
  find $OUT/pmc$i -name "*counter_collection.csv" -exec cp {} $OUT/pmc${i}_counters.csv \;
done
python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $OUT > $OUT/pmc_kernels.json
for cfg in C3 C5; do python $GRAFT_REPO_ROOT/tools/prof_phases.py $cfg $( [ $cfg = C5 ] && echo 1024 || echo 4096 ) > $OUT/phase_cycles_$cfg.txt 2>&1; done
rm -rf $OUT/trace $OUT/pmc[0-9]
ls -la $OUT
# round 6: the phase census of the reference's COMAN stacks on the 40-lane layout
python $GRAFT_REPO_ROOT/tools/prof_phases_coman.py S2 S3 S4 2>&1 | grep -v amdgpu.ids > $OUT/phase_cycles_COMAN35.txt
