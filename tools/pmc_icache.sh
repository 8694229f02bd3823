#!/bin/sh
# developer helper, runs ON the GPU box: instruction-cache counters of the bench's kernels (one --pmc pass, kernel-trace only)
OUT=$PWD/gpurun_out/icache
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2> $OUT/err.txt
find $OUT/p1 -name "*counter_collection.csv" -exec cp {} $OUT/pmc1_counters.csv \;
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $OUT/err.txt
find $OUT/p2 -name "*counter_collection.csv" -exec cp {} $OUT/pmc2_counters.csv \;
rm -rf $OUT/p1 $OUT/p2
python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $OUT > $OUT/icache.json
python - <<PY
import json
d=json.load(open("$OUT/icache.json"))
for r in d["kernels"]:
    p=r["per_launch"]
    if "SQC_ICACHE_REQ" in p:
        print(r["kernel"][:60], r["workgroups"], {k:round(v) for k,v in p.items()})
PY
