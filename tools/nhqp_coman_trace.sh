#!/bin/sh
# developer helper (GPU box): per-kernel average durations of the nHQP front-end on a COMAN35 stack (default S1)
W=${1:-S1}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench; print(bench.time_coman35('$W', 4096, 0, 5, 2, front_end='nHQP')['value'])" 2>&1 | tail -1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs']) / 1000.0, 1), 'us', r['Percentage'])
PY
