#!/bin/sh
# developer helper (GPU box): per-launch durations of the nHQP kernels, in launch order (levels 0, 1, 2 repeat)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench; bench.time_nhqp(4096, 0, steps=3, warmup=1)" > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'osot_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows[-12:]:
    print(r['Kernel_Name'][:50], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0, 'us')
PY
