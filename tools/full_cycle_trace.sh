#!/bin/sh
# developer helper (GPU box): kernel durations and the launch timeline of the full_cycle sub-line (q -> kinematics -> cycle -> q += dq)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trf
rocprofv3 --kernel-trace --output-format csv -d /tmp/trf -o t -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench; r = bench.time_full_cycle(4096, 0, steps=10, warmup=3); print(r['value'], r['ms_per_step'])" 2>&1 | tail -2
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/trf/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(list)
for r in rows: agg[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
for k, v in agg.items(): print(f"{k:60s} n={len(v):4d} mean={sum(v)/len(v):8.1f} us  last10 mean={sum(v[-10:])/len(v[-10:]):8.1f}")
t0 = int(rows[-40]['Start_Timestamp'])
for r in rows[-40:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1000.0:9.1f} {(int(r['End_Timestamp'])-t0)/1000.0:9.1f} q={r.get('Queue_Id','?'):>3s} {r['Kernel_Name'][:50]}")
PY
