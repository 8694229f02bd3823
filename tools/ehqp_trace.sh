export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tre
rocprofv3 --kernel-trace --output-format csv -d /tmp/tre -o t -- python $GRAFT_REPO_ROOT/tools/exp_frontend_lanes.py 2>&1 | grep -E "ehqp|nhqp"
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tre/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'ehqp' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[-12]['Start_Timestamp'])
for r in rows[-12:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1000.0:9.1f} {(int(r['End_Timestamp'])-t0)/1000.0:9.1f} q={r.get('Queue_Id','?')} grid={r.get('Grid_Size','?')} lds={r.get('LDS_Block_Size','?')} {r['Kernel_Name'][:40]}")
PY
